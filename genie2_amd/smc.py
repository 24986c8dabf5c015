"""Twisted-diffusion / SMC sampling with gradients through the denoiser: the algorithm of the fork's
`genie/sampler/unconditional_smc.py` (:25-43 weight helpers, :233-288 systematic resampling, :303-345 the motif twisting
function, :465-576 the loop) without its wandb / file logging.  The gradient torch.autograd takes there through the whole
Denoiser is `GenieEngine.denoise_vjp` here (HIP backward kernels, frames held fixed exactly as `T(rots.detach(), trans.detach())`
does); the potential itself acts on a [B, N, 3] tensor and stays in PyTorch.

The reference module imports wandb / Bio, which are not installed here, so it could not be run: parity of this file is
UNPINNED by reference outputs; what is tested is (a) the helpers against restatements of the quoted lines, (b) that a constant
potential reproduces the ancestral sampler, (c) that the guided gradient equals torch autograd through the oracle."""
import torch

from . import features as F
from . import pack
from .sampler import UnconditionalSampler


def normalize_log_weights(log_weights, dim):
    log_weights = log_weights - log_weights.max(dim=dim, keepdims=True)[0]
    return log_weights - torch.logsumexp(log_weights, dim=dim, keepdims=True)


def normalize_weights(log_weights, dim=0):
    return torch.exp(normalize_log_weights(log_weights, dim=dim))


def compute_ess(w, dim=0):
    return (w.sum(dim=dim)) ** 2 / torch.sum(w ** 2, dim=dim)


def compute_ess_from_log_w(log_w, dim=0):
    return compute_ess(normalize_weights(log_w, dim=dim), dim=dim)


def log_normal_density(sample, mean, var):
    return torch.distributions.normal.Normal(loc=mean, scale=torch.sqrt(var)).log_prob(sample)


def systematic_resampling(particles, weights, u=None):
    """unconditional_smc.py:237-288: one uniform draw u in [0, 1/N), points u + i/N, particle j is taken for every point in
    (cumsum_j, cumsum_{j+1}].  Returns (resampled particles, new log-weights = 0, indices)."""
    n = len(weights)
    weights = weights / torch.sum(weights)
    cumsum = torch.cumsum(weights, dim=0)
    if u is None:
        u = torch.distributions.Uniform(low=0.0, high=1.0 / n).sample()
    points = (torch.as_tensor(u, dtype=cumsum.dtype) + torch.arange(n, dtype=cumsum.dtype) / n).to(cumsum.device)
    # the reference walks j up while points[i] > cumsum[j + 1]: the number of cumulative sums strictly below the point
    indexes = torch.searchsorted(cumsum, points, right=False).clamp_(max=n - 1)
    return particles[indexes], torch.zeros(n, device=particles.device), indexes


def xstart_variance(alphas_cumprod_t, tausq=0.012):
    """unconditional_smc.py:290-302, var_type 6: sigma^2 tau^2 / (sigma^2 + tau^2), sigma^2 = (1 - abar) / abar."""
    sigmasq = (1 - alphas_cumprod_t) / alphas_cumprod_t
    return (sigmasq * tausq) / (sigmasq + tausq)


def motif_twisting_function(x0, motif_index_mask, motif_target, alphas_cumprod_t, tausq=0.012):
    """unconditional_smc.py:303-345: log of the mean over candidate placements of a Gaussian likelihood of the (centred) motif
    coordinates.  x0 [B,N,3]; motif_index_mask [n_placement, N] bool (one segment set per placement, equal counts);
    motif_target [n_motif_res, 3] centred.  Returns [B]."""
    var = xstart_variance(alphas_cumprod_t, tausq)
    scores = []
    for mask in motif_index_mask:
        sel = x0[:, mask]                                           # [B, n_motif_res, 3]
        sel = sel - sel.mean(dim=-2, keepdim=True)
        scores.append(-torch.sum((sel - motif_target[None]) ** 2, dim=(1, 2)) / (2 * var))
    score = torch.stack(scores)                                     # [n_placement, B]
    return torch.logsumexp(score, dim=0) - torch.log(torch.tensor(float(score.shape[0]), device=x0.device))


def get_all_motif_locations(L, segment_lengths, max_offsets=1000, rng=None):
    """unconditional_smc.py:173-215: every placement of the segments, in order and without overlap, inside 0..L-1, as
    [(start, end), ...] per placement (end inclusive), in the reference's order -- ascending in the first start, then the second,
    ...  More than `max_offsets` placements are thinned with one `choice(n, max_offsets, replace=False)` draw (numpy's global
    generator there; `rng` here, default the same global one)."""
    import numpy as np
    k, total = len(segment_lengths), sum(segment_lengths)
    out = []
    starts = [0] * k

    def place(seg, first_free):
        # the segments from `seg` on need this much room; the last admissible start leaves exactly that
        need = sum(segment_lengths[seg:])
        for st in range(first_free, L - need + 1):
            starts[seg] = st
            if seg + 1 == k:
                out.append([(s, s + n - 1) for s, n in zip(starts, segment_lengths)])
            else:
                place(seg + 1, st + segment_lengths[seg])

    if k and total <= L:
        place(0, 0)
    if len(out) > max_offsets:
        pick = (rng if rng is not None else np.random).choice(len(out), max_offsets, replace=False)
        out = [out[i] for i in pick]
    return out


def generate_motif_index_mask(motif_target, n_res, max_offsets=1000, rng=None, device=None):
    """unconditional_smc.py:172-232: bool [n_placement, n_segment, n_res, 3], True over the residues segment j occupies in
    placement i.  `motif_target`: one entry per segment (anything with a length: its residues)."""
    locs = get_all_motif_locations(n_res, [len(seg) for seg in motif_target], max_offsets, rng)
    mask = torch.zeros(len(locs), len(motif_target), n_res, 3, dtype=torch.bool)
    for i, placement in enumerate(locs):
        for j, (st, end) in enumerate(placement):
            mask[i, j, st:end + 1] = True
    return mask.to(device) if device is not None else mask


def placement_masks(motif_index_mask):
    """[n_placement, n_segment, N, 3] -> [n_placement, N]: the residues any segment covers (what motif_twisting_function takes)."""
    return motif_index_mask[..., 0].any(dim=1)


class TwistedSampler(UnconditionalSampler):
    """params: the UnconditionalSampler's + 'twisting_function': callable (x0_pred [B,N,3] requiring grad, step) -> log p(y | x_t) [B];
    or, instead, 'motif_target' (list of [n_i, 3] segments: the placements are enumerated with generate_motif_index_mask and the
    potential is motif_twisting_function over all of them, unconditional_smc.py:303-345); optional 'tausq' with it;
    optional 'noise' [T,B,N,3] (initial draw + one per step, as BaseSampler), 'resample_u' (list of uniforms, tests),
    'guidance_alpha' (default 0.012), 'ess_threshold' (default 0.5), 'last_unguided_steps' (default 50)."""

    def _sample(self, params):
        feats = F.convert_np_features_to_tensor(
            F.batchify_np_features([self.create_np_features(params) for _ in range(params['num_samples'])]), self.device)
        B, N = feats['residue_mask'].shape
        m = self.model
        T = m.config.diffusion['n_timestep']
        sched = {k: v.to(self.device) for k, v in pack.schedule_tensors(T).items()}
        abar, betas = sched['alphas_cumprod'], sched['betas']
        noise = params.get('noise')
        draw = (lambda k: noise[k].to(self.device)) if noise is not None else (lambda k: torch.randn(B, N, 3, device=self.device))
        twist = params.get('twisting_function')
        if twist is None:       # the reference's own potential: every placement of the motif segments (:172-232, 303-345)
            segs = [torch.as_tensor(x, dtype=torch.float32) for x in params['motif_target']]
            pm = placement_masks(generate_motif_index_mask(segs, N)).to(self.device)
            tgt = torch.cat(segs).to(self.device)
            tgt = tgt - tgt.mean(dim=0, keepdim=True)
            tausq = float(params.get('tausq', 0.012))
            twist = lambda x0, step: motif_twisting_function(x0, pm, tgt, abar[step], tausq)      # noqa: E731
        alpha = float(params.get('guidance_alpha', 0.012))
        eng = m.model.bind(feats)
        w = pack.flatten_state_dict(m.model.state_dict(), m.model.dims).to(self.device)
        mask = feats['residue_mask'].unsqueeze(-1).float()
        trans = draw(0)
        log_proposal = log_normal_density(trans, torch.tensor(0., device=self.device), torch.tensor(1., device=self.device)).sum(dim=(1, 2))
        log_w_acc = torch.zeros(B, device=self.device)
        rots = eng.frenet(trans)
        self.ess_trace, self.resampled_at = [], []
        us = list(params.get('resample_u', []))
        for it, step in enumerate(range(T, 0, -1)):
            ts = torch.full((B,), step, dtype=torch.int32, device=self.device)
            c0, c1 = torch.sqrt(abar[step]), torch.sqrt(1 - abar[step])
            z = eng.denoise(trans, rots, ts)['z']
            x0 = ((trans - c1 * z) / c0).detach().requires_grad_(True)          # E[x_0 | x_t] (:474)
            log_prob = twist(x0, step)
            g = torch.autograd.grad(log_prob.mean(), x0)[0] * B                   # (:480-482, "rescale mean back")
            # chain rule through x0(trans) = (trans - c1 z(trans)) / c0 with the frames fixed: the part through z is the HIP VJP
            _, dz_part = eng.denoise_vjp(w, trans, rots, ts, (-c1 / c0) * g)
            grad = g / c0 + dz_part
            norm = grad.double().norm().float()                                   # (f64: the f32 sum of squares overflows before the cap below acts)
            grad = grad * alpha * norm / (alpha + norm)                           # (:483-488)
            x0u = x0.detach()
            x0t = x0u + grad if step >= int(params.get('last_unguided_steps', 50)) else x0u
            coef1 = torch.sqrt(abar[step - 1]) * betas[step] / (1 - abar[step])
            coef2 = sched['sqrt_alphas'][step] * (1.0 - abar[step - 1]) / (1 - abar[step])
            mean_t, mean_u = coef1 * x0t + coef2 * trans, coef1 * x0u + coef2 * trans
            if step == 1:
                trans = mean_t
                break
            sigma = sched['sqrt_betas'][step]
            new = (mean_t + params['scale'] * sigma * draw(it + 1)) * mask
            log_rev = log_normal_density(new, mean_u, sigma ** 2).sum(dim=(1, 2))
            log_tw = log_normal_density(new, mean_t, sigma ** 2).sum(dim=(1, 2))
            log_w = (log_rev + log_prob.detach() - log_tw) - log_proposal
            log_proposal = log_prob.detach()
            log_w_acc = log_w + log_w_acc
            ess = compute_ess_from_log_w(log_w_acc)
            self.ess_trace.append(float(ess))
            if ess < float(params.get('ess_threshold', 0.5)) * B:
                new, log_w_acc, idx = systematic_resampling(new, torch.softmax(log_w_acc, dim=0), us.pop(0) if us else None)
                log_proposal = log_proposal[idx.to(self.device)]
                self.resampled_at.append(step)
            else:
                log_w_acc = normalize_log_weights(log_w_acc, dim=0) + torch.log(torch.tensor(float(B), device=self.device))
            trans = new
            rots = eng.frenet(trans)
        feats['atom_positions'] = trans.detach().cpu()
        return F.debatchify_np_features(F.convert_tensor_features_to_numpy(feats))
