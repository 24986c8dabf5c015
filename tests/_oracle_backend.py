"""Test infrastructure: a CPU stand-in for the device side of GenieTrainer, built on the oracle under torch autograd.  Lets the
world_size-2 gloo test run the real training-loop / DDP code (genie2_amd/training.py) without a GPU."""
import torch

from genie2_amd import pack
from oracle import genie_oracle as O


def small_config(n_pair=1, n_struct=1, n_timestep=50):
    from genie2_amd.config import Config
    cfg = Config()
    cfg.model['n_pair_transform_layer'] = n_pair
    cfg.model['n_structure_layer'] = n_struct
    cfg.diffusion['n_timestep'] = n_timestep
    cfg.io['max_n_res'] = 32
    return cfg


def unflatten(blob, dims):
    out, o = {}, 0
    for k, shp in pack.weight_layout(dims):
        n = 1
        for s in shp:
            n *= s
        out[k] = blob[o:o + n].reshape(shp)
        o += n
    return out


class OracleBackend:
    device = torch.device('cpu')

    def __init__(self, dims):
        self.dims = dims
        self.calls = []

    def bind(self, f):
        self.f = f

    def q_sample(self, x0, z, c0, c1):
        trans = c0.view(-1, 1, 1) * x0 + c1.view(-1, 1, 1) * z
        return trans, O.compute_frenet_frames(trans, self.f['chain_index'], self.f['residue_mask'])

    def forward_backward(self, w, g, trans, rots, s, z, cond_w, seed, opts, event):
        B, N = trans.shape[:2]
        sd = {k: v.clone().requires_grad_(True) for k, v in unflatten(w, self.dims).items()}
        masks = None
        if opts['train_mode']:
            masks = O.train_dropout_masks(self.dims, B, N, seed, opts['tri_dropout'], opts['ipa_dropout'], opts['transition_dropout'])
        out = O.denoiser_forward(sd, self.dims, rots, trans, s, self.f, 'closed', None, None, masks)
        lo = O.training_loss(out['z'], z, O.prepare_features(self.f), cond_w)
        lo['weighted_loss'].backward()
        g.copy_(torch.cat([sd[k].grad.reshape(-1) for k, _ in pack.weight_layout(self.dims)]))
        self.calls.append(dict(trans=trans, rots=rots, s=s, z=z, seed=seed))
        return {k: v.detach() for k, v in lo.items()} | {'z': out['z'].detach(), 'grads': g}

    def adam(self, w, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-8):
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        w.addcdiv_(m / (1 - b1 ** step), (v / (1 - b2 ** step)).sqrt() + eps, value=-lr)
