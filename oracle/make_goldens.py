"""
Golden-fixture generator -- runs ONLY in the development container.

Imports the real reference (marvinli00/genie2 mounted read-only at
/root/reference) and drives its own modules -- `Denoiser`,
`compute_frenet_frames`, `get_betas`, `sinusoidal_encoding`, `quat_to_rot`,
`rot_to_quat`, `UnconditionalSampler._sample`, `save_np_features_to_pdb` --
on seeded inputs, checks the restatement in `oracle/genie_oracle.py` against
them, and writes the inputs + expected outputs as small `.npz` fixtures under
`tests/golden/`.  Only data is written: no reference source travels.

    python oracle/make_goldens.py            # regenerate every fixture

The reference's Lightning-bound `Genie`/`DDPM` classes are not importable here
(no pytorch_lightning); `_Shim` below supplies the handful of attributes
`BaseSampler` reads (sampler/base.py:27-34, 236-270), deriving the schedule
with the reference's own `get_betas` exactly as diffusion/ddpm.py:40-66 does.
"""
import hashlib
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get('GENIE_REFERENCE', '/root/reference')
# The repo root also holds a `genie` compatibility package; keep it OFF sys.path here
# so that `genie.*` below is the real reference.
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location('genie_oracle', os.path.join(HERE, 'genie_oracle.py'))
O = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(O)
sys.path[:] = [p for p in sys.path if os.path.abspath(p or '.') != ROOT]
sys.path.insert(0, REF)

from genie.config import Config  # noqa: E402  (reference)
from genie.model.model import Denoiser  # noqa: E402
from genie.diffusion.schedule import get_betas  # noqa: E402
from genie.utils.affine_utils import T, quat_to_rot, rot_to_quat  # noqa: E402
from genie.utils.geo_utils import compute_frenet_frames  # noqa: E402
from genie.utils.loss import mse as ref_mse  # noqa: E402
from genie.utils.encoding import sinusoidal_encoding  # noqa: E402
from genie.utils import feat_utils  # noqa: E402
from genie.sampler.unconditional import UnconditionalSampler  # noqa: E402
import genie.model.pair_feature_net as ref_pfn  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


def ref_config(n_timestep=1000):
    cfg = Config(os.path.join(REF, 'results', 'base', 'configuration'))
    cfg.diffusion['n_timestep'] = n_timestep
    return cfg


def ref_denoiser(cfg, sd):
    m = Denoiser(**cfg.model, n_timestep=cfg.diffusion['n_timestep'],
                 max_n_res=cfg.io['max_n_res'], max_n_chain=cfg.io['max_n_chain']).eval()
    m.load_state_dict(sd, strict=True)
    return m


class _Shim:
    """Lightning-free stand-in for genie.diffusion.genie.Genie."""

    def __init__(self, cfg, denoiser):
        self.config = cfg
        self.device = torch.device('cpu')
        self.model = denoiser

    def setup_schedule(self):
        self.betas = get_betas(self.config.diffusion['n_timestep'], self.config.diffusion['schedule'])
        self.alphas = 1. - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, 0)
        self.sqrt_betas = torch.sqrt(self.betas)
        self.sqrt_alphas = torch.sqrt(self.alphas)
        self.sqrt_one_minus_alphas_cumprod = torch.sqrt(1. - self.alphas_cumprod)


def sha(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()


def feats_to_np(f):
    return {('f_' + k): v.numpy() for k, v in f.items()}


def maxdiff(a, b):
    return float((a - b).abs().max())


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f'  wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)')


# ------------------------------------------------------------------ KATs

def gen_schedule():
    out = {}
    for Tn in (100, 1000):
        b = get_betas(Tn, 'cosine')
        mine = O.cosine_beta_schedule(Tn)
        assert torch.equal(b, mine), 'schedule restatement differs'
        out[f'betas_{Tn}'] = b
    save('schedule', **out)


def gen_encoding():
    out = {}
    for nm, (vmax, N, D) in dict(pos=(256, 256, 256), chain=(4, 1, 64), t1000=(1001, 1000, 512),
                                 t100=(101, 100, 512)).items():
        v = torch.arange(vmax, dtype=torch.int32)
        e = sinusoidal_encoding(v, N, D)
        mine = O.sinusoidal_encoding(v, N, D)
        assert torch.equal(e, mine), 'encoding restatement differs'
        # tables are large; store a strided sample plus a checksum of the whole
        out[nm + '_rows'] = np.array([0, 1, 2, vmax // 2, vmax - 1])
        out[nm + '_vals'] = e[out[nm + '_rows']]
        out[nm + '_sum'] = e.double().sum()
        out[nm + '_abs_sum'] = e.double().abs().sum()
    save('encoding', **out)


def gen_geometry():
    g = torch.Generator().manual_seed(7)
    out = {}
    # frenet: ragged, multi-chain, B != 3 (torch.cross without dim, see oracle docstring)
    B, N = 4, 24
    coords = 3.0 * torch.randn(B, N, 3, generator=g)
    mask = torch.zeros(B, N, dtype=torch.int32)
    lengths = [24, 17, 9, 5]
    chains = torch.zeros(B, N, dtype=torch.int32)
    for b, L in enumerate(lengths):
        mask[b, :L] = 1
    chains[1, 8:17] = 1          # two chains 8 + 9
    chains[2, 3:6] = 1
    chains[2, 6:9] = 2           # three chains 3 + 3 + 3
    rots = compute_frenet_frames(coords, chains, mask)
    mine = O.compute_frenet_frames(coords, chains, mask)
    assert maxdiff(rots, mine) == 0.0, 'frenet restatement differs'
    det = torch.linalg.det(rots[0, :24])
    assert torch.allclose(det, -torch.ones_like(det), atol=1e-4), 'expected improper frames'
    out.update(frenet_coords=coords, frenet_mask=mask, frenet_chains=chains, frenet_rots=rots)
    # quat_to_rot
    q = torch.randn(64, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    r = quat_to_rot(q)
    assert maxdiff(r, O.quat_to_rot(q)) < 1e-6
    out.update(q2r_q=q, q2r_r=r)
    # rot_to_quat on products of frenet frames (R_j . R_i, proper)
    fr = rots[0]
    rr = torch.matmul(fr.unsqueeze(0), fr.unsqueeze(1))
    qe = rot_to_quat(rr)
    qc = O.rot_to_quat_closed(rr)
    codes = O.quat_sign_codes(qe)
    qa = O.apply_sign_codes(qc, codes)
    print('  rot_to_quat eigh vs closed (sign aligned):', maxdiff(qe, qa))
    assert maxdiff(qe, qa) < 2e-4
    out.update(r2q_r=rr, r2q_q=qe, r2q_codes=codes)
    save('geometry', **out)


def gen_pdb():
    f = feat_utils.create_empty_np_features([7])
    g = np.random.RandomState(3)
    f['atom_positions'] = g.randn(7, 3) * 10
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'x.pdb')
        feat_utils.save_np_features_to_pdb(f, p)
        data = open(p, 'rb').read()
    save('pdb_writer', atom_positions=f['atom_positions'], pdb_bytes=np.frombuffer(data, dtype=np.uint8))


# ------------------------------------------------------------------ single calls

def run_ref_call(model, rots, trans, ts, feats, record):
    """One reference Denoiser.forward with hooks for stage taps + eigh signs."""
    taps = {}
    orig = ref_pfn.rot_to_quat

    def rec_q(r):
        q = orig(r)
        record.append(q)
        return q

    ref_pfn.rot_to_quat = rec_q
    hooks = [
        model.pair_feature_net.register_forward_hook(lambda m, i, o: taps.__setitem__('p_init', o)),
        model.pair_transform_net.net[0].register_forward_hook(lambda m, i, o: taps.__setitem__('p_after_layer0', o[0])),
        model.pair_transform_net.net[0].tri_mul_out.register_forward_hook(lambda m, i, o: taps.__setitem__('trimul_out0', o)),
        model.structure_net.net[0].ipa.register_forward_hook(lambda m, i, o: taps.__setitem__('ipa_out0', o)),
    ]
    try:
        with torch.no_grad():
            out = model(T(rots, trans), ts, feats)
    finally:
        ref_pfn.rot_to_quat = orig
        for h in hooks:
            h.remove()
    return out, taps


def case_inputs(name, g):
    """(features, trans, timestep) for each single-call case."""
    if name == 'uncond_n16_b1_t1000':
        f = O.empty_features([16]); t = 1000
    elif name == 'uncond_n32_b2_t500':
        f = O.empty_features([32, 32]); t = 500
    elif name == 'ragged_n50_b2_t1':
        f = O.empty_features([50, 37]); t = 1
    elif name == 'twochain_n32_b1_t777':
        f = O.empty_features([32], chains_per_sample=[[20, 12]]); t = 777
    elif name == 'motif_n40_b2_t300':
        f = O.empty_features([40, 33])
        seqs, coords = feat_utils.parse_pdb(os.path.join(REF, '6E6R_long_motif.pdb'))
        ca = torch.tensor(coords[0][:10], dtype=torch.float32)
        ca = ca - ca.mean(0, keepdim=True)
        O.add_motif(f, 0, ca, list(range(12, 22)), aatype_idx=torch.tensor(seqs[0][:10]))
        O.add_motif(f, 1, ca[:6], [3, 4, 5, 20, 21, 22])
        t = 300
    else:
        raise KeyError(name)
    B, N = f['residue_mask'].shape
    trans = 2.5 * torch.randn(B, N, 3, generator=g)
    return f, trans, t


def gen_single_calls(sd):
    cfg = ref_config(1000)
    model = ref_denoiser(cfg, sd)
    dims = dict(O.BASE_DIMS)
    g = torch.Generator().manual_seed(11)
    pick = torch.Generator().manual_seed(5)
    for name in ('uncond_n16_b1_t1000', 'uncond_n32_b2_t500', 'ragged_n50_b2_t1',
                 'twochain_n32_b1_t777', 'motif_n40_b2_t300'):
        print(name)
        f, trans, t = case_inputs(name, g)
        B, N = f['residue_mask'].shape
        fr = O.prepare_features(f)
        rots = compute_frenet_frames(trans, fr['chain_index'], fr['residue_mask'])
        ts = torch.full((B,), t, dtype=torch.int32)
        rec = []
        out, taps = run_ref_call(model, rots, trans, ts, fr, rec)
        codes = O.quat_sign_codes(rec[0])

        # restatement check, both quaternion modes
        o_e = O.denoiser_forward(sd, dims, rots, trans, ts, f, 'eigh')
        o_c = O.denoiser_forward(sd, dims, rots, trans, ts, f, 'closed', codes)
        m3 = fr['residue_mask'].unsqueeze(-1).float()
        zs = float(out['z'].abs().max())
        print(f'  |z|max {zs:.3f}  oracle(eigh) dz {maxdiff(out["z"] * m3, o_e["z"] * m3):.2e}'
              f'  oracle(closed+codes) dz {maxdiff(out["z"] * m3, o_c["z"] * m3):.2e}'
              f'  ds {maxdiff(out["s"], o_e["s"]):.2e}  dp {maxdiff(out["p"], o_e["p"]):.2e}')
        assert maxdiff(out['z'] * m3, o_c['z'] * m3) <= 1e-4 * max(1.0, zs)
        assert maxdiff(out['p'], o_c['p']) <= 2e-4 * max(1.0, float(out['p'].abs().max()))

        # sampled pair entries (fixed index set) to keep the fixture small
        n_s = 256
        idx = torch.stack([torch.randint(0, B, (n_s,), generator=pick),
                           torch.randint(0, N, (n_s,), generator=pick),
                           torch.randint(0, N, (n_s,), generator=pick)], dim=1)

        def samp(p):
            return p[idx[:, 0], idx[:, 1], idx[:, 2]]

        save('call_' + name,
             timestep=t, trans=trans, rots=rots, quat_codes=codes,
             z=out['z'], s=out['s'], states=out['states'][[1, -1]],
             rots_out=out['ts'].rots, trans_out=out['ts'].trans,
             p_idx=idx, p_final_samples=samp(out['p']), p_init_samples=samp(taps['p_init']),
             p_layer0_samples=samp(taps['p_after_layer0']), trimul_out0_samples=samp(taps['trimul_out0']),
             p_final_abs_mean=out['p'].abs().mean(), p_final_abs_max=out['p'].abs().max(),
             ipa_out0=taps['ipa_out0'],
             **feats_to_np(f))


# ------------------------------------------------------------------ trajectory

def gen_trajectory(sd):
    """Config 1: N=50, T=100, B=1, scale 0.6 through the reference's real
    UnconditionalSampler._sample (sampler/base.py:169-289)."""
    Tn, N, B, scale, seed = 100, 50, 1, 0.6, 42
    cfg = ref_config(Tn)
    model = ref_denoiser(cfg, sd)
    shim = _Shim(cfg, model)
    sampler = UnconditionalSampler(shim)

    torch.manual_seed(seed)
    noise = torch.stack([torch.randn(B, N, 3) for _ in range(Tn)])   # draw order of base.py:227,269

    rec, traj = [], []
    orig_q = ref_pfn.rot_to_quat

    def rec_q(r):
        q = orig_q(r)
        rec.append(O.quat_sign_codes(q))
        return q

    ref_pfn.rot_to_quat = rec_q
    # record x_t after every step through the frenet call at the end of each iteration
    import genie.sampler.base as ref_base
    orig_f = ref_base.compute_frenet_frames
    ref_base.compute_frenet_frames = lambda c, ch, m: (traj.append(c.clone()) or orig_f(c, ch, m))
    torch.manual_seed(seed)
    t0 = time.time()
    try:
        out = sampler._sample(dict(length=N, scale=scale, num_samples=B, outdir='.', prefix='x', offset=0))
    finally:
        ref_pfn.rot_to_quat = orig_q
        ref_base.compute_frenet_frames = orig_f
    dt = time.time() - t0
    print(f'  reference _sample N={N} T={Tn}: {dt:.1f}s = {Tn / dt:.2f} steps/s ({torch.get_num_threads()} threads)')
    final = torch.tensor(out[0]['atom_positions'], dtype=torch.float32)[None]
    assert torch.equal(traj[0], noise[0]), 'noise replay mismatch'
    traj = torch.stack(traj[1:])                       # [T, B, N, 3] state after each step
    codes = torch.stack(rec)                           # [T, B, N, N]
    assert codes.shape[0] == Tn

    # restatement check with the recorded signs
    dims = dict(O.BASE_DIMS, n_timestep=Tn)
    f = O.empty_features([N] * B)
    mine, _, _ = O.sample_loop(sd, dims, f, noise, scale, 'closed', codes)
    rms = float(final.pow(2).mean().sqrt())
    print(f'  oracle(closed+codes) vs reference: max|dCa| {maxdiff(mine, final):.2e} at coordinate RMS {rms:.2f}')
    assert maxdiff(mine, final) <= 1e-4 * rms
    save('trajectory_n50_t100', noise=noise, quat_codes=codes, final=final, every10=traj[9::10],
         scale=scale, seed=seed, ref_steps_per_s=Tn / dt, ref_threads=torch.get_num_threads())


def gen_motif(sd):
    """Config 4 (scaffold path): the reference's motif-problem parser, mask sampler, feature
    builder and motif-PDB writer on tests/golden/motif_problem_6E6R.pdb (REMARK 999 header
    authored for the tests; ATOM records = the reference's 6E6R_long_motif.pdb data file), and a
    ragged, motif-conditioned B=2 trajectory through its real ScaffoldSampler._sample."""
    from genie.utils import motif_utils
    from genie.sampler.scaffold import ScaffoldSampler
    import genie.sampler.base as ref_base
    path = os.path.join(OUT, 'motif_problem_6E6R.pdb')
    spec = motif_utils.load_motif_spec(path)
    out = dict(spec_name=spec['name'], spec_min=spec['min_total_length'], spec_max=spec['max_total_length'],
               spec_structures=np.array([[s['type'] == 'motif', s.get('min_length', s.get('start_index')),
                                          s.get('max_length', s.get('end_index')),
                                          ord(s.get('group', ' ')), ord(s.get('chain', ' '))]
                                         for s in spec['structures']]))
    seqs, coords = feat_utils.parse_pdb(path)
    out['parse_seq'] = np.array(seqs[0]); out['parse_coords'] = np.array(coords[0])
    for seed in range(4):
        np.random.seed(seed)
        f = feat_utils.create_np_features_from_motif_pdb(path)
        for k, v in f.items():
            out[f'seed{seed}_{k}'] = np.asarray(v)
        with tempfile.TemporaryDirectory() as d:
            motif_utils.save_motif_pdb(path, f['fixed_sequence_mask'], os.path.join(d, 'm.pdb'))
            out[f'seed{seed}_motif_pdb'] = np.frombuffer(open(os.path.join(d, 'm.pdb'), 'rb').read(), dtype=np.uint8)
    # unconditional features of the same file (feat_utils.create_np_features_from_pdb)
    f = feat_utils.create_np_features_from_pdb(path)
    for k, v in f.items():
        out[f'frompdb_{k}'] = np.asarray(v)
    save('motif_features', **out)

    Tn, B, scale, seed = 20, 2, 0.6, 7
    cfg = ref_config(Tn)
    model = ref_denoiser(cfg, sd)
    sampler = ScaffoldSampler(_Shim(cfg, model))
    np.random.seed(seed)
    lens = [len(feat_utils.create_np_features_from_motif_pdb(path)['residue_mask']) for _ in range(B)]
    N = max(lens)
    torch.manual_seed(seed)
    noise = torch.stack([torch.randn(B, N, 3) for _ in range(Tn)])
    rec, traj = [], []
    orig_q = ref_pfn.rot_to_quat
    ref_pfn.rot_to_quat = lambda r: (lambda q: (rec.append(O.quat_sign_codes(q)), q)[1])(orig_q(r))
    orig_f = ref_base.compute_frenet_frames
    ref_base.compute_frenet_frames = lambda c, ch, m: (traj.append(c.clone()) or orig_f(c, ch, m))
    np.random.seed(seed)
    torch.manual_seed(seed)
    try:
        res = sampler._sample(dict(filepath=path, scale=scale, strength=0, num_samples=B, outdir='.', prefix='x', offset=0))
    finally:
        ref_pfn.rot_to_quat = orig_q
        ref_base.compute_frenet_frames = orig_f
    assert [len(r['residue_mask']) for r in res] == lens and torch.equal(traj[0], noise[0])
    final = torch.zeros(B, N, 3)
    for b, r in enumerate(res):
        final[b, :lens[b]] = torch.tensor(r['atom_positions'], dtype=torch.float32)
    codes = torch.stack(rec)
    # restatement check: same features (np seed), recorded signs
    np.random.seed(seed)
    feats = feat_utils.convert_np_features_to_tensor(feat_utils.batchify_np_features(
        [feat_utils.create_np_features_from_motif_pdb(path) for _ in range(B)]), 'cpu')
    dims = dict(O.BASE_DIMS, n_timestep=Tn)
    mine, _, _ = O.sample_loop(sd, dims, feats, noise, scale, 'closed', codes)
    rms = float(final.pow(2).mean().sqrt())
    print(f'  scaffold: lengths {lens}; oracle(closed+codes) vs reference max|dCa| {maxdiff(mine, final):.2e} (RMS {rms:.2f})')
    assert maxdiff(mine, final) <= 1e-4 * max(rms, 1.0)
    save('trajectory_scaffold_t20', noise=noise, quat_codes=codes, final=final, lengths=np.array(lens),
         scale=scale, seed=seed, **feats_to_np(feats))


def gen_train():
    """The two ends of Genie.training_step (diffusion/genie.py:66-105) around the denoiser call, driven line by line with
    the reference's own get_betas / compute_frenet_frames / mse (the LightningModule itself is not importable here):
    forward noising + frames, and the loss with its gradient with respect to the predicted noise (reference autograd)."""
    g = torch.Generator().manual_seed(77)
    T_ = 1000
    betas = get_betas(T_, 'cosine')
    ac = torch.cumprod(1. - betas, 0)                                # ddpm.py:45-46
    sqrt_ac, sqrt_1mac = torch.sqrt(ac), torch.sqrt(1. - ac)         # ddpm.py:54,56
    # (batch of 4, not 3: the reference's torch.cross without dim= picks the batch axis when B == 3)
    f = O.empty_features([24, 17, 9, 20], chains_per_sample=[[10, 14], [17], [4, 5], [20]])
    B, N = f['residue_mask'].shape
    f['fixed_sequence_mask'][0, 3:9] = True                         # a motif-conditioned structure, an unconditional one, ...
    f['fixed_sequence_mask'][2, 0:4] = True
    f['atom_positions'] = torch.randn(B, N, 3, generator=g) * 8 * f['residue_mask'].unsqueeze(-1)
    s = torch.tensor([1, 500, 1000, 37])
    z = torch.randn(B, N, 3, generator=g) * f['residue_mask'].unsqueeze(-1)                                     # genie.py:77
    trans_s = sqrt_ac[s].view(-1, 1, 1) * f['atom_positions'] + sqrt_1mac[s].view(-1, 1, 1) * z                 # genie.py:80-81
    rots_s = compute_frenet_frames(trans_s, f['chain_index'], f['residue_mask'])                                # genie.py:82-86
    z_pred = (z + 0.3 * torch.randn(B, N, 3, generator=g)).requires_grad_(True)     # stands in for output['z']
    w = 2.5                                                          # config.training['condition_loss_weight'] in the test
    rm, fs = f['residue_mask'], f['fixed_sequence_mask']
    condition_mask = rm * fs                                         # genie.py:91-92
    infill_mask = rm * ~fs
    condition_losses = ref_mse(z_pred, z, condition_mask, aggregate='sum')
    infill_losses = ref_mse(z_pred, z, infill_mask, aggregate='sum')
    unweighted_losses = (condition_losses + infill_losses) / f['num_residues']
    weighted_losses = (w * condition_losses + infill_losses) / (w * torch.sum(condition_mask, dim=-1) + torch.sum(infill_mask, dim=-1))
    unweighted_loss, weighted_loss = torch.mean(unweighted_losses), torch.mean(weighted_losses)
    weighted_loss.backward()
    # the restatement agrees with the reference
    sched = O.training_schedule(T_)
    assert maxdiff(sched['sqrt_alphas_cumprod'], sqrt_ac) == 0 and maxdiff(sched['sqrt_one_minus_alphas_cumprod'], sqrt_1mac) == 0
    tr_o, ro_o = O.q_sample(f['atom_positions'], s, z, f['chain_index'], f['residue_mask'], sched)
    assert maxdiff(tr_o, trans_s) == 0 and maxdiff(ro_o, rots_s) < 1e-6
    zp = z_pred.detach().clone().requires_grad_(True)
    lo = O.training_loss(zp, z, f, w)
    lo['weighted_loss'].backward()
    assert maxdiff(lo['weighted_loss'].detach(), weighted_loss.detach()) < 1e-7 and maxdiff(zp.grad, z_pred.grad) < 1e-8
    assert maxdiff(lo['unweighted_loss'].detach(), unweighted_loss.detach()) < 1e-7
    save('train_ends_n24_b4', atom_positions=f['atom_positions'], residue_mask=f['residue_mask'], chain_index=f['chain_index'],
         residue_index=f['residue_index'], fixed_sequence_mask=f['fixed_sequence_mask'], num_residues=f['num_residues'],
         lengths=np.array([24, 17, 9, 20]), s=s, z=z, sqrt_alphas_cumprod_s=sqrt_ac[s], sqrt_one_minus_alphas_cumprod_s=sqrt_1mac[s],
         trans_s=trans_s, rots_s=rots_s, z_pred=z_pred.detach(), condition_loss_weight=np.float32(w),
         condition_losses=condition_losses.detach(), infill_losses=infill_losses.detach(),
         unweighted_loss=unweighted_loss.detach(), weighted_loss=weighted_loss.detach(), grad_z_pred=z_pred.grad)


def gen_train_grads(sd):
    """Gradient of the training loss through the reference Denoiser (eval mode: the dropouts of structure_net.py:63-70 /
    pair_transform_net.py are identities, so the result is a function of the inputs), for the backward kernels of the
    training row to be checked against.  The quaternion signs of the reference's eigh are recorded and replayed."""
    g = torch.Generator().manual_seed(99)
    cfg = ref_config()
    model = ref_denoiser(cfg, sd)
    f = O.empty_features([16, 11], chains_per_sample=[[16], [5, 6]])
    B, N = f['residue_mask'].shape
    f['fixed_sequence_mask'][0, 2:6] = True
    f['atom_positions'] = torch.randn(B, N, 3, generator=g) * 6 * f['residue_mask'].unsqueeze(-1)
    T_ = cfg.diffusion['n_timestep']
    betas = get_betas(T_, 'cosine')
    ac = torch.cumprod(1. - betas, 0)
    s = torch.tensor([700, 40])
    z = torch.randn(B, N, 3, generator=g) * f['residue_mask'].unsqueeze(-1)
    trans_s = torch.sqrt(ac)[s].view(-1, 1, 1) * f['atom_positions'] + torch.sqrt(1. - ac)[s].view(-1, 1, 1) * z
    rots_s = compute_frenet_frames(trans_s, f['chain_index'], f['residue_mask'])
    feats = O.prepare_features(f)
    record = []
    orig = ref_pfn.rot_to_quat

    def rec_q(r):
        q = orig(r)
        record.append(q.detach())
        return q

    ref_pfn.rot_to_quat = rec_q
    try:
        out = model(T(rots_s, trans_s), s.int(), feats)
    finally:
        ref_pfn.rot_to_quat = orig
    w = 2.0
    rm, fs = f['residue_mask'], f['fixed_sequence_mask']
    cm, im = rm * fs, rm * ~fs
    cl = ref_mse(out['z'], z, cm, aggregate='sum')
    il = ref_mse(out['z'], z, im, aggregate='sum')
    loss = torch.mean((w * cl + il) / (w * torch.sum(cm, dim=-1) + torch.sum(im, dim=-1)))
    loss.backward()
    ref_grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    # the oracle restatement under autograd, same quaternion signs
    codes = O.quat_sign_codes(record[0])
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o = O.denoiser_forward(sdg, O.BASE_DIMS, rots_s, trans_s, s.int(), f, 'closed', codes)
    lo = O.training_loss(o['z'], z, f, w)['weighted_loss']
    lo.backward()
    worst = 0.0
    for k in ref_grads:
        scale = max(float(ref_grads[k].abs().max()), 1e-6)
        worst = max(worst, float((sdg[k].grad - ref_grads[k]).abs().max()) / scale)
    print('  loss ref %.6f oracle %.6f; worst relative gradient difference %.2e over %d tensors' % (float(loss), float(lo), worst, len(ref_grads)))
    assert abs(float(loss) - float(lo)) < 1e-5 and worst < 5e-3
    keys = list(ref_grads.keys())
    probe = {k: ref_grads[k].reshape(-1)[:8] for k in keys}
    save('train_grads_n16_b2', atom_positions=f['atom_positions'], residue_mask=f['residue_mask'], chain_index=f['chain_index'],
         residue_index=f['residue_index'], fixed_sequence_mask=f['fixed_sequence_mask'], num_residues=f['num_residues'],
         lengths=np.array([16, 11]), chain_lengths=np.array([16, 0, 5, 6]), s=s, z=z, trans_s=trans_s, rots_s=rots_s,
         quat_codes=codes, condition_loss_weight=np.float32(w), loss=loss.detach(), z_pred=out['z'].detach(),
         keys=np.array(keys), grad_abs_max=np.array([float(ref_grads[k].abs().max()) for k in keys], dtype=np.float32),
         grad_norm=np.array([float(ref_grads[k].norm()) for k in keys], dtype=np.float32),
         grad_probe=np.stack([probe[k].numpy() if probe[k].numel() == 8 else np.pad(probe[k].numpy(), (0, 8 - probe[k].numel())) for k in keys]))


def gen_dataset():
    """genie/data/dataset.py: the reference's own GenieDataset on one of the structure files it ships (results/test001/pdbs/100_0.pdb,
    copied as a data fixture): __getitem__ with motif_prob = 1 under three (numpy, random) seeds, and the unconditional item."""
    import random
    import shutil
    from genie.data.dataset import GenieDataset
    src = os.path.join(REF, 'results', 'test001', 'pdbs', '100_0.pdb')
    shutil.copy(src, os.path.join(OUT, 'dataset_100_0.pdb'))
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(src, os.path.join(d, 'x100.pdb'))
        random.seed(0)
        ds = GenieDataset({'datadir': d, 'names': ['x100']}, 20, 128, 1, 1.0, 0.05, 0.5, 1, 4)
        arrs = {}
        for seed in (1, 2, 3):
            np.random.seed(seed)
            random.seed(seed)
            it = ds[0]
            for k, v in it.items():
                arrs[f's{seed}_{k}'] = np.asarray(v)
        ds.motif_prob = 0.0
        np.random.seed(9)
        it = ds[0]
        for k, v in it.items():
            arrs[f'u_{k}'] = np.asarray(v)
    save('dataset_items', **arrs)


def main():
    torch.manual_seed(0)
    print('weights (synthetic recipe, seed 0)')
    sd = O.synthetic_state_dict(O.BASE_DIMS, seed=0)
    # key/shape layout must equal the reference's own state_dict
    cfg = ref_config()
    ref_sd = ref_denoiser(cfg, sd).state_dict()
    assert list(ref_sd.keys()) == [k for k, _ in O.state_dict_template(O.BASE_DIMS)]
    h = hashlib.sha256()
    for k in sd:
        h.update(sd[k].numpy().tobytes())
    save('weights_recipe', sha256=np.frombuffer(h.digest(), dtype=np.uint8),
         n_params=sum(v.numel() for v in sd.values()),
         keys=np.array(list(ref_sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in ref_sd.values()]),
         probe=sd['structure_net.net.7.ipa.linear_out.weight'][:4, :8])
    if len(sys.argv) > 1 and sys.argv[1] == 'motif':
        print('motif'); gen_motif(sd)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'dataset':
        print('dataset'); gen_dataset()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'train':
        print('train'); gen_train(); gen_train_grads(sd)
        return
    print('schedule'); gen_schedule()
    print('encoding'); gen_encoding()
    print('geometry'); gen_geometry()
    print('pdb'); gen_pdb()
    gen_single_calls(sd)
    print('trajectory'); gen_trajectory(sd)
    print('motif'); gen_motif(sd)
    print('train'); gen_train(); gen_train_grads(sd)
    print('dataset'); gen_dataset()


if __name__ == '__main__':
    main()
