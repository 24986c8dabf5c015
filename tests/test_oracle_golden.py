"""The oracle (oracle/genie_oracle.py) against the fixtures generated from the
real reference (oracle/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import CALL_CASES, golden_features, load_golden
from oracle import genie_oracle as O


def t(x):
    return torch.from_numpy(np.asarray(x))


def test_schedule_matches_reference():
    g = load_golden('schedule')
    for T in (100, 1000):
        b = O.cosine_beta_schedule(T)
        assert b.shape == (T + 1,) and b[0] == 0
        assert torch.equal(b, t(g[f'betas_{T}']))
    s = O.setup_schedule(1000)
    assert abs(float(s['betas'][1]) - 2.5e-6) < 1e-6 and abs(float(s['betas'][1000]) - 0.75003) < 1e-4


def test_encoding_matches_reference():
    g = load_golden('encoding')
    for nm, (vmax, N, D) in dict(pos=(256, 256, 256), chain=(4, 1, 64), t1000=(1001, 1000, 512), t100=(101, 100, 512)).items():
        e = O.sinusoidal_encoding(torch.arange(vmax, dtype=torch.int32), N, D)
        assert torch.equal(e[t(g[nm + '_rows'])], t(g[nm + '_vals']))
        assert abs(float(e.double().sum()) - float(g[nm + '_sum'])) < 1e-6


def test_frenet_matches_reference():
    g = load_golden('geometry')
    r = O.compute_frenet_frames(t(g['frenet_coords']), t(g['frenet_chains']), t(g['frenet_mask']))
    assert torch.equal(r, t(g['frenet_rots']))
    # improper frames (det = -1) inside a chain, identity on padding
    assert torch.allclose(torch.linalg.det(r[0]), -torch.ones(24), atol=1e-4)
    assert torch.equal(r[3, 5:], torch.eye(3).expand(19, 3, 3))


def test_quaternions_match_reference():
    g = load_golden('geometry')
    assert torch.allclose(O.quat_to_rot(t(g['q2r_q'])), t(g['q2r_r']), atol=1e-6)
    q_ref = t(g['r2q_q'])
    q = O.apply_sign_codes(O.rot_to_quat_closed(t(g['r2q_r'])), t(g['r2q_codes']))
    assert (q - q_ref).abs().max() < 2e-4          # fp32 eigh itself is only ~4e-5 accurate
    assert torch.equal(O.quat_sign_codes(q_ref), t(g['r2q_codes']))


@pytest.mark.parametrize('case', CALL_CASES)
def test_denoiser_call_matches_reference(case, base_weights):
    g = load_golden('call_' + case)
    f = golden_features(g)
    B, N = f['residue_mask'].shape
    ts = torch.full((B,), int(g['timestep']), dtype=torch.int32)
    taps = {}
    out = O.denoiser_forward(base_weights, dict(O.BASE_DIMS), t(g['rots']), t(g['trans']), ts, f, 'closed',
                             t(g['quat_codes']), taps)
    m = f['residue_mask'].unsqueeze(-1).float()
    zref = t(g['z'])
    assert ((out['z'] - zref) * m).abs().max() <= 1e-4 * max(1.0, float(zref.abs().max()))
    assert (out['s'] - t(g['s'])).abs().max() < 1e-5
    idx = t(g['p_idx']).long()
    for key, val in (('p_final_samples', out['p']), ('p_init_samples', taps['p_init']),
                     ('p_layer0_samples', taps['p_after_layer0'])):
        got = val[idx[:, 0], idx[:, 1], idx[:, 2]]
        assert (got - t(g[key])).abs().max() <= 2e-4 * max(1.0, float(np.abs(g[key]).max())), key
    assert ((out['s_final'] - t(g['states'])[1]) * m).abs().max() < 2e-3
    # eigh mode reproduces the reference's own call up to eigh's sign choice on this machine
    out_e = O.denoiser_forward(base_weights, dict(O.BASE_DIMS), t(g['rots']), t(g['trans']), ts, f, 'eigh')
    assert torch.isfinite(out_e['z']).all()


def test_trajectory_matches_reference(base_weights):
    """Config 1: N=50, T=100, B=1, scale 0.6 with the reference's own noise and eigh signs."""
    g = load_golden('trajectory_n50_t100')
    dims = dict(O.BASE_DIMS, n_timestep=100)
    f = O.empty_features([50])
    final, _, rec = O.sample_loop(base_weights, dims, f, t(g['noise']), float(g['scale']), 'closed', t(g['quat_codes']),
                                  record_every=10)
    ref = t(g['final'])
    rms = float(ref.pow(2).mean().sqrt())
    assert (final - ref).abs().max() <= 1e-4 * rms
    assert (torch.stack(rec) - t(g['every10'])).abs().max() <= 1e-4 * rms


def _train_features(g):
    f = O.empty_features([int(x) for x in g['lengths']])
    for k in ('residue_mask', 'chain_index', 'residue_index', 'fixed_sequence_mask', 'num_residues'):
        f[k] = t(g[k])
    f['atom_positions'] = t(g['atom_positions'])
    return f


def test_training_step_ends_match_reference_golden():
    """The two ends of Genie.training_step around the denoiser (diffusion/genie.py:77-105): forward noising + frames and
    the loss with its gradient, against what the reference's own get_betas / compute_frenet_frames / mse + autograd gave."""
    g = load_golden('train_ends_n24_b4')
    f = _train_features(g)
    sched = O.training_schedule(1000)
    s = t(g['s'])
    assert torch.equal(sched['sqrt_alphas_cumprod'][s], t(g['sqrt_alphas_cumprod_s']))
    assert torch.equal(sched['sqrt_one_minus_alphas_cumprod'][s], t(g['sqrt_one_minus_alphas_cumprod_s']))
    tr, ro = O.q_sample(f['atom_positions'], s, t(g['z']), f['chain_index'], f['residue_mask'], sched)
    assert torch.equal(tr, t(g['trans_s'])) and (ro - t(g['rots_s'])).abs().max() < 1e-6
    zp = t(g['z_pred']).clone().requires_grad_(True)
    lo = O.training_loss(zp, t(g['z']), f, float(g['condition_loss_weight']))
    lo['weighted_loss'].backward()
    assert abs(float(lo['weighted_loss'].detach()) - float(g['weighted_loss'])) < 1e-7
    assert abs(float(lo['unweighted_loss'].detach()) - float(g['unweighted_loss'])) < 1e-7
    assert (lo['condition_losses'].detach() - t(g['condition_losses'])).abs().max() < 1e-5
    assert (zp.grad - t(g['grad_z_pred'])).abs().max() < 1e-8


def test_training_gradients_match_reference_golden(base_weights):
    """d weighted_loss / d parameter through the whole denoiser (eval mode): the oracle under torch autograd against the
    gradients the reference Denoiser's own autograd produced (tests/golden/train_grads_n16_b2.npz, all 396 tensors:
    largest magnitude, norm and the first 8 entries of each).  This is what the backward kernels of the training row
    will be checked with."""
    g = load_golden('train_grads_n16_b2')
    f = _train_features(g)
    sd = {k: v.clone().requires_grad_(True) for k, v in base_weights.items()}
    o = O.denoiser_forward(sd, O.BASE_DIMS, t(g['rots_s']), t(g['trans_s']), t(g['s']).int(), f, 'closed', t(g['quat_codes']))
    assert (o['z'].detach() - t(g['z_pred'])).abs().max() < 1e-4
    lo = O.training_loss(o['z'], t(g['z']), f, float(g['condition_loss_weight']))['weighted_loss']
    assert abs(float(lo.detach()) - float(g['loss'])) < 1e-5
    lo.backward()
    keys = [str(k) for k in g['keys']]
    assert keys == list(sd.keys())
    for i, k in enumerate(keys):
        gr = sd[k].grad
        scale = max(float(g['grad_abs_max'][i]), 1e-6)
        assert abs(float(gr.abs().max()) - float(g['grad_abs_max'][i])) <= 5e-3 * scale, k
        assert abs(float(gr.norm()) - float(g['grad_norm'][i])) <= 5e-3 * max(float(g['grad_norm'][i]), 1e-6), k
        n = min(8, gr.numel())
        assert (gr.reshape(-1)[:n] - t(g['grad_probe'][i][:n])).abs().max() <= 5e-3 * scale, k
