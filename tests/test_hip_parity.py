"""Parity of the HIP path (through the C ABI, libgenie_hip.so) with the oracle
and with the fixtures generated from the real reference.  GPU only.

Tolerances (SURVEY.md 8c / BASELINE.md 4, fp32 against fp32):
  single call : |dz| <= 1e-4 * max(1, |z|_inf)          (valid residues)
  trajectory  : max|dCa| <= 1e-4 * coordinate RMS       (N=50, T=100, reference noise and eigh signs)
"""
import numpy as np
import pytest
import torch

from conftest import CALL_CASES, golden_features, load_golden
from oracle import genie_oracle as O

pytestmark = pytest.mark.gpu


def t(x):
    return torch.from_numpy(np.asarray(x))


def mdiff(a, b):
    return float((a.detach().cpu().float() - b.detach().cpu().float()).abs().max())


def test_native_library_is_loaded():
    from genie2_amd import capi
    capi.load_library()
    assert any('libgenie_hip.so' in line for line in open('/proc/self/maps'))


# --------------------------------------------------------------- geometry
def test_frenet_matches_reference_golden(base_engine):
    g = load_golden('geometry')
    f = O.empty_features([24] * 4)
    f['residue_mask'] = t(g['frenet_mask'])
    f['chain_index'] = t(g['frenet_chains'])
    base_engine.bind_features(f)
    r = base_engine.frenet(t(g['frenet_coords']))
    assert mdiff(r, t(g['frenet_rots'])) < 2e-6


@pytest.mark.parametrize('lengths,chains', [([40, 3, 2, 17], None), ([30], [[1, 1, 10, 1, 17]]), ([9, 9], [[3, 3, 3], [8, 1]])])
def test_frenet_edge_cases_match_oracle(base_engine, lengths, chains):
    """short structures, single-residue chains, consecutive chain ends"""
    f = O.empty_features(lengths, chains_per_sample=chains)
    B, N = f['residue_mask'].shape
    if chains and chains[-1][-1] == 1 and lengths[-1] == N:
        pytest.skip('reference raises IndexError when the last residue of a full row is its own chain')
    x = torch.randn(B, N, 3, generator=torch.Generator().manual_seed(1)) * 3
    base_engine.bind_features(f)
    ref = O.compute_frenet_frames(x, f['chain_index'], f['residue_mask'])
    assert mdiff(base_engine.frenet(x), ref) < 2e-6


def test_p_sample_matches_oracle(base_engine):
    f = O.empty_features([33, 21])
    g = torch.Generator().manual_seed(5)
    x, z, e = (torch.randn(2, 33, 3, generator=g) * s for s in (5.0, 1.0, 1.0))
    base_engine.bind_features(f)
    sched = O.setup_schedule(1000)
    fr = O.prepare_features(f)
    for step, eps in ((1000, e), (500, e), (2, e), (1, None)):
        nx, nr = O.p_sample_step(sched, step, 0.6, x, z, eps, fr)
        xg = x.clone().cuda()
        rg = base_engine.p_sample(step, 0.6, xg, z.cuda(), eps.cuda() if eps is not None else None)
        assert mdiff(xg, nx) <= 2e-6 * max(1.0, float(nx.abs().max())), step
        assert mdiff(rg, nr) < 5e-6, step
        assert float(xg[1, 21:].abs().max()) == 0.0           # masked residues are zeroed


# --------------------------------------------------------------- single calls vs reference goldens
MATH_MODES = ['hx', 'f32']       # split-f16 MFMA (default) / exact f32 MFMA: same tolerances for both


def base_engine_weights(engine):
    """the synthetic weights the session engine was built with (conftest.base_weights)"""
    return engine._test_weights


@pytest.fixture
def math_engine(request, base_engine):
    base_engine.set_math(request.param)
    assert base_engine.math == request.param
    yield base_engine
    base_engine.set_math('hx')


@pytest.mark.parametrize('math_engine', MATH_MODES, indirect=True)
@pytest.mark.parametrize('case', CALL_CASES)
def test_denoiser_call_matches_reference_golden(case, math_engine):
    base_engine = math_engine
    g = load_golden('call_' + case)
    f = golden_features(g)
    B, N = f['residue_mask'].shape
    ts = torch.full((B,), int(g['timestep']), dtype=torch.int32)
    base_engine.bind_features(f)
    out = base_engine.denoise(t(g['trans']), t(g['rots']), ts, t(g['quat_codes']),
                              taps=('s', 'p', 's_final', 'rots_out', 'trans_out', 'p_init', 'p_layer0', 'states', 'p_trimul_out0',
                                    'ipa_cat0'))
    m = f['residue_mask'].unsqueeze(-1).float()
    zref = t(g['z'])

    def rel(x):
        return 1e-4 * max(1.0, float(np.abs(np.asarray(x)).max()))

    assert mdiff(out['z'].cpu() * m, zref * m) <= rel(g['z'])
    assert mdiff(out['s'], t(g['s'])) < 2e-5
    idx = t(g['p_idx']).long()

    def samp(x):
        return x.cpu()[idx[:, 0], idx[:, 1], idx[:, 2]]

    for key, tap in (('p_final_samples', 'p'), ('p_init_samples', 'p_init'), ('p_layer0_samples', 'p_layer0')):
        assert mdiff(samp(out[tap]), t(g[key])) <= rel(g[key]), key
    # the outgoing triangle multiplication's own output (module hook on net[0].tri_mul_out): p after it minus p before it
    upd = samp(out['p_trimul_out0']) - samp(out['p_init'])
    assert mdiff(upd, t(g['trimul_out0_samples'])) <= rel(g['p_init_samples']), 'trimul_out0'
    assert abs(float(out['p'].abs().mean()) - float(g['p_final_abs_mean'])) < 1e-4 * float(g['p_final_abs_mean'])
    # layer 0's IPA output (module hook on structure_net.net[0].ipa) = linear_out over the concatenation the kernels leave
    w = base_engine_weights(base_engine)
    ipa = torch.nn.functional.linear(out['ipa_cat0'].cpu().double(), w['structure_net.net.0.ipa.linear_out.weight'].double(),
                                     w['structure_net.net.0.ipa.linear_out.bias'].double()).float()
    assert mdiff(ipa * m, t(g['ipa_out0']) * m) <= rel(g['ipa_out0']), 'ipa_out0'
    # the golden keeps states[[1, -1]] of the reference's stack: after the first structure layer and after the last
    st = out['states'].cpu()
    assert torch.equal(st[0], out['s'].cpu()) and torch.equal(st[-1], out['s_final'].cpu())
    assert mdiff(st[1] * m, t(g['states'])[0] * m) <= rel(g['states'][0])
    assert mdiff(out['s_final'].cpu() * m, t(g['states'])[1] * m) <= rel(g['states'][1])
    m4 = m.unsqueeze(-1)
    assert mdiff(out['rots_out'].cpu() * m4, t(g['rots_out']) * m4) <= 1e-4
    assert mdiff(out['trans_out'].cpu() * m, t(g['trans_out']) * m) <= rel(g['trans_out'])


@pytest.mark.parametrize('math', MATH_MODES)
def test_trajectory_matches_reference_golden(base_weights, math):
    """Config 1 (N=50, T=100, batch 1, scale 0.6): the reference's own
    UnconditionalSampler._sample, same noise, eigh signs supplied."""
    from genie2_amd.engine import GenieEngine
    g = load_golden('trajectory_n50_t100')
    dims = dict(O.BASE_DIMS, n_timestep=100)
    eng = GenieEngine(dims, base_weights, 'cuda:0', math=math)
    eng.bind_features(O.empty_features([50]))
    final, _, rec = eng.sample_loop(t(g['noise']), float(g['scale']), quat_codes=t(g['quat_codes']), record=True)
    ref = t(g['final'])
    rms = float(ref.pow(2).mean().sqrt())
    assert mdiff(final, ref) <= 1e-4 * rms
    assert mdiff(rec.cpu()[9::10], t(g['every10'])) <= 1e-4 * rms
    # canonical-sign mode (no codes) is self-consistent with the oracle in the same mode
    f2, _, _ = eng.sample_loop(t(g['noise']), float(g['scale']), first_step=100, last_step=91)
    o2 = t(g['noise'])[0].clone()
    fr = O.prepare_features(O.empty_features([50]))
    r2 = O.compute_frenet_frames(o2, fr['chain_index'], fr['residue_mask'])
    sched = O.setup_schedule(100)
    for it, step in enumerate(range(100, 90, -1)):
        z = O.denoiser_forward(base_weights, dims, r2, o2, torch.full((1,), step, dtype=torch.int32), fr, 'closed')['z']
        o2, r2 = O.p_sample_step(sched, step, float(g['scale']), o2, z, t(g['noise'])[it + 1], fr)
    assert mdiff(f2, o2) <= 1e-4 * float(o2.pow(2).mean().sqrt())
    eng.close()


# --------------------------------------------------------------- HIP vs oracle on seeded inputs
def _case(name, g):
    if name == 'n70_b3_ragged':            # N not a multiple of 32 / 64, B = 3
        f = O.empty_features([70, 64, 33])
    elif name == 'n130_b1':                # spans three 64-tiles, NP = 160
        f = O.empty_features([130])
    elif name == 'n20_multichain_motif':
        f = O.empty_features([20, 18], chains_per_sample=[[8, 12], [18]])
        O.add_motif(f, 0, torch.randn(5, 3, generator=g) * 4, [2, 3, 4, 10, 11])
        O.add_motif(f, 1, torch.randn(4, 3, generator=g) * 4, [0, 1, 16, 17])
    elif name == 'n2_tiny':
        f = O.empty_features([3, 2])
    else:
        raise KeyError(name)
    return f


@pytest.mark.parametrize('name', ['n70_b3_ragged', 'n130_b1', 'n20_multichain_motif', 'n2_tiny'])
@pytest.mark.parametrize('rescale,math', [(1.0, 'hx'), (2.0, 'hx'), (1.0, 'f32')])
def test_denoiser_matches_oracle_small_dims(name, rescale, math):
    from genie2_amd.engine import GenieEngine
    dims = O.small_dims(rescale=rescale)
    sd = O.synthetic_state_dict(dims, seed=3)
    g = torch.Generator().manual_seed(9)
    f = _case(name, g)
    B, N = f['residue_mask'].shape
    trans = 2.5 * torch.randn(B, N, 3, generator=g)
    fr = O.prepare_features(f)
    rots = O.compute_frenet_frames(trans, fr['chain_index'], fr['residue_mask'])
    ts = torch.randint(1, dims['n_timestep'] + 1, (B,), generator=g).int()
    taps = {}
    ref = O.denoiser_forward(sd, dims, rots, trans, ts, f, 'closed', None, taps)
    eng = GenieEngine(dims, sd, 'cuda:0', math=math)
    eng.bind_features(f)
    out = eng.denoise(trans, rots, ts, None, taps=('s', 'p', 'p_init', 'states'))
    m = fr['residue_mask'].unsqueeze(-1).float()
    assert mdiff(out['s'], ref['s']) < 2e-5
    # 'states' (structure_net.py:236-243): the single representation entering the structure net and after each of its layers
    assert out['states'].shape == taps['states'].shape == (1 + dims['n_structure_layer'], B, N, dims['c_s'])
    assert mdiff(out['states'].cpu() * m, taps['states'] * m) <= 1e-4 * max(1.0, float(taps['states'].abs().max()))
    assert mdiff(out['p_init'], taps['p_init']) <= 1e-4 * max(1.0, float(taps['p_init'].abs().max()))
    assert mdiff(out['p'], ref['p']) <= 1e-4 * max(1.0, float(ref['p'].abs().max()))
    assert mdiff(out['z'].cpu() * m, ref['z'] * m) <= 1e-4 * max(1.0, float(ref['z'].abs().max()))
    eng.close()


# --------------------------------------------------------------- full size (BASELINE config 2 shape)
@pytest.mark.parametrize('scale', [1e-3, 1.0, 40.0])
def test_hx_operand_scaling_follows_the_weights(scale):
    """The split-f16 path derives its power-of-two operand scales from the weights at load time
    (hx.h): pair-stack weights 1000x smaller or 40x larger must neither flush nor overflow."""
    from genie2_amd.engine import GenieEngine
    dims = O.small_dims()
    sd = O.synthetic_state_dict(dims, seed=5)
    for k in sd:
        if k.startswith('pair_transform_net') and k.endswith('weight') and 'layer_norm' not in k and sd[k].dim() == 2:
            sd[k] = sd[k] * scale
    g = torch.Generator().manual_seed(11)
    f = O.empty_features([48, 40])
    trans = 2.5 * torch.randn(2, 48, 3, generator=g)
    fr = O.prepare_features(f)
    rots = O.compute_frenet_frames(trans, fr['chain_index'], fr['residue_mask'])
    ts = torch.tensor([60, 3], dtype=torch.int32)
    ref = O.denoiser_forward(sd, dims, rots, trans, ts, f, 'closed')
    assert torch.isfinite(ref['p']).all()
    eng = GenieEngine(dims, sd, 'cuda:0', math='hx')
    eng.bind_features(f)
    out = eng.denoise(trans, rots, ts, None, taps=('p',))
    m = fr['residue_mask'].unsqueeze(-1).float()
    assert mdiff(out['p'], ref['p']) <= 1e-4 * max(1.0, float(ref['p'].abs().max()))
    assert mdiff(out['z'].cpu() * m, ref['z'] * m) <= 1e-4 * max(1.0, float(ref['z'].abs().max()))
    eng.set_math('f32')
    z32 = eng.denoise(trans, rots, ts, None)['z'].cpu()
    assert mdiff(z32 * m, ref['z'] * m) <= 1e-4 * max(1.0, float(ref['z'].abs().max()))
    eng.close()


@pytest.mark.parametrize('math', MATH_MODES)
def test_reference_style_initialisation_with_zero_final_layers(math):
    """The reference initialises every 'final' Linear to zero and gate Linears to (0, 1) (primitives.py:96-160): whole weight
    matrices are exactly zero, which the load-time operand scales of the split-f16 path must survive (no inf / NaN)."""
    from genie2_amd.engine import GenieEngine
    dims = O.small_dims()
    sd = O.synthetic_state_dict(dims, seed=6)
    for k in sd:
        if k.endswith(('linear_z.weight', 'linear_z.bias', 'pair_transition.linear_2.weight', 'pair_transition.linear_2.bias',
                       'ipa.linear_out.weight', 'ipa.linear_out.bias', 'linear_3.weight', 'linear_3.bias')):
            sd[k] = torch.zeros_like(sd[k])
        elif k.endswith(('linear_g.weight', 'linear_a_g.weight', 'linear_b_g.weight')):
            sd[k] = torch.zeros_like(sd[k])
        elif k.endswith(('linear_g.bias', 'linear_a_g.bias', 'linear_b_g.bias')):
            sd[k] = torch.ones_like(sd[k])
    g = torch.Generator().manual_seed(2)
    f = O.empty_features([33, 20])
    trans = 2.5 * torch.randn(2, 33, 3, generator=g)
    fr = O.prepare_features(f)
    rots = O.compute_frenet_frames(trans, fr['chain_index'], fr['residue_mask'])
    ts = torch.tensor([17, 90], dtype=torch.int32)
    ref = O.denoiser_forward(sd, dims, rots, trans, ts, f, 'closed')
    eng = GenieEngine(dims, sd, 'cuda:0', math=math)
    eng.bind_features(f)
    out = eng.denoise(trans, rots, ts, None, taps=('p', 's_final'))
    m = fr['residue_mask'].unsqueeze(-1).float()
    assert torch.isfinite(out['p']).all() and torch.isfinite(out['z']).all()
    assert mdiff(out['p'], ref['p']) <= 1e-4 * max(1.0, float(ref['p'].abs().max()))
    assert mdiff(out['z'].cpu() * m, ref['z'] * m) <= 1e-4 * max(1.0, float(ref['z'].abs().max()))
    eng.close()


def test_oversized_batch_is_refused_not_wrapped(base_engine):
    """[B,N,N,128] f32 must stay below 2 GiB (32-bit buffer offsets in the pair kernels): asking for more raises
    instead of wrapping addresses."""
    from genie2_amd.capi import GenieError
    f = O.empty_features([256] * 64)
    with pytest.raises(GenieError, match='split the batch'):
        base_engine.bind_features(f)
    base_engine.bind_features(O.empty_features([16]))       # the handle stays usable


def test_full_size_n256_matches_oracle_and_batches_are_independent(base_engine, base_weights):
    """N=256 (the metric's length): batch entry 0 against the oracle directly
    (one structure is ~3 s of CPU), then size-independent properties at batch 8:
    every batch entry equals its own batch-1 run bit for bit, a permuted batch
    gives permuted outputs, translation of the input leaves z unchanged."""
    N, B = 256, 8
    g = torch.Generator().manual_seed(2)
    trans = 3.0 * torch.randn(B, N, 3, generator=g)
    f8 = O.empty_features([N] * B)
    fr = O.prepare_features(f8)
    rots = O.compute_frenet_frames(trans, fr['chain_index'], fr['residue_mask'])
    ts = torch.full((B,), 640, dtype=torch.int32)
    base_engine.bind_features(f8)
    z8 = base_engine.denoise(trans, rots, ts)['z'].cpu()
    assert torch.isfinite(z8).all()
    # determinism
    assert torch.equal(z8, base_engine.denoise(trans, rots, ts)['z'].cpu())
    # permutation of the batch
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4])
    assert torch.equal(z8[perm], base_engine.denoise(trans[perm], rots[perm], ts)['z'].cpu())
    # translation invariance (distances and frames do not change; fp32 rounding of the shifted coordinates does)
    shift = torch.tensor([7.0, -3.0, 11.0])
    zs = base_engine.denoise(trans + shift, rots, ts)['z'].cpu()
    assert mdiff(zs, z8) <= 1e-4 * max(1.0, float(z8.abs().max()))
    # batch 1 == entry 0 of batch 8, and both == oracle
    f1 = O.empty_features([N])
    base_engine.bind_features(f1)
    z1 = base_engine.denoise(trans[:1], rots[:1], ts[:1])['z'].cpu()
    assert torch.equal(z1[0], z8[0])
    ref = O.denoiser_forward(base_weights, dict(O.BASE_DIMS), rots[:1], trans[:1], ts[:1], f1, 'closed')['z']
    assert mdiff(z1, ref) <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_padding_does_not_change_valid_residues(base_engine):
    """a length-48 structure alone vs padded to 80 inside a ragged batch"""
    g = torch.Generator().manual_seed(4)
    x = 3.0 * torch.randn(1, 48, 3, generator=g)
    f_a = O.empty_features([48])
    base_engine.bind_features(f_a)
    ra = base_engine.frenet(x)
    za = base_engine.denoise(x, ra, torch.tensor([77], dtype=torch.int32))['z'].cpu()
    f_b = O.empty_features([48, 80])
    xb = torch.zeros(2, 80, 3)
    xb[0, :48] = x[0]
    xb[1] = torch.randn(80, 3, generator=g)
    xb[0, 48:] = torch.randn(32, 3, generator=g)          # garbage in the padding (the reference's initial noise is unmasked)
    base_engine.bind_features(f_b)
    rb = base_engine.frenet(xb)
    zb = base_engine.denoise(xb, rb, torch.tensor([77, 77], dtype=torch.int32))['z'].cpu()
    assert mdiff(zb[0, :48], za[0]) <= 1e-5 * max(1.0, float(za.abs().max()))


# --------------------------------------------------------------- the drop-in API
def test_sampler_api_end_to_end(tmp_path, base_weights):
    """genie.sampler.UnconditionalSampler over the HIP engine: same call
    sequence as the reference CLI, explicit noise, PDB files written."""
    from genie.config import Config
    from genie.sampler.unconditional import UnconditionalSampler
    from genie2_amd.diffusion import Genie
    cfg = Config()
    cfg.diffusion['n_timestep'] = 20
    model = Genie(cfg)
    model.model.load_state_dict(base_weights)
    model = model.eval().to('cuda:0')
    sampler = UnconditionalSampler(model)
    noise = torch.randn(20, 2, 30, 3, generator=torch.Generator().manual_seed(8))
    params = {'length': 30, 'scale': 0.6, 'num_samples': 2, 'outdir': str(tmp_path), 'prefix': '30', 'offset': 4,
              'noise': noise}
    sampler.sample(params)
    files = sorted(p.name for p in (tmp_path / 'pdbs').iterdir())
    assert files == ['30_4.pdb', '30_5.pdb']
    lines = (tmp_path / 'pdbs' / '30_4.pdb').read_text().splitlines()
    assert len(lines) == 30 and lines[0].startswith('ATOM      1  CA  ALA A   1')
    # the same through the oracle
    dims = dict(O.BASE_DIMS, n_timestep=20)
    ref, _, _ = O.sample_loop(base_weights, dims, O.empty_features([30, 30]), noise, 0.6, 'closed')
    got = sampler._sample(params)
    xyz = torch.tensor(np.stack([g['atom_positions'] for g in got]), dtype=torch.float32)
    assert mdiff(xyz, ref) <= 1e-4 * float(ref.pow(2).mean().sqrt())
    # default noise path: device generator, reference draw order
    torch.manual_seed(0)
    del params['noise']
    a = sampler._sample(params)
    torch.manual_seed(0)
    b = sampler._sample(params)
    assert np.array_equal(a[0]['atom_positions'], b[0]['atom_positions'])
    # Denoiser.forward seam: model.model(ts, timesteps, features)['z']
    from genie.utils.affine_utils import T
    from genie2_amd import features as F
    feats = F.convert_np_features_to_tensor(F.batchify_np_features([F.create_empty_np_features([30])]), 'cuda:0')
    x = noise[0, :1].cuda()
    rots = model.model.bind(feats).frenet(x)
    out = model.model(T(rots, x), torch.tensor([20], dtype=torch.int32), feats, outputs=('z', 's', 'p', 'states', 'ts'))
    assert out['z'].shape == (1, 30, 3) and out['p'].shape == (1, 30, 30, 128) and out['states'].shape == (9, 1, 30, 384)
    assert out['ts'].rots.shape == (1, 30, 3, 3)


def test_fused_structure_tail_matches_separate_launches(base_engine, monkeypatch):
    """k_struct_rows_hx (split-K output projection summed on load, LayerNorm, transition, LayerNorm, BackboneUpdate in
    one launch) against the seven separate launches it replaces (GENIE_NO_STRUCT_FUSE): same arithmetic up to summation
    order.  Ragged batch whose row count is not a multiple of the 32-row tile.  The other alternative forms the library keeps
    behind switches are run on the same inputs."""
    f = O.empty_features([45, 23, 38])
    B, N = f['residue_mask'].shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, N, 3, generator=g) * 4
    base_engine.set_math('hx')
    base_engine.bind_features(f)
    r = base_engine.frenet(x)
    ts = torch.tensor([900, 17, 333], dtype=torch.int32)
    monkeypatch.delenv('GENIE_NO_STRUCT_FUSE', raising=False)
    zf = base_engine.denoise(x, r, ts, None, taps=('s_final',))
    monkeypatch.setenv('GENIE_NO_STRUCT_FUSE', '1')
    zs = base_engine.denoise(x, r, ts, None, taps=('s_final',))
    monkeypatch.delenv('GENIE_NO_STRUCT_FUSE', raising=False)
    monkeypatch.setenv('GENIE_IPA_Q8', '1')          # eight queries per 1024-thread attention work-group (opt-in form)
    z8 = base_engine.denoise(x, r, ts, None, taps=('s_final',))
    monkeypatch.delenv('GENIE_IPA_Q8', raising=False)
    monkeypatch.setenv('GENIE_OUT_STREAMED', '1')    # TriMul output kernel with streamed weights, pair bias through the f32 kernel
    monkeypatch.setenv('GENIE_IPA_BIAS_F32', '1')
    zo = base_engine.denoise(x, r, ts, None, taps=('s_final',))
    monkeypatch.delenv('GENIE_OUT_STREAMED', raising=False)
    monkeypatch.delenv('GENIE_IPA_BIAS_F32', raising=False)
    m = f['residue_mask'].bool()           # padded rows attend through an all -1e5 bias: not comparable beyond summation order
    for k in ('z', 's_final'):
        a, b = zf[k].cpu()[m], zs[k].cpu()[m]
        assert torch.isfinite(a).all()
        assert mdiff(a, b) <= 4e-6 * max(1.0, float(b.abs().max())), k
        assert mdiff(z8[k].cpu()[m], a) <= 4e-6 * max(1.0, float(b.abs().max())), k
        assert mdiff(zo[k].cpu()[m], a) <= 2e-5 * max(1.0, float(b.abs().max())), k


def test_training_step_ends_match_reference_golden(base_engine):
    """genie_q_sample / genie_training_loss (diffusion/genie.py:77-105) against the reference's own results
    (tests/golden/train_ends_n24_b4.npz: its get_betas, compute_frenet_frames, mse and autograd)."""
    g = load_golden('train_ends_n24_b4')
    f = O.empty_features([int(x) for x in g['lengths']])
    for k in ('residue_mask', 'chain_index', 'residue_index', 'fixed_sequence_mask', 'num_residues'):
        f[k] = t(g[k])
    base_engine.bind_features(f)
    tr, ro = base_engine.q_sample(t(g['atom_positions']), t(g['z']), t(g['sqrt_alphas_cumprod_s']), t(g['sqrt_one_minus_alphas_cumprod_s']))
    assert mdiff(tr, t(g['trans_s'])) <= 2e-6 * float(np.abs(g['trans_s']).max())
    assert mdiff(ro, t(g['rots_s'])) < 2e-6
    out = base_engine.training_loss(t(g['z_pred']), t(g['z']), float(g['condition_loss_weight']))
    assert abs(float(out['weighted_loss']) - float(g['weighted_loss'])) <= 2e-6 * float(g['weighted_loss'])
    assert abs(float(out['unweighted_loss']) - float(g['unweighted_loss'])) <= 2e-6 * float(g['unweighted_loss'])
    assert mdiff(out['condition_losses'], t(g['condition_losses'])) <= 1e-5
    assert mdiff(out['infill_losses'], t(g['infill_losses'])) <= 1e-5
    assert mdiff(out['grad'], t(g['grad_z_pred'])) <= 2e-6 * float(np.abs(g['grad_z_pred']).max())
    # no gradient requested: same losses
    out2 = base_engine.training_loss(t(g['z_pred']), t(g['z']), float(g['condition_loss_weight']), grad=False)
    assert float(out2['weighted_loss']) == float(out['weighted_loss']) and 'grad' not in out2


def test_adam_step_matches_torch_adam():
    """genie_adam_step against torch.optim.Adam with the reference's settings (ddpm.py:73-77: lr from the config,
    default betas / eps, no weight decay), three consecutive updates of a flat blob."""
    from genie2_amd.engine import adam_step
    g0 = torch.Generator().manual_seed(3)
    n = 100003
    p_ref = torch.nn.Parameter(torch.randn(n, generator=g0))
    opt = torch.optim.Adam([p_ref], lr=1e-4)
    p = p_ref.detach().clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in (1, 2, 3):
        grad = torch.randn(n, generator=g0) * (10.0 ** (step - 2))
        p_ref.grad = grad.clone()
        opt.step()
        adam_step(p, grad.cuda(), m, v, 1e-4, step)
        assert mdiff(p, p_ref.detach()) <= 2e-7, step
    st = opt.state[p_ref]
    assert mdiff(m, st['exp_avg']) <= 1e-6 * float(st['exp_avg'].abs().max())
    assert mdiff(v, st['exp_avg_sq']) <= 1e-6 * float(st['exp_avg_sq'].abs().max())


def test_fused_pair_chains_match_separate_launches(base_engine, monkeypatch):
    """The fused row-local chains (pair_fused_kernels.hip: TriMul output -> [transition ->] next projections in one kernel, z read
    and written once per chain) against the seven separate launches per block they replace (GENIE_NO_PAIR_FUSE): same arithmetic
    up to the order of the f32 sums.  Ragged batch, N not a multiple of 32, motif conditioning live."""
    g = torch.Generator().manual_seed(21)
    f = O.empty_features([70, 41, 64])
    O.add_motif(f, 1, torch.randn(6, 3, generator=g) * 4, [3, 4, 5, 20, 21, 22])
    B, N = f['residue_mask'].shape
    x = torch.randn(B, N, 3, generator=g) * 4
    base_engine.set_math('hx')
    base_engine.bind_features(f)
    r = base_engine.frenet(x)
    ts = torch.tensor([900, 17, 333], dtype=torch.int32)
    taps = ('p', 'p_layer0', 'p_trimul_out0', 's_final')
    monkeypatch.delenv('GENIE_NO_PAIR_FUSE', raising=False)
    a = base_engine.denoise(x, r, ts, None, taps=taps)
    monkeypatch.setenv('GENIE_NO_PAIR_FUSE', '1')
    b = base_engine.denoise(x, r, ts, None, taps=taps)
    monkeypatch.delenv('GENIE_NO_PAIR_FUSE', raising=False)
    pm = (f['residue_mask'][:, :, None] * f['residue_mask'][:, None, :]).bool()
    for k in ('p_trimul_out0', 'p_layer0', 'p'):
        u, w = a[k].cpu()[pm], b[k].cpu()[pm]
        assert torch.isfinite(u).all(), k
        assert mdiff(u, w) <= 2e-5 * max(1.0, float(w.abs().max())), k
    m = f['residue_mask'].bool()
    assert mdiff(a['z'].cpu()[m], b['z'].cpu()[m]) <= 2e-5 * max(1.0, float(b['z'].abs().max()))
    # the end-of-block mask zeroes padded pairs in both forms
    assert float(a['p'].cpu()[~pm].abs().max()) == 0.0


@pytest.mark.parametrize('B,N', [(8, 256), (32, 240)])
def test_fused_pair_chains_match_separate_launches_everywhere_at_full_size(base_engine, monkeypatch, B, N):
    """The fused chains store z through an LDS staging area with `buffer_store_dwordx4` (fz_store_half), whose data registers need
    two wait states before they are rewritten -- a regression shows as a few rows of the NEXT piece inside a tile, every few tiles.
    So: EVERY element of p (and of the first block's taps) of the fused path against the separate launches (GENIE_NO_PAIR_FUSE, which
    shares none of that store code), at the headline size (N = 256, batch 8) and at config 3's batch 32 with N = 240 (not a multiple
    of 32: partial tiles), compared on the device."""
    f = O.empty_features([N] * B)
    g = torch.Generator().manual_seed(100 + B)
    x = torch.randn(B, N, 3, generator=g) * 6
    base_engine.set_math('hx')
    base_engine.bind_features(f)
    r = base_engine.frenet(x)
    ts = torch.randint(1, 1001, (B,), generator=g).int()
    taps = ('p', 'p_layer0', 'p_trimul_out0')
    monkeypatch.delenv('GENIE_NO_PAIR_FUSE', raising=False)
    a = base_engine.denoise(x, r, ts, None, taps=taps)
    a2 = base_engine.denoise(x, r, ts, None, taps=('p',))       # the other tile direction (launch parity alternates)
    monkeypatch.setenv('GENIE_NO_PAIR_FUSE', '1')
    b = base_engine.denoise(x, r, ts, None, taps=taps)
    monkeypatch.delenv('GENIE_NO_PAIR_FUSE', raising=False)
    for k in taps:
        assert torch.isfinite(a[k]).all(), k
        scale = max(1.0, float(b[k].abs().max()))
        d = (a[k] - b[k]).abs()
        worst = float(d.max())
        assert worst <= 2e-5 * scale, (k, worst, scale, int((d > 2e-5 * scale).sum()))
    assert float((a2['p'] - b['p']).abs().max()) <= 2e-5 * max(1.0, float(b['p'].abs().max()))
    assert float((a['z'] - b['z']).abs().max()) <= 2e-5 * max(1.0, float(b['z'].abs().max()))
    del a, a2, b
    torch.cuda.empty_cache()


@pytest.mark.parametrize('math', MATH_MODES)
def test_heavy_tailed_pair_weights_and_large_layernorm_gains(math):
    """What trained checkpoints look like and the synthetic recipe does not: every pair-stack matrix carries a few entries 50x its
    typical size, LayerNorm gains reach 10.  The split-f16 path takes its operand scales from bounds over the weights (hx_bound,
    Cauchy-Schwarz over the LN-folded rows) -- outliers inflate those bounds, which must cost neither range nor the 1e-4 bar."""
    from genie2_amd.engine import GenieEngine
    dims = O.small_dims()
    sd = O.synthetic_state_dict(dims, seed=15)
    g = torch.Generator().manual_seed(77)
    for k in sorted(sd):
        if not k.startswith('pair_transform_net'):
            continue
        if 'layer_norm' in k and k.endswith('weight'):
            sd[k] = sd[k] * torch.exp(torch.rand(sd[k].shape, generator=g) * 2.302585)            # gains spread over 1 .. 10
        elif k.endswith('weight') and sd[k].dim() == 2:
            w = sd[k].clone()
            n_out = max(3, w.numel() // 2000)
            idx = torch.randint(0, w.numel(), (n_out,), generator=g)
            w.view(-1)[idx] = w.view(-1)[idx] * 50.0
            # keep the layer's output scale where it was, so that the test stresses ranges, not the network's conditioning
            sd[k] = w * (sd[k].norm() / w.norm())
    f = O.empty_features([48, 40])
    trans = 2.5 * torch.randn(2, 48, 3, generator=g)
    fr = O.prepare_features(f)
    rots = O.compute_frenet_frames(trans, fr['chain_index'], fr['residue_mask'])
    ts = torch.tensor([60, 3], dtype=torch.int32)
    ref = O.denoiser_forward(sd, dims, rots, trans, ts, f, 'closed')
    assert torch.isfinite(ref['p']).all()
    eng = GenieEngine(dims, sd, 'cuda:0', math=math)
    eng.bind_features(f)
    out = eng.denoise(trans, rots, ts, None, taps=('p',))
    m = fr['residue_mask'].unsqueeze(-1).float()
    print('max|p|', float(ref['p'].abs().max()), 'dp', mdiff(out['p'], ref['p']), 'dz', mdiff(out['z'].cpu() * m, ref['z'] * m))
    assert mdiff(out['p'], ref['p']) <= 1e-4 * max(1.0, float(ref['p'].abs().max()))
    assert mdiff(out['z'].cpu() * m, ref['z'] * m) <= 1e-4 * max(1.0, float(ref['z'].abs().max()))
    eng.close()


def test_two_stream_structure_net_is_bit_identical(base_engine, monkeypatch):
    """From 1024 rows up the structure layers of the two halves of a batch run on two streams (genie_api.hip denoise_internal: every
    kernel of a layer takes a batch range; fork / join by events each call): z, all nine states, the first layer's attention output and
    the frames must equal the single-stream order (GENIE_NO_STRUCT_SPLIT=1) BIT FOR BIT, here on an odd, ragged batch (5 structures,
    halves of 2 and 3) and over consecutive reverse-loop steps (the events are reused every step)."""
    f = O.empty_features([256, 201, 256, 180, 233])
    g = torch.Generator().manual_seed(9)
    B, N = f['residue_mask'].shape
    x = torch.randn(B, N, 3, generator=g) * 6
    base_engine.set_math('hx')
    base_engine.bind_features(f)
    r = base_engine.frenet(x)
    ts = torch.randint(1, 1001, (B,), generator=g).int()
    T = base_engine.dims['n_timestep']
    noise = torch.randn(T, B, N, 3, generator=g)
    taps = ('states', 's_final', 'ipa_cat0', 'rots_out', 'trans_out')
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv('GENIE_NO_STRUCT_SPLIT', '1')
        else:
            monkeypatch.delenv('GENIE_NO_STRUCT_SPLIT', raising=False)
        out = base_engine.denoise(x, r, ts, None, taps=taps)
        tr, ro, _ = base_engine.sample_loop(noise, 0.6, first_step=T, last_step=T - 5)
        out['traj'], out['traj_rots'] = tr.clone(), ro.clone()
        res.append(out)
    monkeypatch.delenv('GENIE_NO_STRUCT_SPLIT', raising=False)
    for k in res[0]:
        assert torch.isfinite(res[0][k]).all(), k
        assert torch.equal(res[0][k], res[1][k]), k


@pytest.mark.gpu
def test_eight_query_attention_form_matches_the_default(base_engine, monkeypatch):
    """GENIE_IPA_Q8 selects the attention kernel's 1024-thread form (eight queries per work-group, one head per wave in the a v / a v_pts
    phase, no partial sums through LDS): same arithmetic in a different summation split, so the step's outputs must agree with the
    four-query default to f32 rounding -- on a ragged batch whose last query group is partial for both forms."""
    f = O.empty_features([250, 203, 117])
    g = torch.Generator().manual_seed(21)
    B, N = f['residue_mask'].shape
    x = torch.randn(B, N, 3, generator=g) * 6
    base_engine.set_math('hx')
    base_engine.bind_features(f)
    r = base_engine.frenet(x)
    ts = torch.randint(1, 1001, (B,), generator=g).int()
    taps = ('states', 'ipa_cat0', 'trans_out')
    monkeypatch.delenv('GENIE_IPA_Q8', raising=False)
    ref = base_engine.denoise(x, r, ts, None, taps=taps)
    monkeypatch.setenv('GENIE_IPA_Q8', '1')
    try:
        out = base_engine.denoise(x, r, ts, None, taps=taps)
    finally:
        monkeypatch.delenv('GENIE_IPA_Q8', raising=False)
    m = f['residue_mask'].float().to(ref['z'].device)
    for k in ref:
        assert torch.isfinite(out[k]).all(), k
        w = m.reshape((1,) * (ref[k].dim() - 3) + (B, N, 1))                 # padded residues carry no meaning
        scale = max(1.0, float((ref[k] * w).abs().max()))
        d = float(((out[k] - ref[k]) * w).abs().max())
        assert d <= 2e-5 * scale, (k, d, scale)
