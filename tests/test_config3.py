"""BASELINE config 3 on the GPU: the unconditional length sweep at batch 32 through the CLI runner
(genie/sample_unconditional.py:33-120, utils/multiprocessor.py:73-100), and the full-size cases of config 2.

At batch 32, N=256 a pair tensor is 1.07 GB (the kernels' 32-bit offsets end at 2 GiB) and the workspace ~6.5 GB:
entry 0 is compared with the oracle directly, the last entry with its own batch-1 run bit for bit.
"""
import os

import numpy as np
import pytest
import torch

from oracle import genie_oracle as O

pytestmark = pytest.mark.gpu

MATH_MODES = ['hx', 'f32']


def mdiff(a, b):
    return float((a.detach().cpu().float() - b.detach().cpu().float()).abs().max())


def rel(x, tol=1e-4):
    return tol * max(1.0, float(x.abs().max()))


_ORACLE_CACHE = {}


def oracle_entry(base_weights, N, seed):
    """inputs of a batch-32 run (seeded) and the oracle's result for entry 0"""
    if N not in _ORACLE_CACHE:
        g = torch.Generator().manual_seed(seed)
        trans = 3.0 * torch.randn(32, N, 3, generator=g)
        f1 = O.empty_features([N])
        fr = O.prepare_features(f1)
        rots0 = O.compute_frenet_frames(trans[:1], fr['chain_index'], fr['residue_mask'])
        ts = torch.full((32,), 411, dtype=torch.int32)
        taps = {}
        ref = O.denoiser_forward(base_weights, dict(O.BASE_DIMS), rots0, trans[:1], ts[:1], f1, 'closed', None, taps)
        idx = torch.stack([torch.randint(0, N, (512,), generator=g), torch.randint(0, N, (512,), generator=g)], 1)
        _ORACLE_CACHE[N] = dict(trans=trans, ts=ts, z=ref['z'], p=ref['p'][0][idx[:, 0], idx[:, 1]].clone(), idx=idx,
                                states=taps['states'].clone())
    return _ORACLE_CACHE[N]


@pytest.mark.parametrize('math', MATH_MODES)
@pytest.mark.parametrize('N', [256, 240, 112])
def test_batch32_sweep_lengths(base_engine, base_weights, N, math):
    """config 3's batch size at its largest length and at two sweep lengths that are not multiples of 32."""
    c = oracle_entry(base_weights, N, 100 + N)
    B = 32
    base_engine.set_math(math)
    try:
        base_engine.bind_features(O.empty_features([N] * B))
        rots = base_engine.frenet(c['trans'])
        out = base_engine.denoise(c['trans'], rots, c['ts'], None, taps=('p', 'states'))
        z = out['z'].cpu()
        assert torch.isfinite(z).all()
        # entry 0 against the oracle: z, 512 sampled rows of p, all nine states
        assert mdiff(z[:1], c['z']) <= rel(c['z'])
        p0 = out['p'][0].cpu()[c['idx'][:, 0], c['idx'][:, 1]]
        assert mdiff(p0, c['p']) <= rel(c['p'])
        assert mdiff(out['states'][:, :1], c['states']) <= rel(c['states'])
        # the last entry (highest addresses of every [B,N,N,128] tensor) equals its own batch-1 run bit for bit
        p_last = out['p'][B - 1].cpu()
        del out
        base_engine.bind_features(O.empty_features([N]))
        o1 = base_engine.denoise(c['trans'][B - 1:], rots[B - 1:], c['ts'][:1], None, taps=('p',))
        assert torch.equal(o1['z'].cpu()[0], z[B - 1])
        assert torch.equal(o1['p'].cpu()[0], p_last)
    finally:
        base_engine.set_math('hx')


@pytest.mark.parametrize('math', MATH_MODES)
def test_full_size_n256_b8_taps_match_oracle(base_engine, base_weights, math):
    """config 2's shape (N=256, batch 8): p (sampled rows) and all states of entry 0 against the oracle, in both arithmetics."""
    c = oracle_entry(base_weights, 256, 356)
    base_engine.set_math(math)
    try:
        base_engine.bind_features(O.empty_features([256] * 8))
        rots = base_engine.frenet(c['trans'][:8])
        out = base_engine.denoise(c['trans'][:8], rots, c['ts'][:8], None, taps=('p', 'states'))
        assert mdiff(out['z'][:1], c['z']) <= rel(c['z'])
        assert mdiff(out['p'][0].cpu()[c['idx'][:, 0], c['idx'][:, 1]], c['p']) <= rel(c['p'])
        assert mdiff(out['states'][:, :1], c['states']) <= rel(c['states'])
    finally:
        base_engine.set_math('hx')


def test_full_reverse_loop_t1000_n256_b8_both_arithmetics(base_engine):
    """The metric's whole job: T=1000 reverse steps at N=256, batch 8, same noise in both arithmetics.  Finite throughout, and
    the hx trajectory stays within 1e-4 x coordinate RMS of the exact-f32 one at steps 100, 500 and 1000."""
    T, B, N = 1000, 8, 256
    noise = torch.randn(T, B, N, 3, generator=torch.Generator().manual_seed(42)).cuda()
    base_engine.bind_features(O.empty_features([N] * B))
    marks = [99, 499, 999]
    rec = {}
    try:
        for math in MATH_MODES:
            base_engine.set_math(math)
            _, _, r = base_engine.sample_loop(noise, 0.6, record=True)
            assert torch.isfinite(r).all(), math
            rec[math] = r[marks].cpu()
            del r
    finally:
        base_engine.set_math('hx')
    for k, it in enumerate(marks):
        rms = float(rec['f32'][k].pow(2).mean().sqrt())
        d = mdiff(rec['hx'][k], rec['f32'][k])
        print(f'iteration {it + 1}: hx-vs-f32 max|dCa| = {d:.3e}, coordinate RMS = {rms:.1f}, ratio {d / rms:.2e}')
        assert d <= 1e-4 * rms, (it, d, rms)


def _write_model_dir(root, name, epoch, base_weights, n_timestep):
    from genie2_amd.config import Config
    from genie2_amd.diffusion import Genie, save_checkpoint
    d = os.path.join(root, name)
    os.makedirs(d)
    with open(os.path.join(d, 'configuration'), 'w') as fh:
        fh.write('name {}\nnumTimesteps {}\n'.format(name, n_timestep))
    g = Genie(Config(os.path.join(d, 'configuration')))
    g.model.load_state_dict(base_weights)
    save_checkpoint(g, os.path.join(d, 'checkpoints', 'epoch.{}.ckpt'.format(epoch)), epoch=epoch)


def test_unconditional_runner_end_to_end(tmp_path, base_weights):
    """UnconditionalRunner().run with num_devices=1 (sample_unconditional.py:33-120): checkpoint written in the reference's
    layout -> load_pretrained_model -> one sampler per device -> batches of at most batch_size -> outdir/pdbs/{length}_{idx}.pdb.
    The files equal a direct sampler.sample on the same draws of the device generator."""
    from genie2_amd.diffusion import load_pretrained_model
    from genie2_amd.sample_unconditional import UnconditionalRunner, build_parser
    from genie2_amd.sampler import UnconditionalSampler
    root = str(tmp_path / 'results')
    _write_model_dir(root, 'base', 7, base_weights, n_timestep=12)
    out = str(tmp_path / 'out')
    args = build_parser().parse_args(['--name', 'base', '--epoch', '7', '--rootdir', root, '--scale', '0.6', '--outdir', out,
                                      '--min_length', '50', '--max_length', '82', '--length_step', '16', '--batch_size', '4',
                                      '--num_samples', '5', '--num_devices', '1'])
    torch.manual_seed(1234)
    UnconditionalRunner().run(vars(args), args.num_devices, args.sequential_order)
    files = sorted(os.listdir(os.path.join(out, 'pdbs')))
    assert files == sorted('{}_{}.pdb'.format(n, i) for n in (82, 66, 50) for i in range(5))
    for n in (82, 66, 50):
        lines = open(os.path.join(out, 'pdbs', '{}_3.pdb'.format(n))).read().splitlines()
        assert len(lines) == n and lines[-1].startswith('ATOM') and 'CA' in lines[-1]
    # the same through the sampler directly: lengths downward from max_length, batches of 4 then 1, same generator state
    model = load_pretrained_model(root, 'base', 7).eval().to('cuda:0')
    sampler = UnconditionalSampler(model)
    out2 = str(tmp_path / 'out2')
    torch.manual_seed(1234)
    for n in (82, 66, 50):
        for batch, offset in ((4, 0), (1, 4)):
            sampler.sample({'length': n, 'scale': 0.6, 'num_samples': batch, 'outdir': out2, 'prefix': str(n), 'offset': offset})
    for name in files:
        assert open(os.path.join(out, 'pdbs', name), 'rb').read() == open(os.path.join(out2, 'pdbs', name), 'rb').read(), name
    # --resume (not in the reference CLI): an interrupted run is continued, finished batches are not sampled again
    victim = os.path.join(out, 'pdbs', '66_4.pdb')
    os.unlink(victim)
    stamps = {n: os.path.getmtime(os.path.join(out, 'pdbs', n)) for n in files if n != '66_4.pdb'}
    args2 = build_parser().parse_args(['--name', 'base', '--epoch', '7', '--rootdir', root, '--scale', '0.6', '--outdir', out,
                                       '--min_length', '50', '--max_length', '82', '--length_step', '16', '--batch_size', '4',
                                       '--num_samples', '5', '--num_devices', '1', '--resume'])
    UnconditionalRunner().run(vars(args2), 1, False)
    assert os.path.exists(victim)
    assert all(os.path.getmtime(os.path.join(out, 'pdbs', n)) == t for n, t in stamps.items())


def test_rebinding_same_shape_cpu_features_is_not_skipped(base_weights):
    """Denoiser.bind must not mistake a new batch for the bound one when a freed tensor's address is reused: two CPU feature
    dicts of the same shape but different masks, the first deleted before the second call."""
    from genie2_amd.affine import T
    from genie2_amd.config import Config
    from genie2_amd.diffusion import Genie
    from genie2_amd.engine import GenieEngine
    cfg = Config()
    model = Genie(cfg)
    model.model.load_state_dict(base_weights)
    model = model.eval().to('cuda:0')
    g = torch.Generator().manual_seed(5)
    x = 3.0 * torch.randn(2, 40, 3, generator=g)
    ts = torch.tensor([300, 300], dtype=torch.int32)

    def feats(lengths):
        return {k: v.clone() for k, v in O.prepare_features(O.empty_features(lengths, n_pad=40)).items()}

    fa = feats([40, 40])
    eng = model.model.bind(fa)
    za = model.model(T(eng.frenet(x), x), ts, fa)['z'].cpu()
    del fa
    fb = feats([40, 25])                                   # same shapes, different residue_mask; may land on the freed storage
    rb = model.model.bind(fb).frenet(x)
    zb = model.model(T(rb, x), ts, fb)['z'].cpu()
    fresh = GenieEngine(dict(O.BASE_DIMS), base_weights, 'cuda:0')
    fresh.bind_features(fb)
    zf = fresh.denoise(x, fresh.frenet(x), ts)['z'].cpu()
    fresh.close()
    assert torch.equal(zb, zf)
    assert not torch.equal(zb[1], za[1])
    # an in-place edit of a bound tensor is seen too (tensor._version)
    fb['residue_mask'][1, 25:30] = 1
    zc = model.model(T(model.model.bind(fb).frenet(x), x), ts, fb)['z'].cpu()
    assert not torch.equal(zc[1], zb[1])


def test_handle_free_frenet_frames_matches_reference_golden():
    """genie.utils.geo_utils.compute_frenet_frames(coords, chains, mask) with the reference's own signature."""
    from conftest import load_golden
    from genie.utils.geo_utils import compute_frenet_frames
    g = load_golden('geometry')
    r = compute_frenet_frames(torch.from_numpy(g['frenet_coords']).cuda(), torch.from_numpy(g['frenet_chains']).cuda(),
                              torch.from_numpy(g['frenet_mask']).cuda())
    assert mdiff(r, torch.from_numpy(g['frenet_rots'])) < 2e-6
    # a long chain (N = 1200 > 682: past the 64-KiB default dynamic-LDS limit)
    x = torch.randn(1, 1200, 3, generator=torch.Generator().manual_seed(3)) * 5
    f = O.empty_features([1200])
    ref = O.compute_frenet_frames(x, f['chain_index'], f['residue_mask'])
    assert mdiff(compute_frenet_frames(x.cuda(), f['chain_index'], f['residue_mask']), ref) < 2e-6
    with pytest.raises(Exception):
        compute_frenet_frames(x, f['chain_index'], f['residue_mask'])          # CPU tensors: no CPU path


@pytest.mark.gpu
def test_bench_emits_one_contract_line():
    """bench.py's contract with the driver: exactly ONE line on stdout, JSON, with the keys the driver and the judge read (metric /
    value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload,
    the roofline object).  Short run (2 steps), without the CPU-baseline and extra legs; then the same through the 1-rank RCCL path."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    for extra_env in ({}, {'GENIE_BENCH_FORCE_DIST': '1', 'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29597'}):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-extra-legs',
                            '--profile-steps', '1'], capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra_env), cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, lines[:5]
        d = json.loads(lines[0])
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                  'config', 'roofline'):
            assert k in d, k
        assert d['steps'] == 2 and d['warmup'] == 1 and d['n_gpus'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak'
        assert d['vs_baseline'] is None and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
        assert abs(d['value'] - 1e3 / d['ms_per_step']) < 1e-6 * d['value'] and d['value'] > 10
        rf = d['roofline']
        for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
            assert k in rf, k
        assert rf['bound'] in ('hbm', 'mfma') and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9 and 0 < rf['frac'] < 1
