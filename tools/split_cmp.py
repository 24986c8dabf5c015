"""Developer aid (GPU box): the two-stream structure net against the single-stream order (GENIE_NO_STRUCT_SPLIT=1): bit-identical
z / states expected; reverse-loop rate of both."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genie2_amd import features as F, pack          # noqa: E402
from genie2_amd.engine import GenieEngine           # noqa: E402

dev = torch.device('cuda', 0)
dims = dict(pack.BASE_DIMS)
T = dims['n_timestep']
for (B, N) in ((8, 256),):
    eng = GenieEngine(dims, pack.random_state_dict(dims, seed=0), dev)
    feats = F.convert_np_features_to_tensor(F.batchify_np_features([F.create_empty_np_features([N - (3 * i) % 7]) for i in range(B)]), dev)
    eng.bind_features(feats)
    g = torch.Generator().manual_seed(1)
    N = feats['residue_mask'].shape[1]
    x = torch.randn(B, N, 3, generator=g) * 5
    r = eng.frenet(x)
    ts = torch.randint(1, 1001, (B,), generator=g).int()
    noise = torch.randn(T, B, N, 3, generator=g).to(dev)
    res, rate = [], []
    for env in ('', '1'):
        if env:
            os.environ['GENIE_NO_STRUCT_SPLIT'] = env
        else:
            os.environ.pop('GENIE_NO_STRUCT_SPLIT', None)
        out = eng.denoise(x, r, ts, None, taps=('states', 's_final', 'ipa_cat0', 'rots_out', 'trans_out'))
        tr, ro, _ = eng.sample_loop(noise, 0.6, first_step=T, last_step=T - 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr, ro, _ = eng.sample_loop(noise, 0.6, first_step=T - 5, last_step=T - 34, state=(tr, ro))
        torch.cuda.synchronize()
        rate.append(30 / (time.perf_counter() - t0))
        out['traj'] = tr
        res.append(out)
    os.environ.pop('GENIE_NO_STRUCT_SPLIT', None)
    d = {k: float((res[0][k] - res[1][k]).abs().max()) for k in res[0]}
    print(f'B={B} N={N}: split vs single max|d| {d} finite {bool(torch.isfinite(res[0]["traj"]).all())} | batch-steps/s split {rate[0]:.2f} single {rate[1]:.2f}', flush=True)
    eng.close()
