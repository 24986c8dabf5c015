"""Developer aid (GPU box): time + check one build of the library (GENIE_HIP_LIB) on the headline shape.
    GENIE_HIP_LIB=genie2_amd/lib/abl/libgenie_fz_x.so python tools/fz_try.py [tag]
Prints batch-steps/s over 30 steps, the per-launch times of the pair-stack kernel classes, and max |p_fused - p_unfused| over ALL of p."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genie2_amd import features as F, pack          # noqa: E402
from genie2_amd.engine import GenieEngine           # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get('GENIE_HIP_LIB', 'default'))
dev = torch.device('cuda', 0)
dims = dict(pack.BASE_DIMS)
B, N, T = 8, 256, dims['n_timestep']
eng = GenieEngine(dims, pack.random_state_dict(dims, seed=0), dev)
feats = F.convert_np_features_to_tensor(F.batchify_np_features([F.create_empty_np_features([N]) for _ in range(B)]), dev)
eng.bind_features(feats)
g = torch.Generator().manual_seed(42)
noise = torch.randn(T, B, N, 3, generator=g).to(dev)
tr, ro, _ = eng.sample_loop(noise, 0.6, first_step=T, last_step=T - 4)
torch.cuda.synchronize()
best = 0.0
for rep in range(3):
    t0 = time.perf_counter()
    tr2, ro2, _ = eng.sample_loop(noise, 0.6, first_step=T - 5, last_step=T - 34, state=(tr.clone(), ro.clone()))
    torch.cuda.synchronize()
    best = max(best, 30 / (time.perf_counter() - t0))
eng.profile(True)
eng.sample_loop(noise, 0.6, first_step=T - 35, last_step=T - 37, state=(tr2, ro2))
torch.cuda.synchronize()
eng.profile(False)
prof = eng.profile_read()
x = torch.randn(B, N, 3, generator=g) * 6
r = eng.frenet(x)
ts = torch.randint(1, 1001, (B,), generator=g).int()
a = eng.denoise(x, r, ts, None, taps=('p',))
a2 = eng.denoise(x, r, ts, None, taps=('p',))
os.environ['GENIE_NO_PAIR_FUSE'] = '1'
b = eng.denoise(x, r, ts, None, taps=('p',))
del os.environ['GENIE_NO_PAIR_FUSE']
d = max(float((a['p'] - b['p']).abs().max()), float((a2['p'] - b['p']).abs().max()))
ks = ' '.join(f'{k}={prof[k][0] / max(prof[k][1], 1):.4f}' for k in ('pair_fused_a', 'pair_fused_b', 'trimul_contract', 'ipa_attn', 'gemm_rows', 'struct_rows') if k in prof)
print(f'{tag}: {best:.2f} batch-steps/s | ms/launch {ks} | max|dp| fused vs separate {d:.2e} (|p| {float(b["p"].abs().max()):.1f}) finite {bool(torch.isfinite(a["p"]).all())}', flush=True)
