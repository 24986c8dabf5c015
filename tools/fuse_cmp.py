"""Fused structure-layer tail (k_struct_rows_hx) against the six separate launches it replaces: bit-identical by construction."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genie2_amd import features as F, pack          # noqa: E402
from genie2_amd.engine import GenieEngine           # noqa: E402

dev = torch.device('cuda', 0)
dims = dict(pack.BASE_DIMS)
for (B, N) in ((8, 256), (2, 50), (1, 37)):
    eng = GenieEngine(dims, pack.random_state_dict(dims, seed=0), dev)
    feats = F.convert_np_features_to_tensor(F.batchify_np_features([F.create_empty_np_features([N]) for _ in range(B)]), dev)
    eng.bind_features(feats)
    g = torch.Generator().manual_seed(1)
    noise = torch.randn(6, B, N, 3, generator=g).to(dev)
    T = dims['n_timestep']
    res = []
    for env in ('', '1'):
        if env:
            os.environ['GENIE_NO_STRUCT_FUSE'] = env
        else:
            os.environ.pop('GENIE_NO_STRUCT_FUSE', None)
        tr, ro, _ = eng.sample_loop(noise, 0.6, first_step=T, last_step=T - 3)
        torch.cuda.synchronize()
        res.append((tr.clone(), ro.clone()))
    os.environ.pop('GENIE_NO_STRUCT_FUSE', None)
    d = (res[0][0] - res[1][0]).abs().max().item()
    print(f'B={B} N={N}: fused vs separate max|dtrans| = {d:.3e}, rots {((res[0][1] - res[1][1]).abs().max().item()):.3e}, finite {bool(torch.isfinite(res[0][0]).all())}')
