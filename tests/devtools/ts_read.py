"""Developer aid (GPU box): read the in-kernel timestamps of an HX_ABL=128 build after one denoiser call and
print the mean duration of each phase of k_trimul_proj_hx (the last launch overwrites earlier ones).

    GENIE_MATH=hx GENIE_HIP_LIB=genie2_amd/lib/abl/libgenie_abl128.so python tests/devtools/ts_read.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import genie_oracle as O  # noqa: E402
from genie2_amd import capi  # noqa: E402
from genie2_amd.engine import GenieEngine  # noqa: E402

dims = dict(O.BASE_DIMS)
sd = O.synthetic_state_dict(dims, seed=1)
B, N = 8, 256
f = O.empty_features([N] * B)
eng = GenieEngine(dims, sd, 'cuda:0')
eng.bind_features(f)
x = torch.randn(B, N, 3)
r = eng.frenet(x)
ts = torch.full((B,), 500, dtype=torch.int32)
for _ in range(3):
    eng.denoise(x, r, ts, None)
torch.cuda.synchronize()
lib = C.CDLL(capi.LIB_PATH)
buf = np.zeros((24, 4096), dtype=np.uint64)
rc = lib.genie_hx_debug_read(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
t = buf.astype(np.int64)
# projection (variant 0 = incoming, 1 = outgoing): per tile 1 + 8 x (start, after mfma, after vmcnt, after barrier)
for var in (0, 1):
    rows = []
    for w in range(8):
        x = t[var * 8 + w]
        n = int((x > 0).sum()) // 33
        if n == 0:
            continue
        x = x[:n * 33].reshape(n, 33)
        st = x[:, 1:].reshape(n, 8, 4)
        d = np.diff(st, axis=2).mean(axis=(0, 1))
        rows.append((w, (x[:, 1] - x[:, 0]).mean(), *d, np.diff(x[:, 0]).mean() if n > 1 else 0))
    print(f'projection variant {var}: wave  prologue  mfma+epi  vmcnt  barrier  tile period')
    for r in rows:
        print('   ', ' '.join(f'{v:9.0f}' for v in r))
# pair transition (variant 2): per stage (start, after GEMM1, after split + GEMM2, after barrier)
print('transition: wave  gemm1  split+gemm2  barrier  stage period  max stage->stage gap (tile boundary)')
for w in range(8):
    x = t[16 + w]
    n = int((x > 0).sum()) // 4 * 4
    if n < 8:
        continue
    st = x[:n].reshape(-1, 4)
    d = np.diff(st, axis=1)
    gap = st[1:, 0] - st[:-1, 3]
    print('   ', w, f'{d[:, 0].mean():9.0f} {d[:, 1].mean():9.0f} {d[:, 2].mean():9.0f} {np.median(np.diff(st[:, 0])):9.0f} {gap.max():9.0f}')
