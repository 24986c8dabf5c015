"""Developer aid (GPU box): in-kernel phase stamps of the fused pair chains (-DFZ_TS build of pair_fused_kernels.hip).

    GENIE_HIP_LIB=genie2_amd/lib/libgenie_fzts.so python tests/devtools/ts_fused.py

Stamps per tile: start | inputs landed | O stages x4 | [T end] | store + split done | P passes x8 | drain  (cycles, work-group 0)
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
eng = GenieEngine(dims, sd, 'cuda:0')
eng.bind_features(O.empty_features([N] * B))
x = torch.randn(B, N, 3)
r = eng.frenet(x)
ts = torch.full((B,), 500, dtype=torch.int32)
for _ in range(3):
    eng.denoise(x, r, ts, None)
torch.cuda.synchronize()
lib = C.CDLL(capi.LIB_PATH)
buf = np.zeros((16, 2048), dtype=np.uint64)
assert lib.genie_fz_debug_read(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.astype(np.int64)
FINE = os.environ.get('FZ_FINE') == '1'
F3 = os.environ.get('FZ_FINE') == '3'      # -DFZ_TS=3 build: LN + split of x, LN statistics of z, the two halves of the z store      # -DFZ_TS=2 build: each projection pass as (mfma + epilogue, vmcnt wait, requests + barrier)
PP = 24 if FINE else 8
for var, (name, per) in enumerate((('chain A (O + P)', 2 + 4 + 1 + PP + 1 + 4 * F3), ('chain B (O + T + P)', 2 + 4 + 1 + 1 + PP + 1 + 4 * F3))):
    labels = ['load'] + (['xnorm'] if F3 else []) + ['Wz0', 'Wz1'] + (['zstat'] if F3 else []) + ['Wg0', 'Wg1'] + (['T'] if var else []) + (['st0', 'st1', 'split'] if F3 else ['store']) + (['P%d%s' % (i, c) for i in range(8) for c in 'mwb'] if FINE else ['P%d' % i for i in range(8)]) + ['drain']
    print(name)
    print('  wave ' + ' '.join(f'{l:>7}' for l in labels) + '   tile period')
    for w in range(8):
        v = t[var * 8 + w]
        n = int((v > 0).sum()) // per
        if n < 2:
            continue
        v = v[:n * per].reshape(n, per)
        d = np.diff(v, axis=1)[1:].mean(axis=0)          # skip the first tile
        print(f'  {w:4d} ' + ' '.join(f'{x:7.0f}' for x in d) + f'   {np.diff(v[:, 0]).mean():9.0f}')
