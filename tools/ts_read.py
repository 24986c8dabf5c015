"""Developer aid (GPU box): read the in-kernel timestamps of an HX_ABL=128 build after one denoiser call and
print the mean duration of each phase of k_trimul_proj_hx (the last launch overwrites earlier ones).

    GENIE_MATH=hx GENIE_HIP_LIB=genie2_amd/lib/abl/libgenie_abl128.so python tools/ts_read.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
buf = np.zeros((8, 4096), dtype=np.uint64)
rc = lib.genie_hx_debug_read(buf.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
PER_TILE = 1 + 8 * 4
names = ['mfma+epilogue', 'vmcnt wait', 'barrier', 'to next stage start']
for w in (0, 2, 4, 6):
    t = buf[w].astype(np.int64)
    n = int((t > 0).sum())
    tiles = n // PER_TILE
    if tiles == 0:
        continue
    t = t[:tiles * PER_TILE].reshape(tiles, PER_TILE)
    pro = (t[:, 1] - t[:, 0]).mean()
    st = t[:, 1:].reshape(tiles, 8, 4)
    d = np.diff(st, axis=2).mean(axis=(0, 1))            # within-stage phases
    nxt = (st[:, 1:, 0] - st[:, :-1, 3]).mean()          # stage end -> next stage start
    tile_total = (t[1:, 0] - t[:-1, 0]).mean() if tiles > 1 else float('nan')
    print(f'variant {w // 4} wg {64 * (w % 4)}: tiles {tiles}  prologue {pro:.0f}  ' + '  '.join(f'{nm} {v:.0f}' for nm, v in zip(names, list(d) + [nxt]))
          + f'  | tile period {tile_total:.0f} (100 MHz ticks x?)')
    print('   per-stage mfma phase by pass:', np.diff(st, axis=2)[:, :, 0].mean(axis=0).round(0))
    print('   per-stage wait by pass      :', np.diff(st, axis=2)[:, :, 1].mean(axis=0).round(0))
    print('   per-stage barrier by pass   :', np.diff(st, axis=2)[:, :, 2].mean(axis=0).round(0))
