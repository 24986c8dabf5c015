"""Developer aid: fine stamps inside the transition stage (HX_ABL=384 build): start, after DMA issue, after bias reads,
after k-chunk 0, after k-chunk 3, after GEMM1, after split+GEMM2, after barrier."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import genie_oracle as O
from genie2_amd import capi
from genie2_amd.engine import GenieEngine
dims = dict(O.BASE_DIMS); sd = O.synthetic_state_dict(dims, seed=1)
B, N = 8, 256
eng = GenieEngine(dims, sd, 'cuda:0'); eng.bind_features(O.empty_features([N] * B))
x = torch.randn(B, N, 3); r = eng.frenet(x); ts = torch.full((B,), 500, dtype=torch.int32)
for _ in range(2): eng.denoise(x, r, ts, None)
torch.cuda.synchronize()
lib = C.CDLL(capi.LIB_PATH)
buf = np.zeros((24, 4096), dtype=np.uint64)
assert lib.genie_hx_debug_read(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.astype(np.int64)
per_stage = 8
names = ['issue DMA', 'bias reads', 'k-chunk 0', 'k-chunks 1-3', 'k-chunks 4-7', 'split+GEMM2', 'barrier']
for w in range(8):
    x = t[16 + w]; n = int((x > 0).sum())
    per_tile = 16 * per_stage
    nt = n // per_tile
    if nt < 2: continue
    st = x[:nt * per_tile].reshape(nt, 16, per_stage)[1:]
    d = np.diff(st, axis=2).mean(axis=(0, 1))
    print('wave', w, '  '.join(f'{nm} {v:.0f}' for nm, v in zip(names, d)))
