import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
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
per = 1 + 8 * (4 + 8)
for w in range(8):
    x = t[w]; n = int((x > 0).sum()) // per
    x = x[:n * per].reshape(n, per)[1:]          # skip first tile
    st = x[:, 1:].reshape(-1, 8, 12)              # per stage: start, 8 kc stamps, after loop, after wait, after barrier
    kc = np.diff(st[:, :, 0:9], axis=2).mean(axis=(0, 1))
    print('wave', w, 'per-kc cycles:', kc.round(0), ' loop->wait', (st[:, :, 10] - st[:, :, 9]).mean().round(0), ' barrier', (st[:, :, 11] - st[:, :, 10]).mean().round(0))
for w in (0, 4, 8, 12):
    x = t[w]; n = int((x > 0).sum()) // per
    x = x[:n * per].reshape(n, per)[1:]
    st = x[:, 1:].reshape(-1, 8, 12)
    d = np.diff(st[:, :, 0:9], axis=2).mean(axis=0)          # [pass][kc]
    print('wave', w, 'cycles of k-chunks 2 and 3 by pass:', d[:, 2].round(0), d[:, 3].round(0), ' k-chunk 0 by pass:', d[:, 0].round(0))
