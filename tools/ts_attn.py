import os, sys, torch
sys.path.insert(0, '/root/repo')
from genie2_amd import features as F, pack
from genie2_amd.engine import GenieEngine
dev = torch.device('cuda', 0)
dims = dict(pack.BASE_DIMS)
eng = GenieEngine(dims, pack.random_state_dict(dims, seed=0), dev)
B, N = 8, 256
feats = F.convert_np_features_to_tensor(F.batchify_np_features([F.create_empty_np_features([N]) for _ in range(B)]), dev)
eng.bind_features(feats)
x = torch.randn(B, N, 3) * 5
r = eng.frenet(x)
ts = torch.full((B,), 500, dtype=torch.int32)
for _ in range(2):
    eng.denoise(x, r, ts, None)
torch.cuda.synchronize()
