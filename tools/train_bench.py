"""Time one training step (forward + backward + Adam) of the base model at N residues, batch B, on one GPU (BASELINE config 5's
shape with a synthetic batch).  python tools/train_bench.py [N] [B] [fast_math]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genie2_amd import pack  # noqa: E402
from genie2_amd.engine import GenieEngine, adam_step  # noqa: E402
from genie2_amd import features as F  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
fast = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dims = dict(pack.BASE_DIMS)
sd = pack.random_state_dict(dims, seed=0)
eng = GenieEngine(dims, sd, 'cuda:0')
feats = F.convert_np_features_to_tensor(F.batchify_np_features([F.create_empty_np_features([N]) for _ in range(B)]), 'cuda:0')
eng.bind_features(feats)
w = pack.flatten_state_dict(sd, dims).cuda()
g, m, v = torch.zeros_like(w), torch.zeros_like(w), torch.zeros_like(w)
gen = torch.Generator().manual_seed(0)
x0 = (torch.randn(B, N, 3, generator=gen) * 8).cuda()
z = torch.randn(B, N, 3, generator=gen).cuda()
s = torch.randint(1, 1001, (B,), generator=gen).int().cuda()
sched = pack.schedule_tensors(1000)
trans, rots = eng.q_sample(x0, z, sched['sqrt_alphas_cumprod'].cuda()[s.long()], sched['sqrt_one_minus_alphas_cumprod'].cuda()[s.long()])
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = eng.train_forward_backward(w, trans, rots, s, z, 1.0, grads=g, seed=it, fast_math=fast)
    adam_step(w, g, m, v, 1e-4, it + 1)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'N={N} B={B} fast_math={fast} step {it}: {dt * 1e3:.1f} ms, loss {float(out["weighted_loss"]):.4f}, workspace {eng.lib.genie_train_workspace_bytes(eng._h) / 2**30:.2f} GiB', flush=True)
eng.profile(True)
out = eng.train_forward_backward(w, trans, rots, s, z, 1.0, grads=g, seed=9, fast_math=fast)
torch.cuda.synchronize()
eng.profile(False)
prof = {k: v for k, v in eng.profile_read().items() if k.startswith('train_')}
print('per class (ms, launches):', {k: (round(ms, 2), n) for k, (ms, n) in prof.items()}, 'gemm TFLOP', eng.lib.genie_train_gemm_flop(eng._h) / 1e12, flush=True)
