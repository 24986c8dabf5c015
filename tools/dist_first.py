"""Developer probe (GPU box): the reverse loop with the NCCL process group initialised BEFORE the engine is created (the order that
made the two-stream structure net 9 % slower)."""
import os
import sys
import time
import torch
import torch.distributed as td
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29592')
td.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from genie2_amd import features as F, pack          # noqa: E402
from genie2_amd.engine import GenieEngine           # noqa: E402
dims = dict(pack.BASE_DIMS)
B, N, T = 8, 256, dims['n_timestep']
eng = GenieEngine(dims, pack.random_state_dict(dims, seed=0), dev)
feats = F.convert_np_features_to_tensor(F.batchify_np_features([F.create_empty_np_features([N]) for _ in range(B)]), dev)
eng.bind_features(feats)
noise = torch.randn(T, B, N, 3, generator=torch.Generator().manual_seed(42)).to(dev)
for rep in range(2):
    tr, ro, _ = eng.sample_loop(noise, 0.6, first_step=T, last_step=T - 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.sample_loop(noise, 0.6, first_step=T - 5, last_step=T - 44, state=(tr, ro))
    torch.cuda.synchronize()
    print(f'NCCL first, GENIE_ST2_PRIO={os.environ.get("GENIE_ST2_PRIO", "0")} GENIE_NO_STRUCT_SPLIT={os.environ.get("GENIE_NO_STRUCT_SPLIT", "")}: {40 / (time.perf_counter() - t0):.2f} batch-steps/s', flush=True)
td.destroy_process_group()
