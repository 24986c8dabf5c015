"""Developer probe (GPU box): why is the reverse loop slower inside an initialised NCCL process group (1 rank)?"""
import os
import sys
import time
import torch
import torch.distributed as td
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genie2_amd import features as F, pack          # noqa: E402
from genie2_amd.engine import GenieEngine           # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dims = dict(pack.BASE_DIMS)
B, N, T = 8, 256, dims['n_timestep']
eng = GenieEngine(dims, pack.random_state_dict(dims, seed=0), dev)
feats = F.convert_np_features_to_tensor(F.batchify_np_features([F.create_empty_np_features([N]) for _ in range(B)]), dev)
eng.bind_features(feats)
noise = torch.randn(T, B, N, 3, generator=torch.Generator().manual_seed(42)).to(dev)


def leg(tag, pre=None, post=None):
    tr, ro, _ = eng.sample_loop(noise, 0.6, first_step=T, last_step=T - 4)
    if pre:
        pre()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.sample_loop(noise, 0.6, first_step=T - 5, last_step=T - 44, state=(tr, ro))
    if post:
        post()
    torch.cuda.synchronize()
    print(f'{tag}: {40 / (time.perf_counter() - t0):.2f} batch-steps/s', flush=True)


leg('no process group')
leg('no process group (again)')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29591')
td.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
leg('process group initialised, no collective')
leg('... td.barrier() before and after', td.barrier, td.barrier)
leg('... td.barrier() before only', td.barrier, None)
leg('process group initialised, no collective (again)')
x = torch.zeros(1, device=dev)
leg('... all_reduce of one float after', None, lambda: td.all_reduce(x))
# an engine created while the process group exists (bench.py's order)
eng.close()
eng = GenieEngine(dims, pack.random_state_dict(dims, seed=0), dev)
eng.bind_features(feats)
leg('engine created inside the process group')
noise = torch.randn(T, B, N, 3, generator=torch.Generator().manual_seed(42)).to(dev)
leg('... and the noise tensor re-uploaded')
td.destroy_process_group()
leg('process group destroyed')
