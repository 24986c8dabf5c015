"""Developer probe (GPU box): does running two half-batches as independent reverse loops on two streams beat one full batch?
The structure track of one half (latency-bound, small kernels) could run under the pair stack of the other (throughput-bound)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genie2_amd import features as F, pack          # noqa: E402
from genie2_amd.engine import GenieEngine           # noqa: E402

dev = torch.device('cuda', 0)
dims = dict(pack.BASE_DIMS)
N, T = 256, dims['n_timestep']
sd = pack.random_state_dict(dims, seed=0)


def make(B):
    eng = GenieEngine(dims, sd, dev)
    feats = F.convert_np_features_to_tensor(F.batchify_np_features([F.create_empty_np_features([N]) for _ in range(B)]), dev)
    eng.bind_features(feats)
    noise = torch.randn(T, B, N, 3, generator=torch.Generator().manual_seed(B)).to(dev)
    return eng, noise


def run(engs, streams, steps=40):
    states = []
    for (eng, noise), st in zip(engs, streams):
        with torch.cuda.stream(st):
            tr, ro, _ = eng.sample_loop(noise, 0.6, first_step=T, last_step=T - 4)
            states.append((tr, ro))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for (eng, noise), st, state in zip(engs, streams, states):
        with torch.cuda.stream(st):
            eng.sample_loop(noise, 0.6, first_step=T - 5, last_step=T - 4 - steps, state=state)
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t0)


one = make(8)
print('one engine, batch 8          : %.2f batch-steps/s' % run([one], [torch.cuda.current_stream()]), flush=True)
two = [make(4), make(4)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print('two engines, batch 4, 1 stream: %.2f batch-of-8-steps/s' % (run(two, [s1, s1])), flush=True)
print('two engines, batch 4, 2 streams: %.2f batch-of-8-steps/s' % (run(two, [s1, s2])), flush=True)
four = [make(2) for _ in range(4)]
ss = [torch.cuda.Stream() for _ in range(4)]
print('four engines, batch 2, 4 streams: %.2f batch-of-8-steps/s' % (run(four, ss)), flush=True)
