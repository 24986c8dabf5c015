"""Developer aid (GPU box): the whole reverse loop (T = 1000, N = 256, batch 8, random-init base weights) in both arithmetics
from the same noise: finiteness, and how far the f32-MFMA and split-f16 trajectories drift apart."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genie2_amd import pack, features as F
from genie2_amd.engine import GenieEngine

dims = dict(pack.BASE_DIMS)
B, N, T = 8, 256, dims['n_timestep']
sd = pack.random_state_dict(dims, seed=0)
feats = F.convert_np_features_to_tensor(F.batchify_np_features([F.create_empty_np_features([N]) for _ in range(B)]), 'cuda:0')
noise = torch.randn(T, B, N, 3, generator=torch.Generator().manual_seed(1)).cuda()
out = {}
for math in ('hx', 'f32'):
    eng = GenieEngine(dims, sd, 'cuda:0', math=math)
    eng.bind_features(feats)
    torch.cuda.synchronize(); t0 = time.time()
    tr, ro, rec = eng.sample_loop(noise, 0.6, record=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    out[math] = rec.cpu()
    print(f'{math}: {T} steps in {dt:.2f} s = {T / dt:.1f} batch-steps/s; finite: {bool(torch.isfinite(tr).all())}; '
          f'final coordinate RMS {float(tr.pow(2).mean().sqrt()):.3f}')
    eng.close()
d = (out['hx'] - out['f32']).abs().amax(dim=(1, 2, 3))
rms = out['f32'].pow(2).mean(dim=(1, 2, 3)).sqrt()
for k in (0, 9, 99, 499, 899, 999):
    print(f'after step {k + 1:4d}: max |dCa| hx vs f32 = {float(d[k]):.3e}  (coordinate RMS {float(rms[k]):.3f}, ratio {float(d[k] / rms[k]):.2e})')
