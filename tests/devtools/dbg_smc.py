import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle import genie_oracle as O
from genie.config import Config
from genie2_amd.diffusion import Genie
from genie2_amd.sampler import UnconditionalSampler
from genie2_amd.smc import TwistedSampler
from genie2_amd import pack
cfg = Config(); cfg.diffusion['n_timestep'] = 12
model = Genie(cfg); model.model.load_state_dict(O.synthetic_state_dict(O.BASE_DIMS, seed=0)); model = model.eval().to('cuda:0')
B, N, T = 4, 24, 12
noise = torch.randn(T, B, N, 3, generator=torch.Generator().manual_seed(4))
base = {'length': N, 'scale': 0.6, 'num_samples': B, 'outdir': '/tmp/o', 'prefix': 'x', 'offset': 0, 'noise': noise}
us = UnconditionalSampler(model)
from genie2_amd import features as F
feats = F.convert_np_features_to_tensor(F.batchify_np_features([us.create_np_features(base) for _ in range(B)]), 'cuda:0')
eng = model.model.bind(feats)
tr, ro, rec = eng.sample_loop(noise.cuda(), 0.6, record=True)
sched = {k: v.cuda() for k, v in pack.schedule_tensors(T).items()}
abar, betas = sched['alphas_cumprod'], sched['betas']
trans = noise[0].cuda(); rots = eng.frenet(trans)
for it, step in enumerate(range(T, 0, -1)):
    ts = torch.full((B,), step, dtype=torch.int32, device='cuda')
    z = eng.denoise(trans, rots, ts)['z']
    c0, c1 = torch.sqrt(abar[step]), torch.sqrt(1 - abar[step])
    x0 = (trans - c1 * z) / c0
    coef1 = torch.sqrt(abar[step - 1]) * betas[step] / (1 - abar[step]); coef2 = sched['sqrt_alphas'][step] * (1 - abar[step - 1]) / (1 - abar[step])
    mean = coef1 * x0 + coef2 * trans
    wz = (1 - sched['alphas'][step]) / sched['sqrt_one_minus_alphas_cumprod'][step]
    mean_b = (trans - wz * z) / sched['sqrt_alphas'][step]
    new = mean if step == 1 else mean + 0.6 * sched['sqrt_betas'][step] * noise[it + 1].cuda()
    print(step, 'mean diff', float((mean - mean_b).abs().max()), 'vs loop', float((new - rec[it]).abs().max()), float(abar[step]), float(coef1), float(coef2))
    trans = new; rots = eng.frenet(trans)
print('---- with vjp calls')
w = pack.flatten_state_dict(model.model.state_dict(), model.model.dims).cuda()
trans = noise[0].cuda(); rots = eng.frenet(trans)
for it, step in enumerate(range(T, 0, -1)):
    ts = torch.full((B,), step, dtype=torch.int32, device='cuda')
    z = eng.denoise(trans, rots, ts)['z']
    z2, dt = eng.denoise_vjp(w, trans, rots, ts, torch.zeros_like(z))
    z3 = eng.denoise(trans, rots, ts)['z']
    print(step, 'z train-path vs sampling', float((z2 - z).abs().max()), 'z after vjp', float((z3 - z).abs().max()), 'dt', float(dt.abs().max()), bool(torch.isfinite(dt).all()))
    wz = (1 - sched['alphas'][step]) / sched['sqrt_one_minus_alphas_cumprod'][step]
    mean_b = (trans - wz * z) / sched['sqrt_alphas'][step]
    new = mean_b if step == 1 else mean_b + 0.6 * sched['sqrt_betas'][step] * noise[it + 1].cuda()
    print('   vs loop', float((new - rec[it]).abs().max()))
    trans = new; rots = eng.frenet(trans)
