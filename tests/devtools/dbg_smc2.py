import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle import genie_oracle as O
from genie.config import Config
from genie2_amd.diffusion import Genie
from genie2_amd.smc import TwistedSampler, motif_twisting_function
from genie2_amd import pack
import genie2_amd.smc as S
cfg = Config(); cfg.diffusion['n_timestep'] = 12
model = Genie(cfg); model.model.load_state_dict(O.synthetic_state_dict(O.BASE_DIMS, seed=0)); model = model.eval().to('cuda:0')
B, N, T = 4, 24, 12
noise = torch.randn(T, B, N, 3, generator=torch.Generator().manual_seed(4))
base = {'length': N, 'scale': 0.6, 'num_samples': B, 'outdir': '/tmp/o', 'prefix': 'x', 'offset': 0, 'noise': noise}
tw = TwistedSampler(model)
orig = tw.model.model.engine().denoise_vjp
def spy(w, trans, rots, ts, dz, quat_codes=None):
    z, dt = orig(w, trans, rots, ts, dz, quat_codes)
    print('step', int(ts[0]), 'dz max', float(dz.abs().max()), 'dt max', float(dt.abs().max()), 'finite', bool(torch.isfinite(dt).all()), 'trans max', float(trans.abs().max()))
    return z, dt
tw.model.model.engine().denoise_vjp = spy
try:
    got = tw._sample(dict(base, twisting_function=lambda x0, step: (x0 * 0).sum(dim=(1, 2)), last_unguided_steps=0, ess_threshold=0.0))
    print('a ok', np.isfinite(np.stack([g['atom_positions'] for g in got])).all())
except Exception as e:
    print('a failed', type(e))
g = torch.Generator().manual_seed(9)
target = torch.randn(6, 3, generator=g) * 3
target = (target - target.mean(0, keepdim=True)).cuda()
mask = torch.zeros(1, N, dtype=torch.bool); mask[0, 5:11] = True
abar = pack.schedule_tensors(T)['alphas_cumprod'].cuda()
twist = lambda x0, step: motif_twisting_function(x0, mask.cuda(), target, abar[step], tausq=0.5)
try:
    got = tw._sample(dict(base, twisting_function=twist, last_unguided_steps=0, guidance_alpha=0.05, ess_threshold=0.0))
    print('b ok', np.isfinite(np.stack([g['atom_positions'] for g in got])).all(), tw.ess_trace, tw.resampled_at)
except Exception as e:
    import traceback; traceback.print_exc()

def motif_rmsd(items):
    out = []
    for it in items:
        x = torch.tensor(it['atom_positions'][5:11], dtype=torch.float32)
        out.append(float(((x - x.mean(0, keepdim=True) - target.cpu()) ** 2).sum(-1).mean().sqrt()))
    return out
from genie2_amd.sampler import UnconditionalSampler
ref = UnconditionalSampler(model)._sample(dict(base))
print('guided', motif_rmsd(got), 'unguided', motif_rmsd(ref))
