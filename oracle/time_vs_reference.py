"""Wall time of the oracle against the real reference on the same host (the oracle is a fair CPU stand-in for the
reference on the GPU box only if it runs within +-10 % of it).  Development container only (imports /root/reference).

    python oracle/time_vs_reference.py [n_timed_steps]

One warm-up call, then n timed Denoiser.forward calls each, N=256, batch 8, fp32, torch threads = all cores.  Prints one JSON line.
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.argv = sys.argv[:2]
n_timed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
import make_goldens as MG  # noqa: E402  (puts the reference on sys.path, keeps the repo's genie/ facade off it)

O = MG.O
torch.set_num_threads(os.cpu_count())
sd = O.synthetic_state_dict(O.BASE_DIMS, seed=0)
cfg = MG.ref_config()
ref = MG.ref_denoiser(cfg, sd)
B, N = 8, 256
f = O.empty_features([N] * B)
fr = O.prepare_features(f)
g = torch.Generator().manual_seed(42)
trans = torch.randn(B, N, 3, generator=g)
rots = O.compute_frenet_frames(trans, fr['chain_index'], fr['residue_mask'])
ts = torch.full((B,), 1000, dtype=torch.int32)


def timeit(fn):
    out = []
    with torch.no_grad():
        for i in range(1 + n_timed):
            t0 = time.perf_counter()
            z = fn()
            out.append(time.perf_counter() - t0)
    return out, z


t_ref, z_ref = timeit(lambda: ref(MG.T(rots, trans), ts, fr)['z'])
t_orc, z_orc = timeit(lambda: O.denoiser_forward(sd, dict(O.BASE_DIMS), rots, trans, ts, f, 'eigh')['z'])
med = lambda v: sorted(v[1:])[len(v[1:]) // 2]  # noqa: E731
print(json.dumps({'threads': torch.get_num_threads(), 'n': N, 'batch': B,
                  'reference_s': [round(x, 2) for x in t_ref], 'oracle_s': [round(x, 2) for x in t_orc],
                  'reference_median_s': round(med(t_ref), 2), 'oracle_median_s': round(med(t_orc), 2),
                  'oracle_over_reference': round(med(t_orc) / med(t_ref), 3),
                  'max_abs_dz': float((z_ref - z_orc).abs().max())}))
