"""Developer aid: fused pair chains vs separate launches, tap by tap, for a few shapes (GPU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import genie_oracle as O  # noqa: E402
from genie2_amd.engine import GenieEngine  # noqa: E402

dims = dict(O.BASE_DIMS)
sd = O.synthetic_state_dict(dims, seed=0)
eng = GenieEngine(dims, sd, 'cuda:0')
taps = ('p_init', 'p_trimul_out0', 'p_layer0', 'p')
for lengths in ([16], [32], [32, 32], [16, 16], [40], [64], [70, 41, 64]):
    f = O.empty_features(lengths)
    B, N = f['residue_mask'].shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, N, 3, generator=g) * 4
    eng.bind_features(f)
    r = eng.frenet(x)
    ts = torch.full((B,), 500, dtype=torch.int32)
    os.environ.pop('GENIE_NO_PAIR_FUSE', None)
    a = eng.denoise(x, r, ts, None, taps=taps)
    os.environ['GENIE_NO_PAIR_FUSE'] = '1'
    b = eng.denoise(x, r, ts, None, taps=taps)
    os.environ.pop('GENIE_NO_PAIR_FUSE', None)
    pm = (f['residue_mask'][:, :, None] * f['residue_mask'][:, None, :]).bool()
    msg = []
    for k in taps:
        d = (a[k].cpu() - b[k].cpu()).abs()
        d[~pm] = 0
        idx = torch.nonzero(d > 1e-3 * float(b[k].abs().max()))
        msg.append(f'{k} {float(d.max()):.2e}/{float(b[k].abs().max()):.1f} nbad={len(idx)}' + (f' first={idx[0].tolist()} last={idx[-1].tolist()}' if len(idx) else ''))
    print(lengths, ' | '.join(msg), flush=True)
