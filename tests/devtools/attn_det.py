"""Developer aid (GPU box): run-to-run determinism of the denoiser and 4-query (default) vs 8-query attention work-groups."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import genie_oracle as O          # noqa: E402
from genie2_amd.engine import GenieEngine     # noqa: E402

dims = dict(O.BASE_DIMS)
sd = O.synthetic_state_dict(dims, seed=1)
eng = GenieEngine(dims, sd, 'cuda:0')
for lengths in ([45, 23, 38], [256] * 2, [64]):
    f = O.empty_features(lengths)
    B, N = f['residue_mask'].shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, N, 3, generator=g) * 4
    eng.bind_features(f)
    r = eng.frenet(x)
    ts = torch.full((B,), 500, dtype=torch.int32)
    out = []
    for env in ({}, {}, {'GENIE_IPA_Q8': '1'}, {'GENIE_NO_STRUCT_FUSE': '1'}, {'GENIE_IPA_Q8': '1', 'GENIE_NO_STRUCT_FUSE': '1'}):
        for k in ('GENIE_IPA_Q8', 'GENIE_NO_STRUCT_FUSE'):
            os.environ.pop(k, None)
        os.environ.update(env)
        o = eng.denoise(x, r, ts, None, taps=('s_final',))
        out.append({k: v.cpu() for k, v in o.items()})
    for k in ('GENIE_IPA_Q8', 'GENIE_NO_STRUCT_FUSE'):
        os.environ.pop(k, None)
    names = ['same again', 'Q8', 'no fuse', 'Q8 + no fuse']
    m = f['residue_mask'].bool()
    for n, o in zip(names, out[1:]):
        print(lengths[:3], n, 'valid residues: z %.3e s_final %.3e; all rows: z %.3e' % (
            (o['z'] - out[0]['z'])[m].abs().max().item(), (o['s_final'] - out[0]['s_final'])[m].abs().max().item(),
            (o['z'] - out[0]['z']).abs().max().item()))
