"""Developer aid (GPU box): print per-stage max differences HIP vs oracle.

    python tests/devtools/stage_diff.py [--base] [--n 40]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import genie_oracle as O  # noqa: E402
from genie2_amd.engine import GenieEngine  # noqa: E402


def md(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    d = (a - b).abs().max().item()
    return f'{d:.3e} (ref max {b.abs().max().item():.3e})'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--base', action='store_true')
    ap.add_argument('--n', type=int, default=40)
    args = ap.parse_args()
    dims = dict(O.BASE_DIMS) if args.base else O.small_dims()
    sd = O.synthetic_state_dict(dims, seed=1)
    N = args.n
    g = torch.Generator().manual_seed(3)
    f = O.empty_features([N, N - 7], chains_per_sample=[[N], [N - 20, 13]])
    ca = 4.0 * torch.randn(8, 3, generator=g)
    O.add_motif(f, 0, ca, [5, 6, 7, 8, 20, 21, 22, 23])
    trans = 2.5 * torch.randn(2, N, 3, generator=g)
    fr = O.prepare_features(f)
    rots = O.compute_frenet_frames(trans, fr['chain_index'], fr['residue_mask'])
    ts = torch.tensor([dims['n_timestep'], 3], dtype=torch.int32)

    eng = GenieEngine(dims, sd, 'cuda:0')
    eng.bind_features(f)
    print('workspace MB', eng.workspace_bytes() / 2 ** 20)
    r_gpu = eng.frenet(trans)
    print('frenet          ', md(r_gpu, rots))

    for mode in ('closed', 'eigh'):
        taps = {}
        ref = O.denoiser_forward(sd, dims, rots, trans, ts, f, mode, None, taps)
        codes = O.quat_sign_codes(taps['quat']) if mode == 'eigh' else None
        out = eng.denoise(trans, rots, ts, codes,
                          taps=('s', 'p', 's_final', 'rots_out', 'trans_out', 'p_init', 'p_layer0'))
        torch.cuda.synchronize()
        print(f'--- quat mode {mode}')
        print('s               ', md(out['s'], ref['s']))
        print('p_init          ', md(out['p_init'], taps['p_init']))
        print('p_layer0        ', md(out['p_layer0'], taps['p_after_layer0']))
        print('p               ', md(out['p'], ref['p']))
        print('s_final         ', md(out['s_final'], ref['s_final']))
        print('rots_out        ', md(out['rots_out'], ref['rots']))
        print('trans_out       ', md(out['trans_out'], ref['trans']))
        m = fr['residue_mask'].unsqueeze(-1).float()
        print('z (masked)      ', md(out['z'].cpu() * m, ref['z'] * m))

    # p_sample
    sched = O.setup_schedule(dims['n_timestep'])
    z = torch.randn(2, N, 3, generator=g)
    eps = torch.randn(2, N, 3, generator=g)
    for step, e in ((7, eps), (1, None)):
        nt, nr = O.p_sample_step(sched, step, 0.6, trans, z, e, fr)
        tg = trans.clone().cuda()
        rg = eng.p_sample(step, 0.6, tg, z.cuda(), e.cuda() if e is not None else None)
        print(f'p_sample step {step}: trans', md(tg, nt), ' rots', md(rg, nr))

    # short loop
    T = dims['n_timestep']
    noise = torch.randn(T, 2, N, 3, generator=g)
    t0 = time.time()
    tr, ro, rec = eng.sample_loop(noise, 0.6, first_step=T, last_step=T - 2, record=True)
    torch.cuda.synchronize()
    print('3 loop steps in', time.time() - t0, 's')
    d3 = dict(dims)
    x = noise[0].clone()
    r = O.compute_frenet_frames(x, fr['chain_index'], fr['residue_mask'])
    for it, step in enumerate(range(T, T - 3, -1)):
        zz = O.denoiser_forward(sd, d3, r, x, torch.full((2,), step, dtype=torch.int32), f, 'closed')['z']
        x, r = O.p_sample_step(sched, step, 0.6, x, zz, noise[it + 1], fr)
        print(f'loop it {it}: x', md(rec[it], x))


if __name__ == '__main__':
    main()
