"""Headline benchmark: denoise-steps/sec at N=256, T=1000, batch=8 (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one iteration of the reverse loop (genie/sampler/base.py:239-282)
over one batch of 8 structures: Denoiser.forward + posterior update + Frenet
frames, all inside libgenie_hip (genie_sample_loop), inputs resident in HBM.
Each rank runs its own replica with its own noise seed (structures shard across
GPUs with no data-path collective); the only collectives are the timing
barrier / max and a trivial all_gather of the final C-alpha coordinates.

Rank 0 prints ONE JSON line.  `value` = batch-steps/s summed over all ranks
(x8 = structure-steps/s, also reported).  `roofline` prices the dominant kernel
class (largest time per step, HIP events on the launch stream) against the roof
that bounds it: its algorithmic HBM bytes at 8 TB/s or its matrix FLOPs at the
MFMA peak of the arithmetic in use, whichever takes longer (--math hx: three f16
MFMAs per f32 product at 2.5 PFLOP/s; --math f32: v_mfma_f32 at 157.3 TFLOP/s).
`traffic` = HBM bytes per launch from the committed rocprofv3 FETCH_SIZE /
WRITE_SIZE passes (profiles/), null if that kernel was not profiled.
`cpu_baseline` times the oracle (the CPU restatement of the reference) on this
host for one step of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from genie2_amd import pack  # noqa: E402
from genie2_amd.engine import GenieEngine  # noqa: E402
from genie2_amd import features as F  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0     # same table, "Peak BF16/FP16 MFMA", dense
PEAK_HBM_GBS = 8000.0
PMC_TRAFFIC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic.json')
KERNEL_OF_CLASS = {'trimul_proj': 'k_trimul_proj', 'trimul_contract': 'k_trimul_contract', 'trimul_out': 'k_trimul_out',
                   'pair_transition': 'k_pair_transition'}


def algorithmic_bytes(dims, B, N):
    """HBM bytes per launch each pair-stack kernel class cannot avoid: every
    [B,N,N,128] f32 tensor it consumes or produces once (split f16 pairs are 4 bytes too)."""
    t = 4.0 * B * N * N * dims['c_p']
    return {'trimul_proj': 3 * t, 'trimul_contract': 3 * t, 'trimul_out': 3 * t, 'pair_transition': 2 * t}


def measured_traffic(cls, math):
    """HBM bytes per launch from the committed PMC passes (None when this kernel was not profiled)."""
    try:
        d = json.load(open(PMC_TRAFFIC))[math]
    except (OSError, KeyError, ValueError):
        return None
    v = [k for name, k in d.items() if name.startswith(KERNEL_OF_CLASS[cls])]
    return sum(x['hbm_read_bytes'] + x['hbm_write_bytes'] for x in v) / len(v) if v else None


def algorithmic_flops(dims, B, N):
    """FLOP per launch of each MFMA kernel class."""
    M = B * N * N
    c = dims['c_p']
    return {
        'trimul_proj': 2.0 * M * c * 4 * dims['c_hidden_mul'],
        'trimul_contract': 2.0 * B * dims['c_hidden_mul'] * N ** 3,
        'trimul_out': 2.0 * M * c * c * 2,
        'pair_transition': 2.0 * M * c * c * dims['pair_transition_n'] * 2,
    }


def step_flops(dims, B, N):
    """Algorithmic FLOP per batch-step (2.17e12 at B=8, N=256)."""
    c, L = dims['c_p'], dims['n_pair_transform_layer']
    pair = B * N * N * L * (40 * c * c + 4 * N * c)
    return pair + B * N * N * (8 * 7.9e3 + 38e3) + B * N * 27.9e6


def host_cores():
    """CPU cores this process may actually use: affinity, capped by the cgroup
    quota (a GPU box exposes every core of the host but grants a share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get('GENIE_BENCH_CPU_THREADS')
    if env:
        n = int(env)
    elif n > 64:
        n = 16      # no quota visible on a many-core host: use the documented 1-GPU CPU share
    return n


def cpu_baseline(dims, B, N, seed):
    """One step of the same workload through the oracle on the host cores."""
    from oracle import genie_oracle as O
    torch.set_num_threads(host_cores())
    sd = pack.random_state_dict(dims, seed=0)
    f = O.empty_features([N] * B)
    g = torch.Generator().manual_seed(seed)
    trans = torch.randn(B, N, 3, generator=g)
    fr = O.prepare_features(f)
    rots = O.compute_frenet_frames(trans, fr['chain_index'], fr['residue_mask'])
    sched = O.setup_schedule(dims['n_timestep'])
    step = dims['n_timestep']
    ts = torch.full((B,), step, dtype=torch.int32)
    t0 = time.time()
    with torch.no_grad():
        z = O.denoiser_forward(sd, dims, rots, trans, ts, f, 'eigh')['z']
        O.p_sample_step(sched, step, 0.6, trans, z, torch.randn(B, N, 3, generator=g), fr)
    dt = time.time() - t0
    return {'value': 1.0 / dt, 'unit': 'batch-steps/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'1 reverse-loop step (denoiser + posterior + Frenet) at N={N}, batch={B}, fp32, '
                      f'oracle/genie_oracle.py with torch.linalg.eigh quaternions, {dt:.1f} s wall'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--length', type=int, default=256)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--profile-steps', type=int, default=3)
    ap.add_argument('--math', choices=['hx', 'f32'], default=None, help='pair-stack arithmetic (default: library default, hx)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    dist = world > 1 or os.environ.get('GENIE_BENCH_FORCE_DIST') == '1'     # (single-rank rehearsal of the RCCL path)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    # stdout carries exactly ONE line, the JSON: RCCL prints its version banner to stdout when the communicator is created, so
    # file descriptor 1 points at stderr until that line is written
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    if dist:
        import torch.distributed as td
        td.init_process_group('nccl', device_id=dev)

    dims = dict(pack.BASE_DIMS)
    B, N, T = args.batch, args.length, dims['n_timestep']
    K, W = args.steps, args.warmup
    assert K + W <= T, 'steps + warmup must fit in T'
    eng = GenieEngine(dims, pack.random_state_dict(dims, seed=0), dev, math=args.math)
    math = eng.math
    feats = F.convert_np_features_to_tensor(
        F.batchify_np_features([F.create_empty_np_features([N]) for _ in range(B)]), dev)
    eng.bind_features(feats)
    g = torch.Generator().manual_seed(42 + rank)
    P = max(1, args.profile_steps)
    noise = torch.randn(K + W + 1 + P, B, N, 3, generator=g).to(dev)   # draws of iterations 0..K+W(+P) (base.py:227,269)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize(dev)

    state = None
    if W > 0:
        tr, ro, _ = eng.sample_loop(noise, 0.6, first_step=T, last_step=T - W + 1)
        state = (tr, ro)
    else:
        tr = noise[0].clone()
        state = (tr, eng.frenet(tr))
    barrier()
    t0 = time.perf_counter()
    tr, ro, _ = eng.sample_loop(noise, 0.6, first_step=T - W, last_step=T - W - K + 1, state=state)
    barrier()
    dt = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        dt = float(tmax.item())
        gathered = [torch.empty_like(tr) for _ in range(world)]      # trivial result gather over xGMI
        td.all_gather(gathered, tr)
    finite = bool(torch.isfinite(tr).all().item())

    out = None
    if rank == 0:
        value = K * world / dt
        # per-kernel timing with HIP events on the launch stream (separate, untimed pass)
        eng.profile(True)
        s0 = T - W - K
        if s0 - P < 1:
            s0 = T
        eng.sample_loop(noise, 0.6, first_step=s0, last_step=s0 - P + 1, state=(tr.clone(), ro.clone()))
        torch.cuda.synchronize(dev)
        eng.profile(False)
        prof = eng.profile_read()
        flops = algorithmic_flops(dims, B, N)
        kern = {k: {'ms_per_launch': ms / max(cnt, 1), 'launches_per_step': cnt / P, 'ms_per_step': ms / P}
                for k, (ms, cnt) in prof.items() if cnt}
        dom = max((k for k in kern if k in flops), key=lambda k: kern[k]['ms_per_step'])
        mfma_peak = PEAK_F16_MFMA_TFLOPS if math == 'hx' else PEAK_FP32_MFMA_TFLOPS

        def price(cls):
            """Roofline of one pair-stack kernel class: the roof (HBM bytes or matrix FLOP) that takes longer bounds it."""
            launch_s = kern[cls]['ms_per_launch'] * 1e-3
            nbytes = algorithmic_bytes(dims, B, N)[cls]
            mfma_flop = flops[cls] * (3.0 if math == 'hx' else 1.0)            # matrix-pipe FLOP per launch
            t_mfma, t_hbm = mfma_flop / (mfma_peak * 1e12), nbytes / (PEAK_HBM_GBS * 1e9)
            if t_hbm >= t_mfma:
                ach = nbytes / launch_s / 1e9
                r = {'bound': 'hbm', 'kernel': cls, 'achieved': ach, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': ach / PEAK_HBM_GBS}
            else:
                ach = mfma_flop / launch_s / 1e12
                r = {'bound': 'mfma', 'kernel': cls, 'achieved': ach, 'peak': mfma_peak, 'unit': 'TFLOP/s', 'frac': ach / mfma_peak}
            r.update({'traffic': measured_traffic(cls, math), 'avg_launch_ms': kern[cls]['ms_per_launch'],
                      'algorithmic_bytes_per_launch': nbytes, 'algorithmic_flop_per_launch': flops[cls],
                      'matrix_flop_per_launch': mfma_flop, 'floor_ms': {'hbm': t_hbm * 1e3, 'mfma': t_mfma * 1e3}})
            return r

        roof = price(dom)
        other_roofs = {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in price(k).items()
                           if kk in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms')}
                       for k in flops if k in kern and k != dom}
        whole = step_flops(dims, B, N) * (K / dt) / 1e12
        out = {
            'metric': 'denoise-steps/sec (N=256, T=1000, batch=8)', 'value': value, 'unit': 'batch-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': dt / K * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if math == 'f32' else 'f32 (pair-stack products as 3 f16 MFMAs on 2xf16-split operands, f32 accumulate)',
            'data': 'synthetic', 'math': math,
            'config': {'workload': f'unconditional N={N}, T={T}, batch={B} per GPU, random-init base Denoiser '
                                   f'(15.7M params), scale=0.6, reverse-loop steps {T - W}..{T - W - K + 1}',
                       'parallelism': f'replica per GPU x{world}, no data-path collective'},
            'structure_steps_per_s': value * B,
            'whole_step_tflops': whole,
            'finite': finite, 'roofline': roof, 'other_pair_kernel_rooflines': other_roofs,
            'kernels': {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in kern.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(dims, B, N, 42)
            out['speedup_vs_cpu_baseline'] = value / out['cpu_baseline']['value']
        else:
            out['cpu_baseline'] = None
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == '__main__':
    main()
