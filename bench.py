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
host: 1 warm-up + 3 timed consecutive steps of the same workload, median.
At N=1 the same line also carries `value_f32` (the same K steps in the exact-f32
MFMA kernels), `hx_vs_f32` (distance of the two trajectories after those steps)
and `full_loop` (all T=1000 reverse steps of one batch, timed end to end), and `train_step` (BASELINE config 5's shape on one
GPU: forward + backward + Adam of the base model at N=256, batch 2, f32-grade arithmetic, 1 warm-up + 5 timed steps).
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
PMC_TRAFFIC = [os.path.join(ROOT, 'profiles', n) for n in ('r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json')]   # newest first
KERNEL_OF_CLASS = {'trimul_proj': 'k_trimul_proj', 'trimul_contract': 'k_trimul_contract', 'trimul_out': 'k_trimul_out',
                   'pair_transition': 'k_pair_transition', 'pair_fused_a': 'k_pair_fused<true, false, true>', 'pair_fused_b': 'k_pair_fused<false, true, true>'}


def algorithmic_bytes(dims, B, N):
    """HBM bytes per launch each pair-stack kernel class cannot avoid: every
    [B,N,N,128] f32 tensor it consumes or produces once (split f16 pairs are 4 bytes too)."""
    t = 4.0 * B * N * N * dims['c_p']
    # fused chains (pair_fused_kernels.hip): x and z in; z, a and b out -- the transition inside chain B moves nothing
    return {'trimul_proj': 3 * t, 'trimul_contract': 3 * t, 'trimul_out': 3 * t, 'pair_transition': 2 * t, 'pair_fused_a': 5 * t, 'pair_fused_b': 5 * t}


def measured_traffic(cls, math):
    """(HBM bytes per launch, source file) from the newest committed PMC passes that profiled this kernel (`kernels_sha` in the
    file names the csrc/ state it was taken on; a mismatch with the running library is reported as stale)."""
    for path in PMC_TRAFFIC:
        try:
            blob = json.load(open(path))
            d = blob[math]
        except (OSError, KeyError, ValueError):
            continue
        v = [k for name, k in d.items() if name.startswith(KERNEL_OF_CLASS[cls])]
        if v:
            return sum(x['hbm_read_bytes'] + x['hbm_write_bytes'] for x in v) / len(v), os.path.basename(path), blob.get('kernels_sha')
    return None, None, None


def kernels_sha():
    """sha256 over the sources of the kernels this bench runs (profiles/ files carry it so stale traffic numbers are visible).  The
    training path's own files (train.h, train_kernels.hip, genie_train.hip) hold no kernel of the sampling step and are left out."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'genie2_amd', 'csrc')
    for n in sorted(os.listdir(d)):
        if n in ('train.h', 'train_kernels.hip', 'train_layout_kernels.hip', 'genie_train.hip'):
            continue
        h.update(open(os.path.join(d, n), 'rb').read())
    return h.hexdigest()[:16]


def algorithmic_flops(dims, B, N):
    """FLOP per launch of each MFMA kernel class."""
    M = B * N * N
    c = dims['c_p']
    f = {
        'trimul_proj': 2.0 * M * c * 4 * dims['c_hidden_mul'],
        'trimul_contract': 2.0 * B * dims['c_hidden_mul'] * N ** 3,
        'trimul_out': 2.0 * M * c * c * 2,
        'pair_transition': 2.0 * M * c * c * dims['pair_transition_n'] * 2,
    }
    f['pair_fused_a'] = f['trimul_out'] + f['trimul_proj']                              # TriMul-out output -> TriMul-in projections
    f['pair_fused_b'] = f['trimul_out'] + f['pair_transition'] + f['trimul_proj']       # TriMul-in output -> transition -> next projections
    return f


def step_flops(dims, B, N):
    """Algorithmic FLOP per batch-step (2.17e12 at B=8, N=256)."""
    c, L = dims['c_p'], dims['n_pair_transform_layer']
    pair = B * N * N * L * (40 * c * c + 4 * N * c)
    return pair + B * N * N * (8 * 7.9e3 + 38e3) + B * N * 27.9e6


def host_cores():
    """(cores, how) this process may actually use: scheduler affinity, capped by the cgroup CPU quota (a GPU box exposes every
    core of the host but grants a share).  GENIE_BENCH_CPU_THREADS overrides."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    how = 'sched_getaffinity'
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    q = max(1, int(int(txt[0]) / int(txt[1]) + 0.5))
                    if q < n:
                        n, how = q, 'cgroup cpu.max quota'
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0 and max(1, int(q / per + 0.5)) < n:
                    n, how = max(1, int(q / per + 0.5)), 'cgroup cfs quota'
            break
        except (OSError, ValueError, IndexError):
            continue
    env = os.environ.get('GENIE_BENCH_CPU_THREADS')
    if env:
        n, how = int(env), 'GENIE_BENCH_CPU_THREADS'
    elif n > 64:
        # no quota visible on a many-core host: the pool documents 16 cores per GPU, so that is what the baseline gets
        n, how = 16, 'assumed: the documented 16-core share of a 1-GPU box (affinity shows %d, no cgroup quota)' % n
    return n, how


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(dims, B, N, seed, n_timed=3):
    """The same workload through the oracle (CPU restatement of the reference) on the host cores: consecutive reverse-loop
    steps T, T-1, ... of one batch, 1 untimed warm-up step then `n_timed` timed ones; value = 1 / median step time."""
    from oracle import genie_oracle as O
    cores, how = host_cores()
    torch.set_num_threads(cores)
    sd = pack.random_state_dict(dims, seed=0)
    f = O.empty_features([N] * B)
    g = torch.Generator().manual_seed(seed)
    trans = torch.randn(B, N, 3, generator=g)
    fr = O.prepare_features(f)
    rots = O.compute_frenet_frames(trans, fr['chain_index'], fr['residue_mask'])
    sched = O.setup_schedule(dims['n_timestep'])
    times = []
    with torch.no_grad():
        for i in range(1 + n_timed):
            step = dims['n_timestep'] - i
            ts = torch.full((B,), step, dtype=torch.int32)
            t0 = time.perf_counter()
            z = O.denoiser_forward(sd, dims, rots, trans, ts, f, 'eigh')['z']
            trans, rots = O.p_sample_step(sched, step, 0.6, trans, z, torch.randn(B, N, 3, generator=g), fr)
            times.append(time.perf_counter() - t0)
    timed = sorted(times[1:])
    med = timed[len(timed) // 2]
    return {'value': 1.0 / med, 'unit': 'batch-steps/s', 'cores': torch.get_num_threads(), 'cores_source': how,
            'cpu_model': cpu_model(), 'kind': 'port',
            'step_seconds': {'warmup': round(times[0], 2), 'timed': [round(x, 2) for x in times[1:]], 'median': round(med, 2)},
            'sample': f'{n_timed} consecutive reverse-loop steps (denoiser + posterior + Frenet) after 1 warm-up step at N={N}, '
                      f'batch={B}, fp32, oracle/genie_oracle.py with torch.linalg.eigh quaternions; median {med:.1f} s per step'}


def train_step_leg(eng, dims, dev, N=256, B=2, n_timed=5):
    """BASELINE config 5's shape on one GPU: one training step (genie_train_forward_backward: train-mode forward, loss, backward
    through the whole Denoiser; then genie_adam_step) of the base model on a synthetic batch, f32-grade arithmetic (fast_math 0:
    every f32 operand as three bf16 pieces, six MFMAs per product -- the reference trains in fp32, train.py:54-65).  1 warm-up +
    `n_timed` timed steps; then one more step with per-class HIP events for the roofline of the dominant class."""
    from genie2_amd.engine import adam_step
    sd = pack.random_state_dict(dims, seed=0)
    feats = F.convert_np_features_to_tensor(F.batchify_np_features([F.create_empty_np_features([N]) for _ in range(B)]), dev)
    eng.bind_features(feats)
    w = pack.flatten_state_dict(sd, dims).to(dev)
    g, m, v = torch.zeros_like(w), torch.zeros_like(w), torch.zeros_like(w)
    gen = torch.Generator().manual_seed(7)
    x0 = (torch.randn(B, N, 3, generator=gen) * 8).to(dev)
    z = torch.randn(B, N, 3, generator=gen).to(dev)
    s = torch.randint(1, dims['n_timestep'] + 1, (B,), generator=gen).int().to(dev)
    sched = pack.schedule_tensors(dims['n_timestep'])
    trans, rots = eng.q_sample(x0, z, sched['sqrt_alphas_cumprod'].to(dev)[s.long()], sched['sqrt_one_minus_alphas_cumprod'].to(dev)[s.long()])

    def one(it):
        out = eng.train_forward_backward(w, trans, rots, s, z, 1.0, grads=g, seed=it, fast_math=0)
        adam_step(w, g, m, v, 1e-4, it + 1)
        return out

    one(0)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for it in range(n_timed):
        out = one(1 + it)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / n_timed
    loss = float(out['weighted_loss'])
    eng.profile(True)
    one(1 + n_timed)
    torch.cuda.synchronize(dev)
    eng.profile(False)
    prof = {k: v for k, v in eng.profile_read().items() if k.startswith('train_') and v[1]}
    flop = float(eng.lib.genie_train_gemm_flop(eng._h))
    kern = {k: {'ms_per_step': round(ms, 3), 'launches_per_step': cnt} for k, (ms, cnt) in prof.items()}
    dom = max(prof, key=lambda k: prof[k][0])
    res = {'ms_per_step': dt * 1e3, 'steps': n_timed, 'warmup': 1, 'loss_finite': loss == loss and abs(loss) != float('inf'),
           'config': {'workload': f'training step, base Denoiser (15.7M params), synthetic batch: N={N}, batch={B}, train-mode dropout, '
                                  'forward + backward + Adam', 'arithmetic': 'fast_math 0: f32 operands as 3 bf16 pieces, 6 MFMAs per product, f32 accumulate'},
           'gemm_algorithmic_tflop_per_step': flop / 1e12, 'algorithmic_tflops': flop / dt / 1e12,
           'workspace_gib': eng.lib.genie_train_workspace_bytes(eng._h) / 2 ** 30, 'kept_gib': eng.lib.genie_train_kept_bytes(eng._h) / 2 ** 30,
           'kernels': kern}
    if dom == 'train_gemm':
        ach = 6.0 * flop / (prof[dom][0] * 1e-3) / 1e12
        res['roofline'] = {'bound': 'mfma', 'kernel': 'train_gemm (all GEMMs of the step)', 'achieved': ach, 'peak': PEAK_F16_MFMA_TFLOPS,
                           'unit': 'TFLOP/s', 'frac': ach / PEAK_F16_MFMA_TFLOPS, 'traffic': None,
                           'matrix_flop_per_step': 6.0 * flop, 'algorithmic_flop_per_step': flop, 'class_ms_per_step': prof[dom][0]}
    else:
        res['roofline'] = {'bound': 'hbm', 'kernel': dom, 'achieved': None, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': None, 'traffic': None,
                           'class_ms_per_step': prof[dom][0]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--length', type=int, default=256)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-legs', action='store_true', help='skip the other-arithmetic and full T=1000 legs (N=1 only)')
    ap.add_argument('--no-train-leg', action='store_true', help='skip the training-step leg (N=1 only)')
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
    noise = torch.randn(T, B, N, 3, generator=g).to(dev)   # draws of the whole loop: 1 initial + one per step but the last (base.py:227,269)

    if dist:
        # The process group comes up AFTER the engine (its streams, workspace, the noise): created after RCCL's own streams, a
        # default-priority second stream of the engine -- the structure net's two batch halves, DESIGN.md 4.6 -- shared a hardware
        # queue with the first and the step ran 9 % slower (101.7 vs 110.5 batch-steps/s at one rank).  The library now creates that
        # stream with a non-default priority, which cures it; the order here is kept as the belt to those braces.
        torch.cuda.synchronize(dev)
        td.init_process_group('nccl', device_id=dev)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize(dev)

    def timed_leg():
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides"""
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
        return time.perf_counter() - t0, tr, ro

    dt, tr, ro = timed_leg()
    if dist:
        from genie2_amd.distributed import gather_coordinates, max_over_ranks
        dt = max_over_ranks(dt, dev)
        gathered = gather_coordinates(tr)                            # trivial result gather over xGMI: [world * B, N, 3]
        assert gathered.shape[0] == world * B
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
            traffic, tsrc, tsha = measured_traffic(cls, math)
            r.update({'traffic': traffic, 'traffic_source': tsrc, 'traffic_stale': (tsha != kernels_sha()) if traffic else None,
                      'avg_launch_ms': kern[cls]['ms_per_launch'],
                      'algorithmic_bytes_per_launch': nbytes, 'algorithmic_flop_per_launch': flops[cls],
                      'matrix_flop_per_launch': mfma_flop, 'floor_ms': {'hbm': t_hbm * 1e3, 'mfma': t_mfma * 1e3}})
            return r

        roof = price(dom)
        other_roofs = {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in price(k).items()
                           if kk in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms')}
                       for k in flops if k in kern and k != dom}
        whole = step_flops(dims, B, N) * (K / dt) / 1e12
        # algorithmic pair-tensor passes of the launches this step actually made (+ 8 IPA reads + 1 pair_init write)
        nbytes_step = sum(algorithmic_bytes(dims, B, N)[k] * v['launches_per_step'] for k, v in kern.items() if k in flops) \
            + 9 * 4.0 * B * N * N * dims['c_p']
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
            'whole_step_hbm': {'algorithmic_gb': nbytes_step / 1e9, 'achieved_gbs': nbytes_step / (dt / K) / 1e9,
                               'frac_of_8tbs': nbytes_step / (dt / K) / 1e9 / PEAK_HBM_GBS,
                               'unfused_reference_gb': 109 * 4.0 * B * N * N * dims['c_p'] / 1e9},
            'finite': finite, 'roofline': roof, 'other_pair_kernel_rooflines': other_roofs,
            'kernels': {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in kern.items()},
        }
        if world == 1 and not args.no_extra_legs:
            # (1) the same K steps in the exact-f32 MFMA kernels, from the same noise: rate, and how far the two arithmetics are apart
            other = 'f32' if math == 'hx' else 'hx'
            eng.set_math(other)
            dt2, tr2, _ = timed_leg()
            eng.set_math(math)
            rms = float(tr.pow(2).mean().sqrt())
            dx = float((tr - tr2).abs().max())
            out['value_' + other] = K / dt2
            out['ms_per_step_' + other] = dt2 / K * 1e3
            out['hx_vs_f32'] = {'max_abs_dx_after_steps': dx, 'steps': K + W, 'coordinate_rms': rms, 'ratio': dx / rms,
                                'tolerance_ratio': 1e-4}
            # (2) the metric's whole job once: all T reverse steps of one batch, timed end to end (frames, draws and loop included)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            trf, _, _ = eng.sample_loop(noise, 0.6)
            torch.cuda.synchronize(dev)
            dtf = time.perf_counter() - t0
            out['full_loop'] = {'steps': T, 'seconds': dtf, 'batch_steps_per_s': T / dtf, 'structure_steps_per_s': T * B / dtf,
                                'finite': bool(torch.isfinite(trf).all().item()), 'math': math}
        if world == 1 and not args.no_extra_legs:
            # what dense f16 MFMA work THIS device sustains (the transition stage's instruction stream on every CU for 20 ms): the
            # roofline above is priced against the datasheet peak; under matrix load the chip holds its power budget, not its clock
            import ctypes as C
            tf, pms = C.c_double(0.0), C.c_double(0.0)
            if eng.lib.genie_probe_mfma(C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), 20.0, C.byref(tf), C.byref(pms)) == 0:
                out['roofline']['sustained_f16_mfma'] = {
                    'tflops': tf.value, 'probe_ms': pms.value, 'frac_of_peak': tf.value / PEAK_F16_MFMA_TFLOPS,
                    'kernel_frac_of_sustained': (roof['achieved'] / tf.value) if roof['bound'] == 'mfma' and math == 'hx' else None,
                    'what': 'genie_probe_mfma: 48 v_mfma_f32_32x32x16_f16 per wave and stage from LDS fragments + ReLU / split + barrier, '
                            '512 threads on every CU, random operands, about 20 ms'}
        if world == 1 and not args.no_extra_legs and not args.no_train_leg:
            out['train_step'] = train_step_leg(eng, dims, dev)
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
