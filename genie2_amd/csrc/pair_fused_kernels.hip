// Row-local chains of the pair transform net fused into ONE persistent kernel per chain (hx arithmetic, hx.h):
//
//   chain A (inside a block):   TriMul-outgoing output  ->  TriMul-incoming projections
//   chain B (across blocks):    TriMul-incoming output  ->  PairTransition + end-of-block mask  ->  next block's
//                               TriMul-outgoing projections
//   (pair_transform_net.py:109-119, modules/triangular_multiplicative_update.py:99-108, modules/pair_transition.py:48-56)
//
// Both chains act on one pair row at a time, so a wave keeps its 32-pair tile in registers from the x / z loads to the
// stores: z is read once and written once per chain, the transition's input never exists in HBM, and the matrix-bound
// transition runs inside the same kernel as the memory-bound output / projection phases.
//
// Orientation: lane = pair in EVERY phase.  Activations are always the B operand of v_mfma_f32_32x32x16_f16 and weights the
// A operand (rows = output features), so every result tile is D[feature][pair]: lane (p, h) holds, for its pair p, features
// (r & 3) + 8 (r >> 2) + 4 h of each 32-block.  Registers 8c' .. 8c' + 7 of such a tile are exactly the B fragment of k-chunk
// 2 ob + c' in the "chained" k order  k = 16 c + (e & 3) + 8 (e >> 2) + 4 h  (hx.h), so a result feeds the next GEMM with no
// data movement; the host packs every weight of these kernels with that k permutation, and x / z are loaded straight into it.
// LayerNorm is an in-lane reduction plus one exchange with lane ^ 32.
//
// Tile = 32 pairs along the contiguous index of the channel-major images: ROW tiles (b, i, j0..j0+31) for chain B, COLUMN tiles
// (b, i0..i0+31, j) for chain A (whose TriMul-outgoing contraction is launched with a and b swapped, so that it leaves x^T and
// every x / a / b access of both chains is a 128-B run).  z rows are moved through a per-wave LDS staging area in whole
// 256-B pieces in both directions (hx_zt_dma on the way in, fz_store_half on the way out).
//
// Stage stream per tile (32-KiB LDS-DMA stages, double buffered, one barrier each, 48 MFMAs per wave and stage):
//   O: W_z{0,1}  W_z{2,3}  W_g{0,1}  W_g{2,3}   |   T: n_hb hidden blocks (W1 block + W2 block)   |   P: 8 projection passes
#include <string.h>
#include "hx_fused.h"

#define FZ_LDS_BYTES (2 * HX_STAGE_BYTES + 6144 + 8 * HX_ZT_BYTES)
#ifndef FZ_XPRE
#define FZ_XPRE 0          // 1: chain B fetches half of the next tile's x between the MFMAs of its last transition stages (32 registers held through the projections: 92 spills, 0.877 ms); 0: all of x at the tile's start (55 spills, 0.848 ms)
#endif
#ifndef FZ_ASYM
#define FZ_ASYM 1          // 1: a stage's LDS-DMA pieces are all issued by waves 4..7 (their SIMD partners start the stage's MFMAs at once); 0: four pieces per wave.  +0.5 % on the step
#endif
#ifndef FZ_BIASPRE
#define FZ_BIASPRE 1       // 1: a transition stage's initial accumulators (b1 block) are fetched under the previous stage's second GEMM (their registers are dead there): removes the kernels' last spills
#endif
#ifndef FZ_KO
#define FZ_KO 0            // developer knock-outs of the transition stage (timing only, results wrong): 1 no ReLU / split, 2 no fragment re-reads, 4 no barrier / wait, 8 no weight DMA
#endif
#ifndef FZ_ZEARLY
#define FZ_ZEARLY 0        // 1: the first channel half of z is requested at the tile's start (behind the x loads), not inside the first stage.  Measured: no gain -- the x loads take as much longer as the first stage gets shorter (the tile's bytes, not their latency, set both)
#endif

// developer builds (-DFZ_TS): s_memtime stamps of every wave of work-group 0, read with tests/devtools/ts_fused.py; none in the product
#ifdef FZ_TS
__device__ unsigned long long g_fz_ts[16][2048];      // [variant * 8 + wave]
extern "C" int genie_fz_debug_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fz_ts), sizeof(unsigned long long) * 16 * 2048); }
#define FZ_TS_DECL() const bool ts_on = blockIdx.x == 0 && HAS_P;      /* (the last block's chain without projections would overwrite chain B's slots) */ unsigned long long* ts_p = g_fz_ts[(HAS_T ? 8 : 0) + (threadIdx.x >> 6)]; int ts_n = 0
#define FZ_STAMP() do { if (ts_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0 && ts_n < 2048) ts_p[ts_n] = t_; ++ts_n; } } while (0)
#else
#define FZ_TS_DECL()
#define FZ_STAMP() do { } while (0)
#endif
#if defined(FZ_TS) && FZ_TS == 2
#define FZ_STAMP2() FZ_STAMP()
#else
#define FZ_STAMP2() do { } while (0)
#endif
#if defined(FZ_TS) && FZ_TS == 3
#define FZ_STAMP3() FZ_STAMP()
#else
#define FZ_STAMP3() do { } while (0)
#endif

template <bool COL, bool HAS_T, bool HAS_P = true>
__global__ __launch_bounds__(512, 1) void k_pair_fused(const FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    float* sb = reinterpret_cast<float*>(smb + 2 * HX_STAGE_BYTES);
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int N = A.N, NP = A.NP, n_wtiles = A.n_wtiles;
    const int ntile = (N + 31) >> 5;
    const int n_tiles = (n_wtiles + 7) / 8;
    const int rev = A.rev;
    const int n_hb = HAS_T ? A.n_hb : 0;
    const int PBASE = 4 + n_hb;                              // first projection stage of the image
    const rsrc_t rw = hx_rsrc(A.wimg, (unsigned)((PBASE + (HAS_P ? 8 : 0)) * HX_STAGE_BYTES));
    const rsrc_t rx = hx_rsrc(A.xcm, A.cm_bytes), rz = hx_rsrc(A.z, A.z_bytes);
    const rsrc_t ra = hx_rsrc(A.acm, A.cm_bytes), rb = hx_rsrc(A.bcm, A.cm_bytes);
    const int lane16 = lane * 16;
    const int sstride = NP * NP * 4;                         // bytes per channel of a channel-major image
    const int zstride = COL ? N * 512 : 512;                 // bytes between consecutive pairs of a tile
    auto issue = [&](int s, int buf) {
#if FZ_ASYM
        // the whole stage is requested by waves 4..7 (8 pieces each): their SIMD partners 0..3 start the stage's MFMAs at once, and the
        // two waves of a SIMD stay half a phase apart for the rest of the stage -- one's VALU phases sit beside the other's MFMAs
        if (wave >= 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int p = 8 * (wave - 4) + q;
                hx_dma(rw, smb + buf * HX_STAGE_BYTES + p * 1024, lane16, s * HX_STAGE_BYTES + p * 1024);
            }
        }
#else
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = 4 * wave + q;
            hx_dma(rw, smb + buf * HX_STAGE_BYTES + p * 1024, lane16, s * HX_STAGE_BYTES + p * 1024);
        }
#endif
    };
    unsigned char* zt = smb + 2 * HX_STAGE_BYTES + 6144 + wave * HX_ZT_BYTES;
    // wave-tile 8 tile + wave (clamped): z byte offset of its first row, valid rows, channel-major element offset of (channel 0, pair 0)
    auto tile_geom = [&](int tile, int& zsoff, int& nv, int& cmoff) {
        const int wt = min(HX_PHYS(tile) * 8 + wave, n_wtiles - 1);
        const int st = wt % ntile, line = (wt / ntile) % N, b = wt / (ntile * N);
        nv = min(32, N - st * 32);
        zsoff = COL ? ((b * N + st * 32) * N + line) * 512 : ((b * N + line) * N + st * 32) * 512;
        cmoff = ((b * 128 * NP + line) * NP + st * 32) * 4;
    };
    // (An L2 "touch" of the next tile's remaining lines -- one dword per 128-B line, 64 lines per wave instruction -- was tried
    //  and cost 12 k cycles in the pass that waited for it: 64 lines in as many pages per instruction.)
    const int vx = (4 * h * NP * NP + pl) * 4;              // x_cm voffset: channel block of this half-wave, pair pl
    // x of chunk c: raw[2c] = channels 16c + 4h + {0..3}, raw[2c+1] = 16c + 8 + 4h + {0..3}
#define FZ_XLOAD(c, xoff) do {                                                                                              \
        raw[2 * (c)].x = hx_load(rx, vx, (xoff) + (16 * (c) + 0) * sstride); raw[2 * (c)].y = hx_load(rx, vx, (xoff) + (16 * (c) + 1) * sstride); \
        raw[2 * (c)].z = hx_load(rx, vx, (xoff) + (16 * (c) + 2) * sstride); raw[2 * (c)].w = hx_load(rx, vx, (xoff) + (16 * (c) + 3) * sstride); \
        raw[2 * (c) + 1].x = hx_load(rx, vx, (xoff) + (16 * (c) + 8) * sstride); raw[2 * (c) + 1].y = hx_load(rx, vx, (xoff) + (16 * (c) + 9) * sstride); \
        raw[2 * (c) + 1].z = hx_load(rx, vx, (xoff) + (16 * (c) + 10) * sstride); raw[2 * (c) + 1].w = hx_load(rx, vx, (xoff) + (16 * (c) + 11) * sstride); \
    } while (0)

    // one dword of the same image: element j (0..7) of chunk c
    auto xld1 = [&](float4 (&raw)[16], int c, int j, int xoff) {
        const float t = hx_load(rx, vx, xoff + (16 * c + 8 * (j >> 2) + (j & 3)) * sstride);
        float4& d = raw[2 * c + (j >> 2)];
        if ((j & 3) == 0) d.x = t; else if ((j & 3) == 1) d.y = t; else if ((j & 3) == 2) d.z = t; else d.w = t;
    };
    int tile = blockIdx.x;
    FZ_TS_DECL();
    if (A.stagger > 0) {        // work-groups start out of phase (experiment: GENIE_FZ_STAGGER = cycles between the first and the last group of eight)
        const long long t0 = (long long)__builtin_amdgcn_s_memtime();
        const long long wait = (long long)A.stagger * ((blockIdx.x >> 3) & 7) / 8;
        while ((long long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
    issue(0, 0);
    for (int u = threadIdx.x; u < FZ_SB_FLOATS; u += 512) {
        float v = 0.f;
        if (u < FZ_SB_BG) v = A.bzs[u];
        else if (u < FZ_SB_B1) v = A.bgs[u - FZ_SB_BG];
        else if (u < FZ_SB_B2) { if (HAS_T && u - FZ_SB_B1 < n_hb * 32) v = A.b1s[u - FZ_SB_B1]; }
        else if (u < FZ_SB_BP) { if (HAS_T) v = A.b2s[u - FZ_SB_B2]; }
        else if (HAS_P) v = A.bproj[u - FZ_SB_BP];
        sb[u] = v;
    }
    // Input prefetch, one tile ahead: the first channel half of a tile's z rows (staging area, requested behind projection passes
    // 4 and 5) and, in chain B, the first half of its x (chunks 0..3, 32 registers, one dword behind each MFMA group of the last four
    // transition stages -- the one phase without memory traffic of its own).  The rest is requested at the tile's start.
    float4 raw[16];
    {
        int zs, nv, cm;
        tile_geom(tile, zs, nv, cm);
        if (HAS_T && FZ_XPRE) {
#pragma unroll
            for (int c = 0; c < 4; ++c) FZ_XLOAD(c, cm);
        }
        hx_zt_dma(rz, zt, lane, zs, zstride, nv, 1);
    }
    hx_stage_landed();
    __syncthreads();

#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = HX_PHYS(tile) * 8 + wave;
        const bool act = wt_raw < n_wtiles;
        const int wt = act ? wt_raw : n_wtiles - 1;          // idle waves shadow the last tile (stores dropped)
        const int st = wt % ntile, line = (wt / ntile) % N, b = wt / (ntile * N);
        const int t0i = st * 32;
        const int nvalid = act ? min(32, N - t0i) : 0;
        int zsoff, znv, cmoff;
        tile_geom(tile, zsoff, znv, cmoff);
        // hipcc hoists loop-invariant LDS loads (the accumulators' initial biases: ~60 values) out of the tile loop and then spills
        // them; an offset it can not see through keeps them where they are used
        int bo = 0;
        asm volatile("" : "+v"(bo));
        const float* sbt = sb + bo;
        int lane_t = lane;                                    // likewise for the ~40 swizzled staging addresses derived from the lane id
        asm volatile("" : "+v"(lane_t));
        const int pl_t = lane_t & 31, h_t = lane_t >> 5;
        // the rest of the tile's x (chunks 4..7: their lines were touched into L2 by the previous tile, fz_touch_next)
        FZ_STAMP();
#pragma unroll
        for (int c = (HAS_T && FZ_XPRE) ? 4 : 0; c < 8; ++c) FZ_XLOAD(c, cmoff);      // (chain A: all of x here -- no phase of it has room to fetch ahead)
#if FZ_ZEARLY
        // z: its second channel half (prefetched by the previous tile) moves from the staging area to registers now, and the first half is
        // requested right behind the x loads: it lands under the LayerNorm of x and the first stage instead of being waited for at that
        // stage's end
        float4 rz1[8];                                        // chunks 4..7
#pragma unroll
        for (int q = 0; q < 4; ++q) fz_zt_chunk(rz1[2 * q], rz1[2 * q + 1], zt, pl_t, h_t, q);
        hx_lds_done();
        hx_zt_dma(rz, zt, lane_t, zsoff, zstride, znv, 0);
#endif
        const float msk = (pl < nvalid) ? A.rmask[b * N + line] * A.rmask[b * N + t0i + pl] : 0.f;
        const bool more = tile + (int)gridDim.x < n_tiles;
        int n_zsoff = 0, n_nv = 1, n_cmoff = 0;
        if (more) tile_geom(tile + gridDim.x, n_zsoff, n_nv, n_cmoff);
#if FZ_ZEARLY
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // x has landed: at most the 8 z pieces (and the mask loads) are younger
        __builtin_amdgcn_sched_barrier(0);
#else
        hx_vm_done();
#endif
        FZ_STAMP();
        f32x16 v[4];                                          // the tile's running value: update -> z' -> (z'' accumulators) -> z''

        // ------------------------------------------------------------------ O: z' = z + (W_z LN(x) + b_z) sigmoid(W_g LN(z) + b_g)
        {
            h8 xh[8], xl[8];
            hx_norm_split(xh, xl, raw, A.sx);                 // (raw = x; dead from here on)
            PIPE_FENCE();
            FZ_STAMP3();
            // z: its second channel half (prefetched by the previous tile) goes to registers, its first half is then fetched into the
            // staging area and STAYS there until the gates are done -- chunks of it are read where they are used (32 live registers
            // instead of 64 through the tightest stages).
#if !FZ_ZEARLY
            float4 rz1[8];                                    // chunks 4..7
#endif
            float zsh = 0.f, zs1 = 0.f, zs2 = 0.f;            // shifted sums for the LayerNorm statistics
#pragma unroll
            for (int half = 0; half < 2; ++half) {            // stages 0, 1: update accumulators of channel blocks 2 half, 2 half + 1
                issue(half + 1, half ^ 1);
                const unsigned char* stage = smb + half * HX_STAGE_BYTES;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[2 * half][r] = sbt[FZ_SB_BZ + (2 * half) * 32 + acc_row(r, lane)];
                    v[2 * half + 1][r] = sbt[FZ_SB_BZ + (2 * half + 1) * 32 + acc_row(r, lane)];
                }
                h8 f0 = hx_frag(stage, 0, 0, lane), f0l = hx_frag(stage, 0, 1, lane), f1 = hx_frag(stage, 8, 0, lane),
                   f1l = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const int kn = min(kc + 1, 7);
                    const h8 n0 = hx_frag(stage, kn, 0, lane), n0l = hx_frag(stage, kn, 1, lane), n1 = hx_frag(stage, 8 + kn, 0, lane),
                             n1l = hx_frag(stage, 8 + kn, 1, lane);
                    PIPE_FENCE();
                    MFH3(f0, f0l, xh[kc], xl[kc], v[2 * half]);
                    MFH3(f1, f1l, xh[kc], xl[kc], v[2 * half + 1]);
                    PIPE_FENCE();
                    f0 = n0; f0l = n0l; f1 = n1; f1l = n1l;
#if !FZ_ZEARLY
                    if (half == 0 && kc == 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) fz_zt_chunk(rz1[2 * q], rz1[2 * q + 1], zt, pl_t, h_t, q);
                    }
                    if (half == 0 && kc == 2) { hx_lds_done(); hx_zt_dma(rz, zt, lane_t, zsoff, zstride, znv, 0); }   // first channel half of z
#endif
                    // LayerNorm statistics of z between this stage's MFMAs, as sums shifted by the row's first element
                    if (half == 1 && kc == 1) {
                        zsh = __shfl(rz1[0].x, lane & 31);         // (the same shift in both half-waves of a row)
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float a = rz1[q].x - zsh, bq = rz1[q].y - zsh, c = rz1[q].z - zsh, d = rz1[q].w - zsh;
                            zs1 += (a + bq) + (c + d); zs2 += (a * a + bq * bq) + (c * c + d * d);
                        }
                    }
                    if (half == 1 && kc >= 2 && kc <= 5) {          // (the first half landed: stage 0 ended with vmcnt(0))
                        float4 ta, tb;
                        fz_zt_chunk(ta, tb, zt, pl_t, h_t, kc - 2);
                        const float a = ta.x - zsh, bq = ta.y - zsh, c = ta.z - zsh, d = ta.w - zsh;
                        const float e = tb.x - zsh, f = tb.y - zsh, g = tb.z - zsh, hh = tb.w - zsh;
                        zs1 += ((a + bq) + (c + d)) + ((e + f) + (g + hh));
                        zs2 += ((a * a + bq * bq) + (c * c + d * d)) + ((e * e + f * f) + (g * g + hh * hh));
                    }
                }
                hx_stage_landed();
                hx_stage_barrier();
                FZ_STAMP();
            }
            zs1 += __shfl_xor(zs1, 32); zs2 += __shfl_xor(zs2, 32);
            const float zm_s = zs1 * (1.0f / 128.0f);                                  // mean - shift
            const float zmean = zsh + zm_s;
            const float zsc = A.sx / sqrtf(fmaxf(zs2 * (1.0f / 128.0f) - zm_s * zm_s, 0.f) + GENIE_LN_EPS);
            FZ_STAMP3();
#pragma unroll
            for (int half = 0; half < 2; ++half) {            // stages 2, 3: gates of channel blocks 2 half, 2 half + 1, then z'
                if (half == 0) issue(3, 1);
                else issue(4, 0);                              // first transition / projection stage
                const unsigned char* stage = smb + half * HX_STAGE_BYTES;
                f32x16 ga, gb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ga[r] = sbt[FZ_SB_BG + (2 * half) * 32 + acc_row(r, lane)];
                    gb[r] = sbt[FZ_SB_BG + (2 * half + 1) * 32 + acc_row(r, lane)];
                }
                h8 f0 = hx_frag(stage, 0, 0, lane), f0l = hx_frag(stage, 0, 1, lane), f1 = hx_frag(stage, 8, 0, lane),
                   f1l = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const int kn = min(kc + 1, 7);
                    const h8 n0 = hx_frag(stage, kn, 0, lane), n0l = hx_frag(stage, kn, 1, lane), n1 = hx_frag(stage, 8 + kn, 0, lane),
                             n1l = hx_frag(stage, 8 + kn, 1, lane);
                    float4 ca, cb;
                    if (kc < 4) fz_zt_chunk(ca, cb, zt, pl_t, h_t, kc);
                    else { ca = rz1[2 * (kc - 4)]; cb = rz1[2 * (kc - 4) + 1]; }
                    const float xz[8] = {ca.x - zmean, ca.y - zmean, ca.z - zmean, ca.w - zmean, cb.x - zmean, cb.y - zmean, cb.z - zmean, cb.w - zmean};
                    h8 sh, sl;
                    hx_split8(xz, zsc, sh, sl);
                    PIPE_FENCE();
                    MFH3(f0, f0l, sh, sl, ga);
                    MFH3(f1, f1l, sh, sl, gb);
                    PIPE_FENCE();
                    f0 = n0; f0l = n0l; f1 = n1; f1l = n1l;
                }
                float4 res[8];                              // the residual rows of this stage's 64 channels: chunks 4 half .. 4 half + 3
                if (half == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) fz_zt_chunk(res[2 * q], res[2 * q + 1], zt, pl_t, h_t, q);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) res[q] = rz1[q];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float g0 = A.cz * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ga[r] * A.cgo));
                    const float g1 = A.cz * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gb[r] * A.cgo));
                    // residual: channel block ob = 2 half + {0, 1}, register r <-> local chunk 2 {0, 1} + (r >> 3), slot r & 7
                    const float4 za = res[2 * (r >> 3) + ((r & 7) >> 2)], zb = res[2 * (2 + (r >> 3)) + ((r & 7) >> 2)];
                    const float z0 = (r & 3) == 0 ? za.x : (r & 3) == 1 ? za.y : (r & 3) == 2 ? za.z : za.w;
                    const float z1 = (r & 3) == 0 ? zb.x : (r & 3) == 1 ? zb.y : (r & 3) == 2 ? zb.z : zb.w;
                    v[2 * half][r] = fmaf(v[2 * half][r], g0, z0);
                    v[2 * half + 1][r] = fmaf(v[2 * half + 1][r], g1, z1);
                }
                hx_stage_landed();
                hx_stage_barrier();
                FZ_STAMP();
            }
        }

        h8 zh[8], zl[8];
        // ------------------------------------------------------------------ T: z'' = (z' + W2 relu(W1 LN(z') + b1) + b2) mask
        if constexpr (HAS_T) {
            float mean, sc;
            fz_stats(v, A.sx, mean, sc);
            fz_split_tile(zh, zl, v, mean, sc);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)                    // the residual and b2 are the accumulators' initial value (1 / c2 is a power of two)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[ob][r] = fmaf(v[ob][r], A.inv_c2, sbt[FZ_SB_B2 + ob * 32 + acc_row(r, lane)]);
#if FZ_BIASPRE
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = sbt[FZ_SB_B1 + acc_row(r, lane)];
#endif
            // one hidden block; XC >= 0: the stage also requests chunk XC of the next tile's x, one dword behind each MFMA group of its
            // first GEMM (the transition has no memory traffic of its own; a burst at a stage's end would not be overlapped)
            auto t_stage = [&](int hb, auto xc_tag) {
                constexpr int XC = decltype(xc_tag)::value;
                if ((FZ_KO & 8) && hb + 1 < n_hb) { }
                else if (HAS_P || hb + 1 < n_hb) issue(4 + hb + 1, (hb + 1) & 1);     // (hb = n_hb - 1: the first projection stage ...
                else if (more) issue(0, 0);                                      //  ... or, in the chain without projections, the next tile's first stage)
                const unsigned char* stage = smb + (hb & 1) * HX_STAGE_BYTES;
#if !FZ_BIASPRE
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = sbt[FZ_SB_B1 + hb * 32 + acc_row(r, lane)];
#endif
                {
                    h8 wh = hx_frag(stage, 0, 0, lane), wl = hx_frag(stage, 0, 1, lane);
#pragma unroll
                    for (int kc = 0; kc < 8; ++kc) {
                        const h8 nh = (FZ_KO & 2) ? wh : hx_frag(stage, min(kc + 1, 7), 0, lane), nl = (FZ_KO & 2) ? wl : hx_frag(stage, min(kc + 1, 7), 1, lane);
                        PIPE_FENCE();
                        MFH3(wh, wl, zh[kc], zl[kc], d);
                        PIPE_FENCE();
                        wh = nh; wl = nl;
                        if (FZ_XPRE && XC >= 0 && more) xld1(raw, XC < 0 ? 0 : XC, kc, n_cmoff);
                    }
                }
                h8 ah[2], al[2];
                if (FZ_KO & 1) { ah[0] = zh[0]; al[0] = zl[0]; ah[1] = zh[1]; al[1] = zl[1]; }
                else
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = fmaxf(d[8 * c + e], 0.f);
                    hx_split8(x, A.c1, ah[c], al[c]);
                }
                {
                    h8 bh = hx_frag(stage, 8, 0, lane), bl = hx_frag(stage, 8, 1, lane);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const h8 nh = (FZ_KO & 2) ? bh : hx_frag(stage, 8 + min(u + 1, 7), 0, lane), nl = (FZ_KO & 2) ? bl : hx_frag(stage, 8 + min(u + 1, 7), 1, lane);
                        PIPE_FENCE();
                        MFH3(bh, bl, ah[u >> 2], al[u >> 2], v[u & 3]);
                        PIPE_FENCE();
                        bh = nh; bl = nl;
#if FZ_BIASPRE
                        if (u == 1) {       // d is dead since the split: the next hidden block's biases (block n_hb reads the b2 slots: unused)
#pragma unroll
                            for (int r = 0; r < 16; ++r) d[r] = sbt[FZ_SB_B1 + (hb + 1) * 32 + acc_row(r, lane)];
                        }
#endif
                    }
                }
                if (FZ_XPRE && XC >= 0 && more) {
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // everything older than this stage's 8 x loads: the next stage's weights
                    __builtin_amdgcn_sched_barrier(0);
                } else if (!((FZ_KO & 4) && hb + 1 < n_hb))
                    hx_stage_landed();
                if (!((FZ_KO & 4) && hb + 1 < n_hb)) hx_stage_barrier();
            };
            const int n_plain = (FZ_XPRE && n_hb >= 4) ? n_hb - 4 : n_hb;
#pragma unroll 1
            for (int hb = 0; hb < n_plain; ++hb) t_stage(hb, std::integral_constant<int, -1>{});
            if (FZ_XPRE && n_hb >= 4) {
                t_stage(n_hb - 4, std::integral_constant<int, 0>{});
                t_stage(n_hb - 3, std::integral_constant<int, 1>{});
                t_stage(n_hb - 2, std::integral_constant<int, 2>{});
                t_stage(n_hb - 1, std::integral_constant<int, 3>{});
            }
            FZ_STAMP();
            const float m2 = msk * A.c2;
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[ob][r] *= m2;
        }

        // ------------------------------------------------------------------ z leaves (its only write), LayerNorm for the projections
        fz_store_half(rz, zt, v[0], v[1], lane_t, zsoff, zstride, nvalid, 0);
        FZ_FULL_WAIT(128);
        FZ_STAMP3();
        fz_store_half(rz, zt, v[2], v[3], lane_t, zsoff, zstride, nvalid, 1);
        FZ_FULL_WAIT(64);
        FZ_STAMP3();
        if constexpr (HAS_P) {
            float mean, sc;
            fz_stats(v, A.sx, mean, sc);
            fz_split_tile(zh, zl, v, mean, sc);
        }
        hx_lds_done();                                        // the staging area is free for the next tile's z
        FZ_STAMP();
        if constexpr (!HAS_P) {                               // (last block: no projections follow; the next tile's z half is requested here)
            if (more) hx_zt_dma(rz, zt, lane_t, n_zsoff, zstride, n_nv, 1);
        }

        // ------------------------------------------------------------------ P: a = (W_ap zn + b) sigmoid(W_ag zn + b) mask, b likewise
        if constexpr (HAS_P) {
            const float* sbias = sbt + FZ_SB_BP;
            const float cg = A.cg;
            const float ma = msk * A.cpa, mb = msk * A.cpb;
            const int voff = (pl < nvalid) ? (4 * h * NP * NP + pl) * 4 : FZ_OOR;
            const int sbase = ((b * 128 * NP + line) * NP + t0i) * 4;
            unsigned wst[8];
            f32x16 apA, agA, apB, agB;
#pragma unroll
            for (int r = 0; r < 16; ++r) { apB[r] = 0.f; agB[r] = 0.f; apA[r] = sbias[acc_row(r, lane)]; agA[r] = sbias[256 + acc_row(r, lane)]; }
#pragma unroll 1
            for (int pp = 0; pp < 4; ++pp) {
                {   // even pass 2pp -> set A; epilogue of pass 2pp - 1 (set B; nothing for pp = 0: stores dropped)
                    const int pass = 2 * pp;
                    issue(PBASE + pass + 1, 1);
                    const unsigned char* stage = smb;
                    const rsrc_t e_rd = pp > 2 ? rb : ra;
                    const int e_voff = pp == 0 ? FZ_OOR : voff;
                    const int e_so = sbase + ((pass - 1) & 3) * 32 * sstride;
                    const float e_pm = pp > 2 ? mb : ma;
                    HX_PROJ_STAGE(apA, agA, apB, agB);
                    FZ_STAMP2();
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    FZ_STAMP2();      // the next stage's weights (and what was requested a stage ago) have landed
                    if (more) {                                // next tile: x chunks 0..3 behind passes 4, 5; z half 0 behind passes 6, 7
                        if (pp == 2) hx_zt_dma(rz, zt, lane_t, n_zsoff, zstride, n_nv, 1, 0, 4);
                    }
                    hx_stage_barrier();
                    FZ_STAMP();
                }
                {   // odd pass 2pp + 1 -> set B; epilogue of pass 2pp (set A)
                    const int pass = 2 * pp + 1;
                    if (pp < 3) issue(PBASE + pass + 1, 0);
                    else if (more) issue(0, 0);
                    const unsigned char* stage = smb + HX_STAGE_BYTES;
                    const rsrc_t e_rd = pp < 2 ? ra : rb;
                    const int e_voff = voff;
                    const int e_so = sbase + ((pass - 1) & 3) * 32 * sstride;
                    const float e_pm = pp < 2 ? ma : mb;
                    HX_PROJ_STAGE(apB, agB, apA, agA);
                    FZ_STAMP2();
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    FZ_STAMP2();
                    if (more) {
                        if (pp == 2) hx_zt_dma(rz, zt, lane_t, n_zsoff, zstride, n_nv, 1, 4, 4);
                    }
                    hx_stage_barrier();
                    FZ_STAMP();
                }
            }
            {   // drain: epilogue of pass 7 (set B)
                const rsrc_t e_rd = rb;
                const int e_voff = voff, e_so = sbase + 3 * 32 * sstride;
                const float e_pm = mb;
                float t[16], u[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    HX_PROJ_PIECE_A(agB, r, t[r]);
                    HX_PROJ_PIECE_B(apB, r, t[r], u[r]);
                }
                PIPE_FENCE();
#pragma unroll
                for (int r = 0; r < 16; ++r) HX_PROJ_PIECE_C_NOW(r, t[r], u[r]);
            }
            FZ_STAMP();
        }
    }
#undef FZ_XLOAD
}

// ---------------------------------------------------------------------------------------------- host side
static int fz_num_cu() {
    static int n = 0;
    if (!n) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&pr, dev); n = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    return n;
}

void pair_fused_kernels_init() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_fused<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, FZ_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_fused<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, FZ_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_fused<false, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, FZ_LDS_BYTES);
}

// chain A of block `w` (col = true):  output of w.out (x^T in xcm) -> projections of w.in
// chain B (col = false):              output of w.in -> transition of w -> projections of next->out
// p == nullptr: the last block's chain (output of w.in -> transition), no projections
void launch_pair_fused(genie_ctx* h, hipStream_t st, const HxFusedW& f, const HxTriW& o, const HxTransW* t, const HxTriW* pp, bool col) {
    const HxTriW& p = pp ? *pp : o;
    const int N = h->N, NP = h->NP, ntile = (N + 31) / 32;
    FusedArgs a;
    a.z = h->p; a.xcm = h->xcm; a.rmask = h->rmaskf; a.wimg = f.img;
    a.bzs = o.bzs; a.bgs = o.bgs; a.b1s = t ? t->b1s : nullptr; a.b2s = t ? t->b2s : nullptr; a.bproj = p.bias_proj;
    a.acm = reinterpret_cast<unsigned*>(h->acm); a.bcm = reinterpret_cast<unsigned*>(h->bcm);
    a.N = N; a.NP = NP; a.n_wtiles = h->B * N * ntile; a.n_hb = t ? h->d.pair_transition_n * 4 : 0;
    a.cm_bytes = (unsigned)((size_t)h->B * 128 * NP * NP * 4);
    a.z_bytes = (unsigned)((size_t)h->B * N * N * 512);
    a.sx = o.sx; a.cgo = o.cgo; a.cz = o.cz;
    a.c1 = t ? t->c1 : 0.f; a.c2 = t ? t->c2 : 1.f; a.inv_c2 = t ? 1.0f / t->c2 : 1.f;
    a.cpa = p.cpa; a.cpb = p.cpb; a.cg = p.cg;
    a.rev = (int)(h->hx_launches++ & 1);
    { static const int stagger = [] { const char* e = getenv("GENIE_FZ_STAGGER"); return e ? atoi(e) : 0; }(); a.stagger = stagger; }      // developer knob, read once
    const long long n_tiles = ((long long)a.n_wtiles + 7) / 8;
    const unsigned grid = (unsigned)(n_tiles < fz_num_cu() ? n_tiles : fz_num_cu());
    if (!pp) hipLaunchKernelGGL((k_pair_fused<false, true, false>), dim3(grid), dim3(512), FZ_LDS_BYTES, st, a);
    else if (col) hipLaunchKernelGGL((k_pair_fused<true, false>), dim3(grid), dim3(512), FZ_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((k_pair_fused<false, true>), dim3(grid), dim3(512), FZ_LDS_BYTES, st, a);
}
