// Pair-stack GEMM kernels, "WL" form: activations in registers, weights through LDS.
//
// Measured on MI355X (profiles/): when every wave streams its own copy of
// the packed weights from L2 (the first, "WI" form) the weight stream itself is the limiter --
// removing it took trimul_proj from 91 to 142 TFLOP/s.  Here the four waves of a work-group
// share ONE copy: each 32-fragment (32 KiB) stage of the weight stream is pulled from L2 once per
// work-group with LDS-DMA (global_load_lds_dwordx4: no VGPRs, 1 KiB lane-linear per instruction,
// which is exactly the packed fragment order), double buffered, one barrier per stage
// (= per 128 MFMAs of every wave).  L2 weight traffic drops 4x; fragment reads become
// conflict-free ds_read_b128.
// Each WAVE still owns one 32-pair tile end to end: the activation tile sits in MFMA fragment
// registers (lane (p, h): row p, k = 8kb + 4h..+3), LayerNorm in registers (affine folded into
// the weights, genie_api.hip fold_ln), chained GEMMs feed accumulators straight back as operands.
#include <stdlib.h>
#include "common.h"

#define STAGE_FRAGS 32
#define STAGE_FLOATS (STAGE_FRAGS * 256)      // 32 KiB
#define WL_LDS_BYTES (2 * STAGE_FLOATS * 4 + 2048)   // double buffer + bias scratch: 66 KiB -> 2 work-groups per CU

__device__ __forceinline__ float fast_sigmoid_(float x) { return __frcp_rn(1.0f + __expf(-x)); }

__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ v4f sfrag(const float* stage, int u, int lane) {
    return *reinterpret_cast<const v4f*>(stage + (u * 64 + lane) * 4);
}

// MFMA groups over 8 k-values with the accumulators interleaved, so that no MFMA depends on the
// one issued right before it (a dependent f32 MFMA waits ~8 cycles beyond the 64-cycle pass).
#define MF1(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
__device__ __forceinline__ void mfma_a2(const v4f a0, const v4f a1, const float4 b, f32x16& c0, f32x16& c1) {
    MF1(a0.x, b.x, c0); MF1(a1.x, b.x, c1); MF1(a0.y, b.y, c0); MF1(a1.y, b.y, c1);
    MF1(a0.z, b.z, c0); MF1(a1.z, b.z, c1); MF1(a0.w, b.w, c0); MF1(a1.w, b.w, c1);
}
__device__ __forceinline__ void mfma_b2(const float4 a, const v4f b0, const v4f b1, f32x16& c0, f32x16& c1) {
    MF1(a.x, b0.x, c0); MF1(a.x, b1.x, c1); MF1(a.y, b0.y, c0); MF1(a.y, b1.y, c1);
    MF1(a.z, b0.z, c0); MF1(a.z, b1.z, c1); MF1(a.w, b0.w, c0); MF1(a.w, b1.w, c1);
}
__device__ __forceinline__ void mfma_b4(const float4 a, const v4f b0, const v4f b1, const v4f b2, const v4f b3,
                                        f32x16& c0, f32x16& c1, f32x16& c2, f32x16& c3) {
    MF1(a.x, b0.x, c0); MF1(a.x, b1.x, c1); MF1(a.x, b2.x, c2); MF1(a.x, b3.x, c3);
    MF1(a.y, b0.y, c0); MF1(a.y, b1.y, c1); MF1(a.y, b2.y, c2); MF1(a.y, b3.y, c3);
    MF1(a.z, b0.z, c0); MF1(a.z, b1.z, c1); MF1(a.z, b2.z, c2); MF1(a.z, b3.z, c3);
    MF1(a.w, b0.w, c0); MF1(a.w, b1.w, c1); MF1(a.w, b2.w, c2); MF1(a.w, b3.w, c3);
}
// hipcc reuses ONE register quad for every LDS fragment (read -> wait -> 4 MFMAs -> read ...),
// which exposes the LDS latency on every group.  These fences pin "read the next fragments,
// then run the MFMAs on the ones read one group earlier"; the lgkmcnt waits stay the compiler's.
#define PIPE_FENCE() __builtin_amdgcn_sched_barrier(0)

// Stage hand-over WITHOUT waiting for this wave's own global stores.  __syncthreads() carries
// s_waitcnt vmcnt(0), which after an epilogue means "wait for the HBM write latency of 16 stores"
// on every stage (measured: 25 % of k_trimul_proj).  Instead:
//   stage_landed():   vmcnt(0) right after the MFMA loop -- everything outstanding there is old
//                     (the LDS-DMA of the next stage issued a whole stage ago, earlier stores);
//   stage_barrier():  bare s_barrier after the epilogue; its stores stay in flight across it.
// The LDS reads of the finished stage were consumed by MFMAs (lgkmcnt-waited) before either.
__device__ __forceinline__ void stage_landed() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void stage_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void norm_frags_(float4 (&zf)[16]) {
    float s = 0.f;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) s += (zf[kb].x + zf[kb].y) + (zf[kb].z + zf[kb].w);
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / 128.0f);
    float ss = 0.f;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        const float a = zf[kb].x - mean, b = zf[kb].y - mean, c = zf[kb].z - mean, d = zf[kb].w - mean;
        ss += (a * a + b * b) + (c * c + d * d);
    }
    ss += __shfl_xor(ss, 32);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 128.0f) + GENIE_LN_EPS);
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        zf[kb].x = (zf[kb].x - mean) * rstd; zf[kb].y = (zf[kb].y - mean) * rstd;
        zf[kb].z = (zf[kb].z - mean) * rstd; zf[kb].w = (zf[kb].w - mean) * rstd;
    }
}
__device__ __forceinline__ void load_norm_tile(float4 (&zf)[16], const float* __restrict__ rowp, int h) {
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) zf[kb] = *reinterpret_cast<const float4*>(rowp + kb * 8 + 4 * h);
    norm_frags_(zf);
}

// ---------------------------------------------------------------------------------------------
// f32 MFMA shares the SIMD's FP32 lanes with the VALU (tools/probe/valu_probe.hip: every vector
// instruction beside the MFMAs costs its full ~4 issue cycles; nothing overlaps), so the roof of
// these kernels is 157 TF x MFMA / (MFMA + VALU) and the game is to issue as few vector
// instructions as possible:
//   * all addressing lives in SGPRs: buffer loads/stores with a per-lane voffset computed once
//     and scalar soffsets (no 64-bit VALU address arithmetic per access);
//   * biases enter as the initial accumulator (C operand) instead of an add per output;
//   * the gate weights / bias are pre-scaled by -log2(e) on the host, so sigmoid is
//     v_exp_f32, v_add, v_rcp_f32;
//   * LayerNorm affine folded into the weights; accumulators ping-pong instead of being copied.
// ---------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void bglds16(rsrc_t r, float* lds_wave_base, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void bstore(rsrc_t r, float v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
#define UNI(x) __builtin_amdgcn_readfirstlane(x)

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, projections (math / layouts: pair_kernels.hip header comment).
// Stage = pass: 16 k-blocks x {p-block, g-block} of 32 channels.  8 stages.
// The gated outputs of pass p-1 (one accumulator row per k-block step) are issued between the
// MFMA groups of pass p.
// ---------------------------------------------------------------------------------------------
template <bool MM, bool EP>
__device__ __forceinline__ void proj_step(const float* stage, const float4 (&zf)[16], const float* sb_cur, int lane,
                                          f32x16& ap, f32x16& ag, const f32x16& pp, const f32x16& pg, float msk,
                                          rsrc_t rdst, int voff, int soff0, int sstride) {
    v4f fp, fg;
    if (MM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { ap[r] = sb_cur[acc_row(r, lane)]; ag[r] = sb_cur[256 + acc_row(r, lane)]; }
        fp = sfrag(stage, 0, lane); fg = sfrag(stage, 1, lane);
    }
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        if (MM) {
            const v4f np = sfrag(stage, min(2 * kb + 2, 30), lane), ng = sfrag(stage, min(2 * kb + 3, 31), lane);
            mfma_a2(fp, fg, zf[kb], ap, ag);
            fp = np; fg = ng;
        }
        if (EP) {   // accumulator register kb holds channel rows (kb&3) + 8(kb>>2) [+4 for the upper half-wave: in voff]
            const float v = pp[kb] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pg[kb])) * msk;
            bstore(rdst, v, voff, soff0 + ((kb & 3) + 8 * (kb >> 2)) * sstride);
        }
    }
}

template <bool OUTGOING>
__global__ __launch_bounds__(256, 2) void k_trimul_proj_wl(const float* __restrict__ z, const float* __restrict__ rmask,
                                                           const float* __restrict__ wp, const float* __restrict__ bias,
                                                           float* __restrict__ acm, float* __restrict__ bcm, int N, int NP,
                                                           int n_wtiles, unsigned cm_bytes) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sbias = sm + 2 * STAGE_FLOATS;                 // [512] biases (gate half pre-scaled by -log2 e)
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int wt_raw = blockIdx.x * 4 + wave;
    const bool act = wt_raw < n_wtiles;
    const int wt = act ? wt_raw : n_wtiles - 1;          // idle waves shadow the last tile (no stores)
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int st = wt % ntile;
    const int line = (wt / ntile) % N;
    const int b = wt / (ntile * N);
    const int t0 = st * 32;
    const int nvalid = act ? min(32, N - t0) : 0;
    const int pr = min(pl, min(32, N - t0) - 1);
    const float* rowp = OUTGOING ? z + (((size_t)b * N + line) * N + t0 + pr) * 128
                                 : z + (((size_t)b * N + t0 + pr) * N + line) * 128;
    const rsrc_t rw = make_rsrc(wp, 512 * 128 * 4);
    const rsrc_t ra = make_rsrc(acm, cm_bytes), rb = make_rsrc(bcm, cm_bytes);
    const int lane16 = lane * 16;
    auto issue = [&](int pass, int buf) {       // this wave's 8 of the stage's 32 fragments: u = 8 wave + q = 2 kb + j
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int u = 8 * wave + q;
            bglds16(rw, sm + buf * STAGE_FLOATS + u * 256, lane16, ((((u & 1) * 8 + pass) * 16 + (u >> 1)) * 1024));
        }
    };
    issue(0, 0);
    sbias[threadIdx.x] = bias[threadIdx.x];
    sbias[256 + threadIdx.x] = bias[256 + threadIdx.x];
    float4 zf[16];
    load_norm_tile(zf, rowp, h);
    const float msk = (pl < nvalid) ? rmask[b * N + line] * rmask[b * N + t0 + pr] : 0.f;
    // channel-major store: element ((b*128 + ch)*NP + line)*NP + t0 + pl ; ch = 32*(pass&3) + row
    const int sstride = NP * NP * 4;                                              // bytes per channel
    const int voff = (pl < nvalid) ? (4 * h * NP * NP + pl) * 4 : 0x7FFFFFF0;      // out-of-range offset: store dropped
    const int sbase = ((b * 128 * NP + line) * NP + t0) * 4;
    __syncthreads();

    f32x16 aP, aG, bP, bG;
#pragma unroll 1
    for (int pass = 0; pass < 8; pass += 2) {
        {   // even pass -> accumulators a*, epilogue of b* (pass - 1)
            if (pass + 1 < 8) issue(pass + 1, 1);
            const int sprev = sbase + ((pass - 1) & 3) * 32 * sstride;
            if (pass == 0) proj_step<true, false>(sm, zf, sbias + pass * 32, lane, aP, aG, bP, bG, msk, ra, voff, 0, sstride);
            else proj_step<true, true>(sm, zf, sbias + pass * 32, lane, aP, aG, bP, bG, msk, (pass - 1) < 4 ? ra : rb, voff, sprev, sstride);
            stage_landed();
            stage_barrier();
        }
        {   // odd pass -> accumulators b*, epilogue of a*
            if (pass + 2 < 8) issue(pass + 2, 0);
            const int sprev = sbase + (pass & 3) * 32 * sstride;
            proj_step<true, true>(sm + STAGE_FLOATS, zf, sbias + (pass + 1) * 32, lane, bP, bG, aP, aG, msk, pass < 4 ? ra : rb, voff, sprev, sstride);
            stage_landed();
            stage_barrier();
        }
    }
    proj_step<false, true>(sm, zf, sbias, lane, aP, aG, bP, bG, msk, rb, voff, sbase + 3 * 32 * sstride, sstride);
}

// ---------------------------------------------------------------------------------------------
// Pair transition + end-of-layer mask (pair_kernels.hip, k_pair_transition), hidden layer chained
// through the accumulator registers.  Stage = hidden block of 32: 16 W1 fragments (k-blocks of
// the input channels) then 16 W2 fragments (4 output blocks x the 4 k-blocks of this hidden block).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_pair_transition_wl(float* __restrict__ z, const float* __restrict__ rmask,
                                                               const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                                               int N, long long M, int n_hb) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_wt = (M + 31) / 32;
    const long long wt_raw = (long long)blockIdx.x * 4 + wave;
    const bool act = wt_raw < n_wt;
    const long long row0 = (act ? wt_raw : n_wt - 1) * 32;
    const int h = lane >> 5, pl = lane & 31;
    const int nrows = (int)min((long long)32, M - row0);
    const int nvalid = act ? nrows : 0;
    const int pr = min(pl, nrows - 1);
    float* zrow = z + row0 * 128;
    const int KB2 = n_hb * 4;
    auto issue = [&](int hb, int buf) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int u = 8 * wave + q;
            const float* g = (u < 16) ? wfrag_ptr(w1, 16, hb, u, lane)
                                      : wfrag_ptr(w2, KB2, (u - 16) >> 2, hb * 4 + ((u - 16) & 3), lane);
            glds16(g, sm + buf * STAGE_FLOATS + u * 256);
        }
    };
    issue(0, 0);
    float4 zn[16];
    load_norm_tile(zn, zrow + (size_t)pr * 128, h);
    float m_own = 0.f;
    if (pl < nvalid) {
        const long long idx = row0 + pl;
        const int bb = (int)(idx / ((long long)N * N));
        const int rem = (int)(idx - (long long)bb * N * N);
        m_own = rmask[bb * N + rem / N] * rmask[bb * N + rem % N];
    }
    __syncthreads();
    f32x16 o0 = zero16(), o1 = zero16(), o2 = zero16(), o3 = zero16();
#pragma unroll 1
    for (int hb = 0; hb < n_hb; ++hb) {
        if (hb + 1 < n_hb) issue(hb + 1, (hb + 1) & 1);
        const float* stage = sm + (hb & 1) * STAGE_FLOATS;
        f32x16 d = zero16(), d2 = zero16();      // two partial sums over alternate k-blocks: D'[hidden][pair]
        v4f f0 = sfrag(stage, 0, lane), f1 = sfrag(stage, 1, lane);
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
            const v4f n0 = sfrag(stage, min(u + 2, 14), lane), n1 = sfrag(stage, min(u + 3, 15), lane);
            PIPE_FENCE();
            MF1(f0.x, zn[u].x, d); MF1(f1.x, zn[u + 1].x, d2); MF1(f0.y, zn[u].y, d); MF1(f1.y, zn[u + 1].y, d2);
            MF1(f0.z, zn[u].z, d); MF1(f1.z, zn[u + 1].z, d2); MF1(f0.w, zn[u].w, d); MF1(f1.w, zn[u + 1].w, d2);
            PIPE_FENCE();
            f0 = n0; f1 = n1;
        }
        v4f w0 = sfrag(stage, 16, lane), w1 = sfrag(stage, 20, lane), w2f = sfrag(stage, 24, lane), w3 = sfrag(stage, 28, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = fmaxf((d[r] + d2[r]) + b1[hb * 32 + acc_row(r, lane)], 0.f);
#pragma unroll
        for (int q = 0; q < 4; ++q) {      // registers 4q..4q+3 of D' are the A fragment of k-block q
            const float4 hf = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
            const int qn = min(q + 1, 3);
            const v4f n0 = sfrag(stage, 16 + qn, lane), n1 = sfrag(stage, 20 + qn, lane), n2 = sfrag(stage, 24 + qn, lane),
                      n3 = sfrag(stage, 28 + qn, lane);
            PIPE_FENCE();
            mfma_b4(hf, w0, w1, w2f, w3, o0, o1, o2, o3);
            PIPE_FENCE();
            w0 = n0; w1 = n1; w2f = n2; w3 = n3;
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = acc_row(r, lane);
        const float m = __shfl(m_own, t);
        if (t < nvalid) {
            float* q = zrow + (size_t)t * 128 + pl;
            q[0]  = ((o0[r] + b2[pl]) * m + q[0]) * m;
            q[32] = ((o1[r] + b2[32 + pl]) * m + q[32]) * m;
            q[64] = ((o2[r] + b2[64 + pl]) * m + q[64]) * m;
            q[96] = ((o3[r] + b2[96 + pl]) * m + q[96]) * m;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, output (pair_kernels.hip, k_trimul_out):
//   z += (W_z LN_out(x) + b_z) * sigmoid(W_g LN_in(z) + b_g).
// Stages 0,1: W_g output blocks {0,1}, {2,3}; stages 2,3: W_z likewise (16 k-blocks each).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_trimul_out_wl(float* __restrict__ z, const float* __restrict__ xcm,
                                                          const float* __restrict__ wg, const float* __restrict__ bg,
                                                          const float* __restrict__ wz, const float* __restrict__ bz,
                                                          int N, int NP, int n_wtiles) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt_raw = blockIdx.x * 4 + wave;
    const bool act = wt_raw < n_wtiles;
    const int wt = act ? wt_raw : n_wtiles - 1;
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int st = wt % ntile;
    const int i = (wt / ntile) % N;
    const int b = wt / (ntile * N);
    const int t0 = st * 32;
    const int nvalid = act ? min(32, N - t0) : 0;
    const int pr = min(pl, min(32, N - t0) - 1);
    float* zrow = z + (((size_t)b * N + i) * N + t0) * 128;
    auto issue = [&](int s, int buf) {          // stage s: matrix (s < 2 ? W_g : W_z), blocks 2(s&1), 2(s&1)+1
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int u = 8 * wave + q;
            glds16(wfrag_ptr(s < 2 ? wg : wz, 16, 2 * (s & 1) + (u >> 4), u & 15, lane), sm + buf * STAGE_FLOATS + u * 256);
        }
    };
    issue(0, 0);
    float4 xf[16];
    {
        const float* xp = xcm + (((size_t)b * 128 + 4 * h) * NP + i) * NP + t0 + pl;     // t0 + pl < NP (NP = ceil32(N))
        const size_t cs = (size_t)NP * NP;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const float* p = xp + (size_t)(kb * 8) * cs;
            xf[kb] = make_float4(p[0], p[cs], p[2 * cs], p[3 * cs]);
        }
    }
    float4 zf[16];
    load_norm_tile(zf, zrow + (size_t)pr * 128, h);
    __syncthreads();

    f32x16 g[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {               // gate: A = zn fragments (i = pair), B = W_g (j = channel)
        issue(s + 1, (s + 1) & 1);
        const float* stage = sm + (s & 1) * STAGE_FLOATS;
        f32x16 ga = zero16(), gb = zero16();
        v4f f0 = sfrag(stage, 0, lane), f1 = sfrag(stage, 16, lane);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const v4f n0 = sfrag(stage, min(kb + 1, 15), lane), n1 = sfrag(stage, 16 + min(kb + 1, 15), lane);
            PIPE_FENCE();
            mfma_b2(zf[kb], f0, f1, ga, gb);
            PIPE_FENCE();
            f0 = n0; f1 = n1;
        }
        const float c0 = bg[(2 * s) * 32 + pl], c1 = bg[(2 * s + 1) * 32 + pl];
#pragma unroll
        for (int r = 0; r < 16; ++r) { ga[r] = fast_sigmoid_(ga[r] + c0); gb[r] = fast_sigmoid_(gb[r] + c1); }
        g[2 * s] = ga; g[2 * s + 1] = gb;
        __syncthreads();
    }
    norm_frags_(xf);
#pragma unroll
    for (int s = 2; s < 4; ++s) {
        if (s + 1 < 4) issue(s + 1, (s + 1) & 1);
        const float* stage = sm + (s & 1) * STAGE_FLOATS;
        f32x16 a0 = zero16(), a1 = zero16();
        v4f f0 = sfrag(stage, 0, lane), f1 = sfrag(stage, 16, lane);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const v4f n0 = sfrag(stage, min(kb + 1, 15), lane), n1 = sfrag(stage, 16 + min(kb + 1, 15), lane);
            PIPE_FENCE();
            mfma_b2(xf[kb], f0, f1, a0, a1);
            PIPE_FENCE();
            f0 = n0; f1 = n1;
        }
        stage_landed();
        const int ob = 2 * (s - 2);
        const float z0 = bz[ob * 32 + pl], z1 = bz[(ob + 1) * 32 + pl];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = acc_row(r, lane);
            if (t < nvalid) {
                float* q = zrow + (size_t)t * 128 + ob * 32 + pl;
                q[0] = (a0[r] + z0) * g[ob][r] + q[0];
                q[32] = (a1[r] + z1) * g[ob + 1][r] + q[32];
            }
        }
        stage_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
void launch_trimul_proj_wl(genie_ctx* h, hipStream_t st, const TriMulW& w, bool outgoing) {
    const int N = h->N, ntile = (N + 31) / 32;
    const int n_wt = h->B * N * ntile;
    dim3 grid((n_wt + 3) / 4);
    const unsigned cm_bytes = (unsigned)((size_t)h->B * 128 * h->NP * h->NP * 4);
    if (outgoing)
        hipLaunchKernelGGL(k_trimul_proj_wl<true>, grid, dim3(256), WL_LDS_BYTES, st, h->p, h->rmaskf, w.proj_w, w.proj_b,
                           h->acm, h->bcm, N, h->NP, n_wt, cm_bytes);
    else
        hipLaunchKernelGGL(k_trimul_proj_wl<false>, grid, dim3(256), WL_LDS_BYTES, st, h->p, h->rmaskf, w.proj_w, w.proj_b,
                           h->acm, h->bcm, N, h->NP, n_wt, cm_bytes);
}

void launch_trimul_out_wl(genie_ctx* h, hipStream_t st, const TriMulW& w) {
    const int N = h->N, ntile = (N + 31) / 32;
    const int n_wt = h->B * N * ntile;
    hipLaunchKernelGGL(k_trimul_out_wl, dim3((n_wt + 3) / 4), dim3(256), WL_LDS_BYTES, st, h->p, h->xcm, w.g_w, w.g_b, w.z_w,
                       w.z_b, N, h->NP, n_wt);
}

void launch_pair_transition_wl(genie_ctx* h, hipStream_t st, const PairLayerW& w) {
    const long long M = (long long)h->B * h->N * h->N;
    const long long n_wt = (M + 31) / 32;
    hipLaunchKernelGGL(k_pair_transition_wl, dim3((unsigned)((n_wt + 3) / 4)), dim3(256), WL_LDS_BYTES, st, h->p, h->rmaskf,
                       w.pt_w1, w.pt_b1, w.pt_w2, w.pt_b2, h->N, M, h->d.pair_transition_n * 4);
}

void pair_wl_kernels_init() {
(void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trimul_proj_wl<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trimul_proj_wl<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_transition_wl), hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trimul_out_wl), hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES);
}
