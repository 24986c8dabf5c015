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
// All three kernels below are PERSISTENT: a work-group walks tiles blockIdx.x, + gridDim.x, ...
// (grid = 2 work-groups per CU), so that per tile only the arithmetic is left on the critical
// path: the next tile's activation rows are prefetched into spare registers in the middle of the
// current tile, and the weight-stage ring never drains (the last stage of a tile issues stage 0
// of the next one).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_raw_tile(float4 (&raw)[16], const float* __restrict__ rowp, int h) {
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) raw[kb] = *reinterpret_cast<const float4*>(rowp + kb * 8 + 4 * h);
}
__device__ __forceinline__ void norm_from_raw(float4 (&zf)[16], const float4 (&raw)[16]) {
    float s = 0.f;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) s += (raw[kb].x + raw[kb].y) + (raw[kb].z + raw[kb].w);
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / 128.0f);
    float ss = 0.f;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        const float a = raw[kb].x - mean, b = raw[kb].y - mean, c = raw[kb].z - mean, d = raw[kb].w - mean;
        zf[kb] = make_float4(a, b, c, d);
        ss += (a * a + b * b) + (c * c + d * d);
    }
    ss += __shfl_xor(ss, 32);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 128.0f) + GENIE_LN_EPS);
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) { zf[kb].x *= rstd; zf[kb].y *= rstd; zf[kb].z *= rstd; zf[kb].w *= rstd; }
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, projections (math / layouts: pair_kernels.hip header comment).
// Stage = pass: 16 k-blocks x {p-block, g-block} of 32 channels, 8 stages per tile.
// ---------------------------------------------------------------------------------------------
template <bool OUTGOING>
__global__ __launch_bounds__(256, 2) void k_trimul_proj_wl(const float* __restrict__ z, const float* __restrict__ rmask,
                                                           const float* __restrict__ wp, const float* __restrict__ bias,
                                                           float* __restrict__ acm, float* __restrict__ bcm, int N, int NP,
                                                           int n_wtiles, unsigned cm_bytes) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sbias = sm + 2 * STAGE_FLOATS;                 // [512] biases (gate half pre-scaled by -log2 e)
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int n_tiles = (n_wtiles + 3) >> 2;
    const rsrc_t rw = make_rsrc(wp, 512 * 128 * 4);
    const rsrc_t ra = make_rsrc(acm, cm_bytes), rb = make_rsrc(bcm, cm_bytes);
    const int lane16 = lane * 16;
    const int sstride = NP * NP * 4;                      // bytes per channel
    auto issue = [&](int pass, int buf) {                 // this wave's 8 of the stage's 32 fragments: u = 8 wave + q = 2 kb + j
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int u = 8 * wave + q;
            bglds16(rw, sm + buf * STAGE_FLOATS + u * 256, lane16, ((((u & 1) * 8 + pass) * 16 + (u >> 1)) * 1024));
        }
    };
    auto row_ptr = [&](int tile) {                        // this lane's row of wave-tile 4 tile + wave (clamped: always readable)
        const int wt = min(tile * 4 + wave, n_wtiles - 1);
        const int st = wt % ntile, line = (wt / ntile) % N, b = wt / (ntile * N);
        const int pr = min(pl, min(32, N - st * 32) - 1);
        return OUTGOING ? z + (((size_t)b * N + line) * N + st * 32 + pr) * 128
                        : z + (((size_t)b * N + st * 32 + pr) * N + line) * 128;
    };
    int tile = blockIdx.x;
    issue(0, 0);
    sbias[threadIdx.x] = bias[threadIdx.x];
    sbias[256 + threadIdx.x] = bias[256 + threadIdx.x];
    float4 raw[16];
    load_raw_tile(raw, row_ptr(tile), h);
    __syncthreads();
#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = tile * 4 + wave;
        const bool act = wt_raw < n_wtiles;
        const int wt = act ? wt_raw : n_wtiles - 1;      // idle waves shadow the last tile (stores dropped)
        const int st = wt % ntile, line = (wt / ntile) % N, b = wt / (ntile * N);
        const int t0 = st * 32;
        const int nvalid = act ? min(32, N - t0) : 0;
        const float msk = (pl < nvalid) ? rmask[b * N + line] * rmask[b * N + t0 + pl] : 0.f;
        // channel-major store: element ((b*128 + ch)*NP + line)*NP + t0 + pl, ch = 32 (pass & 3) + row
        const int voff = (pl < nvalid) ? (4 * h * NP * NP + pl) * 4 : 0x7FFFFFF0;   // out-of-range offset: store dropped
        const int sbase = ((b * 128 * NP + line) * NP + t0) * 4;
        const bool more = tile + (int)gridDim.x < n_tiles;
        float4 zf[16];
        norm_from_raw(zf, raw);
#pragma unroll 1
        for (int pass = 0; pass < 8; ++pass) {
            if (pass + 1 < 8) issue(pass + 1, (pass + 1) & 1);
            else if (more) issue(0, 0);
            if (pass == 3 && more) load_raw_tile(raw, row_ptr(tile + gridDim.x), h);
            const float* stage = sm + (pass & 1) * STAGE_FLOATS;
            const float* sb = sbias + pass * 32;
            f32x16 ap, ag;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ap[r] = sb[acc_row(r, lane)]; ag[r] = sb[256 + acc_row(r, lane)]; }
            v4f fp = sfrag(stage, 0, lane), fg = sfrag(stage, 1, lane);
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) {
                const v4f np = sfrag(stage, min(2 * kb + 2, 30), lane), ng = sfrag(stage, min(2 * kb + 3, 31), lane);
                PIPE_FENCE();
                mfma_a2(fp, fg, zf[kb], ap, ag);
                PIPE_FENCE();
                fp = np; fg = ng;
            }
            stage_landed();
            const rsrc_t rd = pass < 4 ? ra : rb;
            const int so = sbase + (pass & 3) * 32 * sstride;
#pragma unroll
            for (int r = 0; r < 16; ++r) {   // register r holds channel rows (r&3) + 8(r>>2) [+4 for the upper half-wave: in voff]
                const float v = ap[r] * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ag[r])) * msk;
                bstore(rd, v, voff, so + ((r & 3) + 8 * (r >> 2)) * sstride);
            }
            stage_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Pair transition + end-of-layer mask (modules/pair_transition.py:48-56,
// pair_transform_net.py:116-117):  z = (z + mask (W2 relu(W1 LN(z) + b1) + b2)) mask, which for the
// binary mask equals (z + W2 relu(..) + b2) * mask.
// Hidden layer chained through the accumulator registers: with D'[hidden][pair] = W1 zn^T,
// registers 4q..4q+3 of D' are the A fragment of k-block q of the second GEMM.
// Stage = hidden block of 32: 16 W1 fragments (k-blocks of the input channels) then 16 W2
// fragments (4 output blocks x the 4 k-blocks of this hidden block).  b1 enters as the initial
// accumulator of D', b2 as the initial accumulator of the output.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_pair_transition_wl(float* __restrict__ z, const float* __restrict__ rmask,
                                                               const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                                               int N, long long M, int n_hb) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* sb1 = sm + 2 * STAGE_FLOATS;          // [<= 512] hidden biases
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int n_wt = (int)((M + 31) / 32);
    const int n_tiles = (n_wt + 3) >> 2;
    const int KB2 = n_hb * 4;
    const rsrc_t rw1 = make_rsrc(w1, (unsigned)(n_hb * 32 * 128 * 4)), rw2 = make_rsrc(w2, (unsigned)(128 * n_hb * 32 * 4));
    const rsrc_t rz = make_rsrc(z, (unsigned)(M * 512));              // rows >= M fall off the end: loads give 0, stores drop
    const rsrc_t rnull = make_rsrc(z, 0u);
    const int lane16 = lane * 16;
    auto issue = [&](int hb, int buf) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int u = 8 * wave + q;          // waves 0,1: W1 k-blocks 0..15 ; waves 2,3: W2 (ob = (u-16)>>2, q' = (u-16)&3)
            if (u < 16) bglds16(rw1, sm + buf * STAGE_FLOATS + u * 256, lane16, (hb * 16 + u) * 1024);
            else        bglds16(rw2, sm + buf * STAGE_FLOATS + u * 256, lane16, ((((u - 16) >> 2) * KB2) + hb * 4 + ((u - 16) & 3)) * 1024);
        }
    };
    auto row_ptr = [&](int tile) {               // this lane's row of wave-tile 4 tile + wave (clamped: always readable)
        const long long r0 = (long long)min(tile * 4 + wave, n_wt - 1) * 32;
        return z + (r0 + min(pl, (int)min((long long)32, M - r0) - 1)) * 128;
    };
    int tile = blockIdx.x;
    issue(0, 0);
    for (int u = threadIdx.x; u < n_hb * 32; u += 256) sb1[u] = b1[u];
    const float c0 = b2[pl], c1 = b2[32 + pl], c2 = b2[64 + pl], c3 = b2[96 + pl];
    __syncthreads();
#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = tile * 4 + wave;
        const bool act = wt_raw < n_wt;
        const long long row0 = (long long)(act ? wt_raw : n_wt - 1) * 32;
        const int nrows = (int)min((long long)32, M - row0);
        const bool more = tile + (int)gridDim.x < n_tiles;
        float m_own = 0.f;
        if (pl < nrows) {
            const long long idx = row0 + pl;
            const int bb = (int)(idx / ((long long)N * N));
            const int rem = (int)(idx - (long long)bb * N * N);
            m_own = rmask[bb * N + rem / N] * rmask[bb * N + rem % N];
        }
        float4 zn[16];          // (2048 MFMAs per tile: the row load is not worth 64 prefetch registers here)
        load_raw_tile(zn, row_ptr(tile), h);
        norm_from_raw(zn, zn);
        f32x16 o0, o1, o2, o3;
        {
            float e0 = c0, e1 = c1, e2 = c2, e3 = c3;
            asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));   // keep hipcc from hoisting (and spilling) the splats
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] = e0; o1[r] = e1; o2[r] = e2; o3[r] = e3; }
        }
#pragma unroll 1
        for (int hb = 0; hb < n_hb; ++hb) {
            if (hb + 1 < n_hb) issue(hb + 1, (hb + 1) & 1);
            else if (more) issue(0, 0);
            const float* stage = sm + (hb & 1) * STAGE_FLOATS;
            f32x16 d, d2 = zero16();              // two partial sums over alternate k-blocks; d starts at the bias
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = sb1[hb * 32 + acc_row(r, lane)];
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
            v4f w0 = sfrag(stage, 16, lane), w1f = sfrag(stage, 20, lane), w2f = sfrag(stage, 24, lane), w3 = sfrag(stage, 28, lane);
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = fmaxf(d[r] + d2[r], 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 hf = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
                const int qn = min(q + 1, 3);
                const v4f n0 = sfrag(stage, 16 + qn, lane), n1 = sfrag(stage, 20 + qn, lane), n2 = sfrag(stage, 24 + qn, lane),
                          n3 = sfrag(stage, 28 + qn, lane);
                PIPE_FENCE();
                mfma_b4(hf, w0, w1f, w2f, w3, o0, o1, o2, o3);
                PIPE_FENCE();
                w0 = n0; w1f = n1; w2f = n2; w3 = n3;
            }
            __syncthreads();
        }
        // epilogue: row t = acc_row(r, lane) of the tile, channel 32 ob + pl
        const rsrc_t rzz = act ? rz : rnull;
        const int voff = (4 * h * 128 + pl) * 4;
        const int srow0 = (int)(row0 * 512);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rc = (r & 3) + 8 * (r >> 2);
            const float m = __shfl(m_own, rc + 4 * h);
            const int so = srow0 + rc * 512;
            const float z0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rzz, voff, so, 0));
            const float z1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rzz, voff, so + 128, 0));
            const float z2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rzz, voff, so + 256, 0));
            const float z3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rzz, voff, so + 384, 0));
            bstore(rzz, (o0[r] + z0) * m, voff, so);
            bstore(rzz, (o1[r] + z1) * m, voff, so + 128);
            bstore(rzz, (o2[r] + z2) * m, voff, so + 256);
            bstore(rzz, (o3[r] + z3) * m, voff, so + 384);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, output (modules/triangular_multiplicative_update.py:105-108 + residual):
//   z += (W_z LN_out(x) + b_z) * sigmoid(W_g LN_in(z) + b_g).
// The x tile arrives channel-major; lane (p, h) reads x[c][p] for its 64 channels directly in
// fragment order (each load = two 128-B runs).  Stages: W_g blocks {0,1}, W_z {0,1}, W_g {2,3}, W_z {2,3}.  W_g / b_g are pre-scaled by -log2(e); b_g and b_z enter as initial
// accumulators.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_trimul_out_wl(float* __restrict__ z, const float* __restrict__ xcm,
                                                          const float* __restrict__ wg, const float* __restrict__ bg,
                                                          const float* __restrict__ wz, const float* __restrict__ bz,
                                                          int N, int NP, int n_wtiles, unsigned cm_bytes, unsigned z_bytes) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int n_tiles = (n_wtiles + 3) >> 2;
    const rsrc_t rg = make_rsrc(wg, 128 * 128 * 4), rwz = make_rsrc(wz, 128 * 128 * 4);
    const rsrc_t rx = make_rsrc(xcm, cm_bytes), rz = make_rsrc(z, z_bytes);
    const int lane16 = lane * 16;
    auto issue = [&](int s, int buf) {          // stage s: matrix (s even ? W_g : W_z), output blocks 2(s>>1), 2(s>>1)+1
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int u = 8 * wave + q;
            bglds16((s & 1) ? rwz : rg, sm + buf * STAGE_FLOATS + u * 256, lane16, ((2 * (s >> 1) + (u >> 4)) * 16 + (u & 15)) * 1024);
        }
    };
    auto row_ptr = [&](int tile) {              // this lane's z row of wave-tile 4 tile + wave (clamped: always readable)
        const int wt = min(tile * 4 + wave, n_wtiles - 1);
        const int st = wt % ntile, i = (wt / ntile) % N, b = wt / (ntile * N);
        const int pr = min(pl, min(32, N - st * 32) - 1);
        return z + (((size_t)b * N + i) * N + st * 32 + pr) * 128;
    };
    int tile = blockIdx.x;
    issue(0, 0);
    float4 zf[16];
    __syncthreads();
#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = tile * 4 + wave;
        const bool act = wt_raw < n_wtiles;
        const int wt = act ? wt_raw : n_wtiles - 1;
        const int st = wt % ntile, i = (wt / ntile) % N, b = wt / (ntile * N);
        const int t0 = st * 32;
        const int nvalid = act ? min(32, N - t0) : 0;
        const int prow0 = (b * N + i) * N + t0;                  // first pair row of the tile
        const bool more = tile + (int)gridDim.x < n_tiles;
        float4 xf[16];
        {   // x_cm[((b*128 + c)*NP + i)*NP + t0 + pl], c = 8kb + 4h + e   (t0 + pl < NP always); consumed two stages later
            const int cs = NP * NP * 4;
            const int vx = (4 * h * NP * NP + pl) * 4;
            const int sx = ((b * 128 * NP + i) * NP + t0) * 4;
#pragma unroll
            for (int kb = 0; kb < 16; ++kb) {
                xf[kb].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vx, sx + (8 * kb + 0) * cs, 0));
                xf[kb].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vx, sx + (8 * kb + 1) * cs, 0));
                xf[kb].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vx, sx + (8 * kb + 2) * cs, 0));
                xf[kb].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vx, sx + (8 * kb + 3) * cs, 0));
            }
        }
        load_raw_tile(zf, row_ptr(tile), h);
        norm_from_raw(zf, zf);
        // residual rows: t = acc_row(r, lane); rows past the tile's valid pairs belong to the next line -> predicate
        const int voff = (4 * h * 128 + pl) * 4;
        // Stage order W_g{0,1}, W_z{0,1}, W_g{2,3}, W_z{2,3}: only 32 gate registers are live at a time.
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 ga, gb;
            {   // gate: A = zn fragments (i = pair), B = W_g (j = channel)
                issue(2 * half + 1, 1);
                const float* stage = sm;
                float c0 = bg[(2 * half) * 32 + pl], c1 = bg[(2 * half + 1) * 32 + pl];
                asm volatile("" : "+v"(c0), "+v"(c1));      // keep hipcc from hoisting (and spilling) the 16-register splats
#pragma unroll
                for (int r = 0; r < 16; ++r) { ga[r] = c0; gb[r] = c1; }
                v4f f0 = sfrag(stage, 0, lane), f1 = sfrag(stage, 16, lane);
#pragma unroll
                for (int kb = 0; kb < 16; ++kb) {
                    const v4f n0 = sfrag(stage, min(kb + 1, 15), lane), n1 = sfrag(stage, 16 + min(kb + 1, 15), lane);
                    PIPE_FENCE();
                    mfma_b2(zf[kb], f0, f1, ga, gb);
                    PIPE_FENCE();
                    f0 = n0; f1 = n1;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ga[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ga[r]));
                    gb[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gb[r]));
                }
                __syncthreads();
            }
            if (half == 0) norm_frags_(xf);
            {   // update of channel blocks 2 half, 2 half + 1
                if (half == 0) issue(2, 0);
                else if (more) issue(0, 0);
                const float* stage = sm + STAGE_FLOATS;
                const int ob = 2 * half;
                f32x16 a0, a1;
                float c0 = bz[ob * 32 + pl], c1 = bz[(ob + 1) * 32 + pl];
                asm volatile("" : "+v"(c0), "+v"(c1));
#pragma unroll
                for (int r = 0; r < 16; ++r) { a0[r] = c0; a1[r] = c1; }
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
                // residual + store, 8 rows (16 loads) in flight at a time; rows past the tile's valid pairs belong to
                // the next line: their offset is pushed out of the buffer (load gives 0, store is dropped)
                const int h4 = 4 * h;
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    float zr0[8], zr1[8];
                    int vo[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int rc = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        vo[q] = (h4 < nvalid - rc) ? voff : 0x7FFFFFF0;
                        zr0[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, vo[q], so, 0));
                        zr1[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rz, vo[q], so + 128, 0));
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int rc = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        bstore(rz, fmaf(a0[r0 + q], ga[r0 + q], zr0[q]), vo[q], so);
                        bstore(rz, fmaf(a1[r0 + q], gb[r0 + q], zr1[q]), vo[q], so + 128);
                    }
                    PIPE_FENCE();
                }
                stage_barrier();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
static int g_num_cu = 0;
static unsigned persistent_grid(long long n_tiles) {      // 2 work-groups per CU (LDS: 66 KiB each)
    if (!g_num_cu) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&pr, dev); g_num_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    const long long cap = 2LL * g_num_cu;
    return (unsigned)(n_tiles < cap ? n_tiles : cap);
}

void launch_trimul_proj_wl(genie_ctx* h, hipStream_t st, const TriMulW& w, bool outgoing) {
    const int N = h->N, ntile = (N + 31) / 32;
    const int n_wt = h->B * N * ntile;
    dim3 grid(persistent_grid((n_wt + 3) / 4));
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
    hipLaunchKernelGGL(k_trimul_out_wl, dim3(persistent_grid((n_wt + 3) / 4)), dim3(256), WL_LDS_BYTES, st, h->p, h->xcm, w.g_w, w.g_b, w.z_w,
                       w.z_b, N, h->NP, n_wt, (unsigned)((size_t)h->B * 128 * h->NP * h->NP * 4),
                       (unsigned)((size_t)h->B * N * N * 512));
}

void launch_pair_transition_wl(genie_ctx* h, hipStream_t st, const PairLayerW& w) {
    const long long M = (long long)h->B * h->N * h->N;
    const long long n_wt = (M + 31) / 32;
    hipLaunchKernelGGL(k_pair_transition_wl, dim3(persistent_grid((n_wt + 3) / 4)), dim3(256), WL_LDS_BYTES, st, h->p, h->rmaskf,
                       w.pt_w1, w.pt_b1, w.pt_w2, w.pt_b2, h->N, M, h->d.pair_transition_n * 4);
}

void pair_wl_kernels_init() {
(void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trimul_proj_wl<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trimul_proj_wl<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_transition_wl), hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trimul_out_wl), hipFuncAttributeMaxDynamicSharedMemorySize, WL_LDS_BYTES);
}
