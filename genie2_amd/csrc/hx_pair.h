// Device helpers shared by the pair-stack hx kernels (pair_hx_kernels.hip, pair_fused_kernels.hip): buffer resources,
// LDS-DMA, the coalesced row-tile loader, LayerNorm + split of a row tile, and the software-pipelined projection stage.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "hx.h"

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t hx_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void hx_dma(rsrc_t r, unsigned char* lds_wave_base, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void hx_store(rsrc_t r, float v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
__device__ __forceinline__ void hx_store_u(rsrc_t r, unsigned v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, voff, soff, 0);
}
__device__ __forceinline__ float hx_load(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
// Tile order.  Consecutive pair-stack kernels walk their tiles in OPPOSITE directions (`rev` alternates per launch): what one
// kernel wrote last is what the next one reads first, while it is still in L2 / the 256-MiB Infinity Cache.
#define HX_PHYS(t) (rev ? n_tiles - 1 - (t) : (t))
#define PIPE_FENCE() __builtin_amdgcn_sched_barrier(0)
#define HX_ZT_BYTES 8192                                   // per-wave row-tile staging (half a tile: 32 rows x 64 channels)
#define HX_LDS_BYTES (2 * HX_STAGE_BYTES + 2048 + 8 * HX_ZT_BYTES)
#ifndef HX_ABL
#define HX_ABL 0          // developer builds (tools/abl_build.sh): 128 = in-kernel timestamps; 0 in the product
#endif
#if HX_ABL & 128          // in-kernel timestamps of every wave of work-group 0 (tests/devtools/ts_read.py)
__device__ unsigned long long g_hx_ts[24][4096];   // [variant * 8 + wave]
#define HX_TS_DECL(variant) const bool ts_on = blockIdx.x == 0; \
                   unsigned long long* ts_p = g_hx_ts[(variant) * 8 + (threadIdx.x >> 6)]; int ts_n = 0
#define HX_TS() do { if (ts_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0 && ts_n < 4096) ts_p[ts_n] = t_; ++ts_n; } } while (0)
extern "C" int genie_hx_debug_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hx_ts), sizeof(unsigned long long) * 24 * 4096);
}
#if HX_ABL & 256          // finer stamps: per k-chunk in the projection (tests/devtools/ts_kc.py), inside the transition's first GEMM (tests/devtools/ts_tr.py)
#define HX_TS2() HX_TS()
#else
#define HX_TS2() do { } while (0)
#endif
#else
#define HX_TS_DECL(variant)
#define HX_TS() do { } while (0)
#define HX_TS2() do { } while (0)
#endif

__device__ __forceinline__ void hx_stage_landed() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void hx_stage_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// Row tile (raw[2kc], raw[2kc+1] = channels 16kc + 8h .. +7 of this lane's pair row, from hx_zt_read below) -> normalised,
// scaled, split fragments: the B / A operand slots of k-chunk kc.  LayerNorm over the 128 channels =
// this lane's 64 + its partner's (lane ^ 32); affine folded into the weights on the host.
__device__ __forceinline__ void hx_norm_split(h8 (&xh)[8], h8 (&xl)[8], float4 (&raw)[16], float sx) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += (raw[q].x + raw[q].y) + (raw[q].z + raw[q].w);
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / 128.0f);
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        raw[q].x -= mean; raw[q].y -= mean; raw[q].z -= mean; raw[q].w -= mean;
        ss += (raw[q].x * raw[q].x + raw[q].y * raw[q].y) + (raw[q].z * raw[q].z + raw[q].w * raw[q].w);
    }
    ss += __shfl_xor(ss, 32);
    const float sc = sx / sqrtf(ss * (1.0f / 128.0f) + GENIE_LN_EPS);
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
        const float x[8] = {raw[2 * kc].x, raw[2 * kc].y, raw[2 * kc].z, raw[2 * kc].w,
                            raw[2 * kc + 1].x, raw[2 * kc + 1].y, raw[2 * kc + 1].z, raw[2 * kc + 1].w};
        hx_split8(x, sc, xh[kc], xl[kc]);
    }
}

// Coalesced row-tile loader.  A lane needs 8 consecutive channels of ITS pair row per k-chunk; loading
// them straight from global memory makes every wave instruction touch 32 different 128-B lines (32 B
// of each), each line is visited by four instructions, and with 8 waves the 128-KiB tile set thrashes
// the 32-KiB L1: measured ~10k cycles per tile, 20-40 % of these kernels.  Instead the tile goes
// through LDS: LDS-DMA (no VGPRs) fetches whole lines -- instruction j = rows 4j..4j+3, 16 lanes x 16 B
// = 256 contiguous bytes per row -- in two channel halves of 8 KiB, and each lane then reads its row
// fragments with ds_read_b128.  The 16-B granule g of row r sits in slot g ^ (r & 15) (the swizzle is
// applied on the GLOBAL side, LDS-DMA writes are lane-linear), which makes those reads conflict free.
// (j0, nj): which of the half's 8 instructions -- a burst of loads blocks the CU's in-order vector-memory
// pipe (every wave's stores and weight DMA queue behind the misses), so callers spread them over stages.
__device__ __forceinline__ void hx_zt_dma(rsrc_t rz, unsigned char* zt, int lane, int soff, int row_stride, int nvalid, int half,
                                          int j0 = 0, int nj = 8) {
#pragma unroll
    for (int j = j0; j < j0 + nj; ++j) {
        const int r = 4 * j + (lane >> 4);
        const int voff = min(r, nvalid - 1) * row_stride + (((lane & 15) ^ (r & 15)) << 4);     // rows past the tile: clamped
        hx_dma(rz, zt + j * 1024, voff, soff + half * 256);
    }
}
__device__ __forceinline__ void hx_zt_read(float4 (&raw)[16], const unsigned char* zt, int pl, int h, int half) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int g = 4 * q + 2 * h;
        raw[2 * (4 * half + q)] = *reinterpret_cast<const float4*>(zt + pl * 256 + ((g ^ (pl & 15)) << 4));
        raw[2 * (4 * half + q) + 1] = *reinterpret_cast<const float4*>(zt + pl * 256 + (((g + 1) ^ (pl & 15)) << 4));
    }
}
__device__ __forceinline__ void hx_lds_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void hx_vm_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// SOFTWARE PIPELINE: the epilogue of pass s-1 (gate, rescale, split, 16 stores) is issued between
// the MFMAs of pass s (two accumulator sets ping-pong), one dependent piece per MFMA gap -- a wave's
// MFMAs are a dependent chain, so whatever sits between them in program order is free, and the
// stores leave at a steady rate instead of in a burst per stage (unpipelined, the matrix pipe
// and HBM alternated chip-wide: the kernel took the SUM of its MFMA and its memory time).
// vmcnt bookkeeping: per stage a wave issues its LDS-DMA pieces of the next weight stage first, then
// exactly 16 stores (dropped ones count too), then (some stages) up to 3 row-tile requests, so
// `s_waitcnt vmcnt(16)` after the last store = "the weights and the row-tile pieces requested a stage
// ago have landed" without waiting for the stores (vector-memory operations retire in order).
#define HX_PROJ_PIECE_A(EG, r, t)  t = __builtin_amdgcn_exp2f(EG[r] * cg)
#define HX_PROJ_PIECE_B(EP, r, t, u) do { t = __builtin_amdgcn_rcpf(1.0f + t); u = EP[r] * e_pm; } while (0)
#define HX_PROJ_OFF(r) (e_so + (((r) & 3) + 8 * ((r) >> 2)) * sstride)
/* the first 8 packed outputs of a stage wait in registers and leave with the last 8: all 16 stores sit in the second half of
   the stage, the HBM-missing row-tile requests are issued right after the last of them, and a wave's next store is then half a
   stage + a barrier away -- a store queued behind an unreturned load stalls its wave (tools/probe/ldst_probe: 3.08 -> 2.72 us) */
#define HX_PROJ_PIECE_C(r, t, u) do { const unsigned wv_ = hx_pack_prod(u, t);                                     \
        if ((r) < 8) wst[(r)] = wv_;                                                                               \
        else { hx_store_u(e_rd, wst[(r) - 8], e_voff, HX_PROJ_OFF((r) - 8)); hx_store_u(e_rd, wv_, e_voff, HX_PROJ_OFF(r)); } } while (0)
#define HX_PROJ_PIECE_C_NOW(r, t, u) hx_store_u(e_rd, hx_pack_prod(u, t), e_voff, HX_PROJ_OFF(r))
#define HX_PROJ_REINIT(EP, EG, r) do { EP[r] = sbn[acc_row(r, lane)]; EG[r] = sbn[256 + acc_row(r, lane)]; } while (0)
#define HX_PROJ_STAGE(AP, AG, EP, EG)                                                                              \
    do {                                                                                                           \
        const float* sbn = sbias + ((pass + 1) & 7) * 32;   /* EP / EG become the next pass's accumulators */       \
        h8 ph = hx_frag(stage, 0, 0, lane), pq = hx_frag(stage, 0, 1, lane), gh = hx_frag(stage, 1, 0, lane),     \
           gq = hx_frag(stage, 1, 1, lane);                                                                        \
        _Pragma("unroll") for (int kc = 0; kc < 8; ++kc) {                                                         \
            const int kn = min(kc + 1, 7);                                                                         \
            const h8 nph = hx_frag(stage, 2 * kn, 0, lane), npq = hx_frag(stage, 2 * kn, 1, lane),                 \
                     ngh = hx_frag(stage, 2 * kn + 1, 0, lane), ngq = hx_frag(stage, 2 * kn + 1, 1, lane);         \
            float t0, t1, u0, u1;                                                                                  \
            PIPE_FENCE(); MFH(pq, zh[kc], AP); PIPE_FENCE(); HX_PROJ_PIECE_A(EG, 2 * kc, t0);                      \
            PIPE_FENCE(); MFH(ph, zl[kc], AP); PIPE_FENCE(); HX_PROJ_PIECE_B(EP, 2 * kc, t0, u0);                  \
            PIPE_FENCE(); MFH(ph, zh[kc], AP); PIPE_FENCE(); HX_PROJ_PIECE_C(2 * kc, t0, u0); HX_PROJ_REINIT(EP, EG, 2 * kc); \
            PIPE_FENCE(); MFH(gq, zh[kc], AG); PIPE_FENCE(); HX_PROJ_PIECE_A(EG, 2 * kc + 1, t1);                  \
            PIPE_FENCE(); MFH(gh, zl[kc], AG); PIPE_FENCE(); HX_PROJ_PIECE_B(EP, 2 * kc + 1, t1, u1);              \
            PIPE_FENCE(); MFH(gh, zh[kc], AG); PIPE_FENCE(); HX_PROJ_PIECE_C(2 * kc + 1, t1, u1); HX_PROJ_REINIT(EP, EG, 2 * kc + 1); \
            PIPE_FENCE(); HX_TS2();                                                                                \
            ph = nph; pq = npq; gh = ngh; gq = ngq;                                                                \
        }                                                                                                          \
    } while (0)

