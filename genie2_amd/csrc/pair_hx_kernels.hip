// Pair-stack GEMM kernels in "hx" arithmetic (hx.h): the WL structure of pair_wl_kernels.hip
// (a wave owns a 32-pair tile in fragment registers, weights arrive through LDS-DMA stages, the
// work-groups are persistent) with every f32 GEMM carried by three f16 MFMAs on split operands.
// 16x less matrix-pipe time per f32 FLOP than v_mfma_f32_32x32x2_f32 x 3 products = 5.3x, and the
// VALU work (LayerNorm, splits, gates) now overlaps the MFMAs instead of competing for the FP32 lanes.
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

// ---------------------------------------------------------------------------------------------
// Pair transition + end-of-layer mask (modules/pair_transition.py:48-56, pair_transform_net.py:116-117):
//   z = (z + W2 relu(W1 LN(z) + b1) + b2) * mask.
// Stage = hidden block of 32: units 0..7 = W1 k-chunks (A operand: rows = hidden units),
// units 8..15 = W2 (k-chunk c of the block, output block ob) at 8 + 4c + ob (B operand, chained-k
// permutation).  D'[hidden][pair] = Sx S1 (W1 zn + b1) accumulates from the scaled bias; relu and
// the rescale c1 = Sh / (Sx S1) give the second GEMM's A operand; out = acc * c2 (c2 = 1 / (Sh S2)).
// ---------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void k_pair_transition_hx(
    float* __restrict__ z, const float* __restrict__ rmask, const unsigned char* __restrict__ wimg,
    const float* __restrict__ b1s, const float* __restrict__ b2s, int N, long long M, int n_hb, float sx, float c1, float c2, int rev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    float* sb1 = reinterpret_cast<float*>(smb + 2 * HX_STAGE_BYTES);
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int n_wt = (int)((M + 31) / 32);
    const int n_tiles = (n_wt + NW - 1) / NW;
    const rsrc_t rw = hx_rsrc(wimg, (unsigned)(n_hb * HX_STAGE_BYTES));
    const rsrc_t rz = hx_rsrc(z, (unsigned)(M * 512));               // rows >= M fall off the end: loads give 0, stores drop
    const rsrc_t rnull = hx_rsrc(z, 0u);
    const int lane16 = lane * 16;
    constexpr int PW = 32 / NW;                   // 1-KiB pieces of a stage per wave
    auto issue = [&](int hb, int buf) {
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int p = PW * wave + q;
            hx_dma(rw, smb + buf * HX_STAGE_BYTES + p * 1024, lane16, hb * HX_STAGE_BYTES + p * 1024);
        }
    };
    unsigned char* zt = smb + 2 * HX_STAGE_BYTES + 2048 + wave * HX_ZT_BYTES;
    auto tile_geom = [&](int tile, int& soff, int& nv) {      // wave-tile NW tile + wave (clamped): first row (bytes), valid rows
        const long long r0 = (long long)min(HX_PHYS(tile) * NW + wave, n_wt - 1) * 32;
        nv = (int)min((long long)32, M - r0);
        soff = (int)(r0 * 512);
    };
    // next tile's rows: half 0 requested in stage hq0, moved to registers (and half 1 requested) in stage hq1, half 1 read at
    // the tile boundary; with fewer than 2 stages per tile the tile is loaded synchronously instead
    const bool prefetch = n_hb >= 2;
    const int hq0 = max(0, n_hb / 2 - 2), hq1 = min(n_hb - 1, hq0 + 4);
    int tile = blockIdx.x;
    HX_TS_DECL(2);
    issue(0, 0);
    for (int u = threadIdx.x; u < n_hb * 32; u += NW * 64) sb1[u] = b1s[u];
    const float cb0 = b2s[pl], cb1 = b2s[32 + pl], cb2 = b2s[64 + pl], cb3 = b2s[96 + pl];
    float4 raw[16];
    int n_soff, n_nv;
    auto load_sync = [&](int t) {
        tile_geom(t, n_soff, n_nv);
        hx_zt_dma(rz, zt, lane, n_soff, 512, n_nv, 0); hx_vm_done(); hx_zt_read(raw, zt, pl, h, 0); hx_lds_done();
        hx_zt_dma(rz, zt, lane, n_soff, 512, n_nv, 1); hx_vm_done(); hx_zt_read(raw, zt, pl, h, 1);
    };
    load_sync(tile);
    __syncthreads();
    h8 zh[8], zl[8];
    float zres[4][16];
#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = HX_PHYS(tile) * NW + wave;
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
        hx_norm_split(zh, zl, raw, sx);
        f32x16 o[4];
        {
            float e0 = cb0, e1 = cb1, e2 = cb2, e3 = cb3;
            asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));   // keep hipcc from hoisting (and spilling) the splats
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] = e0; o[1][r] = e1; o[2][r] = e2; o[3][r] = e3; }
        }
        const rsrc_t rzz = act ? rz : rnull;
        const int voff = (4 * h * 128 + pl) * 4;
        const int srow0 = (int)(row0 * 512);
        auto stage_body = [&](int hb, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            HX_TS();
            if (more && prefetch) {
                if (hb == hq0) { tile_geom(tile + gridDim.x, n_soff, n_nv); hx_zt_dma(rz, zt, lane, n_soff, 512, n_nv, 0); }
                if (hb == hq1) { hx_zt_read(raw, zt, pl, h, 0); hx_lds_done(); hx_zt_dma(rz, zt, lane, n_soff, 512, n_nv, 1); }
            }
            if (!LAST) issue(hb + 1, (hb + 1) & 1);
            else if (more) issue(0, 0);
            HX_TS2();
            const unsigned char* stage = smb + (hb & 1) * HX_STAGE_BYTES;
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = sb1[hb * 32 + acc_row(r, lane)];
            HX_TS2();
            {
                h8 wh = hx_frag(stage, 0, 0, lane), wl = hx_frag(stage, 0, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const h8 nh = hx_frag(stage, min(kc + 1, 7), 0, lane), nl = hx_frag(stage, min(kc + 1, 7), 1, lane);
                    PIPE_FENCE();
                    MFH3(wh, wl, zh[kc], zl[kc], d);
                    PIPE_FENCE();
                    if (kc == 0 || kc == 3) HX_TS2();
                    wh = nh; wl = nl;
                }
            }
            HX_TS();
            if (LAST) {     // zh / zl are dead from here on: the first half of the residual rows is fetched while the last GEMM runs
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int so = srow0 + ((r & 3) + 8 * (r >> 2)) * 512;
                    zres[0][r] = hx_load(rzz, voff, so); zres[1][r] = hx_load(rzz, voff, so + 128);
                    zres[2][r] = hx_load(rzz, voff, so + 256); zres[3][r] = hx_load(rzz, voff, so + 384);
                }
            }
            h8 ah[2], al[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaxf(d[8 * c + e], 0.f);
                hx_split8(x, c1, ah[c], al[c]);
            }
            {
                h8 bh = hx_frag(stage, 8, 0, lane), bl = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const h8 nh = hx_frag(stage, 8 + min(u + 1, 7), 0, lane), nl = hx_frag(stage, 8 + min(u + 1, 7), 1, lane);
                    PIPE_FENCE();
                    MFH3(ah[u >> 2], al[u >> 2], bh, bl, o[u & 3]);
                    PIPE_FENCE();
                    bh = nh; bl = nl;
                }
            }
            HX_TS();
            __syncthreads();
            HX_TS();
        };
#pragma unroll 1
        for (int hb = 0; hb + 1 < n_hb; ++hb) stage_body(hb, std::false_type{});
        stage_body(n_hb - 1, std::true_type{});
        // epilogue: row t = acc_row(r, lane) of the tile, channel 32 ob + pl; second half of the residual rows requested first
#pragma unroll
        for (int r = 8; r < 16; ++r) {
            const int so = srow0 + ((r & 3) + 8 * (r >> 2)) * 512;
            zres[0][r] = hx_load(rzz, voff, so); zres[1][r] = hx_load(rzz, voff, so + 128);
            zres[2][r] = hx_load(rzz, voff, so + 256); zres[3][r] = hx_load(rzz, voff, so + 384);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rc = (r & 3) + 8 * (r >> 2);
            const float m = __shfl(m_own, rc + 4 * h);
            const int so = srow0 + rc * 512;
            const float v0 = o[0][r], v1 = o[1][r], v2 = o[2][r], v3 = o[3][r];
            hx_store(rzz, fmaf(v0, c2, zres[0][r]) * m, voff, so);
            hx_store(rzz, fmaf(v1, c2, zres[1][r]) * m, voff, so + 128);
            hx_store(rzz, fmaf(v2, c2, zres[2][r]) * m, voff, so + 256);
            hx_store(rzz, fmaf(v3, c2, zres[3][r]) * m, voff, so + 384);
        }
        if (more) {
            if (prefetch) hx_zt_read(raw, zt, pl, h, 1);
            else load_sync(tile + gridDim.x);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, projections (modules/triangular_multiplicative_update.py:93-103):
//   a = (W_ap zn + b) sigmoid(W_ag zn + b) mask,  b likewise;  zn = LN_in(z).
// Stage = pass: units 2kc = the pass's 32 projection rows, 2kc + 1 = its 32 gate rows (pre-scaled by
// -log2 e), k-chunk kc.  D rows = channels, D cols = pairs, so every store is a 128-B run of the
// channel-major operand image the contraction reads; a and b are stored already SPLIT
// (hi | lo << 16 of a S_a), 4 bytes per element as before.
// ---------------------------------------------------------------------------------------------
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

template <bool OUTGOING, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void k_trimul_proj_hx(
    const float* __restrict__ z, const float* __restrict__ rmask, const unsigned char* __restrict__ wimg,
    const float* __restrict__ bias, unsigned* __restrict__ acm, unsigned* __restrict__ bcm, int N, int NP, int n_wtiles,
    unsigned cm_bytes, unsigned z_bytes, float sx, float cpa, float cpb, float cg, int rev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    float* sbias = reinterpret_cast<float*>(smb + 2 * HX_STAGE_BYTES);      // [512]
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int n_tiles = (n_wtiles + NW - 1) / NW;
    const rsrc_t rw = hx_rsrc(wimg, 8 * HX_STAGE_BYTES);
    const rsrc_t ra = hx_rsrc(acm, cm_bytes), rb = hx_rsrc(bcm, cm_bytes);
    const int lane16 = lane * 16;
    const int sstride = NP * NP * 4;                      // bytes per channel
    constexpr int PW = 32 / NW;
    auto issue = [&](int pass, int buf) {
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int p = PW * wave + q;
            hx_dma(rw, smb + buf * HX_STAGE_BYTES + p * 1024, lane16, pass * HX_STAGE_BYTES + p * 1024);
        }
    };
    unsigned char* zt = smb + 2 * HX_STAGE_BYTES + 2048 + wave * HX_ZT_BYTES;
    const rsrc_t rz = hx_rsrc(z, z_bytes);
    const int zstride = OUTGOING ? 512 : N * 512;         // bytes between consecutive pairs of a tile
    auto tile_geom = [&](int tile, int& soff, int& nv) {  // wave-tile NW tile + wave (clamped): byte offset of its first row, valid rows
        const int wt = min(HX_PHYS(tile) * NW + wave, n_wtiles - 1);
        const int st = wt % ntile, line = (wt / ntile) % N, b = wt / (ntile * N);
        nv = min(32, N - st * 32);
        soff = OUTGOING ? ((b * N + line) * N + st * 32) * 512 : ((b * N + st * 32) * N + line) * 512;
    };
    int tile = blockIdx.x;
    HX_TS_DECL(OUTGOING ? 1 : 0);
    issue(0, 0);
    for (int u = threadIdx.x; u < 512; u += NW * 64) sbias[u] = bias[u];
    __syncthreads();
    unsigned wst[8];
    f32x16 apA, agA, apB, agB;                            // set A: even passes, set B: odd passes
#pragma unroll
    for (int r = 0; r < 16; ++r) { apB[r] = 0.f; agB[r] = 0.f; apA[r] = sbias[acc_row(r, lane)]; agA[r] = sbias[256 + acc_row(r, lane)]; }
    h8 zh[8], zl[8];
    float4 raw[16];
    int n_soff, n_nv;
    tile_geom(tile, n_soff, n_nv);
    hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 0); hx_vm_done(); hx_zt_read(raw, zt, pl, h, 0); hx_lds_done();
    hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 1); hx_vm_done(); hx_zt_read(raw, zt, pl, h, 1);
    int p_voff = 0x7FFFFFF0, p_so = 0;                    // epilogue state of the previous tile's pass 7 (none yet: stores dropped)
    float p_pm = 0.f;
#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = HX_PHYS(tile) * NW + wave;
        const bool act = wt_raw < n_wtiles;
        const int wt = act ? wt_raw : n_wtiles - 1;      // idle waves shadow the last tile (stores dropped)
        const int st = wt % ntile, line = (wt / ntile) % N, b = wt / (ntile * N);
        const int t0i = st * 32;
        const int nvalid = act ? min(32, N - t0i) : 0;
        const float msk = (pl < nvalid) ? rmask[b * N + line] * rmask[b * N + t0i + pl] : 0.f;
        const float ma = msk * cpa, mb = msk * cpb;
        // channel-major store: element ((b*128 + ch)*NP + line)*NP + t0 + pl, ch = 32 (pass & 3) + row
        const int voff = (pl < nvalid) ? (4 * h * NP * NP + pl) * 4 : 0x7FFFFFF0;   // out-of-range offset: store dropped
        const int sbase = ((b * 128 * NP + line) * NP + t0i) * 4;
        const bool more = tile + (int)gridDim.x < n_tiles;
        HX_TS();
        hx_norm_split(zh, zl, raw, sx);
#pragma unroll 1
        for (int pp = 0; pp < 4; ++pp) {
            {   // even pass 2pp -> set A; epilogue of pass 2pp - 1 (set B; for pp = 0 the previous tile's pass 7)
                const int pass = 2 * pp;
                HX_TS();
                if (more) {                               // next tile's rows: first half -> raw at pass 4 (requested at the ends of passes 0..2)
                    if (pp == 0) tile_geom(tile + gridDim.x, n_soff, n_nv);
                    if (pp == 2) { hx_zt_read(raw, zt, pl, h, 0); hx_lds_done(); }
                }
                issue(pass + 1, 1);
                const unsigned char* stage = smb;
                const rsrc_t e_rd = (pp == 0 || pp > 2) ? rb : ra;
                const int e_voff = pp == 0 ? p_voff : voff;
                const int e_so = pp == 0 ? p_so : sbase + ((pass - 1) & 3) * 32 * sstride;
                const float e_pm = pp == 0 ? p_pm : (pp > 2 ? mb : ma);
                HX_PROJ_STAGE(apA, agA, apB, agB);
                HX_TS();
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // weights of the next stage + the row-tile pieces requested a stage ago
                if (more) {                               // row-tile requests AFTER the stage's last store: passes 0, 2 -> half 0; 4, 6 -> half 1
                    if (pp == 0) hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 0, 0, 3);
                    else if (pp == 1) hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 0, 6, 2);
                    else if (pp == 2) hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 1, 0, 3);
                    else hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 1, 6, 2);
                }
                HX_TS(); hx_stage_barrier();
                HX_TS();
            }
            {   // odd pass 2pp + 1 -> set B; epilogue of pass 2pp (set A)
                const int pass = 2 * pp + 1;
                HX_TS();
                if (pp < 3) issue(pass + 1, 0);
                else if (more) issue(0, 0);
                const unsigned char* stage = smb + HX_STAGE_BYTES;
                const rsrc_t e_rd = pp < 2 ? ra : rb;
                const int e_voff = voff;
                const int e_so = sbase + ((pass - 1) & 3) * 32 * sstride;
                const float e_pm = pp < 2 ? ma : mb;
                HX_PROJ_STAGE(apB, agB, apA, agA);
                HX_TS();
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                if (more) {                               // passes 1 -> half 0, 5 -> half 1 (none after passes 3 and 7: the halves are read next)
                    if (pp == 0) hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 0, 3, 3);
                    else if (pp == 2) hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 1, 3, 3);
                }
                HX_TS(); hx_stage_barrier();
                HX_TS();
            }
        }
        p_voff = voff; p_so = sbase + 3 * 32 * sstride; p_pm = mb;
        if (more) hx_zt_read(raw, zt, pl, h, 1);          // second half: read at the tile boundary (register pressure)
    }
    {   // drain: epilogue of the last pass 7 (set B)
        const rsrc_t e_rd = rb;
        const int e_voff = p_voff, e_so = p_so;
        const float e_pm = p_pm;
        float t[16], u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            HX_PROJ_PIECE_A(agB, r, t[r]);
            HX_PROJ_PIECE_B(apB, r, t[r], u[r]);
        }
        PIPE_FENCE();           // (the packs are asm: keep them clear of the v_rcp results' forwarding window)
#pragma unroll
        for (int r = 0; r < 16; ++r) HX_PROJ_PIECE_C_NOW(r, t[r], u[r]);
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, contraction (triangular_multiplicative_update.py:57-82):
//   x_cm[bc][i][j] = sum_k a_cm[bc][i][k] b_cm[bc][j][k] / (S_a S_b),  operands stored split.
// WG tile (64 WT)^2, 4 waves of (32 WT)^2, K streamed 16 at a time: each thread loads 16 B (4 split
// values) per 64 rows, de-interleaves them with two v_perm (hi halves / lo halves) and writes the
// two 8-B pieces into the hi / lo planes of the double-buffered LDS stage (row stride 48 B: the
// ds_read_b128 fragment reads are conflict free).  HBM-bound in this arithmetic (36 FLOP/B).
// Persistent + XCD-aware tile order as in pair_kernels.hip.
// ---------------------------------------------------------------------------------------------
#define CX_ROWB 48
template <int WT>
__global__ __launch_bounds__(256, 2) void k_trimul_contract_hx(const unsigned* __restrict__ acm, const unsigned* __restrict__ bcm,
                                                               float* __restrict__ xcm, int NP, int n_mat, unsigned cm_bytes, float cx, int rev) {
    constexpr int TM = 64 * WT;
    constexpr int PLANE = TM * CX_ROWB;                 // bytes of one (operand, half) plane
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];   // [2 buf][A hi | A lo | B hi | B lo]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = UNI(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = (NP + TM - 1) / TM;
    const int T = tiles * tiles;
    const int n_tiles = T * ((n_mat + 7) / 8) * 8;
    const int nk = NP / 16;
    const rsrc_t ra = hx_rsrc(acm, cm_bytes), rb = hx_rsrc(bcm, cm_bytes), rx = hx_rsrc(xcm, cm_bytes);
    const int lr = tid >> 2, c4 = tid & 3;
    const int vload = (lr * NP + c4 * 4) * 4;           // per-lane byte offset inside a 64-row block of a panel
    const int slds = lr * CX_ROWB + c4 * 8;
    u32x4 rA[WT], rB[WT];

    auto decode = [&](int wl, int& mi, int& i0, int& j0) {      // tile id -> (matrix, tile origin); `rev`: last matrices first
        const int w = rev ? n_tiles - 1 - wl : wl;
        const int grp = w / (8 * T), rem = w % (8 * T);
        const int tile = rem >> 3;
        mi = grp * 8 + (rem & 7);
        i0 = (tile / tiles) * TM;
        j0 = (tile % tiles) * TM;
    };
    auto gload = [&](int w, int kc) {
        int mi, i0, j0;
        decode(w, mi, i0, j0);
        const int mclamp = min(mi, n_mat - 1);
        const int mbase = mclamp * NP * NP * 4 + kc * 64;
#pragma unroll
        for (int u = 0; u < WT; ++u) {     // rows past NP: voffset pushed out of the buffer -> the load returns 0
            const int va = (i0 + 64 * u + lr < NP) ? vload : 0x7FFFFFF0, vb = (j0 + 64 * u + lr < NP) ? vload : 0x7FFFFFF0;
            rA[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, va, mbase + (i0 + 64 * u) * NP * 4, 0));
            rB[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, vb, mbase + (j0 + 64 * u) * NP * 4, 0));
        }
    };
    auto swrite = [&](int buf) {
        unsigned char* sa = smb + buf * 4 * PLANE + slds;
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            uint2 ahi, alo, bhi, blo;
            ahi.x = __builtin_amdgcn_perm(rA[u].y, rA[u].x, 0x05040100u); ahi.y = __builtin_amdgcn_perm(rA[u].w, rA[u].z, 0x05040100u);
            alo.x = __builtin_amdgcn_perm(rA[u].y, rA[u].x, 0x07060302u); alo.y = __builtin_amdgcn_perm(rA[u].w, rA[u].z, 0x07060302u);
            bhi.x = __builtin_amdgcn_perm(rB[u].y, rB[u].x, 0x05040100u); bhi.y = __builtin_amdgcn_perm(rB[u].w, rB[u].z, 0x05040100u);
            blo.x = __builtin_amdgcn_perm(rB[u].y, rB[u].x, 0x07060302u); blo.y = __builtin_amdgcn_perm(rB[u].w, rB[u].z, 0x07060302u);
            *reinterpret_cast<uint2*>(sa + 64 * u * CX_ROWB) = ahi;
            *reinterpret_cast<uint2*>(sa + PLANE + 64 * u * CX_ROWB) = alo;
            *reinterpret_cast<uint2*>(sa + 2 * PLANE + 64 * u * CX_ROWB) = bhi;
            *reinterpret_cast<uint2*>(sa + 3 * PLANE + 64 * u * CX_ROWB) = blo;
        }
    };

    f32x16 acc[WT][WT];
#pragma unroll
    for (int m = 0; m < WT; ++m)
#pragma unroll
        for (int n = 0; n < WT; ++n) acc[m][n] = zero16();

    int w = blockIdx.x;
    if (w >= n_tiles) return;
    const int my_tiles = (n_tiles - 1 - w) / gridDim.x + 1;
    const int total = my_tiles * nk;
    gload(w, 0);
    swrite(0);
    __syncthreads();
    int kc = 0;
    const int foff = (lane & 31) * CX_ROWB + (lane >> 5) * 16;
    for (int it = 0; it < total; ++it) {
        const bool last_chunk = kc == nk - 1;
        const int wn_next = last_chunk ? w + gridDim.x : w;
        const int kc_next = last_chunk ? 0 : kc + 1;
        if (it + 1 < total) gload(wn_next, kc_next);
        const unsigned char* sa = smb + (it & 1) * 4 * PLANE + foff;
        h8 ah[WT], al[WT], bh[WT], bl[WT];
#pragma unroll
        for (int m = 0; m < WT; ++m) {
            ah[m] = *reinterpret_cast<const h8*>(sa + (wm * WT + m) * 32 * CX_ROWB);
            al[m] = *reinterpret_cast<const h8*>(sa + PLANE + (wm * WT + m) * 32 * CX_ROWB);
        }
#pragma unroll
        for (int n = 0; n < WT; ++n) {
            bh[n] = *reinterpret_cast<const h8*>(sa + 2 * PLANE + (wn * WT + n) * 32 * CX_ROWB);
            bl[n] = *reinterpret_cast<const h8*>(sa + 3 * PLANE + (wn * WT + n) * 32 * CX_ROWB);
        }
#pragma unroll
        for (int m = 0; m < WT; ++m)
#pragma unroll
            for (int n = 0; n < WT; ++n) MFH3(ah[m], al[m], bh[n], bl[n], acc[m][n]);
        if (last_chunk) {       // store the finished tile (row = acc_row(r): 4 (lane>>5) in voffset, the rest scalar)
            int mi, i0, j0;
            decode(w, mi, i0, j0);
            if (mi < n_mat) {
                const int vst = ((4 * (lane >> 5)) * NP + (lane & 31)) * 4;
#pragma unroll
                for (int m = 0; m < WT; ++m)
#pragma unroll
                    for (int n = 0; n < WT; ++n) {
                        const int ib = i0 + (wm * WT + m) * 32, jb = j0 + (wn * WT + n) * 32;
                        if (ib < NP && jb < NP) {
                            const int sbase = ((mi * NP + ib) * NP + jb) * 4;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = acc[m][n][r] * cx;     // (bit_cast straight from a vector element picks element 0)
                                hx_store(rx, v, vst, sbase + ((r & 3) + 8 * (r >> 2)) * NP * 4);
                            }
                        }
                        acc[m][n] = zero16();
                    }
            }
        }
        if (it + 1 < total) swrite((it + 1) & 1);
        __syncthreads();
        w = wn_next;
        kc = kc_next;
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, output (modules/triangular_multiplicative_update.py:105-108 + residual):
//   z += (W_z LN_out(x) + b_z) * sigmoid(W_g LN_in(z) + b_g).
// A operand = the pair tile (zn, then xn), B = weights; stages W_g{0,1}, W_z{0,1}, W_g{2,3}, W_z{2,3},
// unit = 8 (output block within the stage) + k-chunk.  x arrives channel-major: lane (p, h) reads
// x[c][p] for its 64 channels c = 16kc + 8h + e (each load = two 128-B runs).
// ---------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void k_trimul_out_hx(
    float* __restrict__ z, const float* __restrict__ xcm, const unsigned char* __restrict__ wimg,
    const float* __restrict__ bgs, const float* __restrict__ bzs, int N, int NP, int n_wtiles, unsigned cm_bytes,
    unsigned z_bytes, float sx, float cg, float cz, int rev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int n_tiles = (n_wtiles + NW - 1) / NW;
    const rsrc_t rw = hx_rsrc(wimg, 4 * HX_STAGE_BYTES);
    const rsrc_t rx = hx_rsrc(xcm, cm_bytes), rz = hx_rsrc(z, z_bytes);
    const int lane16 = lane * 16;
    constexpr int PW = 32 / NW;
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int p = PW * wave + q;
            hx_dma(rw, smb + buf * HX_STAGE_BYTES + p * 1024, lane16, s * HX_STAGE_BYTES + p * 1024);
        }
    };
    unsigned char* zt = smb + 2 * HX_STAGE_BYTES + 2048 + wave * HX_ZT_BYTES;
    int tile = blockIdx.x;
    issue(0, 0);
    __syncthreads();
#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = HX_PHYS(tile) * NW + wave;
        const bool act = wt_raw < n_wtiles;
        const int wt = act ? wt_raw : n_wtiles - 1;
        const int st = wt % ntile, i = (wt / ntile) % N, b = wt / (ntile * N);
        const int t0 = st * 32;
        const int nvalid = act ? min(32, N - t0) : 0;
        const int prow0 = (b * N + i) * N + t0;                  // first pair row of the tile
        const bool more = tile + (int)gridDim.x < n_tiles;
        h8 zh[8], zl[8], xh[8], xl[8];
        {   // x_cm[((b*128 + c)*NP + i)*NP + t0 + pl], c = 16kc + 8h + e   (t0 + pl < NP always); the z rows come through the
            // coalesced LDS loader (two halves), the second half's latency is spent on LayerNorm + split of x
            float4 raw[16], rawz[16];
            const int cs = NP * NP * 4;
            const int vx = (8 * h * NP * NP + pl) * 4;
            const int sxo = ((b * 128 * NP + i) * NP + t0) * 4;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                raw[2 * kc].x = hx_load(rx, vx, sxo + (16 * kc + 0) * cs); raw[2 * kc].y = hx_load(rx, vx, sxo + (16 * kc + 1) * cs);
                raw[2 * kc].z = hx_load(rx, vx, sxo + (16 * kc + 2) * cs); raw[2 * kc].w = hx_load(rx, vx, sxo + (16 * kc + 3) * cs);
                raw[2 * kc + 1].x = hx_load(rx, vx, sxo + (16 * kc + 4) * cs); raw[2 * kc + 1].y = hx_load(rx, vx, sxo + (16 * kc + 5) * cs);
                raw[2 * kc + 1].z = hx_load(rx, vx, sxo + (16 * kc + 6) * cs); raw[2 * kc + 1].w = hx_load(rx, vx, sxo + (16 * kc + 7) * cs);
            }
            const int zsoff = prow0 * 512, znv = min(32, N - t0);
            hx_zt_dma(rz, zt, lane, zsoff, 512, znv, 0); hx_vm_done(); hx_zt_read(rawz, zt, pl, h, 0); hx_lds_done();
            hx_zt_dma(rz, zt, lane, zsoff, 512, znv, 1);
            hx_norm_split(xh, xl, raw, sx);
            hx_vm_done(); hx_zt_read(rawz, zt, pl, h, 1);
            hx_norm_split(zh, zl, rawz, sx);
        }
        const int voff = (4 * h * 128 + pl) * 4;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 ga, gb;
            {   // gate: A = zn fragments (i = pair), B = W_g (j = channel)
                issue(2 * half + 1, 1);
                const unsigned char* stage = smb;
                float c0 = bgs[(2 * half) * 32 + pl], c1 = bgs[(2 * half + 1) * 32 + pl];
                asm volatile("" : "+v"(c0), "+v"(c1));      // keep hipcc from hoisting (and spilling) the 16-register splats
#pragma unroll
                for (int r = 0; r < 16; ++r) { ga[r] = c0; gb[r] = c1; }
                h8 f0 = hx_frag(stage, 0, 0, lane), f0l = hx_frag(stage, 0, 1, lane), f1 = hx_frag(stage, 8, 0, lane),
                   f1l = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const int kn = min(kc + 1, 7);
                    const h8 n0 = hx_frag(stage, kn, 0, lane), n0l = hx_frag(stage, kn, 1, lane), n1 = hx_frag(stage, 8 + kn, 0, lane),
                             n1l = hx_frag(stage, 8 + kn, 1, lane);
                    PIPE_FENCE();
                    MFH3(zh[kc], zl[kc], f0, f0l, ga);
                    MFH3(zh[kc], zl[kc], f1, f1l, gb);
                    PIPE_FENCE();
                    f0 = n0; f0l = n0l; f1 = n1; f1l = n1l;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ga[r] = cz * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ga[r] * cg));
                    gb[r] = cz * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gb[r] * cg));
                }
                __syncthreads();
            }
            {   // update of channel blocks 2 half, 2 half + 1
                if (half == 0) issue(2, 0);
                else if (more) issue(0, 0);
                const unsigned char* stage = smb + HX_STAGE_BYTES;
                const int ob = 2 * half;
                f32x16 a0, a1;
                float c0 = bzs[ob * 32 + pl], c1 = bzs[(ob + 1) * 32 + pl];
                asm volatile("" : "+v"(c0), "+v"(c1));
#pragma unroll
                for (int r = 0; r < 16; ++r) { a0[r] = c0; a1[r] = c1; }
                const int h4 = 4 * h;
                float zp0[16], zp1[16];
                if (half == 1) {    // zh / zl are dead after the second gate: the residual rows of this half are requested before its GEMM
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rc = (r & 3) + 8 * (r >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        const int vr = (h4 < nvalid - rc) ? voff : 0x7FFFFFF0;
                        zp0[r] = hx_load(rz, vr, so);
                        zp1[r] = hx_load(rz, vr, so + 128);
                    }
                }
                h8 f0 = hx_frag(stage, 0, 0, lane), f0l = hx_frag(stage, 0, 1, lane), f1 = hx_frag(stage, 8, 0, lane),
                   f1l = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const int kn = min(kc + 1, 7);
                    const h8 n0 = hx_frag(stage, kn, 0, lane), n0l = hx_frag(stage, kn, 1, lane), n1 = hx_frag(stage, 8 + kn, 0, lane),
                             n1l = hx_frag(stage, 8 + kn, 1, lane);
                    PIPE_FENCE();
                    MFH3(xh[kc], xl[kc], f0, f0l, a0);
                    MFH3(xh[kc], xl[kc], f1, f1l, a1);
                    PIPE_FENCE();
                    f0 = n0; f0l = n0l; f1 = n1; f1l = n1l;
                }
                hx_stage_landed();
                // residual + store, 8 rows (16 loads) in flight at a time; rows past the tile's valid pairs belong to
                // the next line: their offset is pushed out of the buffer (load gives 0, store is dropped)
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    float zr0[8], zr1[8];
                    int vo[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int rc = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        vo[q] = (h4 < nvalid - rc) ? voff : 0x7FFFFFF0;
                        if (half == 1) { zr0[q] = zp0[r0 + q]; zr1[q] = zp1[r0 + q]; }
                        else { zr0[q] = hx_load(rz, vo[q], so); zr1[q] = hx_load(rz, vo[q], so + 128); }
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int rc = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        const float u0 = a0[r0 + q], u1 = a1[r0 + q], g0 = ga[r0 + q], g1 = gb[r0 + q];
                        hx_store(rz, fmaf(u0, g0, zr0[q]), vo[q], so);
                        hx_store(rz, fmaf(u1, g1, zr1[q]), vo[q], so + 128);
                    }
                    PIPE_FENCE();
                }
                hx_stage_barrier();
            }
        }
    }
}

// The same kernel with all four weight stages resident in LDS (128 KiB: no weight stream, no barrier inside the tile loop --
// the eight waves of a work-group run their tiles independently) and the z rows loaded straight into operand registers
// (no LDS left for the row-tile loader).
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void k_trimul_out_hx_r(
    float* __restrict__ z, const float* __restrict__ xcm, const unsigned char* __restrict__ wimg,
    const float* __restrict__ bgs, const float* __restrict__ bzs, int N, int NP, int n_wtiles, unsigned cm_bytes,
    unsigned z_bytes, float sx, float cg, float cz, int rev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int n_tiles = (n_wtiles + NW - 1) / NW;
    const rsrc_t rw = hx_rsrc(wimg, 4 * HX_STAGE_BYTES);
    const rsrc_t rx = hx_rsrc(xcm, cm_bytes), rz = hx_rsrc(z, z_bytes);
    const int lane16 = lane * 16;
    constexpr int PW = 32 / NW;
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int p = PW * wave + q;
            hx_dma(rw, smb + buf * HX_STAGE_BYTES + p * 1024, lane16, s * HX_STAGE_BYTES + p * 1024);
        }
    };
    int tile = blockIdx.x;
    issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);      // all four weight stages stay in LDS (128 KiB) for every tile
    hx_stage_landed();
    __syncthreads();
#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = HX_PHYS(tile) * NW + wave;
        const bool act = wt_raw < n_wtiles;
        const int wt = act ? wt_raw : n_wtiles - 1;
        const int st = wt % ntile, i = (wt / ntile) % N, b = wt / (ntile * N);
        const int t0 = st * 32;
        const int nvalid = act ? min(32, N - t0) : 0;
        const int prow0 = (b * N + i) * N + t0;                  // first pair row of the tile
        h8 zh[8], zl[8], xh[8], xl[8];
        {   // x_cm[((b*128 + c)*NP + i)*NP + t0 + pl], c = 16kc + 8h + e   (t0 + pl < NP always); the z rows come through the
            // coalesced LDS loader (two halves), the second half's latency is spent on LayerNorm + split of x
            float4 raw[16], rawz[16];
            const int cs = NP * NP * 4;
            const int vx = (8 * h * NP * NP + pl) * 4;
            const int sxo = ((b * 128 * NP + i) * NP + t0) * 4;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                raw[2 * kc].x = hx_load(rx, vx, sxo + (16 * kc + 0) * cs); raw[2 * kc].y = hx_load(rx, vx, sxo + (16 * kc + 1) * cs);
                raw[2 * kc].z = hx_load(rx, vx, sxo + (16 * kc + 2) * cs); raw[2 * kc].w = hx_load(rx, vx, sxo + (16 * kc + 3) * cs);
                raw[2 * kc + 1].x = hx_load(rx, vx, sxo + (16 * kc + 4) * cs); raw[2 * kc + 1].y = hx_load(rx, vx, sxo + (16 * kc + 5) * cs);
                raw[2 * kc + 1].z = hx_load(rx, vx, sxo + (16 * kc + 6) * cs); raw[2 * kc + 1].w = hx_load(rx, vx, sxo + (16 * kc + 7) * cs);
            }
            {   // this lane's pair row, channels 16 kc + 8 h .. + 7 (rows past the tile: clamped)
                const float* zrow = z + ((size_t)(prow0 + min(pl, min(32, N - t0) - 1))) * 128 + 8 * h;
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    rawz[2 * kc] = *reinterpret_cast<const float4*>(zrow + 16 * kc);
                    rawz[2 * kc + 1] = *reinterpret_cast<const float4*>(zrow + 16 * kc + 4);
                }
            }
            hx_norm_split(xh, xl, raw, sx);
            hx_norm_split(zh, zl, rawz, sx);
        }
        const int voff = (4 * h * 128 + pl) * 4;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 ga, gb;
            {   // gate: A = zn fragments (i = pair), B = W_g (j = channel)
                const unsigned char* stage = smb + (2 * half) * HX_STAGE_BYTES;
                float c0 = bgs[(2 * half) * 32 + pl], c1 = bgs[(2 * half + 1) * 32 + pl];
                asm volatile("" : "+v"(c0), "+v"(c1));      // keep hipcc from hoisting (and spilling) the 16-register splats
#pragma unroll
                for (int r = 0; r < 16; ++r) { ga[r] = c0; gb[r] = c1; }
                h8 f0 = hx_frag(stage, 0, 0, lane), f0l = hx_frag(stage, 0, 1, lane), f1 = hx_frag(stage, 8, 0, lane),
                   f1l = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const int kn = min(kc + 1, 7);
                    const h8 n0 = hx_frag(stage, kn, 0, lane), n0l = hx_frag(stage, kn, 1, lane), n1 = hx_frag(stage, 8 + kn, 0, lane),
                             n1l = hx_frag(stage, 8 + kn, 1, lane);
                    PIPE_FENCE();
                    MFH3(zh[kc], zl[kc], f0, f0l, ga);
                    MFH3(zh[kc], zl[kc], f1, f1l, gb);
                    PIPE_FENCE();
                    f0 = n0; f0l = n0l; f1 = n1; f1l = n1l;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ga[r] = cz * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ga[r] * cg));
                    gb[r] = cz * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gb[r] * cg));
                }
            }
            {   // update of channel blocks 2 half, 2 half + 1
                const unsigned char* stage = smb + (2 * half + 1) * HX_STAGE_BYTES;
                const int ob = 2 * half;
                f32x16 a0, a1;
                float c0 = bzs[ob * 32 + pl], c1 = bzs[(ob + 1) * 32 + pl];
                asm volatile("" : "+v"(c0), "+v"(c1));
#pragma unroll
                for (int r = 0; r < 16; ++r) { a0[r] = c0; a1[r] = c1; }
                const int h4 = 4 * h;
                float zp0[16], zp1[16];
                if (half == 1) {    // zh / zl are dead after the second gate: the residual rows of this half are requested before its GEMM
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rc = (r & 3) + 8 * (r >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        const int vr = (h4 < nvalid - rc) ? voff : 0x7FFFFFF0;
                        zp0[r] = hx_load(rz, vr, so);
                        zp1[r] = hx_load(rz, vr, so + 128);
                    }
                }
                h8 f0 = hx_frag(stage, 0, 0, lane), f0l = hx_frag(stage, 0, 1, lane), f1 = hx_frag(stage, 8, 0, lane),
                   f1l = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const int kn = min(kc + 1, 7);
                    const h8 n0 = hx_frag(stage, kn, 0, lane), n0l = hx_frag(stage, kn, 1, lane), n1 = hx_frag(stage, 8 + kn, 0, lane),
                             n1l = hx_frag(stage, 8 + kn, 1, lane);
                    PIPE_FENCE();
                    MFH3(xh[kc], xl[kc], f0, f0l, a0);
                    MFH3(xh[kc], xl[kc], f1, f1l, a1);
                    PIPE_FENCE();
                    f0 = n0; f0l = n0l; f1 = n1; f1l = n1l;
                }
                // residual + store, 8 rows (16 loads) in flight at a time; rows past the tile's valid pairs belong to
                // the next line: their offset is pushed out of the buffer (load gives 0, store is dropped)
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    float zr0[8], zr1[8];
                    int vo[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int rc = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        vo[q] = (h4 < nvalid - rc) ? voff : 0x7FFFFFF0;
                        if (half == 1) { zr0[q] = zp0[r0 + q]; zr1[q] = zp1[r0 + q]; }
                        else { zr0[q] = hx_load(rz, vo[q], so); zr1[q] = hx_load(rz, vo[q], so + 128); }
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int rc = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        const float u0 = a0[r0 + q], u1 = a1[r0 + q], g0 = ga[r0 + q], g1 = gb[r0 + q];
                        hx_store(rz, fmaf(u0, g0, zr0[q]), vo[q], so);
                        hx_store(rz, fmaf(u1, g1, zr1[q]), vo[q], so + 128);
                    }
                    PIPE_FENCE();
                }
            }
        }
    }
}

// (Tried: the output kernel transposed, D[channel][pair] with A = weights, so that a lane holds its pair row on both sides of
//  the GEMMs -- four v_permlane32_swap per k-chunk turn the loaded z row into the accumulator's row set, which is then both the
//  residual and, with the gate weights packed in that k order, the gate's B operand; LayerNorm applied while splitting, twice
//  per operand, to stay inside 256 registers; one float4 store per lane and 4-channel group.  Parity green, no second read of
//  z (-268 MB of L2 reads per launch), but 0.254 ms against 0.237: the doubled split work and the row-per-lane 16-B stores
//  cost more than the re-read.  tools/probe/swap_probe.hip is the check of the swap's semantics.)
// ---------------------------------------------------------------------------------------------
static int g_hx_cu = 0;
static int hx_num_cu() {
    if (!g_hx_cu) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&pr, dev); g_hx_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    return g_hx_cu;
}
static int hx_nw() { return 8; }   // waves per work-group (one work-group per CU: 130 KiB of LDS)
// `cus`: CUs this launch may fill (all of them, or half when two halves of the batch run side by side)
static unsigned hx_grid(long long n_tiles, int nw, int cus) {
    const long long cap = (nw == 8 ? 1LL : 2LL) * cus;
    return (unsigned)(n_tiles < cap ? n_tiles : cap);
}

// A contiguous slice of the batch (structures b0 .. b0 + nb - 1) on its own stream: every pair-stack kernel is separable over
// the batch (outermost dimension of p, a, b, x and of the mask), so a slice is a pointer offset and a smaller tile count.
struct HxSlice { int b0, nb; hipStream_t st; int cus; unsigned launches; bool prof; };

static void pair_transition_slice(genie_ctx* h, HxSlice& v, const PairLayerW& w) {
    const int N = h->N;
    const long long M = (long long)v.nb * N * N;
    const long long n_wt = (M + 31) / 32;
    const int n_hb = h->d.pair_transition_n * 4;
    const HxTransW& x = w.hx_pt;
    hipLaunchKernelGGL(k_pair_transition_hx<8>, dim3(hx_grid((n_wt + 7) / 8, 8, v.cus)), dim3(512), HX_LDS_BYTES, v.st,
                       h->p + (size_t)v.b0 * N * N * 128, h->rmaskf + (size_t)v.b0 * N, x.img, x.b1s, x.b2s, N, M, n_hb, x.sx, x.c1, x.c2,
                       (int)(v.launches++ & 1));
}

static void trimul_slice(genie_ctx* h, HxSlice& v, const TriMulW& w, bool outgoing) {
    const int N = h->N, NP = h->NP, ntile = (N + 31) / 32;
    const int nw = hx_nw();
    const HxTriW& x = w.hx;
    const size_t cm_off = (size_t)v.b0 * 128 * NP * NP;
    unsigned* acm = reinterpret_cast<unsigned*>(h->acm) + cm_off;
    unsigned* bcm = reinterpret_cast<unsigned*>(h->bcm) + cm_off;
    float* xcm = h->xcm + cm_off;
    const int n_wt = v.nb * N * ntile;
    const unsigned cm_bytes = (unsigned)((size_t)v.nb * 128 * NP * NP * 4);
    const unsigned z_bytes = (unsigned)((size_t)v.nb * N * N * 512);
    float* zs = h->p + (size_t)v.b0 * N * N * 128;
    const float* ms = h->rmaskf + (size_t)v.b0 * N;
    hipStream_t st = v.st;
    {
        ProfScope ps(h, st, KC_TRIMUL_PROJ, v.prof);
        const dim3 grid(hx_grid((n_wt + nw - 1) / nw, nw, v.cus)), block(nw * 64);
        const int rev = (int)(v.launches++ & 1);
#define HX_PROJ(OUT, NWV) hipLaunchKernelGGL((k_trimul_proj_hx<OUT, NWV>), grid, block, HX_LDS_BYTES, st, zs, ms, x.img_proj, \
                                             x.bias_proj, acm, bcm, N, NP, n_wt, cm_bytes, z_bytes, x.sx, x.cpa, x.cpb, x.cg, rev)
        if (outgoing) HX_PROJ(true, 8);
        else HX_PROJ(false, 8);
#undef HX_PROJ
    }
    {
        ProfScope ps(h, st, KC_TRIMUL_CONTRACT, v.prof);
        const int BC = v.nb * h->d.c_hidden_mul;
        const int ncu = v.cus;
        const int rev = (int)(v.launches++ & 1);
        if (NP >= 128) {
            const int tiles = (NP + 127) / 128;
            const int n_tiles = tiles * tiles * ((BC + 7) / 8) * 8;
            hipLaunchKernelGGL(k_trimul_contract_hx<2>, dim3(n_tiles < 3 * ncu ? n_tiles : 3 * ncu), dim3(256), 2 * 4 * 128 * CX_ROWB, st,
                               acm, bcm, xcm, NP, BC, cm_bytes, x.cx, rev);
        } else {
            const int tiles = (NP + 63) / 64;
            const int n_tiles = tiles * tiles * ((BC + 7) / 8) * 8;
            hipLaunchKernelGGL(k_trimul_contract_hx<1>, dim3(n_tiles < 4 * ncu ? n_tiles : 4 * ncu), dim3(256), 2 * 4 * 64 * CX_ROWB, st,
                               acm, bcm, xcm, NP, BC, cm_bytes, x.cx, rev);
        }
    }
    {
        ProfScope ps(h, st, KC_TRIMUL_OUT, v.prof);
        const bool resident = getenv("GENIE_OUT_STREAMED") == nullptr;     // default: weights resident in LDS (0.233 vs 0.240 ms per launch)
        if (resident)
            hipLaunchKernelGGL(k_trimul_out_hx_r<8>, dim3(hx_grid((n_wt + 7) / 8, 8, v.cus)), dim3(512), 4 * HX_STAGE_BYTES, st, zs, xcm,
                               x.img_out, x.bgs, x.bzs, N, NP, n_wt, cm_bytes, z_bytes, x.sx, x.cgo, x.cz, (int)(v.launches++ & 1));
        else
        hipLaunchKernelGGL(k_trimul_out_hx<8>, dim3(hx_grid((n_wt + 7) / 8, 8, v.cus)), dim3(512), HX_LDS_BYTES, st, zs, xcm, x.img_out,
                           x.bgs, x.bzs, N, NP, n_wt, cm_bytes, z_bytes, x.sx, x.cgo, x.cz, (int)(v.launches++ & 1));
    }
}

void launch_pair_transition_hx(genie_ctx* h, hipStream_t st, const PairLayerW& w) {
    HxSlice v{0, h->B, st, hx_num_cu(), h->hx_launches, h->prof};
    pair_transition_slice(h, v, w);
    h->hx_launches = v.launches;
}

// GENIE_HX_SLICE=n runs the three kernels of a triangle multiplication per slice of n structures, so that (n = 2, N = 256:
// a + b + x = 201 MB) the operands stay in the 256-MiB Infinity Cache between producer and consumer.  Measured: no gain
// (82.1 / 80.4 / 83.0 / 80.9 batch-steps/s for n = 8 / 4 / 2 / 1), so the default is the whole batch; kept as a switch for larger N.
static int g_hx_slice = 0;
static int hx_slice() {
    if (!g_hx_slice) { const char* e = getenv("GENIE_HX_SLICE"); g_hx_slice = e && atoi(e) > 0 ? atoi(e) : 1 << 20; }
    return g_hx_slice;
}

void launch_trimul_hx(genie_ctx* h, hipStream_t st, const TriMulW& w, bool outgoing) {
    const int SB = hx_slice();
    for (int b0 = 0; b0 < h->B; b0 += SB) {
        HxSlice v{b0, h->B - b0 < SB ? h->B - b0 : SB, st, hx_num_cu(), h->hx_launches, h->prof};
        trimul_slice(h, v, w, outgoing);
        h->hx_launches = v.launches;
    }
}

// (Tried on top of the slices: the whole pair transform net with the batch in two halves on two streams, each on half of the
//  CUs, the second half one or two kernels behind the first, so that a memory-bound TriMul kernel and the matrix-bound
//  transition run side by side.  89.6 - 94.2 batch-steps/s against 95.1: a kernel on half of the CUs takes twice as long,
//  whether it is bound by memory or by the matrix pipe.)

// ---------------------------------------------------------------------------------------------
// Pair bias of all IPA layers in one pass over p (invariant_point_attention.py:181), hx arithmetic:
//   out[o][b][i][j] = sum_c W[o][c] p[b][i][j][c] + bias[o],  o = layer * H + head  (L H = 96 rows, three 32-row blocks).
// The f32 form (pair_kernels.hip k_ipa_bias) staged p through LDS for 192 v_mfma_f32_32x32x2 per wave and tile.  Here a wave
// takes 32 consecutive j of one (b, i): lane n = j holds its own row -- 8 consecutive channels per k-step are exactly the B
// fragment of v_mfma_f32_32x32x16_f16 (k = 16 s + 8 (lane >> 5) + e), so p goes from HBM to the matrix pipe through
// registers only (a 128-B line is used up by four consecutive k-steps of the same wave), the block-floating-point scale of
// a row is an in-lane maximum plus one exchange with the lane holding the row's other channels, and it comes back out as a
// per-column (= per-lane) factor.  W (48 KiB as 24 hx units) sits in LDS.  By-product: max |p| for the attention kernel's split.
// ---------------------------------------------------------------------------------------------
#define IB_UNITS 24          // 3 row blocks x 8 k-steps
__global__ __launch_bounds__(256) void k_ipa_bias_hx(const float* __restrict__ z, const unsigned char* __restrict__ wimg, float inv_sw,
                                                     const float* __restrict__ bias, float* __restrict__ out, int B, int N, int LH,
                                                     int n_tiles, int rev, unsigned* pmax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ibw[];          // [IB_UNITS][hi 1 KiB | lo 1 KiB], then bias[96]
    float* sbias = reinterpret_cast<float*>(ibw + IB_UNITS * 2048);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int u = tid; u < IB_UNITS * 128; u += 256) reinterpret_cast<uint4*>(ibw)[u] = reinterpret_cast<const uint4*>(wimg)[u];
    if (tid < 96) sbias[tid] = tid < LH ? bias[tid] : 0.f;
    __syncthreads();
    const int n = lane & 31, hf = lane >> 5;
    const int jt = (N + 31) >> 5;                        // 32-wide j tiles per (b, i)
    float amax = 0.f;
    for (int t = blockIdx.x * 4 + wave; t < n_tiles; t += gridDim.x * 4) {
        const int tt = rev ? n_tiles - 1 - t : t;
        const int j0 = (tt % jt) * 32;
        const int bi = tt / jt;                          // b * N + i
        const int j = min(j0 + n, N - 1);
        const float* row = z + ((size_t)bi * N + j) * 128 + 8 * hf;
        float4 x[8][2];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            x[s][0] = *reinterpret_cast<const float4*>(row + 16 * s);
            x[s][1] = *reinterpret_cast<const float4*>(row + 16 * s + 4);
        }
        float m = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            m = fmaxf(m, fmaxf(fmaxf(fabsf(x[s][0].x), fabsf(x[s][0].y)), fmaxf(fabsf(x[s][0].z), fabsf(x[s][0].w))));
            m = fmaxf(m, fmaxf(fmaxf(fabsf(x[s][1].x), fabsf(x[s][1].y)), fmaxf(fabsf(x[s][1].z), fabsf(x[s][1].w))));
        }
        m = fmaxf(m, __shfl_xor(m, 32));                 // the row's other 64 channels
        amax = fmaxf(amax, m);
        const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 255u);
        const bool tiny = e < 16;
        const float sc = tiny ? 1.0f : __builtin_bit_cast(float, (unsigned)(268 - e) << 23);      // max * sc in [2^14, 2^15)
        const float isc = (tiny ? 1.0f : __builtin_bit_cast(float, (unsigned)(e - 14) << 23)) * inv_sw;
        f32x16 acc[3];
#pragma unroll
        for (int blk = 0; blk < 3; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[blk][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float xs[8] = {x[s][0].x, x[s][0].y, x[s][0].z, x[s][0].w, x[s][1].x, x[s][1].y, x[s][1].z, x[s][1].w};
            h8 bh, bl;
            hx_split8(xs, sc, bh, bl);
#pragma unroll
            for (int blk = 0; blk < 3; ++blk) {
                const h8 ah = hx_frag(ibw, blk * 8 + s, 0, lane), al = hx_frag(ibw, blk * 8 + s, 1, lane);
                MFH3(ah, al, bh, bl, acc[blk]);
            }
        }
        if (j0 + n < N) {
            const int b = bi / N, i = bi - b * N;
#pragma unroll
            for (int blk = 0; blk < 3; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    if (o < LH) out[((((size_t)o * B + b) * N + i) * N) + j0 + n] = fmaf(acc[blk][r], isc, sbias[o]);
                }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0 && __float_as_uint(amax) > *reinterpret_cast<volatile unsigned*>(pmax)) atomicMax(pmax, __float_as_uint(amax));
}

bool launch_ipa_bias_hx(genie_ctx* h, hipStream_t st) {
    const int LH = h->d.n_structure_layer * h->d.n_head_ipa;
    if (!h->hx || LH > 96 || h->d.c_p != 128 || getenv("GENIE_IPA_BIAS_F32")) return false;
    const HxGemmW* w = nullptr;
    for (int i = 0; i < h->n_hxg; ++i)
        if (h->hxg[i].w == h->ipa_bias_w) w = &h->hxg[i];
    if (!w) return false;
    const int N = h->N, n_tiles = h->B * N * ((N + 31) / 32);
    const int grid = n_tiles / 4 < 2 * hx_num_cu() ? (n_tiles + 3) / 4 : 2 * hx_num_cu();      // 191 registers: two work-groups per CU
    (void)hipMemsetAsync(h->pmax, 0, sizeof(unsigned), st);
    hipLaunchKernelGGL(k_ipa_bias_hx, dim3(grid), dim3(256), IB_UNITS * 2048 + 96 * 4, st, h->p, w->img, w->inv_s, h->ipa_bias_b, h->ipa_bias,
                       h->B, N, LH, n_tiles, (int)(h->hx_launches & 1), h->pmax);
    return true;
}

void pair_hx_kernels_init() {
#define HX_ATTR(k) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, HX_LDS_BYTES)
    HX_ATTR((k_trimul_proj_hx<true, 8>)); HX_ATTR((k_trimul_proj_hx<false, 8>));
    HX_ATTR(k_trimul_out_hx<8>); HX_ATTR(k_trimul_out_hx_r<8>);
#undef HX_ATTR
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_transition_hx<8>), hipFuncAttributeMaxDynamicSharedMemorySize, HX_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ipa_bias_hx), hipFuncAttributeMaxDynamicSharedMemorySize, IB_UNITS * 2048 + 96 * 4);
}
