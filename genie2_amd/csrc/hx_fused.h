// Shared by the fused pair-chain kernels (pair_fused_kernels.hip; the two-groups-per-CU variant measured in round 3 is kept as text
// under tools/probe/attic/): launch arguments, the bias table's slots, the z staging helpers and the LayerNorm / split of
// a result tile.  Layout conventions: pair_fused_kernels.hip's header comment.
#pragma once
#include "hx_pair.h"

#define FZ_SB_BZ 0
#define FZ_SB_BG 128
#define FZ_SB_B1 256
#define FZ_SB_B2 768
#define FZ_SB_BP 896
#define FZ_SB_FLOATS 1408
#define FZ_OOR 0x7FFFFFF0
#ifndef FZ_SAFE
#define FZ_SAFE 0
#endif
#define FZ_FULL_WAIT(bit) do { if (FZ_SAFE & (bit)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); else asm volatile("" ::: "memory"); } while (0)

struct FusedArgs {
    float* z; const float* xcm; const float* rmask; const unsigned char* wimg;
    const float *bzs, *bgs, *b1s, *b2s, *bproj;
    unsigned *acm, *bcm;
    int N, NP, n_wtiles, n_hb;
    unsigned cm_bytes, z_bytes;
    float sx, cgo, cz, c1, c2, inv_c2, cpa, cpb, cg;
    int rev, stagger;
};

// granules (16 B = 4 channels) g and g + 2 of this lane's row: the chained-k slots e = 0..3 / 4..7 of k-chunk 4 half + q
__device__ __forceinline__ void fz_zt_read(float4 (&raw)[16], const unsigned char* zt, int pl, int h, int half) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int g = 4 * q + h;
        raw[2 * (4 * half + q)] = *reinterpret_cast<const float4*>(zt + pl * 256 + ((g ^ (pl & 15)) << 4));
        raw[2 * (4 * half + q) + 1] = *reinterpret_cast<const float4*>(zt + pl * 256 + (((g + 2) ^ (pl & 15)) << 4));
    }
}

// one k-chunk (q = 0..3 within the half the staging area holds) of this lane's row
__device__ __forceinline__ void fz_zt_chunk(float4& a, float4& b, const unsigned char* zt, int pl, int h, int q) {
    const int g = 4 * q + h;
    a = *reinterpret_cast<const float4*>(zt + pl * 256 + ((g ^ (pl & 15)) << 4));
    b = *reinterpret_cast<const float4*>(zt + pl * 256 + (((g + 2) ^ (pl & 15)) << 4));
}

// One channel half (64 channels = accumulators 2 hf, 2 hf + 1) of a result tile -> the wave's staging area in row order ->
// global memory in whole 256-B pieces (the inverse of hx_zt_dma / fz_zt_read; rows >= nvalid are dropped).
__device__ __forceinline__ void fz_store_half(rsrc_t rz, unsigned char* zt, const f32x16& v0, const f32x16& v1, int lane, int soff,
                                              int row_stride, int nvalid, int hf) {
    const int pl = lane & 31, h = lane >> 5;
    FZ_FULL_WAIT(1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x16& v = (q < 2) ? v0 : v1;
        const int r0 = 8 * (q & 1);
        const int g = 4 * q + h;
        *reinterpret_cast<float4*>(zt + pl * 256 + ((g ^ (pl & 15)) << 4)) = make_float4(v[r0], v[r0 + 1], v[r0 + 2], v[r0 + 3]);
        *reinterpret_cast<float4*>(zt + pl * 256 + (((g + 2) ^ (pl & 15)) << 4)) = make_float4(v[r0 + 4], v[r0 + 5], v[r0 + 6], v[r0 + 7]);
    }
    hx_lds_done();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + (lane >> 4);
        u32x4 d = *reinterpret_cast<const u32x4*>(zt + j * 1024 + lane * 16);
        const int g = (lane & 15) ^ (r & 15);
        const int voff = r < nvalid ? r * row_stride + (g << 4) : FZ_OOR;
        __builtin_amdgcn_raw_buffer_store_b128(d, rz, voff, soff + hf * 256, 0);
        // A store of more than 64 bits reads its data registers for a few cycles after it issues; a VALU write of them in the very
        // next instruction corrupts the stored value (observed on gfx950: hipcc re-used dword 0 of `d` as an address temporary right
        // behind the store and put no wait state between -- some tiles came out with rows of the next piece).  The empty-bodied asm
        // keeps `d` allocated until two wait states behind the store.
        asm volatile("s_nop 1" : "+v"(d) : : "memory");
    }
    FZ_FULL_WAIT(2);
}

// LayerNorm statistics of a result tile (this lane's 64 channels + its partner's): mean and sx / sqrt(var + eps)
__device__ __forceinline__ void fz_stats(const f32x16 (&v)[4], float sx, float& mean, float& sc) {
    float s = 0.f;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[ob][r];
    s += __shfl_xor(s, 32);
    mean = s * (1.0f / 128.0f);
    float ss = 0.f;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = v[ob][r] - mean; ss += d * d; }
    ss += __shfl_xor(ss, 32);
    sc = sx / sqrtf(ss * (1.0f / 128.0f) + GENIE_LN_EPS);
}
__device__ __forceinline__ void fz_split_tile(h8 (&zh)[8], h8 (&zl)[8], const f32x16 (&v)[4], float mean, float sc) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = v[c >> 1][8 * (c & 1) + e] - mean;
        hx_split8(x, sc, zh[c], zl[c]);
    }
}

