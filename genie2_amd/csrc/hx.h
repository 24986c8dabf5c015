// "hx" arithmetic: f32 GEMMs on the f16 matrix pipe, operands split in two halves.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the f16/bf16 MFMA rate on gfx950 and shares the FP32
// lanes with the VALU (tools/probe/valu_probe.hip).  Every f32 operand x of a GEMM is therefore
// carried as an unevaluated sum of two f16 numbers
//     x S = hi + lo,   hi = f16(x S),  lo = f16(x S - hi)            (S = a power of two)
// which keeps 22 significand bits (|x S - hi - lo| <= 2^-22 |x S|), and a product is taken as
//     a b ~= (a_hi b_hi + a_hi b_lo + a_lo b_hi) / (S_a S_b)         (a_lo b_lo ~ 2^-22 dropped)
// = three v_mfma_f32_32x32x16_f16 into ONE f32 accumulator (products of f16 numbers are exact in
// f32; accumulation is f32 as before).  The per-product error (~2^-21) is of the size of the f32
// accumulation error of the K = 128..512 sums it replaces; measured against an f64 GEMM the
// split product is as accurate as the f32 one.  The scales S keep every
// half inside f16's range: LayerNorm outputs are bounded by sqrt(C), weights are known at load
// time, and every other operand (hidden activations, gated projections) is bounded through the
// weights that produce it (Cauchy-Schwarz on the folded rows; genie_api.hip hx_bound()).  Halves
// that fall below f16's normal range lose absolute, not relative, precision: < bound * 2^-40.
//
// Fragment convention of v_mfma_f32_32x32x16_f16 (8 halves = one b128 per lane and operand):
//   A[i = lane&31][k = 8*(lane>>5) + e],  B[k = 8*(lane>>5) + e][j = lane&31],  e = 0..7,
//   D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31],  r = 0..15   (as for the f32 form).
// A weight "unit" is the pair of 1-KiB fragments (hi, then lo) of 32 rows x 16 k-values; the host
// emits units in exactly the order a kernel consumes them (genie_api.hip), one 32-KiB stage =
// 16 units, pulled into LDS by LDS-DMA as before.
// When an accumulator tile is fed back as the A operand of the next GEMM (D^T chaining), registers
// 8c..8c+7 of the lane are the 8 halves of k-chunk c, i.e. MFMA slot (h = lane>>5, e) carries
// k = 16c + (e&3) + 8(e>>2) + 4h; the host packs the consuming weight with the same permutation.
#pragma once
#include "common.h"

#ifdef __HIPCC__
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFH(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
// (a_hi + a_lo)(b_hi + b_lo) without the lo*lo term, small terms first
#define MFH3(ah, al, bh, bl, c) do { MFH(al, bh, c); MFH(ah, bl, c); MFH(ah, bh, c); } while (0)

// hi / lo halves of two f32 values (already scaled), packed: RTN both times
__device__ __forceinline__ void hx_split2(float a, float b, unsigned& hi, unsigned& lo) {
    h2 h;
    h.x = (_Float16)a; h.y = (_Float16)b;
    const float ra = a - (float)h.x, rb = b - (float)h.y;
    h2 l;
    l.x = (_Float16)ra; l.y = (_Float16)rb;
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// 8 consecutive k-values -> one (hi, lo) fragment pair
__device__ __forceinline__ void hx_split8(const float (&x)[8], h8& hi, h8& lo) {
    u32x4 h, l;
    unsigned a, b;
    hx_split2(x[0], x[1], a, b); h.x = a; l.x = b;
    hx_split2(x[2], x[3], a, b); h.y = a; l.y = b;
    hx_split2(x[4], x[5], a, b); h.z = a; l.z = b;
    hx_split2(x[6], x[7], a, b); h.w = a; l.w = b;
    hi = __builtin_bit_cast(h8, h);
    lo = __builtin_bit_cast(h8, l);
}
// one value -> (hi | lo << 16), the storage form of split activations in HBM
__device__ __forceinline__ unsigned hx_pack1(float v) {
    h2 p;
    p.x = (_Float16)v;
    p.y = (_Float16)(v - (float)p.x);
    return __builtin_bit_cast(unsigned, p);
}
// LDS fragment of unit u (2 KiB: hi then lo) of a stage
__device__ __forceinline__ h8 hx_frag(const unsigned char* stage, int u, int part, int lane) {
    return *reinterpret_cast<const h8*>(stage + u * 2048 + part * 1024 + lane * 16);
}
#endif

#define HX_STAGE_BYTES 32768
#define HX_UNITS 16            // units per stage
