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

// The splits are written with v_fma_mix* directly: hi = f16(x s) and lo = f16(x s - hi) are both taken
// from the EXACT product x s (one rounding each), and -- the reason for the asm -- from the SAME hi.
// Left to hipcc, `(_Float16)(x*s)` is sometimes contracted into v_fma_mixlo (exact product) for one
// use and v_cvt_pk_f16_f32 of the rounded f32 product for another; when the two disagree (double
// rounding) the pair is off by a whole f16 ulp of hi, i.e. 2^-11 instead of 2^-22.
// (hi, lo) of a*s and b*s, packed with a in the low half
__device__ __forceinline__ void hx_split2(float a, float b, float s, unsigned& hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(a), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(b), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(a), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(b), "v"(s), "v"(hi));
}
// 8 consecutive k-values, scaled by s -> one (hi, lo) fragment pair
__device__ __forceinline__ void hx_split8(const float (&x)[8], float s, h8& hi, h8& lo) {
    u32x4 h, l;
    unsigned a, b;
    hx_split2(x[0], x[1], s, a, b); h.x = a; l.x = b;
    hx_split2(x[2], x[3], s, a, b); h.y = a; l.y = b;
    hx_split2(x[4], x[5], s, a, b); h.z = a; l.z = b;
    hx_split2(x[6], x[7], s, a, b); h.w = a; l.w = b;
    hi = __builtin_bit_cast(h8, h);
    lo = __builtin_bit_cast(h8, l);
}
// u*t -> (hi | lo << 16), the storage form of split activations in HBM
__device__ __forceinline__ unsigned hx_pack_prod(float u, float t) {
    unsigned w;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(w) : "v"(u), "v"(t));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel_hi:[0,0,1]" : "+v"(w) : "v"(u), "v"(t));
    return w;
}
// LDS fragment of unit u (2 KiB: hi then lo) of a stage
__device__ __forceinline__ h8 hx_frag(const unsigned char* stage, int u, int part, int lane) {
    return *reinterpret_cast<const h8*>(stage + u * 2048 + part * 1024 + lane * 16);
}
#endif

#define HX_STAGE_BYTES 32768
#define HX_UNITS 16            // units per stage
