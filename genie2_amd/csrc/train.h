// Declarations shared by the training path (train_kernels.hip: device kernels; genie_train.hip: the forward / backward sequence).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// C[z][m][n] (op)= alpha sum_k A[z][m][k] B[z][k][n] (+ bias[n]);  z = z1 * nb2 + z2, element strides throughout.
// mode 0: store, 1: add (read-modify-write), 2: atomic add (required when nsplit > 1 or batches share C).
struct GemmP {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    long long am, ak, bk, bn, cm, cn;
    int batch, nb2;
    long long a1, a2, b1, b2, c1, c2;
    int nsplit; float alpha; int mode;
    int relu;         // mode 0: store max(v, 0)                      (a Linear + ReLU in one pass)
    const float* gate;  // mode 0: store v where gate > 0, else 0; gate is laid out like C   (ReLU backward on the dX GEMM)
    float* asum;      // optional (batch 1, A dense [K][M] with am == 1): asum[m] += sum_k A[m][k] -- a Linear's bias gradient rides on its
                      // weight-gradient GEMM, which streams dY anyway
    const float* colscale;   // optional: the product's column n is scaled by colscale[n] before bias / accumulation -- a weight gradient
                             // taken against LayerNorm's xhat instead of xhat * gamma + beta (the beta part is launch_rank1_add's)
    // optional: C as a row of separately placed blocks -- `cblk` output rows (cblk_m = 1) or columns (0) each, block t at C + ctab[t]
    // (element (m, n) of the product at local index m - t cblk resp. n - t cblk inside it), its row sums at asum + atab[t].  Several
    // Linears that share their input run as ONE GEMM this way although their results (forward: the projections; backward: the weight
    // gradients inside the gradient blob) do not sit next to each other.  bias / colscale stay indexed by the product's column.
    int cblk, cblk_m;
    long long ctab[8], atab[8];
    int xcd;          // set by launch_gemm: split-K work-groups renumbered so that the tiles of one K range share an XCD (its L2)
    // 128-tile kernel only (gemm_takes_mask), mode 0, batch 1, cm == N, cn == 1: one bit per element of C, word (m, n / 32) at [m N / 32 + n / 32]
    //   mask_out: bit = (stored value > 0)            (a Linear + ReLU leaves its sign pattern: 1/32 of the bytes of its output)
    //   mask_in:  store v where the bit is set, else 0  (ReLU backward on the dX GEMM without re-reading the forward output as `gate`)
    unsigned* mask_out; const unsigned* mask_in;
};
bool gemm_takes_mask(const GemmP& p);      // launch_gemm would run this product on the 128-tile kernel
void launch_gemm(hipStream_t st, const GemmP& p, int terms);
int gemm_splits(long long M, long long N, long long K, long long batch);

void launch_ln_fwd(hipStream_t st, const float* x, const float* g, const float* b, float* y, float* xhat, float* rstd, long long R, int C);
void launch_ln_bwd(hipStream_t st, const float* dy, const float* xhat, const float* rstd, const float* g, float* dx, long long R, int C, int accumulate,
                   float* dgamma, float* dbeta);
void launch_colsum(hipStream_t st, const float* a, const float* w, long long R, int C, float* out_b, float* out_g, long long lda = 0);      // lda: row stride of a (0: C)
void launch_transpose(hipStream_t st, const float* in, float* out, int B, int R, int C, bool to_cm);
// Fused layout changes around the triangle multiplication's contraction (row-major [B][N N][C] <-> channel-major [B][C][N N]); C = 128 k:
//   gate_to_cm:       a = ap m sigmoid(ag), b = bp m sigmoid(bg)  (row-major in) -> acm, bcm (channel-major out)
//   ln_from_cm:       x (channel-major) -> LayerNorm over channels -> y = xhat g + b, xhat (row-major), rstd
//   ln_bwd_to_cm:     LayerNorm backward of row-major dy -> dx written channel-major; gamma / beta gradients
//   gate_bwd_from_cm: da, db, a, b (channel-major) + ag, bg (row-major) -> d ap, d ag, d bp, d bg (row-major)
bool train_cm_fusable(int C);
void launch_gate_to_cm(hipStream_t st, const float* ap, const float* ag, const float* bp, const float* bg, const float* rmask, float* acm, float* bcm,
                       int B, int N, int C);
void launch_ln_from_cm(hipStream_t st, const float* xcm, const float* g, const float* b, float* y, float* xhat, float* rstd, int B, int R, int C);
void launch_ln_bwd_to_cm(hipStream_t st, const float* dy, const float* xhat, const float* rstd, const float* g, float* dxcm, int B, int R, int C,
                         float* dgamma, float* dbeta);
void launch_gate_bwd_from_cm(hipStream_t st, const float* dacm, const float* dbcm, const float* acm, const float* bcm, const float* ag, const float* bg,
                             const float* rmask, float* dap, float* dag, float* dbp, float* dbg, int B, int N, int C, int ldo);     // ldo: row stride of the four results
// A Linear applied to LayerNorm's output without materialising it:  (xhat gamma + beta) W^T + b  =  xhat (W gamma)^T + (W beta + b):
//   Wf[o][k] = W[o][k] gamma[k],   bf[o] = b[o] + sum_k W[o][k] beta[k]          (b may be NULL)
void launch_fold_ln(hipStream_t st, const float* W, const float* b, const float* gamma, const float* beta, float* Wf, float* bf, int O, int K);
// ... for a table of Linears in one launch (all offsets into the weight blob `wts` / the scratch `dst`; b < 0: no bias)
struct FoldEntry { long long w, b, g, beta, dst, bdst, raw; int O, K; };  // Wf at dst, bf at bdst; raw >= 0: W itself copied to dst-buffer + raw
                                                                         // (the unfolded weights of several Linears stacked for one input-gradient GEMM)
void launch_fold_ln_table(hipStream_t st, const float* wts, float* dst, const FoldEntry* table_dev, int n_entries, int max_O);
// dW[o][c] += beta[c] db[o]   (the beta part of a weight gradient taken against xhat, see GemmP::colscale)
void launch_rank1_add(hipStream_t st, float* dW, const float* beta, const float* db, int O, int C);
// ... for every entry of a fold table in one launch, at the end of the backward pass:  grads[w + o K + k] += wts[beta + k] grads[b + o]
void launch_rank1_table(hipStream_t st, float* grads, const float* wts, const FoldEntry* table_dev, int n_entries, int max_OK);
void launch_pair_features(hipStream_t st, const float* trans, const float* rots, const int8_t* codes, const float* rmask, const uint8_t* fstm,
                          const uint8_t* fsm, const float* mpos, const int32_t* ridx, const int32_t* cidx, float* F, int B, int N, int nbin,
                          float dmin, float dstep, int relk);
void launch_pair_features_bwd(hipStream_t st, const float* dF, int ldf, const float* trans, const float* rmask, float* dtr, int B, int N, int nbin,
                              float dmin, float dstep);
void launch_pair_sum_bwd(hipStream_t st, const float* dp, float* dpi, float* dpj, int B, int N, int C);

struct IpaArgs {
    int B, N, H, C, Pq, Pv, cp;
    const float *q, *kv, *qp, *kp, *vp, *bias, *p, *rots, *trans, *rmask, *head_w, *wb;
    float *att, *cat;                                   // forward outputs (kept)
    const float* dcat;                                  // backward input
    float *dlg, *dq, *dqp, *doptg, *dP, *dhead, *dbb, *dR, *dT, *dkv, *dkp, *dvp;
};
void launch_ipa_fwd(hipStream_t st, const IpaArgs& a);
void launch_ipa_bwd(hipStream_t st, const IpaArgs& a);
void launch_points_fwd(hipStream_t st, const float* lin, const float* rots, const float* trans, float* out0, float* out1, int M, int H, int P0, int P1);
void launch_points_bwd(hipStream_t st, const float* lin, const float* rots, const float* dg0, const float* dg1, float* dlin, float* dR, float* dT,
                       int M, int H, int P0, int P1);
void launch_frames_fwd(hipStream_t st, const float* bb, const float* R, const float* T, float* R2, float* T2, int M);
void launch_frames_bwd(hipStream_t st, const float* bb, const float* R, const float* dR2, const float* dT2, float* dbb, float* dR, float* dT, int M);

// elementwise lambdas
template <class F>
__global__ void k_ew(long long n, F f) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) f(i);
}
template <class F>
static inline void launch_ew(hipStream_t st, long long n, F f) {
    if (n > 0) hipLaunchKernelGGL(k_ew<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, f);
}
// dropout keep-mask value (0 or 1 / (1 - rate)) of element `idx` of dropout site `tag`: a counter-based hash, so the forward and
// the backward pass (and a test that rebuilds the masks in numpy) see the same mask without storing it
__host__ __device__ static inline float drop_scale(uint32_t seed, uint32_t tag, uint64_t idx, float rate) {
    uint32_t x = (uint32_t)idx * 0x9E3779B1u ^ (uint32_t)(idx >> 32) * 0x85EBCA77u ^ seed * 0xC2B2AE3Du ^ tag * 0x27D4EB2Fu;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return u < rate ? 0.f : 1.0f / (1.0f - rate);
}
