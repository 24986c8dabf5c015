// Shared declarations for libgenie_hip.so (gfx950 / CDNA4 only).
//
// GEMM convention used by every MFMA kernel here
// ------------------------------------------------
// v_mfma_f32_32x32x2_f32 (exact fp32, one fmaf chain per output):
//   A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31],
//   D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31], r = 0..15.
// Operands are consumed 8 k-values at a time as one float4 per lane:
//   frag.e = X[idx = lane&31][k = 8*kb + 4*(lane>>5) + e],  e = 0..3
// and MFMA e pairs k = 8kb+e (lanes < 32) with k = 8kb+4+e (lanes >= 32).
// Weights are repacked once on the host into that order ("fragment-major"):
//   Wp[((nb*KB + kb)*64 + lane)*4 + e] = W[32nb + (lane&31)][8kb + 4(lane>>5) + e]
// so a whole fragment is ONE coalesced 1-KiB transfer: k_gemm_rows streams them into a register
// ring, the pair-stack kernels pull 32-KiB stages into LDS with LDS-DMA (pair_wl_kernels.hip).
// LDS tiles of activations are K-contiguous with a +4 float row pad, which makes the float4
// fragment read (ds_read_b128, 16-lane groups, bank = (addr/4) % 64) conflict free.
// The default pair-stack arithmetic is NOT this instruction but three f16 MFMAs on split
// operands: hx.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/genie_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GENIE_LN_EPS 1e-5f

// ---------------------------------------------------------------- device helpers
#ifdef __HIPCC__

__device__ __forceinline__ f32x16 mfma_8k(const float4 a, const float4 b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// row index inside a 32x32 accumulator tile held in register r by this lane
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// packed-weight fragment (see header comment)
__device__ __forceinline__ float4 wfrag(const float* __restrict__ wp, int KB, int nb, int kb, int lane) {
    return *reinterpret_cast<const float4*>(wp + ((size_t)(nb * KB + kb) * 64 + lane) * 4);
}

// ---- software-pipelined weight fragments -------------------------------------------------
// hipcc places every load right before its first use (one L2 round trip exposed per k-block) and
// undoes source-level register rotation, so the prefetch is written with loads it cannot see:
//   wf_issue()  : global_load_dwordx4 the compiler does not count (asm), destination "=v";
//   wf_wait<N>(): s_waitcnt vmcnt(N) tied ("+v") to the registers it releases + scheduling fence,
//                 so no consumer can be placed above it.
// Counting rule (vector-memory ops retire in order): N = number of THIS wave's asm loads issued
// after the ones being released.  Compiler-issued loads/stores in between can only make the wait
// stricter, never weaker.  Every asm load must be retired (wf_wait<0>) before its destination
// registers die, otherwise a late write-back would land in a reused register.
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ const float* wfrag_ptr(const float* __restrict__ wp, int KB, int nb, int kb, int lane) {
    return wp + ((size_t)(nb * KB + kb) * 64 + lane) * 4;
}
__device__ __forceinline__ void wf_issue(v4f& d, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void wf_wait(v4f& a) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "i"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void wf_wait(v4f& a, v4f& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ f32x16 mfma_8k(const v4f a, const float4 b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x16 mfma_8k(const float4 a, const v4f b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// K-contiguous LDS tile fragment: rows r0..r0+31, k-block kb
__device__ __forceinline__ float4 lfrag(const float* lds, int ld, int r0, int kb, int lane) {
    return *reinterpret_cast<const float4*>(lds + (r0 + (lane & 31)) * ld + kb * 8 + 4 * (lane >> 5));
}

// M-contiguous LDS tile ([k][ld], idx contiguous): same fragment, 4 scalar reads
__device__ __forceinline__ float4 lfrag_t(const float* lds, int ld, int i0, int kb, int lane) {
    const float* p = lds + (kb * 8 + 4 * (lane >> 5)) * ld + i0 + (lane & 31);
    return make_float4(p[0], p[ld], p[2 * ld], p[3 * ld]);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// In-place LayerNorm of a [rows][ld] LDS tile with `C` channels (C = 128):
// 4 threads per row (256 threads <-> 64 rows).  Two-pass mean / variance.
__device__ __forceinline__ void ln_rows_128(float* tile, int ld, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, int tid) {
    const int row = tid >> 2, part = tid & 3;
    float* p = tile + row * ld + part * 32;
    float v[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float4 t = *reinterpret_cast<float4*>(p + 4 * q);
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += v[q];
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    const float mean = s * (1.0f / 128.0f);
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) { float d = v[q] - mean; ss += d * d; }
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 128.0f) + GENIE_LN_EPS);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = part * 32 + 4 * q;
        float4 g = *reinterpret_cast<const float4*>(gamma + c);
        float4 b = *reinterpret_cast<const float4*>(beta + c);
        float4 o;
        o.x = (v[4 * q] - mean) * rstd * g.x + b.x;
        o.y = (v[4 * q + 1] - mean) * rstd * g.y + b.y;
        o.z = (v[4 * q + 2] - mean) * rstd * g.z + b.z;
        o.w = (v[4 * q + 3] - mean) * rstd * g.w + b.w;
        *reinterpret_cast<float4*>(p + 4 * q) = o;
    }
}


// Quaternion of a proper rotation r (row-major 3x3): normalised column of
// K + I with the largest diagonal (closed form of the eigenvector that
// affine_utils.py:336-355 obtains with eigh), sign pinned by `code`.
__device__ __forceinline__ void rot_to_quat_dev(const float* r, int code, float* q) {
    const float xx = r[0], xy = r[1], xz = r[2], yx = r[3], yy = r[4], yz = r[5], zx = r[6], zy = r[7], zz = r[8];
    const float d0 = 1.f + xx + yy + zz, d1 = 1.f + xx - yy - zz, d2 = 1.f + yy - xx - zz, d3 = 1.f + zz - xx - yy;
    float c0, c1, c2, c3;
    if (d0 >= d1 && d0 >= d2 && d0 >= d3)      { c0 = d0;      c1 = zy - yz; c2 = xz - zx; c3 = yx - xy; }
    else if (d1 >= d2 && d1 >= d3)             { c0 = zy - yz; c1 = d1;      c2 = xy + yx; c3 = xz + zx; }
    else if (d2 >= d3)                         { c0 = xz - zx; c1 = xy + yx; c2 = d2;      c3 = yz + zy; }
    else                                       { c0 = yx - xy; c1 = xz + zx; c2 = yz + zy; c3 = d3; }
    const float inv = 1.0f / sqrtf(c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3);
    q[0] = c0 * inv; q[1] = c1 * inv; q[2] = c2 * inv; q[3] = c3 * inv;
    if (code > 0) {
        const int m = (code - 1) >> 1;
        const bool want_neg = ((code - 1) & 1) != 0;
        const float comp = q[m];
        if ((comp < 0.f) != want_neg) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    }
}

#endif  // __HIPCC__

// ---------------------------------------------------------------- host side

enum KernelClass {
    KC_SINGLE_INPUT = 0, KC_GEMM_ROWS, KC_LAYERNORM, KC_PAIR_STATIC, KC_PAIR_INIT,
    KC_TRIMUL_PROJ, KC_TRIMUL_CONTRACT, KC_TRIMUL_OUT, KC_PAIR_TRANSITION,
    KC_IPA_BIAS, KC_IPA_PREP, KC_IPA_ATTN, KC_BB_UPDATE, KC_STRUCT_ROWS, KC_P_SAMPLE, KC_MISC, KC_PAIR_FUSED_A, KC_PAIR_FUSED_B,
    KC_TR_GEMM, KC_TR_EW, KC_TR_LN, KC_TR_TRANSPOSE, KC_TR_IPA, KC_TR_MISC, KC_COUNT      // training path (genie_train.hip)
};

// "hx" images (hx.h): weights split in f16 halves, in stage order, + the scales that go with them
struct HxTransW { const unsigned char* img; float *b1s, *b2s; float sx, c1, c2; };

struct HxTriW {
    const unsigned char *img_proj, *img_out;   // 8 stages (passes) / 4 stages (W_g{0,1}, W_z{0,1}, W_g{2,3}, W_z{2,3})
    float *bias_proj, *bgs, *bzs;              // scaled biases (initial accumulators)
    float sx, cpa, cpb, cg, cx, cgo, cz;       // operand scale / epilogue rescales
};
// weight stream of one fused row-local chain (pair_fused_kernels.hip): stages O (4) | T (n_hb) | P (8), chained k order
struct HxFusedW { const unsigned char* img; };
struct TriMulW {
    float *proj_w;            // packed [512][128]: a_p | b_p | a_g | b_g
    float *proj_b;            // [512] same order
    float *g_w, *g_b;         // packed [128][128], [128]
    float *z_w, *z_b;         // packed [c_p][c_hidden]
    float *ln_in_g, *ln_in_b, *ln_out_g, *ln_out_b;
    HxTriW hx;
};
struct PairLayerW {
    TriMulW out, in;
    float *pt_ln_g, *pt_ln_b, *pt_w1, *pt_b1, *pt_w2, *pt_b2;   // w1 packed [512][128], w2 packed [128][512]
    HxTransW hx_pt;
    HxFusedW fa, fb;          // chain A: out-output -> in-projections;  chain B: in-output -> transition -> next block's out-projections
};
struct StructLayerW {
    float *proj_w, *proj_b;       // packed [1152][384]: q | kv | q_pts | kv_pts
    float *head_w;                // [H] raw head_weights
    float *out_w, *out_b;         // packed [384][2112]
    float *ln_ipa_g, *ln_ipa_b;
    float *t1_w, *t1_b, *t2_w, *t2_b, *t3_w, *t3_b;
    float *ln_tr_g, *ln_tr_b;
    float *bb_w, *bb_b;           // raw [6][384], [6]
};

struct HxGemmW { const float* w; const unsigned char* img; float inv_s; };   // hx image of a k_gemm_rows weight (keyed by its f32 pack)

struct genie_train_ws;
struct genie_ctx {
    genie_dims_t d;
    genie_train_ws* train;        // activations / scratch of the training path (genie_train.hip)
    int device;
    char err[512];

    // weights
    bool have_weights;
    float* wdev;                  // one device allocation holding everything below
    bool hx;                      // pair-stack GEMMs in split-f16 arithmetic (hx.h); GENIE_MATH=f32 selects the f32-MFMA kernels
    unsigned char* hxdev;         // device allocation of the hx weight images
    HxGemmW hxg[64]; int n_hxg;   // hx images of the row-GEMM weights
    unsigned hx_launches;         // pair-stack launches so far: its parity picks the tile direction of the next one
    size_t wdev_floats;
    float *single_w;              // packed [384][856]
    float *pij_w;                 // packed [256][384]: linear_s_p_i | linear_s_p_j
    float *relpos_t;              // [67][128] transposed raw
    float *templ_w;               // packed [128][48]
    float *motif_w;               // packed [128][40]
    float *ipa_bias_w, *ipa_bias_b;   // packed [ceil32(L*H)][128], [L*H]
    PairLayerW* pair;             // host arrays of device pointers
    StructLayerW* st;

    // tables
    bool have_tables;
    float *pos_tab, *chain_tab, *t_tab, *sched;
    int n_pos, n_chain;
    float* sched_host;            // [4][T+1]

    // bound batch
    int B, N, NP;
    bool have_feats, has_motif;
    int32_t *f_aatype, *f_rmask, *f_ridx, *f_cidx;
    float *f_pos;
    uint8_t *f_fsm, *f_fstm, *f_ifm;
    float *rmaskf;                // [B,N] residue mask as float

    // workspace (one allocation, carved)
    void* ws; size_t ws_bytes;
    float *p, *acm, *bcm, *xcm, *pstatic, *ipa_bias;
    float *spart;                 // [3][B N][c_s] split-K slices of the IPA output projection (SR_KSPLIT)
    unsigned *pmax;               // bits of max |p| over the pair tensor the IPA layers read (k_ipa_bias -> k_ipa_attn_q)
    float *xsingle, *s0, *s, *s1, *s2, *h1, *h2, *pij, *proj, *cat;
    float *kT, *v, *qp, *kpT, *vp;
    float *vf, *vmax;      // hx attention: V / v_pts rows as f32 MFMA B fragments, and each row's largest |v|, |v_pt| (single_kernels.hip k_ipa_prep)
    float *rots_w, *trans_w;      // working frames
    int32_t* tsteps;              // [B] uniform timestep buffer for the loop
    float *loop_z;                // [B,N,3]

    // profiling
    bool prof;
    struct ProfRec { int cls; hipEvent_t a, b; };
    ProfRec* prof_recs; int prof_n, prof_cap;
    double prof_ms[KC_COUNT]; int64_t prof_cnt[KC_COUNT];
    double train_gemm_flop;       // algorithmic FLOP (2 M N K per product) of the GEMMs the last training call launched

    // the structure net's second stream (genie_api.hip: the two halves of a batch run their layers side by side)
    hipStream_t st2; hipEvent_t ev_fork, ev_join;
};

// launchers (each enqueues on `st`, no sync)
void launch_single_input(genie_ctx* h, hipStream_t st, const int32_t* timesteps);
void launch_gemm_rows(genie_ctx* h, hipStream_t st, const float* A, int lda, int M, int K,
                      const float* Wp, int Nout, const float* bias, const float* res, int ldr,
                      const float* rowmask, int relu, float* out, int ldo);
void launch_layernorm_rows(genie_ctx* h, hipStream_t st, const float* in, float* out, int M, int C,
                           const float* g, const float* b);
void launch_pair_static(genie_ctx* h, hipStream_t st);
void launch_pair_init(genie_ctx* h, hipStream_t st, const float* trans, const float* rots,
                      const int8_t* codes);
void launch_trimul(genie_ctx* h, hipStream_t st, const TriMulW& w, bool outgoing);
void launch_pair_transition(genie_ctx* h, hipStream_t st, const PairLayerW& w);
bool launch_pair_stack_fused(genie_ctx* h, hipStream_t st, float* tap_trimul_out0, float* tap_layer0);   // false: not applicable, use the launches above
void launch_ipa_bias(genie_ctx* h, hipStream_t st);
void launch_ipa_prep(genie_ctx* h, hipStream_t st, int b0 = 0, int nb = -1);       // batch entries b0 .. b0 + nb - 1 (nb < 0: all)
void launch_ipa_attn(genie_ctx* h, hipStream_t st, int layer, const float* head_w, int b0 = 0, int nb = -1);
size_t ipa_vf_floats(const genie_dims_t& d, int B, int N);      // size of genie_ctx::vf
bool ipa_attn_splits(const genie_ctx* h);         // the attention kernel in use takes a batch range
void launch_q_sample(genie_ctx* h, hipStream_t st, const float* x0, const float* z, const float* c0, const float* c1, float* trans_out);
void launch_training_loss(genie_ctx* h, hipStream_t st, const float* zp, const float* z, float w, float* losses, float* grad);
void launch_adam(hipStream_t st, size_t n, float* p, const float* g, float* m, float* v, double lr, double b1, double b2, double eps, int step);
void launch_any_nonzero(genie_ctx* h, hipStream_t st, const uint8_t* x, size_t n, unsigned* flag);
bool launch_struct_tail(genie_ctx* h, hipStream_t st, const StructLayerW& S, const float* trans_in, float* z_out, int b0 = 0, int nb = -1);
bool struct_tail_fused(const genie_ctx* h, const StructLayerW& S);
void launch_bb_update(genie_ctx* h, hipStream_t st, const StructLayerW& w, const float* trans_in,
                      float* z_out);
void launch_frenet(genie_ctx* h, hipStream_t st, int mode, int step, float scale, float* trans,
                   float* rots, const float* z, const float* eps);
void launch_fill_i32(genie_ctx* h, hipStream_t st, int32_t* p, int n, int v);
void train_ws_free(genie_ctx* h);

// profiling hooks used by the launchers
void prof_begin(genie_ctx* h, hipStream_t st, int cls);
void prof_end(genie_ctx* h, hipStream_t st);

struct ProfScope {
    genie_ctx* h; hipStream_t st; bool on;
    ProfScope(genie_ctx* h_, hipStream_t st_, int cls, bool enable = true) : h(h_), st(st_), on(enable && h_->prof) { if (on) prof_begin(h, st, cls); }
    ~ProfScope() { if (on) prof_end(h, st); }
};
