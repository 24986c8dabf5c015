// Training path (genie/diffusion/genie.py:60-120): device kernels of the forward pass that keeps what the
// backward pass needs, and of the hand-derived backward pass through the whole Denoiser.  Unlike the sampling path (fused,
// fragment-packed weights) this path reads the plain state_dict blob and is built from a few general pieces:
//   k_gemm         batched, arbitrarily strided C (+)= alpha A B (+ bias) on the bf16 matrix pipe; operands are split on the fly
//                  into one, two or three bf16 pieces (8 / 16 / 24 significand bits, f32's exponent range: gradients of any
//                  magnitude survive); three pieces = six MFMAs per product reproduce f32 products, one piece is the reference's
//                  bf16 autocast;
//                  split-K with float atomics for the tall reductions that weight gradients are
//   k_ln_*         LayerNorm forward / backward over rows
//   k_ew           elementwise lambdas (gates, ReLU, dropout, residuals)
//   k_ipa_*        invariant point attention forward (keeping the attention weights) and its two backward kernels
//   k_frames_*     BackboneUpdate / frame composition, forward and backward
// Reference lines are cited per kernel; the derivatives are checked tensor by tensor against the reference's own autograd
// (tests/golden/train_grads_n16_b2.npz) and against torch autograd over the oracle.
#include <stdint.h>
#include <stdlib.h>
#include "common.h"
#include "train.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------ GEMM
// 64 x 64 output tile per 256-thread work-group (four waves, one 32 x 32 accumulator each), K in steps of 64.  An operand tile is
// 64 rows x 64 k of f32 = four float4 per thread, fetched one step ahead into registers, split into its bf16 pieces on the way into
// LDS ([piece][row][k], row stride 144 B: the fragment reads (16 B per lane, 32 rows) and the 8-byte stores are conflict-free).
// Either dimension of an operand may be the contiguous one: KC = along k (lanes walk k; a float4 is four k of one row), otherwise
// along the row index (a float4 is four rows of one k; a thread holds a 4 x 4 block and stores its transpose).  `vec`: the operand's
// base, its other stride and its batch strides are multiples of four floats, so whole float4 loads are legal; weights inside the
// caller's flat state_dict blob often are not (a 6-float bias shifts everything behind it) and are fetched dword by dword.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#ifndef GM_BK
#define GM_BK 32         // K per step: 32 -> 30 KiB of LDS with three pieces, four work-groups per CU (their phases interleave)
#endif
#define GM_LD (GM_BK + 8)   // row stride 80 B (144 B for 64): fragment reads and packed stores are conflict-free
#define GM_NV (GM_BK / 16)  // float4 per thread and operand tile
#define GM_WPS (GM_BK == 32 ? 4 : 2)
#ifndef GM_EXP
#define GM_EXP 0      // developer knock-outs (tools/probe/gemm_variants.sh): 1 no MFMA, 2 no global loads, 4 no result store, 8 hi piece only
#endif

// `full` (rows_left >= 64 and k_left >= GM_BK) is uniform over the work-group: full tiles -- nearly all of them -- load without a
// per-lane test, so that a step's float4 go out back to back; edge tiles test every element.
// KC: float4 f = tid + 256 q is k quad f % (GM_BK / 4) of row f / (GM_BK / 4).  Otherwise thread (kg, mq) holds rows 4 mq .. + 3 at
// k = GM_NV kg .. + GM_NV - 1 (one float4 per k).
template <bool KC>
__device__ __forceinline__ void gm_load(const float* __restrict__ base, long long sr, long long sk, int rows_left, int k_left, bool vec, int tid,
                                        float4 (&v)[GM_NV]) {
    const bool full = rows_left >= 64 && k_left >= GM_BK;
    constexpr int QK = GM_BK / 4;                           // k quads per row
    if (KC) {
        if (full && vec) {
#pragma unroll
            for (int q = 0; q < GM_NV; ++q) {
                const int f = tid + 256 * q;
                v[q] = *reinterpret_cast<const float4*>(base + (long long)(f / QK) * sr + 4 * (f % QK));
            }
        } else if (full) {
#pragma unroll
            for (int q = 0; q < GM_NV; ++q) {
                const int f = tid + 256 * q;
                const float* s = base + (long long)(f / QK) * sr + (long long)(4 * (f % QK)) * sk;
                v[q].x = s[0]; v[q].y = s[sk]; v[q].z = s[2 * sk]; v[q].w = s[3 * sk];
            }
        } else {
#pragma unroll
            for (int q = 0; q < GM_NV; ++q) {
                const int f = tid + 256 * q, kq = f % QK, rr = f / QK;
                const float* s = base + (long long)rr * sr + (long long)(4 * kq) * sk;
                const bool ro = rr < rows_left;
                v[q].x = (ro && 4 * kq + 0 < k_left) ? s[0] : 0.f;
                v[q].y = (ro && 4 * kq + 1 < k_left) ? s[sk] : 0.f;
                v[q].z = (ro && 4 * kq + 2 < k_left) ? s[2 * sk] : 0.f;
                v[q].w = (ro && 4 * kq + 3 < k_left) ? s[3 * sk] : 0.f;
            }
        }
    } else {
        const int kg = (tid & 7) + 8 * ((tid >> 6) & 1), mq = ((tid >> 3) & 7) + 8 * (tid >> 7);
        if (full && vec) {
#pragma unroll
            for (int i = 0; i < GM_NV; ++i) v[i] = *reinterpret_cast<const float4*>(base + (long long)(GM_NV * kg + i) * sk + 4 * mq);
        } else if (full) {
#pragma unroll
            for (int i = 0; i < GM_NV; ++i) {
                const float* s = base + (long long)(GM_NV * kg + i) * sk + (long long)(4 * mq) * sr;
                v[i].x = s[0]; v[i].y = s[sr]; v[i].z = s[2 * sr]; v[i].w = s[3 * sr];
            }
        } else {
#pragma unroll
            for (int i = 0; i < GM_NV; ++i) {
                const int kk = GM_NV * kg + i;
                const float* s = base + (long long)kk * sk + (long long)(4 * mq) * sr;
                const bool ko = kk < k_left;
                v[i].x = (ko && 4 * mq + 0 < rows_left) ? s[0] : 0.f;
                v[i].y = (ko && 4 * mq + 1 < rows_left) ? s[sr] : 0.f;
                v[i].z = (ko && 4 * mq + 2 < rows_left) ? s[2 * sr] : 0.f;
                v[i].w = (ko && 4 * mq + 3 < rows_left) ? s[3 * sr] : 0.f;
            }
        }
    }
}
// NP consecutive k of one row: split into the bf16 pieces, one packed LDS store per piece
template <int TERMS, int NP, int ROWS = 64>
__device__ __forceinline__ void gm_put(__bf16 (*S)[ROWS][GM_LD], int row, int k, const float (&x)[NP]) {
    typedef __bf16 vec_t __attribute__((ext_vector_type(NP)));
    vec_t h, m, l;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        h[j] = (__bf16)x[j];
        if (GM_EXP & 8) { m[j] = h[j]; l[j] = h[j]; continue; }
        if (TERMS > 1) {
            const float r1 = x[j] - (float)h[j];
            m[j] = (__bf16)r1;
            if (TERMS > 2) l[j] = (__bf16)(r1 - (float)m[j]);
        }
    }
    *reinterpret_cast<vec_t*>(&S[0][row][k]) = h;
    if (TERMS > 1) *reinterpret_cast<vec_t*>(&S[1][row][k]) = m;
    if (TERMS > 2) *reinterpret_cast<vec_t*>(&S[2][row][k]) = l;
}
template <int TERMS, bool KC, int ROWS = 64>
__device__ __forceinline__ void gm_store(__bf16 (*S)[ROWS][GM_LD], int tid, const float4 (&v)[GM_NV]) {
    if (KC) {
        constexpr int QK = GM_BK / 4;
#pragma unroll
        for (int q = 0; q < GM_NV; ++q) {
            const int f = tid + 4 * ROWS * q;               // 4 ROWS threads
            const float x[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
            gm_put<TERMS, 4, ROWS>(S, f / QK, 4 * (f % QK), x);
        }
    } else {
        const int kg = (tid & 7) + 8 * ((tid >> 6) & 1), mq = ((tid >> 3) & 7) + 8 * (tid >> 7);
        float x0[GM_NV], x1[GM_NV], x2[GM_NV], x3[GM_NV];
#pragma unroll
        for (int i = 0; i < GM_NV; ++i) { x0[i] = v[i].x; x1[i] = v[i].y; x2[i] = v[i].z; x3[i] = v[i].w; }
        gm_put<TERMS, GM_NV, ROWS>(S, 4 * mq + 0, GM_NV * kg, x0);
        gm_put<TERMS, GM_NV, ROWS>(S, 4 * mq + 1, GM_NV * kg, x1);
        gm_put<TERMS, GM_NV, ROWS>(S, 4 * mq + 2, GM_NV * kg, x2);
        gm_put<TERMS, GM_NV, ROWS>(S, 4 * mq + 3, GM_NV * kg, x3);
    }
}

// A persistent grid: work-group w walks the tiles wv, wv + G, ... (wv: w renumbered so that consecutive tiles -- the n-tiles that
// share rows of A, the tiles of one K split -- run on one XCD and meet in its L2) and its loop runs over (tile, K step) pairs: the
// first operand tiles of the next output tile are in flight while the current one finishes its MFMAs and stores its result.
struct GmTile { const float* A; const float* B; float* C; int m0, n0, kbeg, kend, sp; };
__device__ __forceinline__ GmTile gm_tile(const GemmP& p, int t, int MT, int NT, int ks) {
    const int per_b = MT * NT * p.nsplit;
    const int bz = t / per_b, r1 = t - bz * per_b;
    const int sp = r1 / (MT * NT), r2 = r1 - sp * (MT * NT);
    const int mt = r2 / NT, nt = r2 - mt * NT;
    const int z1 = bz / p.nb2, z2 = bz - z1 * p.nb2;
    GmTile q;
    q.m0 = mt * 64; q.n0 = nt * 64; q.sp = sp;
    q.kbeg = sp * ks; q.kend = min(p.K, q.kbeg + ks);
    q.A = p.A + z1 * p.a1 + z2 * p.a2 + (long long)q.m0 * p.am;
    q.B = p.B + z1 * p.b1 + z2 * p.b2 + (long long)q.n0 * p.bn;
    q.C = p.C + z1 * p.c1 + z2 * p.c2;
    return q;
}

// The loop is rotated so that each of load / LDS store / MFMA block / result store appears once in the code (hipcc otherwise keeps
// every call site's lane offsets live across the loop: 316 registers and one work-group per CU): an iteration requests step i,
// multiplies step i - 1 out of LDS (and stores the tile that step finished), then moves step i from registers into LDS.
template <int TERMS, bool AKC, bool BKC>
__global__ __launch_bounds__(256, GM_WPS) void k_gemm(const GemmP p, const int vecA, const int vecB, const int total) {
    __shared__ __attribute__((aligned(16))) __bf16 As[TERMS][64][GM_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[TERMS][64][GM_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int G = gridDim.x;                                // a multiple of 8
    const int MT = (p.M + 63) / 64, NT = (p.N + 63) / 64;
    const int ks = ((p.K + p.nsplit - 1) / p.nsplit + GM_BK - 1) / GM_BK * GM_BK;
    // a K split past the end of K has nothing to add (mode 2)
    auto skip_empty = [&](int u) { while (u < total) { const int sp = (u % (MT * NT * p.nsplit)) / (MT * NT); if (sp * ks < p.K) break; u += G; } return u; };
    int t = skip_empty((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3));
    bool valid = t < total;                                 // the step being requested exists
    GmTile rq = gm_tile(p, valid ? t : 0, MT, NT, ks);      // ... and belongs to this tile, at k = rk
    int rk = rq.kbeg;
    bool have = false, last = false;                        // LDS holds a step; it is its tile's last
    GmTile cur = rq;                                        // the tile of the step in LDS
    f32x16 acc = zero16();
    float4 va[GM_NV], vb[GM_NV];
    if (GM_EXP & 2) for (int i = 0; i < GM_NV; ++i) va[i] = vb[i] = make_float4(1.f, 2.f, 3.f, (float)tid);
    while (true) {
        if (valid && !(GM_EXP & 2)) {
            gm_load<AKC>(rq.A + (long long)rk * p.ak, p.am, p.ak, p.M - rq.m0, rq.kend - rk, vecA != 0, tid, va);
            gm_load<BKC>(rq.B + (long long)rk * p.bk, p.bn, p.bk, p.N - rq.n0, rq.kend - rk, vecB != 0, tid, vb);
        }
        if (have) {
#pragma unroll
            for (int c = 0; c < ((GM_EXP & 1) ? 0 : GM_BK / 16); ++c) {
                const int ko = c * 16 + 8 * (lane >> 5);
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&As[0][wm * 32 + (lane & 31)][ko]);
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bs[0][wn * 32 + (lane & 31)][ko]);
                if (TERMS > 1) {     // small terms first
                    const bf16x8 am = *reinterpret_cast<const bf16x8*>(&As[TERMS > 1 ? 1 : 0][wm * 32 + (lane & 31)][ko]);
                    const bf16x8 bm = *reinterpret_cast<const bf16x8*>(&Bs[TERMS > 1 ? 1 : 0][wn * 32 + (lane & 31)][ko]);
                    if (TERMS > 2) {
                        const bf16x8 al = *reinterpret_cast<const bf16x8*>(&As[TERMS > 2 ? 2 : 0][wm * 32 + (lane & 31)][ko]);
                        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bs[TERMS > 2 ? 2 : 0][wn * 32 + (lane & 31)][ko]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
            if (last) {
                const int col = cur.n0 + wn * 32 + (lane & 31);
                if (col < p.N && !((GM_EXP & 4) && acc[0] != 12345.f)) {
                    const float bv = (p.bias && cur.sp == 0) ? p.bias[col] : 0.f;
                    const float al = p.colscale ? p.alpha * p.colscale[col] : p.alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = cur.m0 + wm * 32 + acc_row(r, lane);
                        if (row >= p.M) continue;
                        float* d = cur.C + (long long)row * p.cm + (long long)col * p.cn;
                        float v = acc[r] * al + bv;
                        if (p.mode == 0) {
                            if (p.relu) v = fmaxf(v, 0.f);
                            if (p.gate && !(p.gate[d - p.C] > 0.f)) v = 0.f;
                            *d = v;
                        } else if (p.mode == 1) *d += v;
                        else atomicAdd(d, v);
                    }
                }
                acc = zero16();
            }
            __syncthreads();                                // every wave has read the step out of LDS
        }
        if (!valid) break;
        gm_store<TERMS, AKC>(As, tid, va);
        gm_store<TERMS, BKC>(Bs, tid, vb);
        __syncthreads();
        have = true; cur = rq; last = rk + GM_BK >= rq.kend;
        if (!last) rk += GM_BK;
        else {
            t = skip_empty(t + G);
            valid = t < total;
            if (valid) { rq = gm_tile(p, t, MT, NT, ks); rk = rq.kbeg; }
        }
    }
}

// ---- the fast path: M and N multiples of 64, every K range a multiple of GM_BK, one unit stride per operand, offsets below 2^31.
// Nothing is tested per element, a tile is a grid index, a lane's offsets into its operand tiles are computed once (32 bit) and the
// uniform tile pointers advance by one step; float4 loads are declared 4-byte aligned (weights inside the flat blob).
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int TERMS, bool AKC, bool BKC>
__global__ __launch_bounds__(256, GM_WPS) void k_gemm_fast(const GemmP p) {
    __shared__ __attribute__((aligned(16))) __bf16 As[TERMS][64][GM_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[TERMS][64][GM_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // p.xcd (split-K, batch 1, nsplit a multiple of 8): work-groups are handed to the eight XCDs round robin in launch order, so the
    // tiles of ONE K range are renumbered onto ONE XCD -- the operand rows they share are then fetched into that XCD's L2 once, not
    // into all eight
    // (the same for the n-tiles of one row block of an unsplit GEMM -- they share rows of A -- changed nothing: 164 vs 161 us for the
    // five n-tiles of the stacked forward projection; not kept)
    int bx = blockIdx.x, by = blockIdx.y, bzs = blockIdx.z;
    if (p.xcd) {
        const int tiles = gridDim.x * gridDim.y, id = bx + gridDim.x * (by + gridDim.y * bzs);
        const int k = id >> 3, tile = k % tiles;
        bzs = (k / tiles) * 8 + (id & 7); bx = tile % gridDim.x; by = tile / gridDim.x;
    }
    const int n0 = bx * 64, m0 = by * 64;
    const int bz = bzs / p.nsplit, sp = bzs - bz * p.nsplit;
    const int z1 = bz / p.nb2, z2 = bz - z1 * p.nb2;
    const int ks = ((p.K + p.nsplit - 1) / p.nsplit + GM_BK - 1) / GM_BK * GM_BK;
    const int kbeg = sp * ks, kend = min(p.K, kbeg + ks);
    if (kbeg >= kend) return;
    const float* A = p.A + z1 * p.a1 + z2 * p.a2 + (long long)m0 * p.am + (long long)kbeg * p.ak;
    const float* B = p.B + z1 * p.b1 + z2 * p.b2 + (long long)n0 * p.bn + (long long)kbeg * p.bk;
    constexpr int QK = GM_BK / 4;
    const int kg = (tid & 7) + 8 * ((tid >> 6) & 1), mq = ((tid >> 3) & 7) + 8 * (tid >> 7);
    unsigned offA[GM_NV], offB[GM_NV];
#pragma unroll
    for (int q = 0; q < GM_NV; ++q) {
        const int f = tid + 256 * q;
        offA[q] = AKC ? (unsigned)((f / QK) * (int)p.am + 4 * (f % QK)) : (unsigned)((GM_NV * kg + q) * (int)p.ak + 4 * mq);
        offB[q] = BKC ? (unsigned)((f / QK) * (int)p.bn + 4 * (f % QK)) : (unsigned)((GM_NV * kg + q) * (int)p.bk + 4 * mq);
    }
    const long long stepA = (long long)GM_BK * p.ak, stepB = (long long)GM_BK * p.bk;
    float4 va[GM_NV], vb[GM_NV];
    auto load = [&]() {
#pragma unroll
        for (int q = 0; q < GM_NV; ++q) {
            const f4u a = *reinterpret_cast<const f4u*>(A + offA[q]);
            const f4u b = *reinterpret_cast<const f4u*>(B + offB[q]);
            va[q] = make_float4(a[0], a[1], a[2], a[3]);
            vb[q] = make_float4(b[0], b[1], b[2], b[3]);
        }
    };
    f32x16 acc = zero16();
    const int nsteps = (kend - kbeg) / GM_BK;
    const bool rowsum = !AKC && p.asum != nullptr && bx == 0;     // uniform
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    load();
    for (int it = 0; it < nsteps; ++it) {
        if (rowsum) {
#pragma unroll
            for (int q = 0; q < GM_NV; ++q) { rs[0] += va[q].x; rs[1] += va[q].y; rs[2] += va[q].z; rs[3] += va[q].w; }
        }
        gm_store<TERMS, AKC>(As, tid, va);
        gm_store<TERMS, BKC>(Bs, tid, vb);
        __syncthreads();
        if (it + 1 < nsteps) { A += stepA; B += stepB; load(); }
#pragma unroll
        for (int c = 0; c < GM_BK / 16; ++c) {
            const int ko = c * 16 + 8 * (lane >> 5);
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&As[0][wm * 32 + (lane & 31)][ko]);
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bs[0][wn * 32 + (lane & 31)][ko]);
            if (TERMS > 1) {     // small terms first
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(&As[TERMS > 1 ? 1 : 0][wm * 32 + (lane & 31)][ko]);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(&Bs[TERMS > 1 ? 1 : 0][wn * 32 + (lane & 31)][ko]);
                if (TERMS > 2) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(&As[TERMS > 2 ? 2 : 0][wm * 32 + (lane & 31)][ko]);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bs[TERMS > 2 ? 2 : 0][wn * 32 + (lane & 31)][ko]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (rowsum) {     // a thread holds rows 4 mq .. + 3 over its k; the eight lanes tid & 7 of two waves share mq.  The 64 sums meet
                      // in LDS and leave as ONE full-wave atomic: what an atomic costs at the L2 is its instruction, not its lanes
#pragma unroll
        for (int j = 0; j < 4; ++j) { rs[j] += __shfl_xor(rs[j], 1); rs[j] += __shfl_xor(rs[j], 2); rs[j] += __shfl_xor(rs[j], 4); }
        float* red = reinterpret_cast<float*>(&As[0][0][0]);            // [2][64]; the last step's barrier has passed
        if ((tid & 7) == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[((tid >> 6) & 1) * 64 + 4 * mq + j] = rs[j];
        }
        __syncthreads();
        if (tid < 64) atomicAdd(p.asum + (p.cblk && p.cblk_m ? p.atab[m0 / p.cblk] - (long long)(m0 / p.cblk) * p.cblk : 0) + m0 + tid, red[tid] + red[64 + tid]);
    }
    const int col = n0 + wn * 32 + (lane & 31);
    const float bv = (p.bias && sp == 0) ? p.bias[col] : 0.f;
    const float al = p.colscale ? p.alpha * p.colscale[col] : p.alpha;
    long long cofs = 0;                          // C in blocks: the tile lies inside one (cblk is a multiple of the tile)
    if (p.cblk) { const int t = (p.cblk_m ? m0 : n0) / p.cblk; cofs = p.ctab[t] - (long long)t * p.cblk * (p.cblk_m ? p.cm : p.cn); }
    float* d0 = p.C + cofs + z1 * p.c1 + z2 * p.c2 + (long long)(m0 + wm * 32 + 4 * (lane >> 5)) * p.cm + (long long)col * p.cn;
    if (p.mode == 0) {
        const float* g0 = p.gate ? p.gate + (d0 - p.C) : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long o = (long long)((r & 3) + 8 * (r >> 2)) * p.cm;
            float v = acc[r] * al + bv;
            if (p.relu) v = fmaxf(v, 0.f);
            if (g0 && !(g0[o] > 0.f)) v = 0.f;
            d0[o] = v;
        }
    } else if (p.mode == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) d0[(long long)((r & 3) + 8 * (r >> 2)) * p.cm] += acc[r] * al + bv;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) atomicAdd(d0 + (long long)((r & 3) + 8 * (r >> 2)) * p.cm, acc[r] * al + bv);
    }
}
// ---- the same for M and N multiples of 128: a 128 x 128 tile per 512-thread work-group, eight waves of 32 x 64 (two accumulators).
// Per MFMA half the split arithmetic and LDS stores of the 64 x 64 tile and three quarters of its fragment reads; A is read once for
// N = 128.  60 KiB of LDS: two work-groups = sixteen waves per CU, as before.
template <int TERMS, bool AKC, bool BKC>
__global__ __launch_bounds__(512, 4) void k_gemm_big(const GemmP p) {
    __shared__ __attribute__((aligned(16))) __bf16 As[TERMS][128][GM_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[TERMS][128][GM_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // p.xcd (split-K, batch 1, nsplit a multiple of 8): work-groups are handed to the eight XCDs round robin in launch order, so the
    // tiles of ONE K range are renumbered onto ONE XCD -- the operand rows they share are then fetched into that XCD's L2 once, not
    // into all eight
    // (the same for the n-tiles of one row block of an unsplit GEMM -- they share rows of A -- changed nothing: 164 vs 161 us for the
    // five n-tiles of the stacked forward projection; not kept)
    int bx = blockIdx.x, by = blockIdx.y, bzs = blockIdx.z;
    if (p.xcd) {
        const int tiles = gridDim.x * gridDim.y, id = bx + gridDim.x * (by + gridDim.y * bzs);
        const int k = id >> 3, tile = k % tiles;
        bzs = (k / tiles) * 8 + (id & 7); bx = tile % gridDim.x; by = tile / gridDim.x;
    }
    const int n0 = bx * 128, m0 = by * 128;
    const int bz = bzs / p.nsplit, sp = bzs - bz * p.nsplit;
    const int z1 = bz / p.nb2, z2 = bz - z1 * p.nb2;
    const int ks = ((p.K + p.nsplit - 1) / p.nsplit + GM_BK - 1) / GM_BK * GM_BK;
    const int kbeg = sp * ks, kend = min(p.K, kbeg + ks);
    if (kbeg >= kend) return;
    const float* A = p.A + z1 * p.a1 + z2 * p.a2 + (long long)m0 * p.am + (long long)kbeg * p.ak;
    const float* B = p.B + z1 * p.b1 + z2 * p.b2 + (long long)n0 * p.bn + (long long)kbeg * p.bk;
    constexpr int QK = GM_BK / 4;
    const int kg = (tid & 7) + 8 * ((tid >> 6) & 1), mq = ((tid >> 3) & 7) + 8 * (tid >> 7);       // mq 0 .. 31
    unsigned offA[GM_NV], offB[GM_NV];
#pragma unroll
    for (int q = 0; q < GM_NV; ++q) {
        const int f = tid + 512 * q;
        offA[q] = AKC ? (unsigned)((f / QK) * (int)p.am + 4 * (f % QK)) : (unsigned)((GM_NV * kg + q) * (int)p.ak + 4 * mq);
        offB[q] = BKC ? (unsigned)((f / QK) * (int)p.bn + 4 * (f % QK)) : (unsigned)((GM_NV * kg + q) * (int)p.bk + 4 * mq);
    }
    const long long stepA = (long long)GM_BK * p.ak, stepB = (long long)GM_BK * p.bk;
    float4 va[GM_NV], vb[GM_NV];
    auto load = [&]() {
#pragma unroll
        for (int q = 0; q < GM_NV; ++q) {
            const f4u a = *reinterpret_cast<const f4u*>(A + offA[q]);
            const f4u b = *reinterpret_cast<const f4u*>(B + offB[q]);
            va[q] = make_float4(a[0], a[1], a[2], a[3]);
            vb[q] = make_float4(b[0], b[1], b[2], b[3]);
        }
    };
    f32x16 acc0 = zero16(), acc1 = zero16();
    const int nsteps = (kend - kbeg) / GM_BK;
    const bool rowsum = !AKC && p.asum != nullptr && bx == 0;     // uniform
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    load();
    for (int it = 0; it < nsteps; ++it) {
        if (rowsum) {
#pragma unroll
            for (int q = 0; q < GM_NV; ++q) { rs[0] += va[q].x; rs[1] += va[q].y; rs[2] += va[q].z; rs[3] += va[q].w; }
        }
        gm_store<TERMS, AKC, 128>(As, tid, va);
        gm_store<TERMS, BKC, 128>(Bs, tid, vb);
        __syncthreads();
        if (it + 1 < nsteps) { A += stepA; B += stepB; load(); }
#pragma unroll
        for (int c = 0; c < GM_BK / 16; ++c) {
            const int ko = c * 16 + 8 * (lane >> 5);
            const int ar = wm * 32 + (lane & 31), br = wn * 64 + (lane & 31);
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&As[0][ar][ko]);
            const bf16x8 b0h = *reinterpret_cast<const bf16x8*>(&Bs[0][br][ko]);
            const bf16x8 b1h = *reinterpret_cast<const bf16x8*>(&Bs[0][br + 32][ko]);
            if (TERMS > 1) {     // small terms first
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(&As[TERMS > 1 ? 1 : 0][ar][ko]);
                const bf16x8 b0m = *reinterpret_cast<const bf16x8*>(&Bs[TERMS > 1 ? 1 : 0][br][ko]);
                const bf16x8 b1m = *reinterpret_cast<const bf16x8*>(&Bs[TERMS > 1 ? 1 : 0][br + 32][ko]);
                if (TERMS > 2) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(&As[TERMS > 2 ? 2 : 0][ar][ko]);
                    const bf16x8 b0l = *reinterpret_cast<const bf16x8*>(&Bs[TERMS > 2 ? 2 : 0][br][ko]);
                    const bf16x8 b1l = *reinterpret_cast<const bf16x8*>(&Bs[TERMS > 2 ? 2 : 0][br + 32][ko]);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b0h, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b1h, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0l, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1l, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0m, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1m, acc1, 0, 0, 0);
                }
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b0h, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, b1h, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0m, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1m, acc1, 0, 0, 0);
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b0h, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, b1h, acc1, 0, 0, 0);
        }
        __syncthreads();
    }
    if (rowsum) {     // as in k_gemm_fast: 128 sums, two contributing waves each
#pragma unroll
        for (int j = 0; j < 4; ++j) { rs[j] += __shfl_xor(rs[j], 1); rs[j] += __shfl_xor(rs[j], 2); rs[j] += __shfl_xor(rs[j], 4); }
        float* red = reinterpret_cast<float*>(&As[0][0][0]);            // [2][128]
        if ((tid & 7) == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[((tid >> 6) & 1) * 128 + 4 * mq + j] = rs[j];
        }
        __syncthreads();
        if (tid < 128) atomicAdd(p.asum + (p.cblk && p.cblk_m ? p.atab[m0 / p.cblk] - (long long)(m0 / p.cblk) * p.cblk : 0) + m0 + tid, red[tid] + red[128 + tid]);
    }
    long long cofs = 0;                          // C in blocks: the tile lies inside one (cblk is a multiple of the tile)
    if (p.cblk) { const int t = (p.cblk_m ? m0 : n0) / p.cblk; cofs = p.ctab[t] - (long long)t * p.cblk * (p.cblk_m ? p.cm : p.cn); }
    float* dbase = p.C + cofs + z1 * p.c1 + z2 * p.c2 + (long long)(m0 + wm * 32 + 4 * (lane >> 5)) * p.cm;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const f32x16& acc = half ? acc1 : acc0;
        const int col = n0 + wn * 64 + 32 * half + (lane & 31);
        const float bv = (p.bias && sp == 0) ? p.bias[col] : 0.f;
        const float al = p.colscale ? p.alpha * p.colscale[col] : p.alpha;
        float* d0 = dbase + (long long)col * p.cn;
        if (p.mode == 0) {
            const float* g0 = p.gate ? p.gate + (d0 - p.C) : nullptr;
            const long long mw = (long long)(m0 + wm * 32 + 4 * (lane >> 5)) * (p.N >> 5) + (col >> 5);      // mask word of this lane's first row
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long o = (long long)((r & 3) + 8 * (r >> 2)) * p.cm;
                float v = acc[r] * al + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                if (g0 && !(g0[o] > 0.f)) v = 0.f;
                if (p.mask_in && !((p.mask_in[mw + (long long)((r & 3) + 8 * (r >> 2)) * (p.N >> 5)] >> (lane & 31)) & 1u)) v = 0.f;
                if (p.mask_out) {         // lanes 0..31 hold one row, lanes 32..63 the row four below
                    const unsigned long long bal = __ballot(v > 0.f);
                    if ((lane & 31) == 0) p.mask_out[mw + (long long)((r & 3) + 8 * (r >> 2)) * (p.N >> 5)] = (unsigned)(lane ? bal >> 32 : bal);
                }
                d0[o] = v;
            }
        } else if (p.mode == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) d0[(long long)((r & 3) + 8 * (r >> 2)) * p.cm] += acc[r] * al + bv;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) atomicAdd(d0 + (long long)((r & 3) + 8 * (r >> 2)) * p.cm, acc[r] * al + bv);
        }
    }
}
template <int TERMS>
static void launch_gemm_big_t(hipStream_t st, const GemmP& p, dim3 grid, bool akc, bool bkc) {
    if (akc && bkc) hipLaunchKernelGGL((k_gemm_big<TERMS, true, true>), grid, dim3(512), 0, st, p);
    else if (akc) hipLaunchKernelGGL((k_gemm_big<TERMS, true, false>), grid, dim3(512), 0, st, p);
    else if (bkc) hipLaunchKernelGGL((k_gemm_big<TERMS, false, true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_big<TERMS, false, false>), grid, dim3(512), 0, st, p);
}

template <int TERMS>
static void launch_gemm_fast_t(hipStream_t st, const GemmP& p, dim3 grid, bool akc, bool bkc) {
    if (akc && bkc) hipLaunchKernelGGL((k_gemm_fast<TERMS, true, true>), grid, dim3(256), 0, st, p);
    else if (akc) hipLaunchKernelGGL((k_gemm_fast<TERMS, true, false>), grid, dim3(256), 0, st, p);
    else if (bkc) hipLaunchKernelGGL((k_gemm_fast<TERMS, false, true>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_fast<TERMS, false, false>), grid, dim3(256), 0, st, p);
}

template <int TERMS>
static void launch_gemm_t(hipStream_t st, const GemmP& p, dim3 grid, bool akc, bool bkc, int vecA, int vecB, int total) {
    if (akc && bkc) hipLaunchKernelGGL((k_gemm<TERMS, true, true>), grid, dim3(256), 0, st, p, vecA, vecB, total);
    else if (akc) hipLaunchKernelGGL((k_gemm<TERMS, true, false>), grid, dim3(256), 0, st, p, vecA, vecB, total);
    else if (bkc) hipLaunchKernelGGL((k_gemm<TERMS, false, true>), grid, dim3(256), 0, st, p, vecA, vecB, total);
    else hipLaunchKernelGGL((k_gemm<TERMS, false, false>), grid, dim3(256), 0, st, p, vecA, vecB, total);
}
// terms: bf16 pieces per operand -- 1: plain bf16 (one MFMA per product), 2: 16 significand bits (three MFMAs), 3: 24 bits (six)
// which kernel launch_gemm runs a product on: 2 the 128-tile kernel, 1 the 64-tile kernel, 0 the generic one
static int gemm_kernel_choice(const GemmP& p) {
    const bool ua = p.ak == 1 || p.am == 1, ub = p.bk == 1 || p.bn == 1;
    const long long ks = ((p.K + p.nsplit - 1) / p.nsplit + GM_BK - 1) / GM_BK * GM_BK;
    const long long lim = 1LL << 24;              // a lane offset is at most 64 rows (or GM_BK k) of such a stride
    static const bool off = getenv("GENIE_GEMM_GENERIC") != nullptr;
    const bool fast = !off && ua && ub && p.M % 64 == 0 && p.N % 64 == 0 && p.K % GM_BK == 0 && ks % GM_BK == 0 && p.am < lim && p.ak < lim && p.bk < lim &&
                      p.bn < lim && (long long)p.batch * p.nsplit < 65536 && p.M / 64 < 65536;
    if (!fast) return 0;
    static const bool nobig = getenv("GENIE_GEMM_NO_BIG") != nullptr;
    const long long big_tiles = (long long)(p.M / 128) * (p.N / 128) * p.batch * p.nsplit;
    const bool use_big = !nobig && p.M % 128 == 0 && p.N % 128 == 0 && (big_tiles >= 256 || (big_tiles >= 192 && (p.M / 128) * (p.N / 128) >= 2));
    return use_big ? 2 : 1;
}
bool gemm_takes_mask(const GemmP& p) { return p.cblk == 0 && p.batch == 1 && p.mode == 0 && p.cn == 1 && p.cm == p.N && gemm_kernel_choice(p) == 2; }

void launch_gemm(hipStream_t st, const GemmP& p_in, int terms) {
    GemmP p = p_in;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.batch <= 0) return;
    if ((p.mask_in || p.mask_out) && !gemm_takes_mask(p)) { fprintf(stderr, "launch_gemm: mask_in / mask_out on a product the 128-tile kernel does not run\n"); abort(); }
    {   // the fast path
        const int choice = gemm_kernel_choice(p);
        const bool fast = choice > 0;
        if (p.cblk > 0 && !(fast && p.cblk % 128 == 0 && !p.gate && p.batch == 1)) {
            // C in blocks is a property of the two tiled kernels: otherwise one GEMM per block
            const int nblk = ((p.cblk_m ? p.M : p.N) + p.cblk - 1) / p.cblk;
            for (int t = 0; t < nblk; ++t) {
                GemmP q = p;
                q.cblk = 0;
                q.C = p.C + p.ctab[t];
                const int w = std::min(p.cblk, (p.cblk_m ? p.M : p.N) - t * p.cblk);
                if (p.cblk_m) { q.M = w; q.A = p.A + (long long)t * p.cblk * p.am; if (p.asum) q.asum = p.asum + p.atab[t]; }
                else { q.N = w; q.B = p.B + (long long)t * p.cblk * p.bn; if (p.bias) q.bias = p.bias + (long long)t * p.cblk;
                       if (p.colscale) q.colscale = p.colscale + (long long)t * p.cblk; }
                if (q.mode == 2) q.nsplit = gemm_splits(q.M, q.N, q.K, q.batch);
                launch_gemm(st, q, terms);
            }
            return;
        }
        if (fast) {
            static const bool noxcd = getenv("GENIE_GEMM_NO_XCD") != nullptr;
            p.xcd = (!noxcd && p.batch == 1 && p.nsplit >= 8 && p.nsplit % 8 == 0) ? 1 : 0;

            const bool akc = p.ak == 1, bkc = p.bk == 1;
            if (p.asum && (akc || p.batch != 1)) { launch_colsum(st, p.A, nullptr, p.K, p.M, p.asum, nullptr, p.ak); p.asum = nullptr; }
            if (choice == 2) {
                const dim3 gb(p.N / 128, p.M / 128, p.batch * p.nsplit);
                if (terms <= 1) launch_gemm_big_t<1>(st, p, gb, akc, bkc);
                else if (terms == 2) launch_gemm_big_t<2>(st, p, gb, akc, bkc);
                else launch_gemm_big_t<3>(st, p, gb, akc, bkc);
                return;
            }
            const dim3 grid(p.N / 64, p.M / 64, p.batch * p.nsplit);
            if (terms <= 1) launch_gemm_fast_t<1>(st, p, grid, akc, bkc);
            else if (terms == 2) launch_gemm_fast_t<2>(st, p, grid, akc, bkc);
            else launch_gemm_fast_t<3>(st, p, grid, akc, bkc);
            return;
        }
    }
    if (p.asum) { launch_colsum(st, p.A, nullptr, p.K, p.M, p.asum, nullptr, p.ak); p.asum = nullptr; }      // the generic kernel does not carry the row sums (A: [K][M], row stride ak)
    const long long tiles = (long long)((p.N + 63) / 64) * ((p.M + 63) / 64) * p.batch * p.nsplit;
    const int total = (int)tiles;
    const int wgs_per_cu = GM_WPS;                      // what __launch_bounds__ and the LDS tiles admit per CU
    long long g = tiles < 256LL * wgs_per_cu ? tiles : 256LL * wgs_per_cu;
    g = (g + 7) / 8 * 8;
    const dim3 grid((unsigned)g);
    // lanes walk k unless the row index is the operand's only unit stride
    const bool akc = p.ak == 1 || p.am != 1, bkc = p.bk == 1 || p.bn != 1;
    auto mult4 = [](long long v) { return (v & 3) == 0; };
    const int vecA = ((akc ? p.ak == 1 : p.am == 1) && mult4(akc ? p.am : p.ak) && mult4(p.a1) && mult4(p.a2) && ((uintptr_t)p.A & 15) == 0) ? 1 : 0;
    const int vecB = ((bkc ? p.bk == 1 : p.bn == 1) && mult4(bkc ? p.bn : p.bk) && mult4(p.b1) && mult4(p.b2) && ((uintptr_t)p.B & 15) == 0) ? 1 : 0;
    if (terms <= 1) launch_gemm_t<1>(st, p, grid, akc, bkc, vecA, vecB, total);
    else if (terms == 2) launch_gemm_t<2>(st, p, grid, akc, bkc, vecA, vecB, total);
    else launch_gemm_t<3>(st, p, grid, akc, bkc, vecA, vecB, total);
}
// split-K factor for a reduction of length K into `tiles` output tiles (x batch): enough work-groups to fill the chip, at least 128 of
// K each.  Only for mode 2 (atomic accumulation).
#ifndef GM_SPLIT_WGS
#define GM_SPLIT_WGS 512      // work-groups of a multi-tile split-K GEMM: two per CU, so that one's operand staging runs under the other's MFMAs
#endif
int gemm_splits(long long M, long long N, long long K, long long batch) {
    // several 128 x 128 output tiles: one work-group of the 128-tile kernel per CU (its LDS admits one) -- the 64-tile kernel spends
    // more vector-ALU time splitting its operand tiles into bf16 pieces than the matrix pipe spends on them (a 64 x 64 tile has half
    // the MFMAs per split element), which is what a long-K weight gradient is bound by, not its bytes
    if (batch == 1 && M % 128 == 0 && N % 128 == 0 && (M / 128) * (N / 128) >= 2 && (M / 128) * (N / 128) <= 32 && K >= 16384) {
        const long long t = (M / 128) * (N / 128);
        return (int)std::max<long long>(8, GM_SPLIT_WGS / t / 8 * 8);
    }
    const long long tiles = ((M + 63) / 64) * ((N + 63) / 64) * batch;
    long long s = (768 + tiles - 1) / tiles;
    const long long smax = (K + 127) / 128;
    if (s > smax) s = smax;
    if (s > 512) s = 512;
    if (s >= 8 && batch == 1) s = std::min((s + 7) / 8 * 8, smax / 8 * 8 > 0 ? smax / 8 * 8 : s);      // a multiple of 8: one K range per XCD (GemmP::xcd)
    return (int)(s < 1 ? 1 : s);
}

// ------------------------------------------------------------------------------------------------ LayerNorm (eps 1e-5, affine)
// one wave per row, C <= 512; keeps xhat and 1/sigma
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ rstd, long long R, int C) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* xr = x + row * C;
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int c = lane + 64 * q; v[q] = c < C ? xr[c] : 0.f; s += v[q]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int c = lane + 64 * q; const float d = c < C ? v[q] - mean : 0.f; ss += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float rs = 1.0f / sqrtf(ss / (float)C + GENIE_LN_EPS);
    if (lane == 0 && rstd) rstd[row] = rs;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = lane + 64 * q;
        if (c < C) {
            const float xh = (v[q] - mean) * rs;
            if (xhat) xhat[row * C + c] = xh;
            if (y) y[row * C + c] = xh * g[c] + b[c];
        }
    }
}
// dx = rstd (dy g - mean(dy g) - xhat mean(dy g xhat)); `accumulate`: dx += (residual path already holds a gradient).
// dgamma / dbeta (optional): += sum_r dy xhat / sum_r dy -- a block walks rows_per_block rows (a wave every fourth), its lanes keep
// their columns' sums in registers, the four waves meet in LDS and the block adds once per column: no second pass over dy and xhat.
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                const float* __restrict__ g, float* __restrict__ dx, long long R, int C, int accumulate,
                                                int rows_per_block, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[2][4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    float ag[8], ab[8], gv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { ag[q] = 0.f; ab[q] = 0.f; const int c = lane + 64 * q; gv[q] = c < C ? g[c] : 0.f; }
    // two rows per iteration (rows wave and wave + 4 of an eight-row group): twice the loads in flight
    for (long long rb = r0 + wave; rb < r1; rb += 8) {
        float t[2][8], xh[2][8], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long row = rb + 4 * u;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = lane + 64 * q;
                const bool ok = c < C && row < r1;
                const float dv = ok ? dy[row * C + c] : 0.f;
                xh[u][q] = ok ? xhat[row * C + c] : 0.f;
                t[u][q] = dv * gv[q];
                ag[q] += dv * xh[u][q]; ab[q] += dv;
                s1[u] += t[u][q]; s2[u] += t[u][q] * xh[u][q];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1[0] += __shfl_xor(s1[0], o); s2[0] += __shfl_xor(s2[0], o);
            s1[1] += __shfl_xor(s1[1], o); s2[1] += __shfl_xor(s2[1], o);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long row = rb + 4 * u;
            if (row >= r1) continue;
            const float m1 = s1[u] / (float)C, m2 = s2[u] / (float)C;
            const float rs = rstd[row];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c = lane + 64 * q;
                if (c < C) {
                    const float v = rs * (t[u][q] - m1 - xh[u][q] * m2);
                    if (accumulate) dx[row * C + c] += v; else dx[row * C + c] = v;
                }
            }
        }
    }
    if (!dgamma) return;                                    // uniform
#pragma unroll
    for (int q = 0; q < 8; ++q) { red[0][wave][lane + 64 * q] = ag[q]; red[1][wave][lane + 64 * q] = ab[q]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}
// ... C = 128 (every LayerNorm of the pair stack): a row is one half-wave of float4 lanes, eight rows per wave and iteration (the generic
// form above has 2 KB per wave in flight and ran at 2.5 TB/s); gamma / beta sums in registers over the block's rows, one atomic per
// column and block
__global__ __launch_bounds__(256) void k_ln_bwd128(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                   const float* __restrict__ g, float* __restrict__ dx, long long R, int accumulate,
                                                   int rows_per_block, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    constexpr int C = 128, U = 4;
    __shared__ float red[2][8][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, hw = lane >> 5;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    const float4 gv = *reinterpret_cast<const float4*>(g + 4 * l32);
    float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
    for (long long rb = r0 + wave * 2 * U; rb < r1; rb += 8 * U) {
        float4 d[U], xh[U], old[U];
        float rs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long row = rb + 2 * u + hw;
            const bool ok = row < r1;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            d[u] = ok ? *reinterpret_cast<const float4*>(dy + row * C + 4 * l32) : z4;
            xh[u] = ok ? *reinterpret_cast<const float4*>(xhat + row * C + 4 * l32) : z4;
            old[u] = (ok && accumulate) ? *reinterpret_cast<const float4*>(dx + row * C + 4 * l32) : z4;
            rs[u] = ok ? rstd[row] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long row = rb + 2 * u + hw;
            ag.x += d[u].x * xh[u].x; ag.y += d[u].y * xh[u].y; ag.z += d[u].z * xh[u].z; ag.w += d[u].w * xh[u].w;
            ab.x += d[u].x; ab.y += d[u].y; ab.z += d[u].z; ab.w += d[u].w;
            const float4 t = make_float4(d[u].x * gv.x, d[u].y * gv.y, d[u].z * gv.z, d[u].w * gv.w);
            float s1 = (t.x + t.y) + (t.z + t.w), s2 = (t.x * xh[u].x + t.y * xh[u].y) + (t.z * xh[u].z + t.w * xh[u].w);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            const float m1 = s1 * (1.0f / C), m2 = s2 * (1.0f / C);
            if (row < r1) {
                float4 v = make_float4(rs[u] * (t.x - m1 - xh[u].x * m2), rs[u] * (t.y - m1 - xh[u].y * m2), rs[u] * (t.z - m1 - xh[u].z * m2),
                                       rs[u] * (t.w - m1 - xh[u].w * m2));
                v.x += old[u].x; v.y += old[u].y; v.z += old[u].z; v.w += old[u].w;
                *reinterpret_cast<float4*>(dx + row * C + 4 * l32) = v;
            }
        }
    }
    if (!dgamma) return;                                    // uniform
    const int slot = wave * 2 + hw;
    *reinterpret_cast<float4*>(&red[0][slot][4 * l32]) = ag;
    *reinterpret_cast<float4*>(&red[1][slot][4 * l32]) = ab;
    __syncthreads();
    {
        const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a += red[which][k][c];
        atomicAdd((which ? dbeta : dgamma) + c, a);
    }
}
// column sums over rows: out_b[c] += sum_r a[r,c];  out_g[c] += sum_r a[r,c] w[r,c]   (LayerNorm gamma / beta and Linear bias gradients)
// 256 threads = (256 / C) row groups x C columns (C < 256), or one row group looping over the columns; rows of a block are summed
// in registers, row groups through LDS, blocks with one atomic per column.
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ a, const float* __restrict__ w, long long R, int C, int rows_per_block,
                                                float* __restrict__ out_b, float* __restrict__ out_g, long long lda) {
    __shared__ float red[2][256];
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < R ? r0 + rows_per_block : R;
    const int tid = threadIdx.x;
    if (C >= 256) {
        for (int c = tid; c < C; c += 256) {
            float sb = 0.f, sg = 0.f;
            for (long long r = r0; r < r1; ++r) {
                const float v = a[r * lda + c];
                sb += v;
                if (w) sg += v * w[r * C + c];
            }
            if (out_b) atomicAdd(out_b + c, sb);
            if (out_g) atomicAdd(out_g + c, sg);
        }
        return;
    }
    const int nrg = 256 / C, c = tid % C, rg = tid / C;
    float sb = 0.f, sg = 0.f;
    if (rg < nrg)
        for (long long r = r0 + rg; r < r1; r += nrg) {
            const float v = a[r * lda + c];
            sb += v;
            if (w) sg += v * w[r * C + c];
        }
    red[0][tid] = sb; red[1][tid] = sg;
    __syncthreads();
    if (rg == 0) {
        for (int g = 1; g < nrg; ++g) { sb += red[0][g * C + c]; sg += red[1][g * C + c]; }
        if (out_b) atomicAdd(out_b + c, sb);
        if (out_g) atomicAdd(out_g + c, sg);
    }
}
// ... C = 128 with no affine output wanted (the pair stack's LayerNorms keep xhat only): a row per half-wave of float4 lanes
__global__ __launch_bounds__(256) void k_ln_fwd128(const float* __restrict__ x, float* __restrict__ xhat, float* __restrict__ rstd, long long R) {
    constexpr int C = 128;
    const int lane = threadIdx.x & 63, l32 = lane & 31;
    const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    if (row >= R) return;           // (a whole half-wave; the shuffles below stay inside it)
    const float4 v = *reinterpret_cast<const float4*>(x + row * C + 4 * l32);
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / C);
    const float4 dv = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
    float ss = (dv.x * dv.x + dv.y * dv.y) + (dv.z * dv.z + dv.w * dv.w);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float rs = 1.0f / sqrtf(ss * (1.0f / C) + GENIE_LN_EPS);
    if (l32 == 0 && rstd) rstd[row] = rs;
    *reinterpret_cast<float4*>(xhat + row * C + 4 * l32) = make_float4(dv.x * rs, dv.y * rs, dv.z * rs, dv.w * rs);
}
void launch_ln_fwd(hipStream_t st, const float* x, const float* g, const float* b, float* y, float* xhat, float* rstd, long long R, int C) {
    if (C == 128 && !y && xhat && R >= 4096) { hipLaunchKernelGGL(k_ln_fwd128, dim3((unsigned)((R + 7) / 8)), dim3(256), 0, st, x, xhat, rstd, R); return; }
    hipLaunchKernelGGL(k_ln_fwd, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, x, g, b, y, xhat, rstd, R, C);
}
void launch_ln_bwd(hipStream_t st, const float* dy, const float* xhat, const float* rstd, const float* g, float* dx, long long R, int C, int accumulate,
                   float* dgamma, float* dbeta) {
    int rpb = (int)((R + 1023) / 1024);                     // about 1024 blocks; a multiple of eight rows each
    rpb = (rpb + 7) / 8 * 8;
    if (C == 128 && R >= 4096) {
        rpb = (rpb + 31) / 32 * 32;
        hipLaunchKernelGGL(k_ln_bwd128, dim3((unsigned)((R + rpb - 1) / rpb)), dim3(256), 0, st, dy, xhat, rstd, g, dx, R, accumulate, rpb, dgamma, dbeta);
        return;
    }
    hipLaunchKernelGGL(k_ln_bwd, dim3((unsigned)((R + rpb - 1) / rpb)), dim3(256), 0, st, dy, xhat, rstd, g, dx, R, C, accumulate, rpb, dgamma, dbeta);
}
void launch_colsum(hipStream_t st, const float* a, const float* w, long long R, int C, float* out_b, float* out_g, long long lda) {
    int rpb = (int)((R + 1023) / 1024);
    if (rpb < 16) rpb = 16;
    hipLaunchKernelGGL(k_colsum, dim3((unsigned)((R + rpb - 1) / rpb)), dim3(256), 0, st, a, w, R, C, rpb, out_b, out_g, lda > 0 ? lda : C);
}

// ------------------------------------------------------------------------------------------------ [B][R][C] <-> [B][C][R]
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ in, float* __restrict__ out, int R, int C, int to_cm) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const float* src = in + (size_t)b * R * C;
    float* dst = out + (size_t)b * R * C;
    if (to_cm) {        // in [R][C] -> out [C][R]
        for (int q = ty; q < 32; q += 8) tile[q][tx] = (r0 + q < R && c0 + tx < C) ? src[(size_t)(r0 + q) * C + c0 + tx] : 0.f;
        __syncthreads();
        for (int q = ty; q < 32; q += 8) if (c0 + q < C && r0 + tx < R) dst[(size_t)(c0 + q) * R + r0 + tx] = tile[tx][q];
    } else {            // in [C][R] -> out [R][C]
        for (int q = ty; q < 32; q += 8) tile[q][tx] = (c0 + q < C && r0 + tx < R) ? src[(size_t)(c0 + q) * R + r0 + tx] : 0.f;
        __syncthreads();
        for (int q = ty; q < 32; q += 8) if (r0 + q < R && c0 + tx < C) dst[(size_t)(r0 + q) * C + c0 + tx] = tile[tx][q];
    }
}
void launch_transpose(hipStream_t st, const float* in, float* out, int B, int R, int C, bool to_cm) {
    hipLaunchKernelGGL(k_transpose, dim3((R + 31) / 32, (C + 31) / 32, B), dim3(256), 0, st, in, out, R, C, to_cm ? 1 : 0);
}

// ------------------------------------------------------------------------------------------------ pair features (pair_feature_net.py:117-301)
// F[p] = [ template: nbin soft bins | 4 quat | fsm | fsm ] [ motif: nbin bins * fsm | fsm | fsm ] [ relpos one-hot (2k+2) | same chain ]
// one thread per pair (b, i, j); the same closed-form quaternion and sign codes as the sampling path (common.h)
__global__ __launch_bounds__(256) void k_pair_features(const float* __restrict__ trans, const float* __restrict__ rots, const int8_t* __restrict__ codes,
                                                       const float* __restrict__ rmask, const uint8_t* __restrict__ fstm, const uint8_t* __restrict__ fsm,
                                                       const float* __restrict__ mpos, const int32_t* __restrict__ ridx, const int32_t* __restrict__ cidx,
                                                       float* __restrict__ F, int B, int N, int nbin, float dmin, float dstep, int relk) {
    const long long pidx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pidx >= (long long)B * N * N) return;
    const int j = (int)(pidx % N), i = (int)((pidx / N) % N), b = (int)(pidx / ((long long)N * N));
    const int nf = (nbin + 6) + (nbin + 2) + (2 * relk + 3);
    float* f = F + pidx * nf;
    const float pm = rmask[b * N + i] * rmask[b * N + j];
    const float fs = fstm[pidx] ? 1.f : 0.f;
    auto bins = [&](const float* xi, const float* xj, float scale, float* out) {
        const float dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
        const float d = sqrtf(1e-10f + dx * dx + dy * dy + dz * dz);
        float mx = -3.0e38f;
        for (int k = 0; k < nbin; ++k) mx = fmaxf(mx, -4.0f * fabsf(d - (dmin + (float)k * dstep)));
        float s = 0.f;
        for (int k = 0; k < nbin; ++k) { const float e = expf(-4.0f * fabsf(d - (dmin + (float)k * dstep)) - mx); out[k] = e; s += e; }
        const float inv = scale / s;
        for (int k = 0; k < nbin; ++k) out[k] *= inv;
    };
    bins(trans + (size_t)(b * N + i) * 3, trans + (size_t)(b * N + j) * 3, pm, f);
    {   // r = R_j R_i (pair_feature_net.py:288-291), quaternion * pair mask
        const float* Ri = rots + (size_t)(b * N + i) * 9;
        const float* Rj = rots + (size_t)(b * N + j) * 9;
        float r[9];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c) r[a * 3 + c] = Rj[a * 3] * Ri[c] + Rj[a * 3 + 1] * Ri[3 + c] + Rj[a * 3 + 2] * Ri[6 + c];
        float q[4];
        rot_to_quat_dev(r, codes ? (int)codes[pidx] : 0, q);
        for (int a = 0; a < 4; ++a) f[nbin + a] = q[a] * pm;
        f[nbin + 4] = fs; f[nbin + 5] = fs;
    }
    float* fm = f + nbin + 6;
    {   // motif template: bins of the motif coordinates under the fixed-sequence pair mask, times the fixed-structure mask
        const float mm = (fsm[b * N + i] ? 1.f : 0.f) * (fsm[b * N + j] ? 1.f : 0.f);
        bins(mpos + (size_t)(b * N + i) * 3, mpos + (size_t)(b * N + j) * 3, mm * fs, fm);
        fm[nbin] = fs; fm[nbin + 1] = fs;
    }
    float* fr = fm + nbin + 2;
    {
        const bool same = cidx[b * N + i] == cidx[b * N + j];
        int d = ridx[b * N + i] - ridx[b * N + j] + relk;
        d = d < 0 ? 0 : (d > 2 * relk ? 2 * relk : d);
        const int sel = same ? d : 2 * relk + 1;
        for (int k = 0; k < 2 * relk + 2; ++k) fr[k] = k == sel ? 1.f : 0.f;
        fr[2 * relk + 2] = same ? 1.f : 0.f;
    }
}
void launch_pair_features(hipStream_t st, const float* trans, const float* rots, const int8_t* codes, const float* rmask, const uint8_t* fstm,
                          const uint8_t* fsm, const float* mpos, const int32_t* ridx, const int32_t* cidx, float* F, int B, int N, int nbin,
                          float dmin, float dstep, int relk) {
    const long long P = (long long)B * N * N;
    hipLaunchKernelGGL(k_pair_features, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, trans, rots, codes, rmask, fstm, fsm, mpos, ridx, cidx,
                       F, B, N, nbin, dmin, dstep, relk);
}
// gradient of the template distance bins wrt the (scaled) translations: F_k = pm softmax_k(-4 |d - v_k|), d = sqrt(1e-10 + |x_i - x_j|^2)
// (pair_feature_net.py:223-269, geo_utils.py:4-19).  dF: [P][ldf], the first nbin columns.  The quaternion features depend on the
// frames only, which the guidance samplers detach (unconditional_smc.py:466, 570-576).
__global__ __launch_bounds__(256) void k_pair_features_bwd(const float* __restrict__ dF, int ldf, const float* __restrict__ trans,
                                                           const float* __restrict__ rmask, float* __restrict__ dtr, int B, int N, int nbin,
                                                           float dmin, float dstep) {
    const long long pidx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pidx >= (long long)B * N * N) return;
    const int j = (int)(pidx % N), i = (int)((pidx / N) % N), b = (int)(pidx / ((long long)N * N));
    const float pm = rmask[b * N + i] * rmask[b * N + j];
    if (pm == 0.f || i == j) return;
    const float* xi = trans + (size_t)(b * N + i) * 3;
    const float* xj = trans + (size_t)(b * N + j) * 3;
    const float dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
    const float d = sqrtf(1e-10f + dx * dx + dy * dy + dz * dz);
    float mx = -3.0e38f;
    for (int k = 0; k < nbin; ++k) mx = fmaxf(mx, -4.0f * fabsf(d - (dmin + (float)k * dstep)));
    float s = 0.f, sg = 0.f, sfg = 0.f, sf = 0.f;
    const float* g = dF + pidx * ldf;
    for (int k = 0; k < nbin; ++k) {
        const float v = dmin + (float)k * dstep;
        const float e = expf(-4.0f * fabsf(d - v) - mx);
        const float dl = d > v ? -4.0f : (d < v ? 4.0f : 0.f);      // d logit_k / d d
        s += e; sg += e * dl; sfg += e * g[k] * dl; sf += e * g[k];
    }
    // y_k = e_k / s;  dL/dd = pm sum_k g_k y_k (dl_k - sum_m y_m dl_m)
    const float dd = pm * (sfg / s - (sf / s) * (sg / s));
    const float c = dd / d;
    atomicAdd(dtr + (size_t)(b * N + i) * 3, c * dx); atomicAdd(dtr + (size_t)(b * N + i) * 3 + 1, c * dy); atomicAdd(dtr + (size_t)(b * N + i) * 3 + 2, c * dz);
    atomicAdd(dtr + (size_t)(b * N + j) * 3, -c * dx); atomicAdd(dtr + (size_t)(b * N + j) * 3 + 1, -c * dy); atomicAdd(dtr + (size_t)(b * N + j) * 3 + 2, -c * dz);
}
void launch_pair_features_bwd(hipStream_t st, const float* dF, int ldf, const float* trans, const float* rmask, float* dtr, int B, int N, int nbin,
                              float dmin, float dstep) {
    const long long P = (long long)B * N * N;
    hipLaunchKernelGGL(k_pair_features_bwd, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, dF, ldf, trans, rmask, dtr, B, N, nbin, dmin, dstep);
}

// p[b,i,j,:] = (p[b,i,j,:] + pi[b,i,:] + pj[b,j,:]) * mask   and its transpose: dpi[b,i,:] = sum_j dp, dpj[b,j,:] = sum_i dp
__global__ __launch_bounds__(128) void k_pair_sum_bwd(const float* __restrict__ dp, float* __restrict__ dpi, float* __restrict__ dpj, int N, int C) {
    const int bi = blockIdx.x;                  // (b, i): row sum over j -> dpi[b, i]; and column role: (b, j = i) sum over rows
    const int b = bi / N, i = bi % N;
    for (int c = threadIdx.x; c < C; c += 128) {
        float s = 0.f, t = 0.f;
        for (int j = 0; j < N; ++j) {
            s += dp[((size_t)bi * N + j) * C + c];
            t += dp[(((size_t)b * N + j) * N + i) * C + c];
        }
        dpi[(size_t)bi * C + c] = s;
        dpj[(size_t)bi * C + c] = t;
    }
}
void launch_pair_sum_bwd(hipStream_t st, const float* dp, float* dpi, float* dpj, int B, int N, int C) {
    hipLaunchKernelGGL(k_pair_sum_bwd, dim3(B * N), dim3(128), 0, st, dp, dpi, dpj, N, C);
}

// ------------------------------------------------------------------------------------------------ invariant point attention
// (modules/invariant_point_attention.py:100-260).  Row-major inputs straight from the projection GEMMs:
//   q [M][H C], kv [M][H][2C] (k, then v), global-frame points qp [M][H][Pq][3], kp [M][H][Pq][3], vp [M][H][Pv][3],
//   bias [B N N][H] = linear_b(p), p [B N N][cp].  One work-group per query (b, i).
struct IpaDims { int B, N, H, C, Pq, Pv, cp; float c_qk, c_b; int skip; };     // skip: developer knock-out mask (GENIE_IPA_SKIP), 0 in use
__device__ __forceinline__ float softplus_dev(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// The IPA training kernels run one block per (b, i) query or key row -- B N blocks, two per CU at N = 256, batch 2 -- and are bound by
// the latency of their loops: IPA_NT threads per block decide how many waves a CU has to hide it with.
#ifndef IPA_NT
#define IPA_NT 1024
#endif
template <int HT>      // head count at compile time (0: d.H) -- the per-head loops are unrolled over registers
__global__ __launch_bounds__(IPA_NT) void k_ipa_fwd(IpaDims d, const float* __restrict__ q, const float* __restrict__ kv, const float* __restrict__ qp,
                                                 const float* __restrict__ kp, const float* __restrict__ vp, const float* __restrict__ bias,
                                                 const float* __restrict__ p, const float* __restrict__ rots, const float* __restrict__ trans,
                                                 const float* __restrict__ rmask, const float* __restrict__ head_w, float* __restrict__ att_out,
                                                 float* __restrict__ cat) {
    extern __shared__ float sm[];
    const int N = d.N, H = HT > 0 ? HT : d.H, C = d.C, Pq = d.Pq, Pv = d.Pv, cp = d.cp;
    float* att = sm;                        // [H][N]
    float* sq = att + H * N;                // [H C]
    float* sqp = sq + H * C;                // [H Pq 3]
    float* sopt = sqp + H * Pq * 3;         // [H Pv 3] global-frame output points
    const int bi = blockIdx.x, b = bi / N, i = bi % N, tid = threadIdx.x;
    for (int u = tid; u < H * C; u += IPA_NT) sq[u] = q[(size_t)bi * H * C + u];
    for (int u = tid; u < H * Pq * 3; u += IPA_NT) sqp[u] = qp[(size_t)bi * H * Pq * 3 + u];
    __syncthreads();
    const float mi = rmask[bi];
    const float cpt = sqrtf(1.0f / (3.0f * ((float)Pq * 9.0f / 2.0f)));
    for (int u = tid; u < H * N; u += IPA_NT) {
        const int h = u / N, j = u % N;
        const size_t mj = (size_t)b * N + j;
        const float* kr = kv + (mj * H + h) * 2 * C;
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) a += sq[h * C + c] * kr[c];
        a = a * d.c_qk + d.c_b * bias[((size_t)bi * N + j) * H + h];
        float ds = 0.f;
        const float* kpr = kp + (mj * H + h) * Pq * 3;
#pragma unroll 4
        for (int t = 0; t < Pq * 3; ++t) { const float df = sqp[h * Pq * 3 + t] - kpr[t]; ds += df * df; }
        a += -0.5f * softplus_dev(head_w[h]) * cpt * ds + 1e5f * (mi * rmask[mj] - 1.0f);
        att[u] = a;
    }
    __syncthreads();
    {   // softmax over j, one wave per head at a time
        const int wave = tid >> 6, lane = tid & 63;
        for (int h = wave; h < H; h += IPA_NT / 64) {
            float mx = -3.0e38f;
            for (int j = lane; j < N; j += 64) mx = fmaxf(mx, att[h * N + j]);
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float s = 0.f;
            for (int j = lane; j < N; j += 64) { const float e = expf(att[h * N + j] - mx); att[h * N + j] = e; s += e; }
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float inv = 1.0f / s;
            for (int j = lane; j < N; j += 64) { const float a = att[h * N + j] * inv; att[h * N + j] = a; att_out[(((size_t)b * H + h) * N + i) * N + j] = a; }
        }
    }
    __syncthreads();
    const int ncat = H * (C + 4 * Pv + cp);
    float* crow = cat + (size_t)bi * ncat;
    for (int u = tid; u < H * C; u += IPA_NT) {          // o
        const int h = u / C, c = u % C;
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < N; ++j) s += att[h * N + j] * kv[(((size_t)b * N + j) * H + h) * 2 * C + C + c];
        crow[u] = s;
    }
    for (int u = tid; u < H * Pv * 3; u += IPA_NT) {     // o_pt, global frame
        const int h = u / (Pv * 3), t = u % (Pv * 3);
        float s = 0.f;
#pragma unroll 8
        for (int j = 0; j < N; ++j) s += att[h * N + j] * vp[(((size_t)b * N + j) * H + h) * Pv * 3 + t];
        sopt[u] = s;
    }
    if (IPA_NT % cp == 0 && H <= 16) {                // o_pair: a thread keeps its channel and walks every (IPA_NT / cp)-th j -- each pair
        const int c = tid % cp, part = tid / cp, nparts = IPA_NT / cp;    // row is read once, for all heads; the parts are then summed
        float* sbuf = sopt + H * Pv * 3;              // pairwise through LDS ([nparts / 2][H][cp]; a fixed order: no float atomics)
        float acc[16];
#pragma unroll
        for (int h = 0; h < 16; ++h) acc[h] = 0.f;
#pragma unroll 8
        for (int j = part; j < N; j += nparts) {
            const float pv = p[((size_t)bi * N + j) * cp + c];
#pragma unroll
            for (int h = 0; h < 16; ++h) if (h < H) acc[h] += att[h * N + j] * pv;
        }
        for (int half = nparts >> 1; half >= 1; half >>= 1) {
            if (part >= half && part < 2 * half) {
#pragma unroll
                for (int h = 0; h < 16; ++h) if (h < H) sbuf[((part - half) * H + h) * cp + c] = acc[h];
            }
            __syncthreads();
            if (part < half) {
#pragma unroll
                for (int h = 0; h < 16; ++h) if (h < H) acc[h] += sbuf[(part * H + h) * cp + c];
            }
            __syncthreads();
        }
        if (part == 0) {
#pragma unroll
            for (int h = 0; h < 16; ++h) if (h < H) crow[H * C + 4 * H * Pv + h * cp + c] = acc[h];
        }
    } else {
        for (int u = tid; u < H * cp; u += IPA_NT) {
            const int h = u / cp, c = u % cp;
            float s = 0.f;
#pragma unroll 8
            for (int j = 0; j < N; ++j) s += att[h * N + j] * p[((size_t)bi * N + j) * cp + c];
            crow[H * C + 4 * H * Pv + u] = s;
        }
    }
    __syncthreads();
    const float* R = rots + (size_t)bi * 9;
    const float* T = trans + (size_t)bi * 3;
    for (int u = tid; u < H * Pv; u += IPA_NT) {         // local frame: R^T (g - t), and its norm
        const float w0 = sopt[u * 3] - T[0], w1 = sopt[u * 3 + 1] - T[1], w2 = sopt[u * 3 + 2] - T[2];
        const float l0 = R[0] * w0 + R[3] * w1 + R[6] * w2, l1 = R[1] * w0 + R[4] * w1 + R[7] * w2, l2 = R[2] * w0 + R[5] * w1 + R[8] * w2;
        crow[H * C + u] = l0; crow[H * C + H * Pv + u] = l1; crow[H * C + 2 * H * Pv + u] = l2;
        crow[H * C + 3 * H * Pv + u] = sqrtf(l0 * l0 + l1 * l1 + l2 * l2 + 1e-8f);
    }
}

// backward, per query (b, i): d logits (kept for the per-key kernel and the linear_b gradient), dq, dq_pts (global), the pair
// gradient rows (b, i, :, :), d head_weights, the query's own frame gradient from the output points
template <int HT>      // head count at compile time (0: d.H) -- the per-head loops are unrolled over registers
__global__ __launch_bounds__(IPA_NT) void k_ipa_bwd_q(IpaDims d, const float* __restrict__ q, const float* __restrict__ kv, const float* __restrict__ qp,
                                                   const float* __restrict__ kp, const float* __restrict__ vp, const float* __restrict__ p,
                                                   const float* __restrict__ rots, const float* __restrict__ trans, const float* __restrict__ head_w,
                                                   const float* __restrict__ wb, const float* __restrict__ att_in, const float* __restrict__ cat,
                                                   const float* __restrict__ dcat, float* __restrict__ dlg, float* __restrict__ dq, float* __restrict__ dqp,
                                                   float* __restrict__ doptg, float* __restrict__ dP, float* __restrict__ dhead, float* __restrict__ dbb,
                                                   float* __restrict__ dR, float* __restrict__ dT) {
    extern __shared__ float sm[];
    const int N = d.N, H = HT > 0 ? HT : d.H, C = d.C, Pq = d.Pq, Pv = d.Pv, cp = d.cp;
    float* att = sm;                        // [H][N]
    float* dat = att + H * N;               // [H][N] d att -> d logits
    float* sdo = dat + H * N;               // [H C]
    float* sdg = sdo + H * C;               // [H Pv 3] d o_pt (global)
    float* sdop = sdg + H * Pv * 3;         // [H cp] d o_pair
    float* sqp = sdop + H * cp;             // [H Pq 3]
    float* sq = sqp + H * Pq * 3;           // [H C]
    float* red = sq + H * C;                // [16] scratch: frame gradient (9 + 3)
    const int bi = blockIdx.x, b = bi / N, i = bi % N, tid = threadIdx.x;
    const int ncat = H * (C + 4 * Pv + cp);
    const float* crow = cat + (size_t)bi * ncat;
    const float* drow = dcat + (size_t)bi * ncat;
    const float* R = rots + (size_t)bi * 9;
    if (tid < 16) red[tid] = 0.f;
    for (int u = tid; u < H * N; u += IPA_NT) { const int h = u / N, j = u % N; att[u] = att_in[(((size_t)b * H + h) * N + i) * N + j]; }
    for (int u = tid; u < H * C; u += IPA_NT) { sdo[u] = drow[u]; sq[u] = q[(size_t)bi * H * C + u]; }
    for (int u = tid; u < H * cp; u += IPA_NT) sdop[u] = drow[H * C + 4 * H * Pv + u];
    for (int u = tid; u < H * Pq * 3; u += IPA_NT) sqp[u] = qp[(size_t)bi * H * Pq * 3 + u];
    __syncthreads();
    for (int u = tid; u < H * Pv; u += IPA_NT) {         // through norm and local frame: l = R^T w, w = g - t
        const float l0 = crow[H * C + u], l1 = crow[H * C + H * Pv + u], l2 = crow[H * C + 2 * H * Pv + u], nr = crow[H * C + 3 * H * Pv + u];
        const float dn = drow[H * C + 3 * H * Pv + u] / nr;
        const float g0 = drow[H * C + u] + dn * l0, g1 = drow[H * C + H * Pv + u] + dn * l1, g2 = drow[H * C + 2 * H * Pv + u] + dn * l2;
        // d w = R d l
        const float w0 = R[0] * g0 + R[1] * g1 + R[2] * g2, w1 = R[3] * g0 + R[4] * g1 + R[5] * g2, w2 = R[6] * g0 + R[7] * g1 + R[8] * g2;
        sdg[u * 3] = w0; sdg[u * 3 + 1] = w1; sdg[u * 3 + 2] = w2;
        doptg[((size_t)bi * H * Pv + u) * 3] = w0; doptg[((size_t)bi * H * Pv + u) * 3 + 1] = w1; doptg[((size_t)bi * H * Pv + u) * 3 + 2] = w2;
        // dR[a][c] += w_a dl_c with w = R l;  dt -= d w
        const float x0 = R[0] * l0 + R[1] * l1 + R[2] * l2, x1 = R[3] * l0 + R[4] * l1 + R[5] * l2, x2 = R[6] * l0 + R[7] * l1 + R[8] * l2;
        atomicAdd(red + 0, x0 * g0); atomicAdd(red + 1, x0 * g1); atomicAdd(red + 2, x0 * g2);
        atomicAdd(red + 3, x1 * g0); atomicAdd(red + 4, x1 * g1); atomicAdd(red + 5, x1 * g2);
        atomicAdd(red + 6, x2 * g0); atomicAdd(red + 7, x2 * g1); atomicAdd(red + 8, x2 * g2);
        atomicAdd(red + 9, -w0); atomicAdd(red + 10, -w1); atomicAdd(red + 11, -w2);
    }
    __syncthreads();
    if (tid < 9) dR[(size_t)bi * 9 + tid] += red[tid];
    else if (tid < 12) dT[(size_t)bi * 3 + tid - 9] += red[tid];
    if (!(d.skip & 1)) {   // d att[h][j] = <d o[h], v_j> + <d o_pt[h], v_pt_j> + <d o_pair[h], p[b,i,j,:]>.  The pair rows come through LDS in tiles
        // of 64 (coalesced; a lane reading its own 512-B row channel by channel touches 64 lines per instruction); thread (j, wave)
        // accumulates heads wave, wave + 4, ... so that one LDS read of p feeds every head of the thread
        float* pt = red + 16;               // [64][cp + 1]
        const int jj = tid & 63, hg = tid >> 6;
        // cp = 128, at most 16 heads: the pair term <d o_pair[h], p[b,i,j,:]> on the matrix pipe -- per wave 16 keys at a time, A = d o_pair
        // (heads as rows, padded to 16), B = the keys' pair rows straight from global memory (a lane's two float4 are the 8 channels of
        // its B fragment), both split into three bf16 pieces (six v_mfma_f32_16x16x32_bf16 per product: f32-grade like the GEMMs);
        // the result lands in dat, the v / v_pt terms are added below.  (As a per-thread loop over the 128 channels out of an LDS tile
        // this term was the kernel's largest phase.)
        const bool pair_mfma = cp == 128 && H <= 16;
        if (pair_mfma) {
            const int lane = tid & 63, wave = tid >> 6, m = lane & 15, g = lane >> 4;
            auto split3 = [](const float (&x)[8], bf16x8& h, bf16x8& mm, bf16x8& l) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    h[e] = (__bf16)x[e];
                    const float r1 = x[e] - (float)h[e];
                    mm[e] = (__bf16)r1;
                    l[e] = (__bf16)(r1 - (float)mm[e]);
                }
            };
            for (int j0 = wave * 16; j0 < N; j0 += (IPA_NT / 64) * 16) {
                const int j = min(j0 + m, N - 1);
                const float* prow = p + ((size_t)bi * N + j) * cp + 8 * g;
                float4 x0[4], x1[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) { x0[ks] = *reinterpret_cast<const float4*>(prow + ks * 32); x1[ks] = *reinterpret_cast<const float4*>(prow + ks * 32 + 4); }
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    float xa[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) xa[e] = m < H ? sdop[m * cp + ks * 32 + 8 * g + e] : 0.f;
                    const float xb[8] = {x0[ks].x, x0[ks].y, x0[ks].z, x0[ks].w, x1[ks].x, x1[ks].y, x1[ks].z, x1[ks].w};
                    bf16x8 ah, am, al, bh, bm, bl;
                    split3(xa, ah, am, al);
                    split3(xb, bh, bm, bl);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
                }
                if (j0 + m < N) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (4 * g + r < H) dat[(4 * g + r) * N + j0 + m] = acc[r];        // D row 4 g + r = head, column m = key
                }
            }
            __syncthreads();
        }
        for (int j0 = 0; j0 < N; j0 += 64) {
            if (!pair_mfma) {
#pragma unroll 8
                for (int u = tid; u < 64 * cp; u += IPA_NT) {
                    const int r = u / cp, c = u - r * cp;
                    pt[r * (cp + 1) + c] = j0 + r < N ? p[((size_t)bi * N + j0 + r) * cp + c] : 0.f;
                }
                __syncthreads();
            }
            const int j = j0 + jj;
            if (j < N) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                const float* prow = pt + jj * (cp + 1);
                if (pair_mfma) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) if (hg + (IPA_NT / 64) * t < H) acc[t] = dat[(hg + (IPA_NT / 64) * t) * N + j];
                } else {
#pragma unroll 8
                    for (int c = 0; c < cp; ++c) {
                        const float pv = prow[c];
#pragma unroll
                        for (int t = 0; t < 4; ++t) if (hg + (IPA_NT / 64) * t < H) acc[t] += sdop[(hg + (IPA_NT / 64) * t) * cp + c] * pv;
                    }
                }
                const size_t mj = (size_t)b * N + j;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int h = hg + (IPA_NT / 64) * t;
                    if (h >= H) continue;
                    float sacc = acc[t];
                    const float* vr = kv + (mj * H + h) * 2 * C + C;
                    const float* vpr = vp + (mj * H + h) * Pv * 3;
                    if (C % 4 == 0 && (Pv * 3) % 4 == 0) {      // a lane owns these rows: 16-byte loads instead of one round trip per float
#pragma unroll 4
                        for (int c = 0; c < C; c += 4) {
                            const float4 x = *reinterpret_cast<const float4*>(vr + c);
                            sacc += (sdo[h * C + c] * x.x + sdo[h * C + c + 1] * x.y) + (sdo[h * C + c + 2] * x.z + sdo[h * C + c + 3] * x.w);
                        }
#pragma unroll 6
                        for (int q3 = 0; q3 < Pv * 3; q3 += 4) {
                            const float4 x = *reinterpret_cast<const float4*>(vpr + q3);
                            sacc += (sdg[h * Pv * 3 + q3] * x.x + sdg[h * Pv * 3 + q3 + 1] * x.y) + (sdg[h * Pv * 3 + q3 + 2] * x.z + sdg[h * Pv * 3 + q3 + 3] * x.w);
                        }
                    } else {
#pragma unroll 8
                        for (int c = 0; c < C; ++c) sacc += sdo[h * C + c] * vr[c];
#pragma unroll 4
                        for (int q3 = 0; q3 < Pv * 3; ++q3) sacc += sdg[h * Pv * 3 + q3] * vpr[q3];
                    }
                    dat[h * N + j] = sacc;
                }
            }
            if (!pair_mfma) __syncthreads();
        }
        if (pair_mfma) __syncthreads();
    }
    const float cpt = sqrtf(1.0f / (3.0f * ((float)Pq * 9.0f / 2.0f)));
    if (!(d.skip & 2)) {   // d logits = att (d att - sum_j att d att); d head_weights, d bias of linear_b
        const int wave = tid >> 6, lane = tid & 63;
        for (int h = wave; h < H; h += IPA_NT / 64) {
            float s = 0.f;
            for (int j = lane; j < N; j += 64) s += att[h * N + j] * dat[h * N + j];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            float sb = 0.f, sh = 0.f;
            for (int j = lane; j < N; j += 64) {
                const float g = att[h * N + j] * (dat[h * N + j] - s);
                dat[h * N + j] = g;
                dlg[(((size_t)b * H + h) * N + i) * N + j] = g;
                sb += g;
                float ds = 0.f;
                const float* kpr = kp + (((size_t)b * N + j) * H + h) * Pq * 3;
                if ((Pq * 3) % 4 == 0) {
#pragma unroll 3
                    for (int t = 0; t < Pq * 3; t += 4) {
                        const float4 x = *reinterpret_cast<const float4*>(kpr + t);
                        const float d0 = sqp[h * Pq * 3 + t] - x.x, d1 = sqp[h * Pq * 3 + t + 1] - x.y, d2 = sqp[h * Pq * 3 + t + 2] - x.z, d3 = sqp[h * Pq * 3 + t + 3] - x.w;
                        ds += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                    }
                } else {
#pragma unroll 4
                    for (int t = 0; t < Pq * 3; ++t) { const float df = sqp[h * Pq * 3 + t] - kpr[t]; ds += df * df; }
                }
                sh += g * ds;
            }
            for (int o = 32; o > 0; o >>= 1) { sb += __shfl_xor(sb, o); sh += __shfl_xor(sh, o); }
            if (lane == 0) {
                atomicAdd(dbb + h, d.c_b * sb);
                const float hw = head_w[h];
                atomicAdd(dhead + h, -0.5f * cpt * sh / (1.0f + expf(-hw)));          // d softplus = sigmoid
            }
        }
    }
    __syncthreads();
    {   // dq and dq_pts (global): sums over the N keys of d logit x (k_j | q_pt - k_pt_j).  One thread per (output, third of the key
        // range) -- every output alone on a thread walked all N keys as a chain of L2 round trips with 336 of the 1024 threads busy --
        // partial sums meet in LDS (the pair-row tile of the first phase is free by now)
        const int nq = H * C, np = H * Pq * 3, nout = nq + np;
        const int G = IPA_NT / nout;                                   // key-range groups (3 for the released models)
        float* part = red + 16;                                        // [G][nout]
        const int o = tid % nout, g = tid / nout;
        if (G == 0) {           // more outputs than threads (not a released shape): every output walks all keys
            for (int u = tid; u < ((d.skip & 4) ? 0 : nq); u += IPA_NT) {
                const int h = u / C, c = u % C;
                float sacc = 0.f;
                for (int j = 0; j < N; ++j) sacc += dat[h * N + j] * kv[(((size_t)b * N + j) * H + h) * 2 * C + c];
                dq[(size_t)bi * nq + u] = sacc * d.c_qk;
            }
            for (int u = tid; u < ((d.skip & 8) ? 0 : np); u += IPA_NT) {
                const int h = u / (Pq * 3), t = u % (Pq * 3);
                float sacc = 0.f;
                for (int j = 0; j < N; ++j) sacc += dat[h * N + j] * (sqp[u] - kp[(((size_t)b * N + j) * H + h) * Pq * 3 + t]);
                dqp[(size_t)bi * np + u] = -softplus_dev(head_w[h]) * cpt * sacc;
            }
        } else if (g < G && !(o < nq ? (d.skip & 4) : (d.skip & 8))) {
            const int jb = (int)((long long)N * g / G), je = (int)((long long)N * (g + 1) / G);
            float sacc = 0.f;
            if (o < nq) {
                const int h = o / C, c = o % C;
#pragma unroll 8
                for (int j = jb; j < je; ++j) sacc += dat[h * N + j] * kv[(((size_t)b * N + j) * H + h) * 2 * C + c];
            } else {
                const int u = o - nq, h = u / (Pq * 3), t = u % (Pq * 3);
                const float qv = sqp[u];
#pragma unroll 8
                for (int j = jb; j < je; ++j) sacc += dat[h * N + j] * (qv - kp[(((size_t)b * N + j) * H + h) * Pq * 3 + t]);
            }
            part[g * nout + o] = sacc;
        }
        __syncthreads();
        if (G > 0 && tid < nout && !(tid < nq ? (d.skip & 4) : (d.skip & 8))) {
            float sacc = 0.f;
            for (int gg = 0; gg < G; ++gg) sacc += part[gg * nout + tid];
            if (tid < nq) dq[(size_t)bi * nq + tid] = sacc * d.c_qk;
            else {
                const int u = tid - nq, h = u / (Pq * 3);
                dqp[(size_t)bi * np + u] = -softplus_dev(head_w[h]) * cpt * sacc;
            }
        }
        __syncthreads();
    }
    // pair gradient rows (b, i, j, :) += sum_h att do_pair + c_b dlogit W_b
    if (d.skip & 16) return;
    if (IPA_NT % cp == 0 && H <= 16) {         // a thread keeps its channel: its column of d o_pair and W_b stays in registers
        const int c = tid % cp;
        float so[16], sw[16];
#pragma unroll
        for (int h = 0; h < 16; ++h) { so[h] = h < H ? sdop[h * cp + c] : 0.f; sw[h] = h < H ? d.c_b * wb[h * cp + c] : 0.f; }
#pragma unroll 8
        for (int j = tid / cp; j < N; j += IPA_NT / cp) {
            float sacc = 0.f;
#pragma unroll
            for (int h = 0; h < 16; ++h) if (h < H) sacc += att[h * N + j] * so[h] + dat[h * N + j] * sw[h];
            dP[((size_t)bi * N + j) * cp + c] += sacc;
        }
    } else {
        for (int c = tid; c < cp; c += IPA_NT) {
#pragma unroll 8
            for (int j = 0; j < N; ++j) {
                float sacc = 0.f;
                for (int h = 0; h < H; ++h) sacc += att[h * N + j] * sdop[h * cp + c] + d.c_b * dat[h * N + j] * wb[h * cp + c];
                dP[((size_t)bi * N + j) * cp + c] += sacc;
            }
        }
    }
}
// backward, per key (b, j): dk, dv (into dkv), dk_pts, dv_pts (global)
template <int HT>      // head count at compile time (0: d.H) -- the per-head loops are unrolled over registers
__global__ __launch_bounds__(IPA_NT) void k_ipa_bwd_k(IpaDims d, const float* __restrict__ q, const float* __restrict__ qp, const float* __restrict__ kp,
                                                   const float* __restrict__ head_w, const float* __restrict__ att_in, const float* __restrict__ dlg,
                                                   const float* __restrict__ dcat, const float* __restrict__ doptg, float* __restrict__ dkv,
                                                   float* __restrict__ dkp, float* __restrict__ dvp) {
    extern __shared__ float sm[];
    const int N = d.N, H = HT > 0 ? HT : d.H, C = d.C, Pq = d.Pq, Pv = d.Pv, cp = d.cp;
    float* att = sm;                // [H][N] over queries i
    float* dl = att + H * N;        // [H][N]
    const int bj = blockIdx.x, b = bj / N, j = bj % N, tid = threadIdx.x;
    const int ncat = H * (C + 4 * Pv + cp);
    for (int u = tid; u < H * N; u += IPA_NT) {
        const int h = u / N, i = u % N;
        att[u] = att_in[(((size_t)b * H + h) * N + i) * N + j];
        dl[u] = dlg[(((size_t)b * H + h) * N + i) * N + j];
    }
    __syncthreads();
    const float cpt = sqrtf(1.0f / (3.0f * ((float)Pq * 9.0f / 2.0f)));
    for (int u = tid; u < H * C; u += IPA_NT) {
        const int h = u / C, c = u % C;
        float sk = 0.f, sv = 0.f;
        for (int i = 0; i < N; ++i) {
            const size_t mi = (size_t)b * N + i;
            sk += dl[h * N + i] * q[mi * H * C + u];
            sv += att[h * N + i] * dcat[mi * ncat + u];
        }
        dkv[((size_t)bj * H + h) * 2 * C + c] = sk * d.c_qk;
        dkv[((size_t)bj * H + h) * 2 * C + C + c] = sv;
    }
    for (int u = tid; u < H * Pq * 3; u += IPA_NT) {
        const int h = u / (Pq * 3);
        float s = 0.f;
        const float kpv = kp[(size_t)bj * H * Pq * 3 + u];
        for (int i = 0; i < N; ++i) s += dl[h * N + i] * (qp[((size_t)b * N + i) * H * Pq * 3 + u] - kpv);
        dkp[(size_t)bj * H * Pq * 3 + u] = softplus_dev(head_w[h]) * cpt * s;
    }
    for (int u = tid; u < H * Pv * 3; u += IPA_NT) {
        const int h = u / (Pv * 3);
        float s = 0.f;
        for (int i = 0; i < N; ++i) s += att[h * N + i] * doptg[((size_t)b * N + i) * H * Pv * 3 + u];
        dvp[(size_t)bj * H * Pv * 3 + u] = s;
    }
}
// developer knock-out of sections of the IPA kernels (wrong gradients by design): only in -DGENIE_DEV builds (GENIE_EXTRA_FLAGS)
#ifdef GENIE_DEV
static int ipa_skip() { static const int v = getenv("GENIE_IPA_SKIP") ? atoi(getenv("GENIE_IPA_SKIP")) : 0; return v; }
#else
static int ipa_skip() { return 0; }
#endif
size_t ipa_train_lds(int N, int H, int C, int Pq, int Pv, int cp) {
    return (size_t)(2 * H * N + 2 * H * C + H * Pv * 3 + H * cp + H * Pq * 3 + 16) * sizeof(float);
}
template <int HT>
static void launch_ipa_fwd_t(hipStream_t st, const IpaArgs& a, const IpaDims& d, size_t lds) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ipa_fwd<HT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_ipa_fwd<HT>, dim3(a.B * a.N), dim3(IPA_NT), lds, st, d, a.q, a.kv, a.qp, a.kp, a.vp, a.bias, a.p, a.rots, a.trans, a.rmask,
                       a.head_w, a.att, a.cat);
}
void launch_ipa_fwd(hipStream_t st, const IpaArgs& a) {
    IpaDims d{a.B, a.N, a.H, a.C, a.Pq, a.Pv, a.cp, sqrtf(1.0f / (3.0f * a.C)), sqrtf(1.0f / 3.0f), ipa_skip()};
    // the forward kernel lays out att [H N], q, q points, output points, then the o_pair exchange buffer [IPA_NT / cp / 2][H][cp]
    const size_t lds = (size_t)(a.H * a.N + a.H * a.C + a.H * a.Pq * 3 + a.H * a.Pv * 3 + (IPA_NT % a.cp == 0 ? (IPA_NT / a.cp / 2) * a.H * a.cp : 0)) * sizeof(float);
    if (a.H == 12) launch_ipa_fwd_t<12>(st, a, d, lds);     // the released models' head count
    else launch_ipa_fwd_t<0>(st, a, d, lds);
}
template <int HT>
static void launch_ipa_bwd_t(hipStream_t st, const IpaArgs& a, const IpaDims& d, size_t lds, size_t lds_q) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ipa_bwd_q<HT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ipa_bwd_k<HT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_ipa_bwd_q<HT>, dim3(a.B * a.N), dim3(IPA_NT), lds_q, st, d, a.q, a.kv, a.qp, a.kp, a.vp, a.p, a.rots, a.trans, a.head_w, a.wb,
                       a.att, a.cat, a.dcat, a.dlg, a.dq, a.dqp, a.doptg, a.dP, a.dhead, a.dbb, a.dR, a.dT);
    hipLaunchKernelGGL(k_ipa_bwd_k<HT>, dim3(a.B * a.N), dim3(IPA_NT), lds, st, d, a.q, a.qp, a.kp, a.head_w, a.att, a.dlg, a.dcat, a.doptg, a.dkv,
                       a.dkp, a.dvp);
}
void launch_ipa_bwd(hipStream_t st, const IpaArgs& a) {
    IpaDims d{a.B, a.N, a.H, a.C, a.Pq, a.Pv, a.cp, sqrtf(1.0f / (3.0f * a.C)), sqrtf(1.0f / 3.0f), ipa_skip()};
    const size_t lds = ipa_train_lds(a.N, a.H, a.C, a.Pq, a.Pv, a.cp);
    const size_t lds_q = lds + (size_t)64 * (a.cp + 1) * sizeof(float);            // + the staged tile of pair rows
    if (a.H == 12) launch_ipa_bwd_t<12>(st, a, d, lds, lds_q);
    else launch_ipa_bwd_t<0>(st, a, d, lds, lds_q);
}

// ------------------------------------------------------------------------------------------------ points and frames
// linear output [M][3 H P] = [x block | y block | z block]  <->  global-frame points [M][H][P][3] = R l + t
// (invariant_point_attention.py:137-174); `kvsplit`: the kv points' (Pq + Pv) are split into k points and v points
__global__ void k_points_fwd(const float* __restrict__ lin, const float* __restrict__ rots, const float* __restrict__ trans, float* __restrict__ out0,
                             float* __restrict__ out1, int M, int H, int P0, int P1) {
    const int PT = P0 + P1;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)M * H * PT) return;
    const int pt = (int)(idx % PT), h = (int)((idx / PT) % H);
    const long long m = idx / ((long long)PT * H);
    const float* l = lin + m * 3 * H * PT + h * PT + pt;
    const float l0 = l[0], l1 = l[H * PT], l2 = l[2 * H * PT];
    const float* R = rots + m * 9;
    const float* T = trans + m * 3;
    const float g0 = R[0] * l0 + R[1] * l1 + R[2] * l2 + T[0], g1 = R[3] * l0 + R[4] * l1 + R[5] * l2 + T[1], g2 = R[6] * l0 + R[7] * l1 + R[8] * l2 + T[2];
    float* o = pt < P0 ? out0 + ((m * H + h) * P0 + pt) * 3 : out1 + ((m * H + h) * P1 + (pt - P0)) * 3;
    o[0] = g0; o[1] = g1; o[2] = g2;
}
// d lin = R^T d g;  dR[a][c] += dg_a l_c;  dt += dg     (frame gradients through atomics: H P contributions per residue)
// one block per residue: its H (P0 + P1) points are reduced in the block (waves by shuffles, then LDS) and the frame gradient is
// added once -- every point adding to the residue's twelve numbers atomically serialises at the L2
__global__ __launch_bounds__(256) void k_points_bwd(const float* __restrict__ lin, const float* __restrict__ rots, const float* __restrict__ dg0,
                                                    const float* __restrict__ dg1, float* __restrict__ dlin, float* __restrict__ dR,
                                                    float* __restrict__ dT, int M, int H, int P0, int P1) {
    __shared__ float red[4][12];
    const int PT = P0 + P1;
    const long long m = blockIdx.x;
    const float* R = rots + m * 9;
    float a[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) a[t] = 0.f;
    for (int u = threadIdx.x; u < H * PT; u += 256) {
        const int pt = u % PT, h = u / PT;
        const float* g = pt < P0 ? dg0 + ((m * H + h) * P0 + pt) * 3 : dg1 + ((m * H + h) * P1 + (pt - P0)) * 3;
        const float g0 = g[0], g1 = g[1], g2 = g[2];
        const long long lo = m * 3 * H * PT + h * PT + pt;
        const float l0 = lin[lo], l1 = lin[lo + H * PT], l2 = lin[lo + 2 * H * PT];
        dlin[lo] = R[0] * g0 + R[3] * g1 + R[6] * g2;
        dlin[lo + H * PT] = R[1] * g0 + R[4] * g1 + R[7] * g2;
        dlin[lo + 2 * H * PT] = R[2] * g0 + R[5] * g1 + R[8] * g2;
        a[0] += g0 * l0; a[1] += g0 * l1; a[2] += g0 * l2;
        a[3] += g1 * l0; a[4] += g1 * l1; a[5] += g1 * l2;
        a[6] += g2 * l0; a[7] += g2 * l1; a[8] += g2 * l2;
        a[9] += g0; a[10] += g1; a[11] += g2;
    }
#pragma unroll
    for (int t = 0; t < 12; ++t)
        for (int o = 32; o > 0; o >>= 1) a[t] += __shfl_xor(a[t], o);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int t = 0; t < 12; ++t) red[threadIdx.x >> 6][t] = a[t];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (threadIdx.x < 9) dR[m * 9 + threadIdx.x] += v; else dT[m * 3 + threadIdx.x - 9] += v;
    }
}
void launch_points_fwd(hipStream_t st, const float* lin, const float* rots, const float* trans, float* out0, float* out1, int M, int H, int P0, int P1) {
    const long long n = (long long)M * H * (P0 + P1);
    hipLaunchKernelGGL(k_points_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, lin, rots, trans, out0, out1, M, H, P0, P1);
}
void launch_points_bwd(hipStream_t st, const float* lin, const float* rots, const float* dg0, const float* dg1, float* dlin, float* dR, float* dT,
                       int M, int H, int P0, int P1) {
    hipLaunchKernelGGL(k_points_bwd, dim3((unsigned)M), dim3(256), 0, st, lin, rots, dg0, dg1, dlin, dR, dT, M, H, P0, P1);
}

// BackboneUpdate + compose (backbone_update.py:40-66, affine_utils.py:109-116, 299-334):
//   quat = (1, b, c, d) / sqrt(1 + b^2 + c^2 + d^2);  U = rot(quat);  R' = R U;  t' = R t_u + t
__device__ __forceinline__ void quat_rot(const float w, const float x, const float y, const float z, float* U) {
    U[0] = w * w + x * x - y * y - z * z; U[1] = 2 * (x * y - w * z); U[2] = 2 * (x * z + w * y);
    U[3] = 2 * (x * y + w * z); U[4] = w * w - x * x + y * y - z * z; U[5] = 2 * (y * z - w * x);
    U[6] = 2 * (x * z - w * y); U[7] = 2 * (y * z + w * x); U[8] = w * w - x * x - y * y + z * z;
}
__global__ void k_frames_fwd(const float* __restrict__ bb, const float* __restrict__ R, const float* __restrict__ T, float* __restrict__ R2,
                             float* __restrict__ T2, int M) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float* u = bb + (size_t)m * 6;
    const float inv = 1.0f / sqrtf(1.0f + u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    float U[9];
    quat_rot(inv, u[0] * inv, u[1] * inv, u[2] * inv, U);
    const float* r = R + (size_t)m * 9;
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) R2[(size_t)m * 9 + a * 3 + c] = r[a * 3] * U[c] + r[a * 3 + 1] * U[3 + c] + r[a * 3 + 2] * U[6 + c];
    for (int a = 0; a < 3; ++a) T2[(size_t)m * 3 + a] = r[a * 3] * u[3] + r[a * 3 + 1] * u[4] + r[a * 3 + 2] * u[5] + T[(size_t)m * 3 + a];
}
// given dR', dt' (gradients wrt the composed frame): d bb [M][6], and dR += ..., dt += ... of the input frame
__global__ void k_frames_bwd(const float* __restrict__ bb, const float* __restrict__ R, const float* __restrict__ dR2, const float* __restrict__ dT2,
                             float* __restrict__ dbb, float* __restrict__ dR, float* __restrict__ dT, int M) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float* u = bb + (size_t)m * 6;
    const float n2 = 1.0f + u[0] * u[0] + u[1] * u[1] + u[2] * u[2];
    const float inv = 1.0f / sqrtf(n2);
    const float w = inv, x = u[0] * inv, y = u[1] * inv, z = u[2] * inv;
    float U[9];
    quat_rot(w, x, y, z, U);
    const float* r = R + (size_t)m * 9;
    const float* g = dR2 + (size_t)m * 9;
    const float* gt = dT2 + (size_t)m * 3;
    float dU[9];            // dU = R^T dR'
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c) dU[a * 3 + c] = r[a] * g[c] + r[3 + a] * g[3 + c] + r[6 + a] * g[6 + c];
    for (int a = 0; a < 3; ++a) {
        for (int c = 0; c < 3; ++c)      // dR += dR' U^T + dt' (x) t_u
            dR[(size_t)m * 9 + a * 3 + c] += g[a * 3] * U[c * 3] + g[a * 3 + 1] * U[c * 3 + 1] + g[a * 3 + 2] * U[c * 3 + 2] + gt[a] * u[3 + c];
        dT[(size_t)m * 3 + a] += gt[a];
    }
    float* o = dbb + (size_t)m * 6;
    for (int a = 0; a < 3; ++a) o[3 + a] = r[a] * gt[0] + r[3 + a] * gt[1] + r[6 + a] * gt[2];       // d t_u = R^T dt'
    // d (w, x, y, z) of the homogeneous quadratic form
    const float dw = 2 * (w * dU[0] - z * dU[1] + y * dU[2] + z * dU[3] + w * dU[4] - x * dU[5] - y * dU[6] + x * dU[7] + w * dU[8]);
    const float dx = 2 * (x * dU[0] + y * dU[1] + z * dU[2] + y * dU[3] - x * dU[4] - w * dU[5] + z * dU[6] + w * dU[7] - x * dU[8]);
    const float dy = 2 * (-y * dU[0] + x * dU[1] + w * dU[2] + x * dU[3] + y * dU[4] + z * dU[5] - w * dU[6] + z * dU[7] - y * dU[8]);
    const float dz = 2 * (-z * dU[0] - w * dU[1] + x * dU[2] + w * dU[3] - z * dU[4] + y * dU[5] + x * dU[6] + y * dU[7] + z * dU[8]);
    // quat = (1, b, c, d) inv:  d b = inv dx - b inv^3 (dw + b dx + c dy + d dz)
    const float dot = dw + u[0] * dx + u[1] * dy + u[2] * dz;
    const float i3 = inv * inv * inv;
    o[0] = inv * dx - u[0] * i3 * dot;
    o[1] = inv * dy - u[1] * i3 * dot;
    o[2] = inv * dz - u[2] * i3 * dot;
}
void launch_frames_fwd(hipStream_t st, const float* bb, const float* R, const float* T, float* R2, float* T2, int M) {
    hipLaunchKernelGGL(k_frames_fwd, dim3((M + 255) / 256), dim3(256), 0, st, bb, R, T, R2, T2, M);
}
void launch_frames_bwd(hipStream_t st, const float* bb, const float* R, const float* dR2, const float* dT2, float* dbb, float* dR, float* dT, int M) {
    hipLaunchKernelGGL(k_frames_bwd, dim3((M + 255) / 256), dim3(256), 0, st, bb, R, dR2, dT2, dbb, dR, dT, M);
}
