// Pair-track kernels: pair feature net, the triangle-multiplication contraction, IPA pair
// bias (the row-tile GEMM kernels -- TriMul projections / output, pair transition -- live in
// pair_wl_kernels.hip).  All GEMMs are exact-fp32 MFMA (v_mfma_f32_32x32x2_f32); see common.h
// for the fragment convention.  Reference lines are cited per kernel.
//
// TriMul data flow (modules/triangular_multiplicative_update.py:57-110):
//   k_trimul_proj_wl : zn = LN_in(z); a = (W_ap zn + b) sigmoid(W_ag zn + b) mask; b likewise.
//       Output channel-major a_cm[b][c][line][pos] (pos contiguous) so the contraction is ONE
//       "NT" batched GEMM for both directions:
//         outgoing: tile = row i of z,    line = i, pos = k = j   (a[i,k,c])
//         incoming: tile = column j of z, line = j, pos = k = i   (a[k,i,c] stored as aT[i][k])
//       MFMA orientation D rows = channels, D cols = pairs -> every store is a 128-B run.
//   k_trimul_contract: x_cm[bc][i][j] = sum_k a_cm[bc][i][k] b_cm[bc][j][k]
//   k_trimul_out_wl  : z += (W_z LN_out(x) + b_z) sigmoid(W_g LN_in(z) + b_g)
#include "common.h"

#define LDZ 132   // 128-channel tile row stride (floats): conflict-free ds_read_b128


// Load 64 pair rows x 128 channels (row t at src + t*row_stride) into tile[64][LDZ].
template <bool AMAX = false>
__device__ __forceinline__ float load_tile64(float* tile, const float* __restrict__ src, size_t row_stride,
                                             int nvalid, int tid) {
    // All eight row loads are issued before the first LDS write (unconditional loads from a
    // clamped row, zeroed afterwards): one HBM round trip per tile instead of eight.
    const int c4 = tid & 31;
    const int r0 = tid >> 5;
    float4 v[8];
    const int last = max(nvalid - 1, 0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = min(r0 + 8 * u, last);
        v[u] = *reinterpret_cast<const float4*>(src + (size_t)r * row_stride + c4 * 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = r0 + 8 * u;
        if (r >= nvalid) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(tile + r * LDZ + c4 * 4) = v[u];
    }
    float am = 0.f;                // largest magnitude among this thread's values (AMAX only)
    if constexpr (AMAX) {
#pragma unroll
        for (int u = 0; u < 8; ++u) am = fmaxf(fmaxf(am, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    return am;
}

// ---------------------------------------------------------------------------
// Triangle multiplication, contraction (triangular_multiplicative_update.py:57-82):
//   x_cm[bc][i][j] = sum_k a_cm[bc][i][k] * b_cm[bc][j][k]   for bc = b*C + c.
// WG tile (64*WT)^2, 4 waves of (32*WT)^2, K streamed in 32-wide chunks through
// double-buffered LDS with register prefetch (one barrier per chunk).
// PERSISTENT: a work-group walks tiles w = blockIdx.x, + gridDim.x, ... as one flat
// (tile, chunk) sequence, so the first chunk of the next tile is already in flight while the
// last chunk of the current one is multiplied and its result stored.
// XCD-aware tile order: work-groups are dealt round-robin over the 8 XCDs, so ids with equal
// id % 8 share an XCD; the T tiles of one (b, c) matrix take ids 8 apart and therefore share
// each A / B panel through that XCD's L2 (placement affects speed only).
// All global addressing is buffer + scalar offset (f32 MFMA and VALU share the FP32 lanes:
// every vector instruction saved is MFMA time, see pair_wl_kernels.hip).
// ---------------------------------------------------------------------------
#define LDK 36
typedef __amdgpu_buffer_rsrc_t crsrc_t;
template <int WT>
__global__ __launch_bounds__(256, 2) void k_trimul_contract(const float* __restrict__ acm, const float* __restrict__ bcm,
                                                            float* __restrict__ xcm, int NP, int n_mat, unsigned cm_bytes) {
    constexpr int TM = 64 * WT;
    constexpr int NU = TM / 32;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [2 buf][A|B][TM*LDK]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = (NP + TM - 1) / TM;
    const int T = tiles * tiles;
    const int n_tiles = T * ((n_mat + 7) / 8) * 8;
    const int nk = NP / 32;
    const crsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(acm), 0, cm_bytes, 0x00020000);
    const crsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bcm), 0, cm_bytes, 0x00020000);
    const crsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xcm, 0, cm_bytes, 0x00020000);
    const int lr = tid >> 3, c4 = tid & 7;
    const int vload = (lr * NP + c4 * 4) * 4;          // per-lane byte offset inside a 32-row block of a panel
    const int slds = lr * LDK + c4 * 4;
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 rA[NU], rB[NU];

    auto decode = [&](int w, int& mi, int& i0, int& j0) {       // tile id -> (matrix, tile origin)
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
        const int mbase = mclamp * NP * NP * 4 + kc * 128;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int ia = min(i0 + 32 * u, NP - 32), jb = min(j0 + 32 * u, NP - 32);
            rA[u] = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(ra, vload, mbase + ia * NP * 4, 0));
            rB[u] = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rb, vload, mbase + jb * NP * 4, 0));
            if (i0 + 32 * u >= NP) rA[u] = v4{0.f, 0.f, 0.f, 0.f};
            if (j0 + 32 * u >= NP) rB[u] = v4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto swrite = [&](int buf) {
        float* sa = sm + buf * 2 * TM * LDK + slds;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            *reinterpret_cast<v4*>(sa + 32 * u * LDK) = rA[u];
            *reinterpret_cast<v4*>(sa + TM * LDK + 32 * u * LDK) = rB[u];
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
    for (int it = 0; it < total; ++it) {
        const bool last_chunk = kc == nk - 1;
        const int wn_next = last_chunk ? w + gridDim.x : w;
        const int kc_next = last_chunk ? 0 : kc + 1;
        if (it + 1 < total) gload(wn_next, kc_next);
        const float* sa = sm + (it & 1) * 2 * TM * LDK;
        const float* sb = sa + TM * LDK;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float4 af[WT], bf[WT];
#pragma unroll
            for (int m = 0; m < WT; ++m) af[m] = lfrag(sa, LDK, (wm * WT + m) * 32, kb, lane);
#pragma unroll
            for (int n = 0; n < WT; ++n) bf[n] = lfrag(sb, LDK, (wn * WT + n) * 32, kb, lane);
#pragma unroll
            for (int m = 0; m < WT; ++m)
#pragma unroll
                for (int n = 0; n < WT; ++n) acc[m][n] = mfma_8k(af[m], bf[n], acc[m][n]);
        }
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
                                const float v = acc[m][n][r];     // (bit_cast straight from a vector element picks element 0)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rx, vst,
                                                                      sbase + ((r & 3) + 8 * (r >> 2)) * NP * 4, 0);
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

// ---------------------------------------------------------------------------
// IPA pair bias for ALL structure layers in one pass over p
// (modules/invariant_point_attention.py:181): bias[l*H+h][b][i][j] = W_b^l z_ij + b.
// 128 pairs per WG; D rows = (layer, head), D cols = pairs -> j-contiguous rows
// for the attention kernel.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ipa_bias(const float* __restrict__ z, const float* __restrict__ wp,
                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                  int B, int N, int LH, int rev, unsigned* pmax) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [128][LDZ]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntile = (N + 127) >> 7;
    const int bid = rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;     // opposite to the kernel that wrote p last
    const int st = bid % ntile;
    const int i = (bid / ntile) % N;
    const int b = bid / (ntile * N);
    const int t0 = st * 128;
    const int nvalid = min(128, N - t0);
    const float* src = z + (((size_t)b * N + i) * N + t0) * 128;
    float am = load_tile64<true>(sm, src, 128, nvalid, tid);
    if (nvalid > 64) {
        am = fmaxf(am, load_tile64<true>(sm + 64 * LDZ, src + (size_t)64 * 128, 128, nvalid - 64, tid));
    } else {
        for (int u = tid; u < 64 * LDZ / 4; u += 256) reinterpret_cast<float4*>(sm + 64 * LDZ)[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // max |p| of the whole tensor for the attention kernel's f16 split (single_kernels.hip k_ipa_attn_q): this pass is
    // the one that sees every element before the first IPA layer.  Magnitudes order like their bit patterns.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    if (lane == 0 && __float_as_uint(am) > *reinterpret_cast<volatile unsigned*>(pmax)) atomicMax(pmax, __float_as_uint(am));
    __syncthreads();
    const int nbo = (LH + 31) >> 5;
    const int t = wave * 32 + (lane & 31);
    for (int ob = 0; ob < nbo; ++ob) {
        f32x16 acc = zero16();
#pragma unroll 4
        for (int kb = 0; kb < 16; ++kb)
            acc = mfma_8k(wfrag(wp, 16, ob, kb, lane), lfrag(sm, LDZ, wave * 32, kb, lane), acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = ob * 32 + acc_row(r, lane);
            if (o < LH && t < nvalid) out[(((size_t)o * B + b) * N + i) * N + t0 + t] = acc[r] + bias[o];
        }
    }
}

// ---------------------------------------------------------------------------
// Pair feature net.  Shared feature builder: softmax(-4|d - v_k|) soft bins
// (pair_feature_net.py:239-269, fork-specific) for 64 pairs (b, i, t0..t0+63):
// 4 lanes per pair, bins interleaved, reductions by wave shuffles.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void soft_bins(float* frow, float d, float dmin, float dstep, int nbin, float pm, int part) {
    float x[10];
    float mx = -3.0e38f;
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        const int k = part + 4 * q;
        x[q] = (k < nbin) ? -4.0f * fabsf(d - (dmin + (float)k * dstep)) : -3.0e38f;
        mx = fmaxf(mx, x[q]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        const int k = part + 4 * q;
        x[q] = (k < nbin) ? expf(x[q] - mx) : 0.f;
        s += x[q];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        const int k = part + 4 * q;
        if (k < nbin) frow[k] = (x[q] / s) * pm;
    }
}

// p = (p_i + p_j + static + W_t [bins | quat | fsm | fsm]) * mask
// (pair_feature_net.py:117-160).  static = relpos + motif term (k_pair_static).
#define LDF 52   // 48 features + 4 pad
// TABLE: no motif / template terms in this batch -- the step-invariant part is then one of 2 k + 2 rows of the relative-position
// table per pair (+ the same-chain row) and is looked up here instead of being read back from the 268-MB pstatic tensor
// (the same expression as k_pair_static<false>, so the result is bit-identical).
template <bool TABLE>
__global__ __launch_bounds__(256) void k_pair_init(const float* __restrict__ trans, const float* __restrict__ rots,
                                                   const int8_t* __restrict__ codes, const float* __restrict__ rmask,
                                                   const uint8_t* __restrict__ fstm, const float* __restrict__ pij,
                                                   const float* __restrict__ pstatic, const float* __restrict__ wt,
                                                   float* __restrict__ p, int N, float dmin, float dstep, int nbin,
                                                   const int32_t* __restrict__ ridx, const int32_t* __restrict__ cidx,
                                                   const float* __restrict__ relpos_t, int relpos_k) {
    __shared__ __attribute__((aligned(16))) float ft[64 * LDF];
    __shared__ float pmk[64];
    __shared__ int dsel[64];
    __shared__ float same[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntile = (N + 63) >> 6;
    const int st = blockIdx.x % ntile;
    const int i = (blockIdx.x / ntile) % N;
    const int b = blockIdx.x / (ntile * N);
    const int t0 = st * 64;
    const int nvalid = min(64, N - t0);
    {
        const int jl = tid >> 2, part = tid & 3;
        float* frow = ft + jl * LDF;
        if (jl < nvalid) {
            const int j = t0 + jl;
            const float pm = rmask[b * N + i] * rmask[b * N + j];
            const float* xi = trans + ((size_t)b * N + i) * 3;
            const float* xj = trans + ((size_t)b * N + j) * 3;
            const float dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
            const float d = sqrtf(1e-10f + ((dx * dx + dy * dy) + dz * dz));
            soft_bins(frow, d, dmin, dstep, nbin, pm, part);
            if (part == 0) {
                const float* Ri = rots + ((size_t)b * N + i) * 9;
                const float* Rj = rots + ((size_t)b * N + j) * 9;
                float r[9];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        r[a * 3 + c] = Rj[a * 3 + 0] * Ri[0 * 3 + c] + Rj[a * 3 + 1] * Ri[1 * 3 + c] + Rj[a * 3 + 2] * Ri[2 * 3 + c];
                float q[4];
                const size_t pidx = ((size_t)b * N + i) * N + j;
                rot_to_quat_dev(r, codes ? (int)codes[pidx] : 0, q);
                const float f = fstm[pidx] ? 1.f : 0.f;
                frow[nbin + 0] = q[0] * pm; frow[nbin + 1] = q[1] * pm; frow[nbin + 2] = q[2] * pm; frow[nbin + 3] = q[3] * pm;
                frow[nbin + 4] = f; frow[nbin + 5] = f;
                for (int k = nbin + 6; k < 48; ++k) frow[k] = 0.f;
                pmk[jl] = pm;
                if (TABLE) {
                    const bool sc = cidx[b * N + i] == cidx[b * N + j];
                    int dd = ridx[b * N + i] - ridx[b * N + j] + relpos_k;
                    dd = max(0, min(dd, 2 * relpos_k));
                    dsel[jl] = sc ? dd : 2 * relpos_k + 1;
                    same[jl] = sc ? 1.f : 0.f;
                }
            }
        } else {
            for (int k = part; k < 48; k += 4) frow[k] = 0.f;
            if (part == 0) pmk[jl] = 0.f;
        }
    }
    __syncthreads();
    f32x16 a0 = zero16(), a1 = zero16();
#pragma unroll
    for (int kb = 0; kb < 6; ++kb) {
        const float4 w = wfrag(wt, 6, wave, kb, lane);
        a0 = mfma_8k(lfrag(ft, LDF, 0, kb, lane), w, a0);
        a1 = mfma_8k(lfrag(ft, LDF, 32, kb, lane), w, a1);
    }
    const int ch = wave * 32 + (lane & 31);
    const float pi_c = pij[((size_t)b * N + i) * 256 + ch];
    const float w_same = TABLE ? relpos_t[(2 * relpos_k + 2) * 128 + ch] : 0.f;
    const size_t prow0 = ((size_t)b * N + i) * N + t0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int t = acc_row(r, lane) + 32 * hf;
            if (t < nvalid) {
                const float acc = hf ? a1[r] : a0[r];
                const float pj_c = pij[((size_t)b * N + t0 + t) * 256 + 128 + ch];
                const size_t o = (prow0 + t) * 128 + ch;
                const float stat = TABLE ? (relpos_t[dsel[t] * 128 + ch] + same[t] * w_same) + 0.f : pstatic[o];
                p[o] = (acc + pi_c + pj_c + stat) * pmk[t];
            }
        }
    }
}

// Step-invariant pair terms (pair_feature_net.py:134,149-158,166-221):
//   static = W_relpos[:, d_ij] + same_chain * W_relpos[:, 66]
//          + W_motif [bins(atom_positions) * fsm_i fsm_j * fstm_ij | fstm | fstm]
#define LDM 44   // 40 features + 4 pad
template <bool MOTIF>
__global__ __launch_bounds__(256) void k_pair_static(const float* __restrict__ pos, const int32_t* __restrict__ ridx,
                                                     const int32_t* __restrict__ cidx, const uint8_t* __restrict__ fsm,
                                                     const uint8_t* __restrict__ fstm, const float* __restrict__ relpos_t,
                                                     const float* __restrict__ wm, float* __restrict__ pstatic, int N,
                                                     int relpos_k, float dmin, float dstep, int nbin) {
    __shared__ __attribute__((aligned(16))) float ft[64 * LDM];
    __shared__ int dsel[64];
    __shared__ float same[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntile = (N + 63) >> 6;
    const int st = blockIdx.x % ntile;
    const int i = (blockIdx.x / ntile) % N;
    const int b = blockIdx.x / (ntile * N);
    const int t0 = st * 64;
    const int nvalid = min(64, N - t0);
    {
        const int jl = tid >> 2, part = tid & 3;
        float* frow = ft + jl * LDM;
        if (jl < nvalid) {
            const int j = t0 + jl;
            const size_t pidx = ((size_t)b * N + i) * N + j;
            if (MOTIF) {
                const float f = fstm[pidx] ? 1.f : 0.f;
                const float pm = ((fsm[b * N + i] && fsm[b * N + j]) ? 1.f : 0.f) * f;
                const float* xi = pos + ((size_t)b * N + i) * 3;
                const float* xj = pos + ((size_t)b * N + j) * 3;
                const float dx = xi[0] - xj[0], dy = xi[1] - xj[1], dz = xi[2] - xj[2];
                const float d = sqrtf(1e-10f + ((dx * dx + dy * dy) + dz * dz));
                soft_bins(frow, d, dmin, dstep, nbin, pm, part);
                if (part == 0) { frow[nbin] = f; frow[nbin + 1] = f; for (int k = nbin + 2; k < 40; ++k) frow[k] = 0.f; }
            }
            if (part == 0) {
                const bool sc = cidx[b * N + i] == cidx[b * N + j];
                int dd = ridx[b * N + i] - ridx[b * N + j] + relpos_k;
                dd = max(0, min(dd, 2 * relpos_k));
                dsel[jl] = sc ? dd : 2 * relpos_k + 1;
                same[jl] = sc ? 1.f : 0.f;
            }
        } else if (MOTIF) {
            for (int k = part; k < 40; k += 4) frow[k] = 0.f;
        }
    }
    __syncthreads();
    f32x16 a0 = zero16(), a1 = zero16();
    if (MOTIF) {
#pragma unroll
        for (int kb = 0; kb < 5; ++kb) {
            const float4 w = wfrag(wm, 5, wave, kb, lane);
            a0 = mfma_8k(lfrag(ft, LDM, 0, kb, lane), w, a0);
            a1 = mfma_8k(lfrag(ft, LDM, 32, kb, lane), w, a1);
        }
    }
    const int ch = wave * 32 + (lane & 31);
    const float w_same = relpos_t[(2 * relpos_k + 2) * 128 + ch];
    const size_t prow0 = ((size_t)b * N + i) * N + t0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int t = acc_row(r, lane) + 32 * hf;
            if (t < nvalid) {
                const float acc = hf ? a1[r] : a0[r];
                pstatic[(prow0 + t) * 128 + ch] = (relpos_t[dsel[t] * 128 + ch] + same[t] * w_same) + acc;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
__global__ void k_any_nonzero(const uint8_t* __restrict__ x, size_t n, unsigned* __restrict__ flag) {
    unsigned any = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) any |= x[i];
    if (any) atomicOr(flag, 1u);
}
void launch_any_nonzero(genie_ctx* h, hipStream_t st, const uint8_t* x, size_t n, unsigned* flag) {
    ProfScope ps(h, st, KC_MISC);
    hipLaunchKernelGGL(k_any_nonzero, dim3(256), dim3(256), 0, st, x, n, flag);
}

void launch_pair_static(genie_ctx* h, hipStream_t st) {
    ProfScope ps(h, st, KC_PAIR_STATIC);
    const int N = h->N, ntile = (N + 63) / 64;
    dim3 grid(h->B * N * ntile);
    if (h->has_motif)
        hipLaunchKernelGGL(k_pair_static<true>, grid, dim3(256), 0, st, h->f_pos, h->f_ridx, h->f_cidx, h->f_fsm, h->f_fstm,
                           h->relpos_t, h->motif_w, h->pstatic, N, h->d.relpos_k, h->d.template_dist_min,
                           h->d.template_dist_step, h->d.template_dist_n_bin);
    else
        hipLaunchKernelGGL(k_pair_static<false>, grid, dim3(256), 0, st, h->f_pos, h->f_ridx, h->f_cidx, h->f_fsm, h->f_fstm,
                           h->relpos_t, h->motif_w, h->pstatic, N, h->d.relpos_k, h->d.template_dist_min,
                           h->d.template_dist_step, h->d.template_dist_n_bin);
}

void launch_pair_init(genie_ctx* h, hipStream_t st, const float* trans, const float* rots, const int8_t* codes) {
    ProfScope ps(h, st, KC_PAIR_INIT);
    const int N = h->N, ntile = (N + 63) / 64;
    if (!h->has_motif && !getenv("GENIE_PAIR_STATIC_READ"))
        hipLaunchKernelGGL(k_pair_init<true>, dim3(h->B * N * ntile), dim3(256), 0, st, trans, rots, codes, h->rmaskf, h->f_fstm,
                           h->pij, h->pstatic, h->templ_w, h->p, N, h->d.template_dist_min, h->d.template_dist_step,
                           h->d.template_dist_n_bin, h->f_ridx, h->f_cidx, h->relpos_t, h->d.relpos_k);
    else
        hipLaunchKernelGGL(k_pair_init<false>, dim3(h->B * N * ntile), dim3(256), 0, st, trans, rots, codes, h->rmaskf, h->f_fstm,
                           h->pij, h->pstatic, h->templ_w, h->p, N, h->d.template_dist_min, h->d.template_dist_step,
                           h->d.template_dist_n_bin, h->f_ridx, h->f_cidx, h->relpos_t, h->d.relpos_k);
}

void launch_trimul_proj_wl(genie_ctx* h, hipStream_t st, const TriMulW& w, bool outgoing);
void launch_trimul_out_wl(genie_ctx* h, hipStream_t st, const TriMulW& w);
void launch_pair_transition_wl(genie_ctx* h, hipStream_t st, const PairLayerW& w);

void launch_trimul_hx(genie_ctx* h, hipStream_t st, const TriMulW& w, bool outgoing);
void launch_trimul(genie_ctx* h, hipStream_t st, const TriMulW& w, bool outgoing) {
    if (h->hx) { launch_trimul_hx(h, st, w, outgoing); return; }
    const int NP = h->NP;
    {
        ProfScope ps(h, st, KC_TRIMUL_PROJ);
        launch_trimul_proj_wl(h, st, w, outgoing);
    }
    {
        ProfScope ps(h, st, KC_TRIMUL_CONTRACT);
        const int BC = h->B * h->d.c_hidden_mul;
        const unsigned cm_bytes = (unsigned)((size_t)BC * NP * NP * 4);
        static int num_cu = 0;
        if (!num_cu) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&pr, dev); num_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
        if (NP >= 128) {
            const int tiles = (NP + 127) / 128;
            const int n_tiles = tiles * tiles * ((BC + 7) / 8) * 8;
            const size_t lds = 2 * 2 * 128 * LDK * sizeof(float);
            hipLaunchKernelGGL(k_trimul_contract<2>, dim3(n_tiles < 2 * num_cu ? n_tiles : 2 * num_cu), dim3(256), lds, st, h->acm,
                               h->bcm, h->xcm, NP, BC, cm_bytes);
        } else {
            const int tiles = (NP + 63) / 64;
            const int n_tiles = tiles * tiles * ((BC + 7) / 8) * 8;
            const size_t lds = 2 * 2 * 64 * LDK * sizeof(float);
            hipLaunchKernelGGL(k_trimul_contract<1>, dim3(n_tiles < 4 * num_cu ? n_tiles : 4 * num_cu), dim3(256), lds, st, h->acm,
                               h->bcm, h->xcm, NP, BC, cm_bytes);
        }
    }
    {
        ProfScope ps(h, st, KC_TRIMUL_OUT);
        launch_trimul_out_wl(h, st, w);
    }
}

void launch_pair_transition_hx(genie_ctx* h, hipStream_t st, const PairLayerW& w);
void launch_pair_transition(genie_ctx* h, hipStream_t st, const PairLayerW& w) {
    ProfScope ps(h, st, KC_PAIR_TRANSITION);
    if (h->hx) launch_pair_transition_hx(h, st, w);
    else launch_pair_transition_wl(h, st, w);
}

bool launch_ipa_bias_hx(genie_ctx* h, hipStream_t st);
void launch_ipa_bias(genie_ctx* h, hipStream_t st) {
    ProfScope ps(h, st, KC_IPA_BIAS);
    if (launch_ipa_bias_hx(h, st)) return;
    const int N = h->N, ntile = (N + 127) / 128;
    const int LH = h->d.n_structure_layer * h->d.n_head_ipa;
    const size_t lds = 128 * LDZ * sizeof(float);
    (void)hipMemsetAsync(h->pmax, 0, sizeof(unsigned), st);
    hipLaunchKernelGGL(k_ipa_bias, dim3(h->B * N * ntile), dim3(256), lds, st, h->p, h->ipa_bias_w, h->ipa_bias_b, h->ipa_bias,
                       h->B, N, LH, h->hx ? (int)(h->hx_launches & 1) : 0, h->pmax);
}

void pair_wl_kernels_init();
void pair_hx_kernels_init();
// One-time opt-in to > 64 KiB dynamic LDS.
void pair_kernels_init() {
    pair_wl_kernels_init();
    pair_hx_kernels_init();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trimul_contract<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        2 * 2 * 128 * LDK * sizeof(float));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ipa_bias), hipFuncAttributeMaxDynamicSharedMemorySize,
                        128 * LDZ * sizeof(float));
}
