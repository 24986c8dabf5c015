// Single-track kernels: single feature net, row GEMMs, LayerNorm, invariant
// point attention, backbone update.  Reference lines are cited per kernel.
#include <algorithm>
#include <type_traits>
#include "hx.h"

// ---------------------------------------------------------------------------
// Single feature net input (single_feature_net.py:103-142): one row per
// residue, [pos | chain | t | aatype*fsm | fsm | fsm | interface | 0-pad].
// The sinusoidal tables come from the host (encoding.py:5-25 evaluated with
// the reference's own torch expression), so no sin/cos/pow runs on device.
// ---------------------------------------------------------------------------
__global__ void k_single_input(float* __restrict__ x, int ldx, const float* __restrict__ pos_tab, int n_pos, int c_pos,
                               const float* __restrict__ chain_tab, int n_chain, int c_chain,
                               const float* __restrict__ t_tab, int c_t, const int32_t* __restrict__ timesteps,
                               const int32_t* __restrict__ ridx, const int32_t* __restrict__ cidx,
                               const int32_t* __restrict__ aatype, const uint8_t* __restrict__ fsm,
                               const uint8_t* __restrict__ ifm, int N, int n_timestep) {
    const int row = blockIdx.x;           // b*N + n
    const int b = row / N;
    float* xr = x + (size_t)row * ldx;
    const int ri = min(max(ridx[row], 0), n_pos - 1);
    const int ci = min(max(cidx[row], 0), n_chain - 1);
    const int ts = min(max(timesteps[b], 0), n_timestep);
    const float f = fsm[row] ? 1.f : 0.f;
    const int o1 = c_pos, o2 = o1 + c_chain, o3 = o2 + c_t, o4 = o3 + 20;
    for (int c = threadIdx.x; c < ldx; c += blockDim.x) {
        float v;
        if (c < o1) v = pos_tab[(size_t)ri * c_pos + c];
        else if (c < o2) v = chain_tab[(size_t)ci * c_chain + (c - o1)];
        else if (c < o3) v = t_tab[(size_t)ts * c_t + (c - o2)];
        else if (c < o4) v = (float)aatype[(size_t)row * 20 + (c - o3)] * f;
        else if (c < o4 + 2) v = f;
        else if (c == o4 + 2) v = ifm[row] ? 1.f : 0.f;
        else v = 0.f;
        xr[c] = v;
    }
}

// ---------------------------------------------------------------------------
// out[M][Nout] = epi(A[M][K] W^T):  epi = (+bias) (relu) (+res) (*rowmask).
// 32 rows x 128 cols per WG (4 waves, one 32x32 MFMA tile each); A streamed in
// 64-wide K chunks through double-buffered LDS, W fragments straight from L2.
// Serves primitives.Linear (primitives.py:96-160) wherever the row count is
// B*N: single feature net, p_i/p_j, IPA projections and output, transition.
// ---------------------------------------------------------------------------
#define LDA 68
#define GR_PD 8
__global__ __launch_bounds__(256) void k_gemm_rows(const float* __restrict__ A, int lda, int M, int K,
                                                   const float* __restrict__ Wp, int Nout,
                                                   const float* __restrict__ bias, const float* __restrict__ res, int ldr,
                                                   const float* __restrict__ rowmask, int relu, float* __restrict__ out, int ldo) {
    // Every global load of the main loop is an asm load (common.h, "software-pipelined weight
    // fragments"): the weight ring stays GR_PD k-blocks ahead, the next A chunk is in flight
    // during the current chunk's MFMAs, and no compiler-inserted vmcnt(0) drains either.
    // In-order retirement makes the wait counts static:
    //   releasing ring slot kb : younger = (7-kb) old slots + 2 A loads + kb refills  = GR_PD + 1
    //   releasing the A loads  : younger = the 8 refills of this chunk                = GR_PD
    __shared__ __attribute__((aligned(16))) float sa[2][32 * LDA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * 32;
    const int nb = min((int)(blockIdx.y * 4 + wave), (Nout + 31) / 32 - 1);   // idle waves shadow the last block
    const bool active = (int)(blockIdx.y * 4 + wave) * 32 < Nout;
    const int KB = (K + 7) >> 3;
    const int nkc = (K + 63) >> 6;
    const int lr = tid >> 4, c4 = tid & 15;      // 16 rows x 16 float4 per pass, 2 passes
    const float* wbase = Wp + ((size_t)nb * KB * 64 + lane) * 4;

    v4f wq[GR_PD];
#pragma unroll
    for (int s = 0; s < GR_PD; ++s) wf_issue(wq[s], wbase + (size_t)min(s, KB - 1) * 256);

    v4f ra0, ra1;
    auto a_addr = [&](int kc, int u) {
        const int r = min(r0 + lr + 16 * u, M - 1), k = min(kc * 64 + c4 * 4, K - 4);
        return A + (size_t)r * lda + k;
    };
    auto a_store = [&](int kc, int buf) {
        const bool kok = kc * 64 + c4 * 4 < K;
        const v4f z4 = {0.f, 0.f, 0.f, 0.f};
        const v4f v0 = (kok && r0 + lr < M) ? ra0 : z4;
        const v4f v1 = (kok && r0 + lr + 16 < M) ? ra1 : z4;
        *reinterpret_cast<v4f*>(&sa[buf][lr * LDA + c4 * 4]) = v0;
        *reinterpret_cast<v4f*>(&sa[buf][(lr + 16) * LDA + c4 * 4]) = v1;
    };
    wf_issue(ra0, a_addr(0, 0));
    wf_issue(ra1, a_addr(0, 1));
    wf_wait<0>(ra0, ra1);
    a_store(0, 0);
    __syncthreads();

    f32x16 acc = zero16();
    for (int kc = 0; kc < nkc; ++kc) {
        const int kcn = min(kc + 1, nkc - 1);            // last chunk: harmless re-load keeps the counts uniform
        wf_issue(ra0, a_addr(kcn, 0));
        wf_issue(ra1, a_addr(kcn, 1));
        const float* cur = sa[kc & 1];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const float4 af = lfrag(cur, LDA, 0, kb, lane);
            wf_wait<GR_PD + 1>(wq[kb]);
            if (kc * 8 + kb < KB) acc = mfma_8k(af, wq[kb], acc);
            __builtin_amdgcn_sched_barrier(0);
            wf_issue(wq[kb], wbase + (size_t)min(kc * 8 + kb + GR_PD, KB - 1) * 256);
        }
        wf_wait<GR_PD>(ra0, ra1);
        a_store(kcn, (kc + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < GR_PD; ++s) wf_wait<0>(wq[s]);
    if (!active) return;
    const int col = nb * 32 + (lane & 31);
    if (col >= Nout) return;
    const float bc = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = r0 + acc_row(r, lane);
        if (row < M) {
            float v = acc[r] + bc;
            if (relu) v = fmaxf(v, 0.f);
            if (res) v += res[(size_t)row * ldr + col];
            if (rowmask) v *= rowmask[row];
            out[(size_t)row * ldo + col] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// The same Linear in hx arithmetic (hx.h): three f16 MFMAs per product on split operands.
// The activations of the single track have no load-time bound (IPA outputs carry pair values and
// coordinates), so they are split BLOCK-FLOATING-POINT: each (row, 64-wide K chunk) gets its own
// power-of-two scale from its largest magnitude (16 lanes share a row chunk: 4 shuffles), the chunk
// is accumulated in a fresh accumulator and folded into the running sum with the inverse scale
// (16 FMAs per 12 MFMAs).  Weight units (hi + lo fragment of 32 columns x 16 k) stream from L2
// through an asm-issued register ring of 8 units; every wait count is static (in-order retirement):
//   releasing a ring unit : younger = 7 other units (14 loads) + 2 x 2 A loads          = 18
//   releasing the A loads : younger = the 4 refills of this chunk (8 loads)             = 8
// (Measured: deeper rings / activation prefetch / split accumulator chains do not help -- at M = B N rows these launches
//  are bounded by fixed latencies, tools/trace_gemm.sh: 13 us for 192 work-groups of 72 MFMAs per wave.)
// ---------------------------------------------------------------------------
#define GH_ROWB 144          // bytes per row of a 64-k plane of halves (128 + 16 pad: conflict-free b128 reads)
#define GH_PD 8
#define GH_AD 4            // activation chunks in flight
__global__ __launch_bounds__(256) void k_gemm_rows_hx(const float* __restrict__ A, int lda, int M, int K,
                                                      const unsigned char* __restrict__ Wx, int Nout, float inv_sw,
                                                      const float* __restrict__ bias, const float* __restrict__ res, int ldr,
                                                      const float* __restrict__ rowmask, int relu, float* __restrict__ out, int ldo,
                                                      size_t zstride) {
    __shared__ __attribute__((aligned(16))) unsigned char sa[2][2][32 * GH_ROWB];   // [buffer][hi | lo]
    __shared__ __attribute__((aligned(16))) float sinv[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * 32;
    const int nb = min((int)(blockIdx.y * 4 + wave), (Nout + 31) / 32 - 1);   // idle waves shadow the last block
    const bool active = (int)(blockIdx.y * 4 + wave) * 32 < Nout;
    // split-K (gridDim.z > 1; the 33 chunks of the IPA output projection): slice z owns chunks [c0, c1) and writes its raw partial
    // product to out + z * zstride (callers pass no bias / residual / mask and sum the slices where they read them)
    const int KCt = (K + 15) >> 4, nkct = (K + 63) >> 6;
    const int c0 = (int)blockIdx.z * nkct / (int)gridDim.z, c1 = ((int)blockIdx.z + 1) * nkct / (int)gridDim.z;
    const float* wbase = reinterpret_cast<const float*>(Wx + ((size_t)nb * KCt + (size_t)c0 * 4) * 2048) + lane * 4;   // unit u: + u*512 floats; lo: + 256
    A += c0 * 64;
    K = min(K, c1 * 64) - c0 * 64;
    out += (size_t)blockIdx.z * zstride;
    const int KC = (K + 15) >> 4;                // 16-wide weight units
    const int nkc = (K + 63) >> 6;               // 64-wide activation chunks
    const int lr = tid >> 4, c4 = tid & 15;      // 16 rows x 16 float4 per pass, 2 passes

    v4f wh[GH_PD], wl[GH_PD];
#pragma unroll
    for (int s = 0; s < GH_PD; ++s) {
        const float* p = wbase + (size_t)min(s, KC - 1) * 512;
        wf_issue(wh[s], p);
        wf_issue(wl[s], p + 256);
    }
    // activation chunks are fetched GH_AD = 4 chunks ahead (compile-time slots kc & 3): one chunk is only 12 MFMAs per wave,
    // far less than an L2 / HBM round trip, and with K = 2112 (IPA output projection) there are 33 of them in a row
    v4f ra[GH_AD][2];
    auto a_addr = [&](int kc, int u) {
        const int r = min(r0 + lr + 16 * u, M - 1), k = min(kc * 64 + c4 * 4, K - 4);
        return A + (size_t)r * lda + k;
    };
    auto split_store = [&](v4f v, int row, int buf) {
        float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4)); m = fmaxf(m, __shfl_xor(m, 8));
        const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 255u);          // biased exponent of the chunk's largest |a|
        const bool tiny = e < 16;                                                      // (all ~0: any scale will do)
        const float sc = tiny ? 1.0f : __builtin_bit_cast(float, (unsigned)(268 - e) << 23);    // max * sc in [2^14, 2^15)
        const float isc = tiny ? 1.0f : __builtin_bit_cast(float, (unsigned)(e - 14) << 23);
        unsigned h01, l01, h23, l23;
        hx_split2(v.x, v.y, sc, h01, l01);
        hx_split2(v.z, v.w, sc, h23, l23);
        *reinterpret_cast<uint2*>(&sa[buf][0][row * GH_ROWB + c4 * 8]) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(&sa[buf][1][row * GH_ROWB + c4 * 8]) = make_uint2(l01, l23);
        if (c4 == 0) sinv[buf][row] = isc;
    };
    auto a_store = [&](v4f v0, v4f v1, int kc, int buf) {
        const bool kok = kc * 64 + c4 * 4 < K;
        const v4f z4 = {0.f, 0.f, 0.f, 0.f};
        split_store((kok && r0 + lr < M) ? v0 : z4, lr, buf);
        split_store((kok && r0 + lr + 16 < M) ? v1 : z4, lr + 16, buf);
    };
    // chunk c sits in slot c & 3: chunks 0..3 now (0 is consumed at once), chunk kc + 4 at the start of chunk kc
#pragma unroll
    for (int c = 0; c < GH_AD; ++c) {
        wf_issue(ra[c][0], a_addr(min(c, nkc - 1), 0));
        wf_issue(ra[c][1], a_addr(min(c, nkc - 1), 1));
    }
    wf_wait<2 * (GH_AD - 1)>(ra[0][0], ra[0][1]);
    a_store(ra[0][0], ra[0][1], 0, 0);
    __syncthreads();

    f32x16 acc = zero16();
    const int foff = (lane & 31) * GH_ROWB + (lane >> 5) * 16;
    // chunk body with COMPILE-TIME slots (weight ring half 4 PAR .. 4 PAR + 3, activation slots): an asm load into a dynamically
    // indexed register array would land in a temporary the compiler has already copied from
    auto chunk = [&](int kc, auto slot_tag, auto first_tag) {
        constexpr int SL = decltype(slot_tag)::value, PAR = SL & 1, NX = (SL + 1) & (GH_AD - 1);
        constexpr bool FIRST = decltype(first_tag)::value;          // kc = SL (the first group of four)
        const int kcn = min(kc + 1, nkc - 1);            // last chunk: harmless re-split keeps the barrier count uniform
        // slot SL held chunk kc, split at the end of chunk kc - 1: free for chunk kc + 4
        wf_issue(ra[SL][0], a_addr(min(kc + GH_AD, nkc - 1), 0));
        wf_issue(ra[SL][1], a_addr(min(kc + GH_AD, nkc - 1), 1));
        const unsigned char* cur = &sa[PAR][0][0] + foff;
        f32x16 part = zero16();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const h8 ah = *reinterpret_cast<const h8*>(cur + q * 32);
            const h8 al = *reinterpret_cast<const h8*>(cur + 32 * GH_ROWB + q * 32);
            wf_wait<2 * GH_PD + 2>(wh[PAR * 4 + q], wl[PAR * 4 + q]);
            const h8 bh = __builtin_bit_cast(h8, wh[PAR * 4 + q]), bl = __builtin_bit_cast(h8, wl[PAR * 4 + q]);
            MFH3(ah, al, bh, bl, part);
            __builtin_amdgcn_sched_barrier(0);
            const float* p = wbase + (size_t)min(kc * 4 + q + GH_PD, KC - 1) * 512;
            wf_issue(wh[PAR * 4 + q], p);
            wf_issue(wl[PAR * 4 + q], p + 256);
        }
        {   // fold the chunk in with the inverse of its row scales (rows 8g + 4h .. + 3 are contiguous)
            const float* si = sinv[PAR] + 4 * (lane >> 5);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 iv = *reinterpret_cast<const float4*>(si + 8 * g);
                acc[4 * g] = fmaf(part[4 * g], iv.x, acc[4 * g]); acc[4 * g + 1] = fmaf(part[4 * g + 1], iv.y, acc[4 * g + 1]);
                acc[4 * g + 2] = fmaf(part[4 * g + 2], iv.z, acc[4 * g + 2]); acc[4 * g + 3] = fmaf(part[4 * g + 3], iv.w, acc[4 * g + 3]);
            }
        }
        // chunk kc + 1 (slot NX) was issued at the start of chunk kc - 3: younger = 8 refills there + 3 x (2 + 8) since;
        // chunks 1..3 were issued before the loop: younger = chunks kc + 2 .. 3 (2 loads each) + (kc + 1) x (2 + 8)
        constexpr int YOUNGER = (FIRST && SL < GH_AD - 1) ? 2 * (GH_AD - 2 - SL) + 10 * (SL + 1) : 8 + 10 * (GH_AD - 1);
        wf_wait<YOUNGER>(ra[NX][0], ra[NX][1]);
        a_store(ra[NX][0], ra[NX][1], kcn, PAR ^ 1);
        __syncthreads();
    };
    auto group = [&](int kc, auto first_tag) {
        chunk(kc, std::integral_constant<int, 0>{}, first_tag);
        if (kc + 1 < nkc) chunk(kc + 1, std::integral_constant<int, 1>{}, first_tag);
        if (kc + 2 < nkc) chunk(kc + 2, std::integral_constant<int, 2>{}, first_tag);
        if (kc + 3 < nkc) chunk(kc + 3, std::integral_constant<int, 3>{}, first_tag);
    };
    group(0, std::true_type{});
    for (int kc = 4; kc < nkc; kc += 4) group(kc, std::false_type{});
#pragma unroll
    for (int c = 0; c < GH_AD; ++c) wf_wait<0>(ra[c][0], ra[c][1]);
#pragma unroll
    for (int s = 0; s < GH_PD; ++s) wf_wait<0>(wh[s], wl[s]);
    if (!active) return;
    const int col = nb * 32 + (lane & 31);
    if (col >= Nout) return;
    const float bc = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = r0 + acc_row(r, lane);
        if (row < M) {
            float v = fmaf(acc[r], inv_sw, bc);
            if (relu) v = fmaxf(v, 0.f);
            if (res) v += res[(size_t)row * ldr + col];
            if (rowmask) v *= rowmask[row];
            out[(size_t)row * ldo + col] = v;
        }
    }
}

// nn.LayerNorm over the last dim, one wave per row (structure_net.py:64,111;
// structure_transition.py:61).  C <= 64*8.
__global__ __launch_bounds__(256) void k_layernorm_rows(const float* __restrict__ in, float* __restrict__ out, int M, int C,
                                                        const float* __restrict__ g, const float* __restrict__ bta) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* x = in + (size_t)row * C;
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int c = lane + 64 * q; v[q] = (c < C) ? x[c] : 0.f; s += v[q]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int c = lane + 64 * q; if (c < C) { const float d = v[q] - mean; ss += d * d; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float rstd = 1.0f / sqrtf(ss / (float)C + GENIE_LN_EPS);
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int c = lane + 64 * q; if (c < C) out[(size_t)row * C + c] = (v[q] - mean) * rstd * g[c] + bta[c]; }
}

// ---------------------------------------------------------------------------
// IPA, activations -> attention operands
// (modules/invariant_point_attention.py:124-174).  proj row layout:
//   [ q (H*C) | kv (H*2C: per head k then v) | q_pts (3 blocks of H*Pq)
//     | kv_pts (3 blocks of H*(Pq+Pv)) ].
// Points are moved to the global frame (T.apply, affine_utils.py:118-121).
// Keys go out j-contiguous (kT, kpT) so the logits pass reads them coalesced.
// ---------------------------------------------------------------------------
__global__ void k_ipa_prep(const float* __restrict__ proj, int ldp, const float* __restrict__ rots, const float* __restrict__ trans,
                           float* __restrict__ kT, float* __restrict__ v, float* __restrict__ qp, float* __restrict__ kpT,
                           float* __restrict__ vp, int N, int H, int C, int Pq, int Pv, int row0) {
    const int row = row0 + blockIdx.x, b = row / N, n = row % N;
    const float* pr = proj + (size_t)row * ldp;
    const float* R = rots + (size_t)row * 9;
    const float* t = trans + (size_t)row * 3;
    const int HC = H * C, oq = 0, okv = HC, oqp = 3 * HC, okp = 3 * HC + 3 * H * Pq;
    (void)oq;
    for (int u = threadIdx.x; u < HC; u += blockDim.x) {
        const int hh = u / C, c = u % C;
        kT[(((size_t)b * H + hh) * C + c) * N + n] = pr[okv + hh * 2 * C + c];
        v[(size_t)row * HC + u] = pr[okv + hh * 2 * C + C + c];
    }
    const int nq = H * Pq, nkv = H * (Pq + Pv);
    for (int u = threadIdx.x; u < nq + nkv; u += blockDim.x) {
        const bool isq = u < nq;
        const int idx = isq ? u : u - nq;
        const int blk = isq ? nq : nkv;
        const float* src = pr + (isq ? oqp : okp);
        const float x = src[idx], y = src[blk + idx], zc = src[2 * blk + idx];
        const float gx = R[0] * x + R[1] * y + R[2] * zc + t[0];
        const float gy = R[3] * x + R[4] * y + R[5] * zc + t[1];
        const float gz = R[6] * x + R[7] * y + R[8] * zc + t[2];
        if (isq) {
            float* d = qp + ((size_t)row * nq + idx) * 3;
            d[0] = gx; d[1] = gy; d[2] = gz;
        } else {
            const int hh = idx / (Pq + Pv), pp = idx % (Pq + Pv);
            if (pp < Pq) {
                float* d = kpT + ((((size_t)b * H + hh) * Pq + pp) * 3) * N + n;
                d[0] = gx; d[N] = gy; d[2 * N] = gz;
            } else {
                float* d = vp + (((size_t)row * H + hh) * Pv + (pp - Pq)) * 3;
                d[0] = gx; d[1] = gy; d[2] = gz;
            }
        }
    }
}

// The same for the hx attention kernel (k_ipa_attn_q<.., MF = 1>, C = 16), which takes a v and a v_pts on the matrix pipe: a work-group
// owns EIGHT consecutive residues of one structure, so that everything that goes out key-contiguous leaves as 32-byte runs (kT, kpT --
// 4-byte stores N apart in the one-row form), and V and v_pts go out as the B fragments of v_mfma_f32_16x16x32_f16 instead of row-major:
//   vf[b][h][block][j / 32][lane = 16 ((j & 31) >> 3) + column][j & 7],   block 0 = the head's 16 channels of v, blocks 1.. = its 3 Pv
// point coordinates in steps of 16 (padding columns and rows beyond N stay zero) -- still f32: the f16 split needs ONE scale per
// structure, known only when every row is through; each work-group leaves its largest |v| and |v_pt| in vmax[b][group][2] for that.
__global__ __launch_bounds__(256) void k_ipa_prep_frag(const float* __restrict__ proj, int ldp, const float* __restrict__ rots,
                                                       const float* __restrict__ trans, float* __restrict__ kT, float* __restrict__ qp,
                                                       float* __restrict__ kpT, float* __restrict__ vf, float* __restrict__ vmax,
                                                       int N, int H, int C, int Pq, int Pv, int b0) {
    const int G = (N + 7) >> 3, b = b0 + blockIdx.x / G, grp = blockIdx.x % G, j0 = grp * 8;
    const int KS = (N + 31) >> 5, NBK = 1 + (3 * Pv + 15) / 16;
    const int HC = H * C, okv = HC, oqp = 3 * HC, okp = 3 * HC + 3 * H * Pq;
    __shared__ float sR[8][12];
    __shared__ unsigned smax[2];
    if (threadIdx.x < 96) {
        const int e = threadIdx.x / 12, k = threadIdx.x % 12, row = b * N + min(j0 + e, N - 1);
        sR[e][k] = k < 9 ? rots[(size_t)row * 9 + k] : trans[(size_t)row * 3 + (k - 9)];
    }
    if (threadIdx.x < 2) smax[threadIdx.x] = 0u;
    __syncthreads();
    const float* pr = proj + (size_t)(b * N + j0) * ldp;
    const bool vec = (N & 3) == 0;                          // then a run of 8 keys is 16-byte aligned and inside the row
    auto put8 = [&](float* dst, const float (&x)[8]) {      // keys j0 .. j0 + 7 of a key-contiguous row
        if (vec && j0 + 8 <= N) {
            *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(x[4], x[5], x[6], x[7]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (j0 + e < N) dst[e] = x[e];
        }
    };
    auto put_frag = [&](int hh, int cb, int c, const float (&x)[8]) {        // rows beyond N as zeros
        float* dst = vf + (((((size_t)b * H + hh) * NBK + cb) * KS + (j0 >> 5)) * 64 + ((j0 & 31) >> 3) * 16 + c) * 8;
        *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(x[4], x[5], x[6], x[7]);
    };
    float mv = 0.f, mp = 0.f;
    for (int u = threadIdx.x; u < HC; u += 256) {
        const int hh = u / C, c = u % C;
        float k8[8], v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = j0 + e < N;
            k8[e] = ok ? pr[(size_t)e * ldp + okv + hh * 2 * C + c] : 0.f;
            v8[e] = ok ? pr[(size_t)e * ldp + okv + hh * 2 * C + C + c] : 0.f;
            mv = fmaxf(mv, fabsf(v8[e]));
        }
        put8(kT + (((size_t)b * H + hh) * C + c) * N + j0, k8);
        put_frag(hh, 0, c, v8);
    }
    const int nq = H * Pq, nkv = H * (Pq + Pv);
    for (int u = threadIdx.x; u < nq + nkv; u += 256) {
        const bool isq = u < nq;
        const int idx = isq ? u : u - nq;
        const int blk = isq ? nq : nkv;
        const float* src = pr + (isq ? oqp : okp);
        float gx[8], gy[8], gz[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = j0 + e < N;
            const float x = ok ? src[(size_t)e * ldp + idx] : 0.f, y = ok ? src[(size_t)e * ldp + blk + idx] : 0.f;
            const float zc = ok ? src[(size_t)e * ldp + 2 * blk + idx] : 0.f;
            const float* R = sR[e];
            gx[e] = ok ? R[0] * x + R[1] * y + R[2] * zc + R[9] : 0.f;
            gy[e] = ok ? R[3] * x + R[4] * y + R[5] * zc + R[10] : 0.f;
            gz[e] = ok ? R[6] * x + R[7] * y + R[8] * zc + R[11] : 0.f;
        }
        if (isq) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (j0 + e < N) {
                    float* d = qp + ((size_t)(b * N + j0 + e) * nq + idx) * 3;
                    d[0] = gx[e]; d[1] = gy[e]; d[2] = gz[e];
                }
            }
        } else {
            const int hh = idx / (Pq + Pv), pp = idx % (Pq + Pv);
            if (pp < Pq) {
                float* d = kpT + ((((size_t)b * H + hh) * Pq + pp) * 3) * N + j0;
                put8(d, gx); put8(d + N, gy); put8(d + 2 * N, gz);
            } else {
                const int w = (pp - Pq) * 3;
                put_frag(hh, 1 + (w >> 4), w & 15, gx);
                put_frag(hh, 1 + ((w + 1) >> 4), (w + 1) & 15, gy);
                put_frag(hh, 1 + ((w + 2) >> 4), (w + 2) & 15, gz);
#pragma unroll
                for (int e = 0; e < 8; ++e) mp = fmaxf(mp, fmaxf(fabsf(gx[e]), fmaxf(fabsf(gy[e]), fabsf(gz[e]))));
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mv = fmaxf(mv, __shfl_xor(mv, o)); mp = fmaxf(mp, __shfl_xor(mp, o)); }
    if ((threadIdx.x & 63) == 0) { atomicMax(&smax[0], __float_as_uint(mv)); atomicMax(&smax[1], __float_as_uint(mp)); }
    __syncthreads();
    if (threadIdx.x < 2) vmax[((size_t)b * G + grp) * 2 + threadIdx.x] = __uint_as_float(smax[threadIdx.x]);
}


// ---------------------------------------------------------------------------
// IPA attention core, one work-group per query residue (b, i)
// (modules/invariant_point_attention.py:181-249):
//   logit[h][j] = q.k/sqrt(3C) + sqrt(1/3) bias + (-1/2) softplus(g_h) wc sum_p |qp - kp|^2 + 1e5 (m_i m_j - 1)
//   a = softmax_j;  o = a v;  o_pt = R_i^T (a v_pts - t_i);  |o_pt|;  o_pair = a z_i.
// The p row (N x 128 fp32) is streamed exactly once, 512 B per row per wave
// half; everything else is L2 resident.  Output: the linear_out input row
//   [ o | o_pt.x | o_pt.y | o_pt.z | |o_pt| | o_pair ].
// Dynamic LDS: att[H][N] | red[8][H][128] (aliased with q/qp scratch).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ipa_attn(const float* __restrict__ proj, int ldp, const float* __restrict__ kT,
                                                  const float* __restrict__ v, const float* __restrict__ qp,
                                                  const float* __restrict__ kpT, const float* __restrict__ vp,
                                                  const float* __restrict__ bias, const float* __restrict__ z,
                                                  const float* __restrict__ rots, const float* __restrict__ trans,
                                                  const float* __restrict__ rmask, const float* __restrict__ head_w,
                                                  float* __restrict__ cat, int B, int N, int H, int C, int Pq, int Pv,
                                                  int c_p, int layer) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* att = sm;                       // [H][N]
    float* scr = sm + H * N;               // scratch: q[H*C] | qpt[H*Pq*3] | hw[H] ; later red[8][H][c_p]
    float* opt = scr + 8 * H * c_p;        // o_pt global-frame accumulators [H*Pv*3]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x, b = row / N, i = row % N;
    const int HC = H * C, nqp = H * Pq * 3;
    float* sq = scr;
    float* sqp = scr + HC;
    float* shw = sqp + nqp;
    for (int u = tid; u < HC; u += 256) sq[u] = proj[(size_t)row * ldp + u];
    for (int u = tid; u < nqp; u += 256) sqp[u] = qp[(size_t)row * nqp + u];
    if (tid < H) {
        const float g = head_w[tid];
        const float sp = (g > 20.f) ? g : log1pf(expf(g));          // nn.Softplus (beta 1, threshold 20)
        shw[tid] = sp * sqrtf(1.0f / (3.0f * ((float)Pq * 9.0f / 2.0f)));
    }
    __syncthreads();
    const float s_qk = sqrtf(1.0f / (3.0f * (float)C)), s_b = sqrtf(1.0f / 3.0f);
    const float mi = rmask[row];
    for (int j = tid; j < N; j += 256) {
        const float sqm = 1e5f * (mi * rmask[b * N + j] - 1.0f);
        for (int hh = 0; hh < H; ++hh) {
            const float* kc = kT + (((size_t)b * H + hh) * C) * N + j;
            float qk = 0.f;
            for (int c = 0; c < C; ++c) qk += sq[hh * C + c] * kc[(size_t)c * N];
            float a = qk * s_qk;
            a += s_b * bias[((((size_t)layer * H + hh) * B + b) * N + i) * N + j];
            const float* kp = kpT + ((((size_t)b * H + hh) * Pq) * 3) * N + j;
            float pt = 0.f;
            for (int pp = 0; pp < Pq; ++pp) {
                const float dx = sqp[(hh * Pq + pp) * 3 + 0] - kp[(size_t)(pp * 3 + 0) * N];
                const float dy = sqp[(hh * Pq + pp) * 3 + 1] - kp[(size_t)(pp * 3 + 1) * N];
                const float dz = sqp[(hh * Pq + pp) * 3 + 2] - kp[(size_t)(pp * 3 + 2) * N];
                pt += ((dx * dx + dy * dy) + dz * dz) * shw[hh];
            }
            a += pt * (-0.5f);
            a += sqm;
            att[hh * N + j] = a;
        }
    }
    __syncthreads();
    // softmax over j, one wave per head (round-robin)
    for (int hh = wave; hh < H; hh += 4) {
        float* ar = att + hh * N;
        float mx = -3.0e38f;
        for (int j = lane; j < N; j += 64) mx = fmaxf(mx, ar[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float s = 0.f;
        for (int j = lane; j < N; j += 64) { const float e = expf(ar[j] - mx); ar[j] = e; s += e; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        for (int j = lane; j < N; j += 64) ar[j] = ar[j] / s;
    }
    __syncthreads();
    float* crow = cat + (size_t)row * (HC + H * Pv * 4 + H * c_p);
    // o (HC outputs) and o_pt (H*Pv*3 outputs, global frame)
    const int npt = H * Pv * 3;
    for (int u = tid; u < HC + npt; u += 256) {
        float acc = 0.f;
        if (u < HC) {
            const float* ar = att + (u / C) * N;
            const float* vv = v + (size_t)b * N * HC + u;
            for (int j = 0; j < N; ++j) acc += ar[j] * vv[(size_t)j * HC];
            crow[u] = acc;
        } else {
            const int w = u - HC;
            const float* ar = att + (w / (Pv * 3)) * N;
            const float* vv = vp + (size_t)b * N * npt + w;
            for (int j = 0; j < N; ++j) acc += ar[j] * vv[(size_t)j * npt];
            opt[w] = acc;
        }
    }
    // o_pair: thread = (channel quad, j-group); 8 j-groups reduced through LDS
    {
        const int nq4 = c_p >> 2;                 // 32 float4 per row
        const int c4 = tid % nq4, jg = tid / nq4; // jg in [0, 8)
        constexpr int HMAX = 16;
        float4 acc[HMAX];
#pragma unroll
        for (int hh = 0; hh < HMAX; ++hh) acc[hh] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* zr = z + ((size_t)row * N) * c_p + c4 * 4;
        for (int j = jg; j < N; j += 8) {
            const float4 zz = *reinterpret_cast<const float4*>(zr + (size_t)j * c_p);
#pragma unroll
            for (int hh = 0; hh < HMAX; ++hh) {
                if (hh < H) {
                    const float a = att[hh * N + j];
                    acc[hh].x += a * zz.x; acc[hh].y += a * zz.y; acc[hh].z += a * zz.z; acc[hh].w += a * zz.w;
                }
            }
        }
        __syncthreads();     // q/qp scratch is dead, o/o_pt loops are done: reuse scr as red
#pragma unroll
        for (int hh = 0; hh < HMAX; ++hh)
            if (hh < H) *reinterpret_cast<float4*>(scr + ((size_t)jg * H + hh) * c_p + c4 * 4) = acc[hh];
    }
    __syncthreads();
    {
        float* op = crow + HC + H * Pv * 4;
        const int tot = H * c_p;
        for (int u = tid; u < tot; u += 256) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += scr[(size_t)g * tot + u];
            op[u] = s;
        }
    }
    // o_pt -> local frame (T.invert_apply, affine_utils.py:123-126), norms
    {
        const int np = H * Pv;
        const float* R = rots + (size_t)row * 9;
        const float* t = trans + (size_t)row * 3;
        for (int u = tid; u < np; u += 256) {
            const float x = opt[u * 3 + 0] - t[0], y = opt[u * 3 + 1] - t[1], zc = opt[u * 3 + 2] - t[2];
            const float lx = R[0] * x + R[3] * y + R[6] * zc;
            const float ly = R[1] * x + R[4] * y + R[7] * zc;
            const float lz = R[2] * x + R[5] * y + R[8] * zc;
            crow[HC + u] = lx;
            crow[HC + np + u] = ly;
            crow[HC + 2 * np + u] = lz;
            crow[HC + 3 * np + u] = sqrtf(((lx * lx + ly * ly) + lz * lz) + 1e-8f);
        }
    }
}

// ---------------------------------------------------------------------------
// IPA attention core, compile-time shapes (the Genie 2 config H=12, C=16, Pq=4, Pv=8, c_p=128).
// Same maths as k_ipa_attn; every inner loop has a constant trip count so hipcc batches the
// (L2-resident) operand loads instead of paying one round trip per scalar, the j loops are
// unrolled x8 / x4, the o_pair partials of the two half-waves are merged with a lane-32 shuffle
// before LDS (24 KB instead of 48 KB -> 4 work-groups per CU).
// Dynamic LDS: att[H][NP8] | red[4][H][128] (aliases the q / q_pts scratch) | opt[H*Pv*3].
// ---------------------------------------------------------------------------
template <int H, int C, int PQ, int PV>
__global__ __launch_bounds__(256) void k_ipa_attn_t(const float* __restrict__ proj, int ldp, const float* __restrict__ kT,
                                                    const float* __restrict__ v, const float* __restrict__ qp,
                                                    const float* __restrict__ kpT, const float* __restrict__ vp,
                                                    const float* __restrict__ bias, const float* __restrict__ z,
                                                    const float* __restrict__ rots, const float* __restrict__ trans,
                                                    const float* __restrict__ rmask, const float* __restrict__ head_w,
                                                    float* __restrict__ cat, int B, int N, int layer) {
    constexpr int CP = 128, HC = H * C, NQP = H * PQ * 3, NPT = H * PV * 3, NCAT = HC + H * PV * 4 + H * CP;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int NP8 = (N + 7) & ~7;
    float* att = sm;                            // [H][NP8], zero padded
    float* scr = sm + H * NP8;                  // q | q_pts | hw ; later red[4][H][CP]
    float* opt = scr + 4 * H * CP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x, b = row / N, i = row % N;
    float* sq = scr;
    float* sqp = scr + HC;
    float* shw = sqp + NQP;
    for (int u = tid; u < HC; u += 256) sq[u] = proj[(size_t)row * ldp + u];
    for (int u = tid; u < NQP; u += 256) sqp[u] = qp[(size_t)row * NQP + u];
    if (tid < H) {
        const float g = head_w[tid];
        const float sp = (g > 20.f) ? g : log1pf(expf(g));
        shw[tid] = sp * sqrtf(1.0f / (3.0f * ((float)PQ * 9.0f / 2.0f)));
    }
    __syncthreads();
    const float s_qk = sqrtf(1.0f / (3.0f * (float)C)), s_b = sqrtf(1.0f / 3.0f);
    const float mi = rmask[row];
    for (int j = tid; j < NP8; j += 256) {
        const int jc = min(j, N - 1);
        const float sqm = 1e5f * (mi * rmask[b * N + jc] - 1.0f);
        float bia[H];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) bia[hh] = bias[((((size_t)layer * H + hh) * B + b) * N + i) * N + jc];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
            const float* kc = kT + (((size_t)b * H + hh) * C) * N + jc;
            float kv[C];
#pragma unroll
            for (int c = 0; c < C; ++c) kv[c] = kc[(size_t)c * N];
            const float* kp = kpT + ((((size_t)b * H + hh) * PQ) * 3) * N + jc;
            float kpv[PQ * 3];
#pragma unroll
            for (int c = 0; c < PQ * 3; ++c) kpv[c] = kp[(size_t)c * N];
            float qk = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) qk += sq[hh * C + c] * kv[c];
            float a = qk * s_qk;
            a += s_b * bia[hh];
            float pt = 0.f;
#pragma unroll
            for (int pp = 0; pp < PQ; ++pp) {
                const float dx = sqp[(hh * PQ + pp) * 3 + 0] - kpv[pp * 3 + 0];
                const float dy = sqp[(hh * PQ + pp) * 3 + 1] - kpv[pp * 3 + 1];
                const float dz = sqp[(hh * PQ + pp) * 3 + 2] - kpv[pp * 3 + 2];
                pt += ((dx * dx + dy * dy) + dz * dz) * shw[hh];
            }
            a += pt * (-0.5f);
            a += sqm;
            att[hh * NP8 + j] = (j < N) ? a : -3.0e38f;
        }
    }
    __syncthreads();
    for (int hh = wave; hh < H; hh += 4) {
        float* ar = att + hh * NP8;
        float mx = -3.0e38f;
        for (int j = lane; j < N; j += 64) mx = fmaxf(mx, ar[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float s = 0.f;
        for (int j = lane; j < N; j += 64) { const float e = expf(ar[j] - mx); ar[j] = e; s += e; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        for (int j = lane; j < NP8; j += 64) ar[j] = (j < N) ? ar[j] / s : 0.f;
    }
    __syncthreads();
    float* crow = cat + (size_t)row * NCAT;
    // o and o_pt: 480 outputs, 2 per thread handled together, j unrolled x8 (weights: ds_read_b128 pairs,
    // values: 16 independent L2 loads in flight per step)
    {
        const int u0 = tid, u1 = tid + 256;
        const bool has1 = u1 < HC + NPT;
        const bool o0 = u0 < HC, o1 = u1 < HC;                       // u1 >= 256 > HC for the base shape, kept general
        const int w0 = o0 ? u0 : u0 - HC, w1 = has1 ? (o1 ? u1 : u1 - HC) : 0;
        const float* ar0 = att + (o0 ? (u0 / C) : (w0 / (PV * 3))) * NP8;
        const float* ar1 = att + (o1 ? (u1 / C) : (w1 / (PV * 3))) * NP8;
        const float* vv0 = o0 ? v + (size_t)b * N * HC + u0 : vp + (size_t)b * N * NPT + w0;
        const float* vv1 = o1 ? v + (size_t)b * N * HC + u1 : vp + (size_t)b * N * NPT + w1;
        const int ld0 = o0 ? HC : NPT, ld1 = o1 ? HC : NPT;
        float acc0 = 0.f, acc1 = 0.f;
        for (int j0 = 0; j0 < NP8; j0 += 8) {
            float x0[8], x1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int jc = min(j0 + k, N - 1);
                x0[k] = vv0[(size_t)jc * ld0];
                x1[k] = vv1[(size_t)jc * ld1];
            }
            const float4 a0 = *reinterpret_cast<const float4*>(ar0 + j0), a1 = *reinterpret_cast<const float4*>(ar0 + j0 + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(ar1 + j0), b1 = *reinterpret_cast<const float4*>(ar1 + j0 + 4);
            acc0 += a0.x * x0[0]; acc0 += a0.y * x0[1]; acc0 += a0.z * x0[2]; acc0 += a0.w * x0[3];
            acc0 += a1.x * x0[4]; acc0 += a1.y * x0[5]; acc0 += a1.z * x0[6]; acc0 += a1.w * x0[7];
            acc1 += b0.x * x1[0]; acc1 += b0.y * x1[1]; acc1 += b0.z * x1[2]; acc1 += b0.w * x1[3];
            acc1 += b1.x * x1[4]; acc1 += b1.y * x1[5]; acc1 += b1.z * x1[6]; acc1 += b1.w * x1[7];
        }
        if (o0) crow[u0] = acc0; else opt[w0] = acc0;
        if (has1) { if (o1) crow[u1] = acc1; else opt[w1] = acc1; }
    }
    // o_pair: thread = (channel quad c4, j-group jg of 8); the p row is streamed once
    {
        const int c4 = tid & 31, jg = tid >> 5;
        float4 acc[H];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) acc[hh] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* zr = z + ((size_t)row * N) * CP + c4 * 4;
        for (int j0 = jg; j0 < N; j0 += 64) {        // 8 rows (j0, j0+8, ..., j0+56) in flight
            float4 zz[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) zz[k] = *reinterpret_cast<const float4*>(zr + (size_t)min(j0 + 8 * k, N - 1) * CP);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = j0 + 8 * k;
                if (j < N) {
#pragma unroll
                    for (int hh = 0; hh < H; ++hh) {
                        const float a = att[hh * NP8 + j];
                        acc[hh].x += a * zz[k].x; acc[hh].y += a * zz[k].y; acc[hh].z += a * zz[k].z; acc[hh].w += a * zz[k].w;
                    }
                }
            }
        }
        // merge the two half-waves (jg even / odd share a wave), then one LDS slab per wave
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {
            acc[hh].x += __shfl_xor(acc[hh].x, 32); acc[hh].y += __shfl_xor(acc[hh].y, 32);
            acc[hh].z += __shfl_xor(acc[hh].z, 32); acc[hh].w += __shfl_xor(acc[hh].w, 32);
        }
        __syncthreads();     // q / q_pts scratch is dead: reuse as red
        if (lane < 32) {
#pragma unroll
            for (int hh = 0; hh < H; ++hh) *reinterpret_cast<float4*>(scr + ((size_t)wave * H + hh) * CP + c4 * 4) = acc[hh];
        }
    }
    __syncthreads();
    {
        float* op = crow + HC + H * PV * 4;
        constexpr int tot = H * CP;
        for (int u = tid; u < tot; u += 256) op[u] = (scr[u] + scr[tot + u]) + (scr[2 * tot + u] + scr[3 * tot + u]);
    }
    {
        constexpr int np = H * PV;
        const float* R = rots + (size_t)row * 9;
        const float* t = trans + (size_t)row * 3;
        for (int u = tid; u < np; u += 256) {
            const float x = opt[u * 3 + 0] - t[0], y = opt[u * 3 + 1] - t[1], zc = opt[u * 3 + 2] - t[2];
            const float lx = R[0] * x + R[3] * y + R[6] * zc;
            const float ly = R[1] * x + R[4] * y + R[7] * zc;
            const float lz = R[2] * x + R[5] * y + R[8] * zc;
            crow[HC + u] = lx;
            crow[HC + np + u] = ly;
            crow[HC + 2 * np + u] = lz;
            crow[HC + 3 * np + u] = sqrtf(((lx * lx + ly * ly) + lz * lz) + 1e-8f);
        }
    }
}

// ---------------------------------------------------------------------------
// Backbone update + frame composition (modules/backbone_update.py:40-66,
// affine_utils.py:109-116,299-334), one wave per residue:
//   [b c d | t] = W s + bias; q = (1,b,c,d)/sqrt(1+b^2+c^2+d^2);
//   R <- R R(q);  t <- R t_upd + t.       Last layer also writes
//   z = trans_in - t / rescale (model/model.py:184-187).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bb_update(const float* __restrict__ s, int c_s, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ rots,
                                                   float* __restrict__ trans, int M, const float* __restrict__ trans_in,
                                                   float* __restrict__ z_out, float inv_rescale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = lane; c < c_s; c += 64) {
        const float x = s[(size_t)row * c_s + c];
#pragma unroll
        for (int o = 0; o < 6; ++o) acc[o] += x * w[o * c_s + c];
    }
#pragma unroll
    for (int o = 0; o < 6; ++o) {
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) acc[o] += __shfl_xor(acc[o], sft);
        acc[o] += bias[o];
    }
    if (lane == 0) {
        const float qb = acc[0], qc = acc[1], qd = acc[2];
        const float den = sqrtf(((qb * qb + qc * qc) + qd * qd) + 1.0f);
        const float a = 1.0f / den, bq = qb / den, c = qc / den, d = qd / den;
        float U[9];
        U[0] = a * a + bq * bq - c * c - d * d; U[1] = 2 * bq * c - 2 * a * d;          U[2] = 2 * bq * d + 2 * a * c;
        U[3] = 2 * bq * c + 2 * a * d;          U[4] = a * a - bq * bq + c * c - d * d; U[5] = 2 * c * d - 2 * a * bq;
        U[6] = 2 * bq * d - 2 * a * c;          U[7] = 2 * c * d + 2 * a * bq;          U[8] = a * a - bq * bq - c * c + d * d;
        float* R = rots + (size_t)row * 9;
        float* t = trans + (size_t)row * 3;
        float Rn[9], r[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) r[k] = R[k];
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) Rn[x * 3 + y] = r[x * 3 + 0] * U[0 * 3 + y] + r[x * 3 + 1] * U[1 * 3 + y] + r[x * 3 + 2] * U[2 * 3 + y];
        const float tx = r[0] * acc[3] + r[1] * acc[4] + r[2] * acc[5] + t[0];
        const float ty = r[3] * acc[3] + r[4] * acc[4] + r[5] * acc[5] + t[1];
        const float tz = r[6] * acc[3] + r[7] * acc[4] + r[8] * acc[5] + t[2];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = Rn[k];
        t[0] = tx; t[1] = ty; t[2] = tz;
        if (z_out) {
            z_out[(size_t)row * 3 + 0] = trans_in[(size_t)row * 3 + 0] - tx * inv_rescale;
            z_out[(size_t)row * 3 + 1] = trans_in[(size_t)row * 3 + 1] - ty * inv_rescale;
            z_out[(size_t)row * 3 + 2] = trans_in[(size_t)row * 3 + 2] - tz * inv_rescale;
        }
    }
}

// ---------------------------------------------------------------------------
// Row-local tail of a structure layer in ONE launch (hx arithmetic, c_s = 32 NW):
//   s2 = LN_ipa(s1);  s = LN_tr(s2 + W3 relu(W2 relu(W1 s2)));  frames <- frames o BackboneUpdate(s)
// (structure_net.py:108-116, structure_transition.py:34-70, backbone_update.py:40-66).  Everything between the IPA output
// projection and the next layer's input projection touches one residue at a time, and at B N = 2048 rows each of the six
// launches it replaces (three Linears, two LayerNorms, the frame update) was all fixed latency (22 + 22 + 22 + 6 + 6 + 7 us).
// A work-group owns 32 rows and NW waves; wave w owns output columns 32w..32w+31 of every Linear.  Activations stay in two
// f32 LDS tiles; each Linear is the k_gemm_rows_hx loop (same block-floating-point split per (row, 64-wide K chunk), same
// weight-unit ring, same accumulation order -- results are bit-identical to the separate launches) with its A chunks split
// out of the LDS tile instead of HBM, and writes its result tile in place once the last chunk has been split.
// ---------------------------------------------------------------------------
#define SR_PD 4
#define SR_KSPLIT 3          // split-K slices of the IPA output projection summed by the loader (nsplit is 0 or this;
                             // 6 slices: output projection + tail 0.120 -> 0.126 ms per layer, the loader's extra reads cost more)
#define SR_LD 388            // floats per row of an activation tile (384 + 4: the two half-waves of a D store hit disjoint banks)
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_struct_rows_hx(
    // (parameters of later stages are deliberately not __restrict__: hipcc would hoist their loads -- 60 registers of LayerNorm
    //  and BackboneUpdate weights -- to the top of the kernel, across the barriers, and spill the GEMM loop)
    const float* __restrict__ x, int nsplit, size_t zstride, const float* __restrict__ xbias, const float* xres,
    const float* g1, const float* be1, const unsigned char* __restrict__ W1, float i1, const float* b1,
    const unsigned char* __restrict__ W2, float i2, const float* b2, const unsigned char* __restrict__ W3, float i3, const float* b3,
    const float* g2, const float* be2, const float* bbw, const float* bbb, float* s_out, float* rots, float* trans, int M,
    const float* trans_in, float* z_out, float inv_rescale, int nrb, unsigned long long* ts) {
    constexpr int C = NW * 32, NT = NW * 64, KC = C / 16, NKC = C / 64;
    static_assert(C % 64 == 0 && NT >= 512 && NT <= 1024 && C + 4 <= SR_LD, "shape");
    extern __shared__ __attribute__((aligned(16))) float srm[];
    float* T0 = srm;
    float* T1 = T0 + 32 * SR_LD;
    unsigned char* sa = reinterpret_cast<unsigned char*>(T1 + 32 * SR_LD);       // [chunk][hi | lo][32 * GH_ROWB]
    float* sinv = reinterpret_cast<float*>(sa + NKC * 2 * 32 * GH_ROWB);         // [chunk][32]
    float* frm = sinv + NKC * 32;                                                  // [32][12] frames of the rows (R | t)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.x * 32;
    const bool splitter = tid < 512;
    int ts_n = 0;                                  // developer aid (GENIE_SR_TS=1): s_memtime at the phase boundaries of work-group 0
    auto stamp = [&]() { if (ts && blockIdx.x == 0 && tid == 0) ts[ts_n++] = __builtin_amdgcn_s_memtime(); };

    if ((int)blockIdx.x >= nrb) {
        // The layer's three weight images (1.7 MB) were last used a whole denoiser step ago: they come from HBM, and a row
        // work-group streaming them through its small register ring would pay that latency chunk after chunk.  The CUs this
        // launch leaves idle pull them into L2 instead: work-groups are dealt round-robin to the 8 XCDs (one L2 each), so
        // prefetcher pf = 8 slice + xcd reads slice `slice` of every image into the L2 of its own XCD.
        const int pf = (int)blockIdx.x - nrb, nsl = max(1, (int)(gridDim.x - nrb) >> 3), slice = min(pf >> 3, nsl - 1);
        constexpr int IMG = KC * NW * 2048;
        const unsigned char* imgs[3] = {W1, W2, W3};
#pragma unroll
        for (int w = 0; w < 3; ++w)
            for (int o = slice * NT * 16 + tid * 16; o < IMG; o += nsl * NT * 16) {
                const float4 v = *reinterpret_cast<const float4*>(imgs[w] + o);
                asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            }
        return;
    }

    // weight units through a wave-uniform base (SGPR pair) + one lane offset; ring of SR_PD = 4 units (one chunk ahead: three
    // waves per SIMD cover an L2 round trip).  The first units of a Linear are issued before the stage that precedes it.
    const unsigned loff = lane * 16;
    const int wsel = __builtin_amdgcn_readfirstlane(wave);
    v4f wh[SR_PD], wl[SR_PD];
    auto unit_issue = [&](const unsigned char* wb, v4f& hi, v4f& lo, int u) {
        const unsigned char* p = wb + (size_t)min(u, KC - 1) * 2048;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(hi) : "v"(loff), "s"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(lo) : "v"(loff), "s"(p) : "memory");
    };
    auto ring_fill = [&](const unsigned char* Wx) {
        const unsigned char* wb = Wx + (size_t)wsel * KC * 2048;
#pragma unroll
        for (int u = 0; u < SR_PD; ++u) unit_issue(wb, wh[u], wl[u], u);
    };
    stamp();
    ring_fill(W1);

    // s1 = sum of the output projection's split-K slices + its bias + the residual s (nsplit = 0: x is s1 itself); every load
    // of a thread's four float4 is issued before the first is used
    {
        constexpr int NI = 32 * (C / 4) / NT;
        static_assert(32 * (C / 4) % NT == 0, "loader shape");
        constexpr int NB = SR_KSPLIT <= 3 ? NI : 2;   // float4 per thread and batch: (SR_KSPLIT + 2) NB loads in flight
        static_assert(NI % NB == 0, "loader batches");
#pragma unroll
        for (int i0 = 0; i0 < NI; i0 += NB) {
            float4 v[NB], w[SR_KSPLIT - 1][NB], rq[NB], bq[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int u = tid + (i0 + i) * NT, r = u / (C / 4), q = u - r * (C / 4);
                const size_t o = (size_t)min(r0 + r, M - 1) * C + q * 4;
                v[i] = *reinterpret_cast<const float4*>(x + o);
                if (nsplit) {
#pragma unroll
                    for (int zz = 1; zz < SR_KSPLIT; ++zz) w[zz - 1][i] = *reinterpret_cast<const float4*>(x + zz * zstride + o);
                    rq[i] = *reinterpret_cast<const float4*>(xres + o);
                    bq[i] = *reinterpret_cast<const float4*>(xbias + q * 4);
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int u = tid + (i0 + i) * NT, r = u / (C / 4), q = u - r * (C / 4);
                float4 t = v[i];
                if (nsplit) {
#pragma unroll
                    for (int zz = 1; zz < SR_KSPLIT; ++zz) { t.x += w[zz - 1][i].x; t.y += w[zz - 1][i].y; t.z += w[zz - 1][i].z; t.w += w[zz - 1][i].w; }
                    t.x = (t.x + bq[i].x) + rq[i].x; t.y = (t.y + bq[i].y) + rq[i].y;
                    t.z = (t.z + bq[i].z) + rq[i].z; t.w = (t.w + bq[i].w) + rq[i].w;
                }
                *reinterpret_cast<float4*>(T0 + r * SR_LD + q * 4) = t;
            }
        }
    }
    // frames of the tile's rows (R 9 floats, t 3) for the composition at the end: fetched now, used 40 us later
    if (tid < 32 * 12) {
        const int r = tid / 12, k = tid - r * 12, row = min(r0 + r, M - 1);
        frm[tid] = k < 9 ? rots[(size_t)row * 9 + k] : trans[(size_t)row * 3 + (k - 9)];
    }
    __syncthreads();

    // 16 lanes per row, four rows per wave: the 32 rows of the tile in one round on waves 0..7; reductions are four DPP steps
    auto dpp_add = [](float m, auto ctrl_tag) {
        constexpr int CTRL = decltype(ctrl_tag)::value;
        return m + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), CTRL, 0xf, 0xf, false));
    };
    auto row16_sum = [&](float m) {
        m = dpp_add(m, std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
        m = dpp_add(m, std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
        m = dpp_add(m, std::integral_constant<int, 0x124>{});      // row_ror:4
        m = dpp_add(m, std::integral_constant<int, 0x128>{});      // row_ror:8
        return m;
    };
    const int l16 = tid & 15, rrow = tid >> 4;                    // row of this lane group (valid for tid < 512)
    // nn.LayerNorm of the tile's rows in place (two-pass, as k_layernorm_rows; a lane owns columns 4 l16 + 64 q .. + 3)
    auto layernorm = [&](float* T, const float* g, const float* bta, float* gout) {
        if (splitter) {
            float4 v[C / 64];
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < C / 64; ++q) {
                v[q] = *reinterpret_cast<const float4*>(T + rrow * SR_LD + 64 * q + 4 * l16);
                sum += (v[q].x + v[q].y) + (v[q].z + v[q].w);
            }
            const float mean = row16_sum(sum) / (float)C;
            float ss = 0.f;
#pragma unroll
            for (int q = 0; q < C / 64; ++q) {
                const float dx = v[q].x - mean, dy = v[q].y - mean, dz = v[q].z - mean, dw = v[q].w - mean;
                ss += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
            const float rstd = 1.0f / sqrtf(row16_sum(ss) / (float)C + GENIE_LN_EPS);
#pragma unroll
            for (int q = 0; q < C / 64; ++q) {
                const int c = 64 * q + 4 * l16;
                const float4 gq = *reinterpret_cast<const float4*>(g + c), bq = *reinterpret_cast<const float4*>(bta + c);
                float4 y;
                y.x = (v[q].x - mean) * rstd * gq.x + bq.x; y.y = (v[q].y - mean) * rstd * gq.y + bq.y;
                y.z = (v[q].z - mean) * rstd * gq.z + bq.z; y.w = (v[q].w - mean) * rstd * gq.w + bq.w;
                *reinterpret_cast<float4*>(T + rrow * SR_LD + c) = y;
                if (gout && r0 + rrow < M) *reinterpret_cast<float4*>(gout + (size_t)(r0 + rrow) * C + c) = y;
            }
        }
        __syncthreads();
    };

    auto dpp_max = [](float m, auto ctrl_tag) {
        constexpr int CTRL = decltype(ctrl_tag)::value;
        return fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), CTRL, 0xf, 0xf, false)));
    };
    // block-floating-point split of one float4 of A (16 consecutive lanes = one row's 64-wide K chunk) into the operand planes
    auto split_store = [&](float4 v, int row, int kc) {
        float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        m = dpp_max(m, std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
        m = dpp_max(m, std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
        m = dpp_max(m, std::integral_constant<int, 0x124>{});      // row_ror:4
        m = dpp_max(m, std::integral_constant<int, 0x128>{});      // row_ror:8  -> largest |a| of the row's chunk
        const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 255u);
        const bool tiny = e < 16;
        const float sc = tiny ? 1.0f : __builtin_bit_cast(float, (unsigned)(268 - e) << 23);
        const float isc = tiny ? 1.0f : __builtin_bit_cast(float, (unsigned)(e - 14) << 23);
        unsigned h01, l01, h23, l23;
        hx_split2(v.x, v.y, sc, h01, l01);
        hx_split2(v.z, v.w, sc, h23, l23);
        unsigned char* d = sa + kc * (2 * 32 * GH_ROWB) + row * GH_ROWB + l16 * 8;
        *reinterpret_cast<uint2*>(d) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(d + 32 * GH_ROWB) = make_uint2(l01, l23);
        if (l16 == 0) sinv[kc * 32 + row] = isc;
    };

    // out = act(A W^T inv_sw + bias) (+ res), tiles in LDS; `out` may alias A.  All NKC chunks of A are split up front (one
    // barrier), then every wave runs its 72 MFMAs against the weight ring without meeting the others again.
    auto linear = [&](const float* A, const unsigned char* __restrict__ Wx, const unsigned char* Wnext, float inv_sw,
                      const float* bias, bool relu, const float* res, float* out) {
        const unsigned char* wb = Wx + (size_t)wsel * KC * 2048;
#pragma unroll
        for (int i = 0; i < 32 * (C / 4) / NT; ++i) {
            const int u = tid + i * NT, rc = u >> 4, row = rc / NKC, kc = rc - row * NKC;      // u = ((row NKC + kc) 16 + l16)
            split_store(*reinterpret_cast<const float4*>(A + row * SR_LD + kc * 64 + l16 * 4), row, kc);
        }
        __syncthreads();
        f32x16 acc = zero16();
        const int foff = (lane & 31) * GH_ROWB + (lane >> 5) * 16;
#pragma unroll 1
        for (int kc = 0; kc < NKC; ++kc) {
            const unsigned char* cur = sa + kc * (2 * 32 * GH_ROWB) + foff;
            f32x16 part = zero16();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const h8 ah = *reinterpret_cast<const h8*>(cur + q * 32);
                const h8 al = *reinterpret_cast<const h8*>(cur + 32 * GH_ROWB + q * 32);
                wf_wait<2 * SR_PD - 2>(wh[q], wl[q]);                           // younger: the 3 other ring units
                const h8 bh = __builtin_bit_cast(h8, wh[q]), bl = __builtin_bit_cast(h8, wl[q]);
                MFH3(ah, al, bh, bl, part);
                __builtin_amdgcn_sched_barrier(0);
                unit_issue(wb, wh[q], wl[q], kc * 4 + q + SR_PD);
            }
            const float* si = sinv + kc * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 iv = *reinterpret_cast<const float4*>(si + 8 * g);
                acc[4 * g] = fmaf(part[4 * g], iv.x, acc[4 * g]); acc[4 * g + 1] = fmaf(part[4 * g + 1], iv.y, acc[4 * g + 1]);
                acc[4 * g + 2] = fmaf(part[4 * g + 2], iv.z, acc[4 * g + 2]); acc[4 * g + 3] = fmaf(part[4 * g + 3], iv.w, acc[4 * g + 3]);
            }
        }
#pragma unroll
        for (int u = 0; u < SR_PD; ++u) wf_wait<0>(wh[u], wl[u]);
        if (Wnext) ring_fill(Wnext);
        // A was consumed by the split before the barrier above: the result may overwrite it
        const int col = wave * 32 + (lane & 31);
        const float bc = bias[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lane);
            float v = fmaf(acc[r], inv_sw, bc);
            if (relu) v = fmaxf(v, 0.f);
            if (res) v += res[row * SR_LD + col];
            out[row * SR_LD + col] = v;
        }
        __syncthreads();
    };

    stamp();
    layernorm(T0, g1, be1, nullptr);                      // T0 = s2
    stamp();
    linear(T0, W1, W2, i1, b1, true, nullptr, T1);
    stamp();
    linear(T1, W2, W3, i2, b2, true, nullptr, T1);
    stamp();
    linear(T1, W3, nullptr, i3, b3, false, T0, T1);
    stamp();
    for (int u = tid; u < 6 * C / 4; u += NT)             // BackboneUpdate weights -> T0 (visible after LayerNorm's barrier)
        reinterpret_cast<float4*>(T0)[u] = reinterpret_cast<const float4*>(bbw)[u];
    layernorm(T1, g2, be2, s_out);                        // T1 = s
    stamp();

    // BackboneUpdate + frame composition (k_bb_update's arithmetic), 16 lanes per row; the update weights (6 x C) sit in T0,
    // whose last reader was the third Linear
    if (splitter) {
        float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < C / 64; ++q) {
            const int c = 64 * q + 4 * l16;
            const float4 xv = *reinterpret_cast<const float4*>(T1 + rrow * SR_LD + c);
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                const float4 wv = *reinterpret_cast<const float4*>(T0 + o * C + c);
                acc[o] += (xv.x * wv.x + xv.y * wv.y) + (xv.z * wv.z + xv.w * wv.w);
            }
        }
#pragma unroll
        for (int o = 0; o < 6; ++o) acc[o] = row16_sum(acc[o]) + bbb[o];
        const int row = r0 + rrow;
        if (l16 == 0 && row < M) {
            const float qb = acc[0], qc = acc[1], qd = acc[2];
            const float den = sqrtf(((qb * qb + qc * qc) + qd * qd) + 1.0f);
            const float a = 1.0f / den, bq = qb / den, c = qc / den, d = qd / den;
            float U[9];
            U[0] = a * a + bq * bq - c * c - d * d; U[1] = 2 * bq * c - 2 * a * d;          U[2] = 2 * bq * d + 2 * a * c;
            U[3] = 2 * bq * c + 2 * a * d;          U[4] = a * a - bq * bq + c * c - d * d; U[5] = 2 * c * d - 2 * a * bq;
            U[6] = 2 * bq * d - 2 * a * c;          U[7] = 2 * c * d + 2 * a * bq;          U[8] = a * a - bq * bq - c * c + d * d;
            float* R = rots + (size_t)row * 9;
            float* t = trans + (size_t)row * 3;
            float Rn[9], rr[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) rr[k] = frm[rrow * 12 + k];
#pragma unroll
            for (int xx = 0; xx < 3; ++xx)
#pragma unroll
                for (int y = 0; y < 3; ++y) Rn[xx * 3 + y] = rr[xx * 3 + 0] * U[0 * 3 + y] + rr[xx * 3 + 1] * U[1 * 3 + y] + rr[xx * 3 + 2] * U[2 * 3 + y];
            const float tx = rr[0] * acc[3] + rr[1] * acc[4] + rr[2] * acc[5] + frm[rrow * 12 + 9];
            const float ty = rr[3] * acc[3] + rr[4] * acc[4] + rr[5] * acc[5] + frm[rrow * 12 + 10];
            const float tz = rr[6] * acc[3] + rr[7] * acc[4] + rr[8] * acc[5] + frm[rrow * 12 + 11];
#pragma unroll
            for (int k = 0; k < 9; ++k) R[k] = Rn[k];
            t[0] = tx; t[1] = ty; t[2] = tz;
            if (z_out) {
                z_out[(size_t)row * 3 + 0] = trans_in[(size_t)row * 3 + 0] - tx * inv_rescale;
                z_out[(size_t)row * 3 + 1] = trans_in[(size_t)row * 3 + 1] - ty * inv_rescale;
                z_out[(size_t)row * 3 + 2] = trans_in[(size_t)row * 3 + 2] - tz * inv_rescale;
            }
        }
    }
    stamp();
}

// ---------------------------------------------------------------------------
// Reverse-loop update + Frenet frames, one work-group per structure
// (sampler/base.py:249-282, utils/geo_utils.py:21-85).
//   mode 0: frames of `trans` only.
//   mode 1: trans <- ((trans - w_z z)/sqrt(alpha_t)) * mask [+ scale sqrt(beta_t) eps, * mask], then frames.
// Chain starts/ends copy their neighbour exactly as the reference's two
// sequential fix-up loops do (including what they do for consecutive chain ends).
// Dynamic LDS: x[N][3] | tv[N][3] | base[N][9] | st1[N][9] | flags.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_p_sample_frenet(int mode, float* __restrict__ trans, float* __restrict__ rots,
                                                         const float* __restrict__ z, const float* __restrict__ eps,
                                                         const int32_t* __restrict__ rmask, const int32_t* __restrict__ cidx,
                                                         int N, float alpha, float sqrt_alpha, float somac, float sqrt_beta,
                                                         float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* x = sm;
    float* tv = x + 3 * N;
    float* base = tv + 3 * N;
    float* st1 = base + 9 * N;
    __shared__ int s_len;
    const int b = blockIdx.x, tid = threadIdx.x;
    float* tr = trans + (size_t)b * N * 3;
    if (tid == 0) s_len = 0;
    __syncthreads();
    int cnt = 0;
    for (int n = tid; n < N; n += 256) cnt += rmask[b * N + n];
    atomicAdd(&s_len, cnt);
    if (mode == 1) {
        const float w_z = (1.0f - alpha) / somac;
        const float inv_sa = 1.0f / sqrt_alpha;
        for (int u = tid; u < 3 * N; u += 256) {
            const float m = (float)rmask[b * N + u / 3];
            float v = inv_sa * (tr[u] - w_z * z[(size_t)b * N * 3 + u]);
            v = v * m;
            if (eps) { v = v + scale * sqrt_beta * eps[(size_t)b * N * 3 + u]; v = v * m; }
            x[u] = v;
            tr[u] = v;
        }
    } else {
        for (int u = tid; u < 3 * N; u += 256) x[u] = tr[u];
    }
    __syncthreads();
    const int len = s_len;
    for (int n = tid; n < N - 1; n += 256) {
        const float dx = x[3 * n + 3] - x[3 * n], dy = x[3 * n + 4] - x[3 * n + 1], dz = x[3 * n + 5] - x[3 * n + 2];
        const float nr = sqrtf(1e-10f + ((dx * dx + dy * dy) + dz * dz));
        tv[3 * n] = dx / nr; tv[3 * n + 1] = dy / nr; tv[3 * n + 2] = dz / nr;
    }
    __syncthreads();
    for (int n = tid; n < N; n += 256) {
        float R[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
        // rots_[1:len-1] = tbn[0:len-2]; tbn[m] built from t[m], t[m+1]
        if (n >= 1 && n < len - 1) {
            const int m = n - 1;
            const float ax = tv[3 * m], ay = tv[3 * m + 1], az = tv[3 * m + 2];
            const float bx = tv[3 * m + 3], by = tv[3 * m + 4], bz = tv[3 * m + 5];
            float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
            const float nr = sqrtf(1e-10f + ((cx * cx + cy * cy) + cz * cz));
            cx /= nr; cy /= nr; cz /= nr;
            const float nx = cy * bz - cz * by, ny = cz * bx - cx * bz, nz = cx * by - cy * bx;
            R[0] = bx; R[1] = cx; R[2] = nx;
            R[3] = by; R[4] = cy; R[5] = ny;
            R[6] = bz; R[7] = cz; R[8] = nz;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) base[9 * n + k] = R[k];
    }
    __syncthreads();
    const int32_t* ch = cidx + b * N;
    // start-of-chain pass: every read sees the pre-pass value (the reference walks j upward and reads j+1)
    for (int n = tid; n < N; n += 256) {
        int srcn = n;
        if (n < len && (n == 0 || ch[n] != ch[n - 1])) srcn = n + 1;
#pragma unroll
        for (int k = 0; k < 9; ++k) st1[9 * n + k] = (srcn < N) ? base[9 * srcn + k] : ((k % 4 == 0) ? 1.f : 0.f);
    }
    __syncthreads();
    // end-of-chain pass: sequential dependence j <- j-1 resolved by walking back over consecutive ends
    float* ro = rots + (size_t)b * N * 9;
    for (int n = tid; n < N; n += 256) {
        int k = n;
        if (n < len) {
            while (k >= 0 && (k == len - 1 || ch[k] != ch[k + 1])) --k;
            if (k < 0) k = N - 1;        // python's rots_[-1]
        }
#pragma unroll
        for (int q = 0; q < 9; ++q) ro[9 * n + q] = st1[9 * k + q];
    }
}

__global__ void k_fill_i32(int32_t* p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_scale_copy(const float* __restrict__ in, float* __restrict__ out, int n, float s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * s;
}

// ---------------------------------------------------------------------------
// IPA attention core, Q queries per work-group (same maths as k_ipa_attn_t, which stays as the
// single-query form).  A work-group of 512 threads handles queries i0..i0+Q-1 of one structure:
// every K / K-point / V / V-point value fetched from L2 is used for Q queries, which divides the
// kernel's dominant traffic (0.85 MB of L2 reads per query in the single-query form: 10.7 TB/s of
// L2 bandwidth at 163 us) by Q.  Logits: thread = (j, half of the heads); o / o_pt: thread = output
// column, Q accumulators; o_pair: the Q p-rows are streamed one after the other.
// ---------------------------------------------------------------------------
//
// MF = 1 (hx arithmetic): o_pair[h][c] = sum_j a[h][j] p[j][c] runs on the matrix pipe.  Wave w owns query w>>1 and
// channel half w&1: per 32 values of j one v_mfma_f32_16x16x32_f16 triple per 16-channel block, A = the attention
// weights (12 heads padded to 16 rows, scaled by 2^14), B = p split into f16 halves under ONE power-of-two scale for
// the whole tensor (`pmax` = max |p|, left by k_ipa_bias, which reads all of p just before the first layer).  A lane's
// dwordx4 load carries channels 4m..4m+3 of row j = j0 + 8g + e (m = lane&15, g = lane>>4): register r of the eight
// loads e = 0..7 is exactly the B fragment of the block holding channels {4m + r}, so the stream needs no LDS and no
// shuffles, and the accumulators of the four blocks form the float4 a lane stores.  No cross-wave reduction, no barrier.
template <int H, int C, int PQ, int PV, int Q, int MF>
__global__ __launch_bounds__(MF ? 128 * Q : 512, MF ? 4 : 1) void k_ipa_attn_q(const float* __restrict__ proj, int ldp, const float* __restrict__ kT,
                                                    const float* __restrict__ v, const float* __restrict__ qp,
                                                    const float* __restrict__ kpT, const float* __restrict__ vp,
                                                    const float* __restrict__ bias, const float* __restrict__ z,
                                                    const float* __restrict__ rots, const float* __restrict__ trans,
                                                    const float* __restrict__ rmask, const float* __restrict__ head_w,
                                                    float* __restrict__ cat, int B, int N, int layer, int rev,
                                                    const unsigned* __restrict__ pmax, unsigned long long* ts, int b0,
                                                    const float* __restrict__ vf, const float* __restrict__ vmax) {
    constexpr int NT = MF ? 128 * Q : 512;       // 512 threads (Q = 4), or 1024 with Q = 8: two waves per query in o_pair, and every key / value
                                                 // fetched through the L1 then serves eight queries
    constexpr int CP = 128, HC = H * C, NQP = H * PQ * 3, NPT = H * PV * 3, NCAT = HC + H * PV * 4 + H * CP, HH = H / (NT / 256);
    static_assert(H % (NT / 256) == 0 && C % 4 == 0 && HC + NPT <= 512, "shape");
    static_assert(!MF || H <= 16, "matrix-pipe o_pair: 2 Q waves = Q queries x 2 channel halves");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int NP8 = (N + 7) & ~7;
    const int NPA = NP8 + 4;                    // row stride of att: rows 1 KiB apart (N = 256) would put the 16 rows of a matrix-pipe A
                                                // fragment (one b128 read per lane) in the same banks
    float* att = sm;                            // [Q][H][NPA], zero padded up to NP8
    float* sq = att + Q * H * NPA;              // [Q][HC]
    float* sqp = sq + Q * HC;                   // [Q][NQP]
    float* shw = sqp + Q * NQP;                 // [H] (16 reserved)
    float* opt = shw + 16;                      // [Q][NPT]
    float* red = opt + Q * NPT;                 // [4][H][CP]   (MF = 0 only)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int groups = (N + Q - 1) / Q;
    // `rev` alternates per layer (starting opposite to the pair-bias kernel's pass): p (268 MB at N = 256, batch 8) is re-read
    // by every layer, and a fixed direction would evict from the 256-MiB Infinity Cache exactly what the next reader needs first
    const int bid = rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int b = b0 + bid / groups, i0 = (bid % groups) * Q;       // the grid covers batch entries b0 .. (B stays the tensor's batch count)
    const int nq = min(Q, N - i0);
    int ts_n = 0;                                  // developer aid (GENIE_SR_TS=1): s_memtime at the phase boundaries of work-group 0
    auto stamp = [&]() { if (ts && blockIdx.x == 0 && tid == 0) ts[ts_n++] = __builtin_amdgcn_s_memtime(); };
    stamp();
    if (ts && tid == 0 && (blockIdx.x & 63) == 0) ts[16 + (blockIdx.x >> 6)] = __builtin_amdgcn_s_memtime();     // start of work-groups 0, 64, ...
    for (int u = tid; u < Q * HC; u += NT) { const int q = u / HC; sq[u] = proj[(size_t)(b * N + min(i0 + q, N - 1)) * ldp + (u - q * HC)]; }
    for (int u = tid; u < Q * NQP; u += NT) { const int q = u / NQP; sqp[u] = qp[(size_t)(b * N + min(i0 + q, N - 1)) * NQP + (u - q * NQP)]; }
    if (tid < H) {
        const float g = head_w[tid];
        const float sp = (g > 20.f) ? g : log1pf(expf(g));
        shw[tid] = sp * sqrtf(1.0f / (3.0f * ((float)PQ * 9.0f / 2.0f)));
    }
    __syncthreads();
    stamp();
    const float s_qk = sqrtf(1.0f / (3.0f * (float)C)), s_b = sqrtf(1.0f / 3.0f);
    {
        const int hg = tid >> 8;                 // heads hg*HH .. hg*HH + HH - 1
        for (int j = tid & 255; j < NP8; j += 256) {
            const int jc = min(j, N - 1);
            const float mj = rmask[b * N + jc];
            float sqm[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) sqm[q] = 1e5f * (rmask[b * N + min(i0 + q, N - 1)] * mj - 1.0f);
#pragma unroll 1
            for (int h2 = 0; h2 < HH; ++h2) {
                const int hh = hg * HH + h2;
                const float* kc = kT + (((size_t)b * H + hh) * C) * N + jc;
                float kv[C];
#pragma unroll
                for (int c = 0; c < C; ++c) kv[c] = kc[(size_t)c * N];
                const float* kp = kpT + ((((size_t)b * H + hh) * PQ) * 3) * N + jc;
                float kpv[PQ * 3];
#pragma unroll
                for (int c = 0; c < PQ * 3; ++c) kpv[c] = kp[(size_t)c * N];
                float bia[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) bia[q] = bias[((((size_t)layer * H + hh) * B + b) * N + min(i0 + q, N - 1)) * N + jc];
                const float hw = shw[hh];
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const float* qv = sq + q * HC + hh * C;
                    float qk = 0.f;
#pragma unroll
                    for (int c = 0; c < C; ++c) qk += qv[c] * kv[c];
                    float a = qk * s_qk;
                    a += s_b * bia[q];
                    const float* qpv = sqp + q * NQP + hh * PQ * 3;
                    float pt = 0.f;
#pragma unroll
                    for (int pp = 0; pp < PQ; ++pp) {
                        const float dx = qpv[pp * 3 + 0] - kpv[pp * 3 + 0];
                        const float dy = qpv[pp * 3 + 1] - kpv[pp * 3 + 1];
                        const float dz = qpv[pp * 3 + 2] - kpv[pp * 3 + 2];
                        pt += ((dx * dx + dy * dy) + dz * dz) * hw;
                    }
                    a += pt * (-0.5f);
                    a += sqm[q];
                    att[(q * H + hh) * NPA + j] = (j < N) ? a : -3.0e38f;
                }
            }
        }
    }
    __syncthreads();
    stamp();
    // softmax over j: 16 lanes per row (four rows per wave and round), float4 passes over LDS, reductions as four DPP steps
    // within the 16-lane row -- the one-wave-per-row form spent 12 dependent ds_bpermute round trips per row
    {
        auto dpp = [](float m, auto ctrl_tag) {
            constexpr int CTRL = decltype(ctrl_tag)::value;
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), CTRL, 0xf, 0xf, false));
        };
        const int l16 = lane & 15;
        for (int rr = wave * 4 + (lane >> 4); rr < Q * H; rr += NT / 16) {
            float* ar = att + rr * NPA;
            float mx = -3.0e38f;
            for (int j = 4 * l16; j < NP8; j += 64) {
                const float4 v = *reinterpret_cast<const float4*>(ar + j);
                mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            }
            mx = fmaxf(mx, dpp(mx, std::integral_constant<int, 0xB1>{})); mx = fmaxf(mx, dpp(mx, std::integral_constant<int, 0x4E>{}));
            mx = fmaxf(mx, dpp(mx, std::integral_constant<int, 0x124>{})); mx = fmaxf(mx, dpp(mx, std::integral_constant<int, 0x128>{}));
            float sum = 0.f;
            for (int j = 4 * l16; j < NP8; j += 64) {
                float4 v = *reinterpret_cast<const float4*>(ar + j);
                v.x = expf(v.x - mx); v.y = expf(v.y - mx); v.z = expf(v.z - mx); v.w = expf(v.w - mx);     // padding (-3e38) -> 0
                *reinterpret_cast<float4*>(ar + j) = v;
                sum += (v.x + v.y) + (v.z + v.w);
            }
            sum += dpp(sum, std::integral_constant<int, 0xB1>{}); sum += dpp(sum, std::integral_constant<int, 0x4E>{});
            sum += dpp(sum, std::integral_constant<int, 0x124>{}); sum += dpp(sum, std::integral_constant<int, 0x128>{});
            for (int j = 4 * l16; j < NP8; j += 64) {
                float4 v = *reinterpret_cast<const float4*>(ar + j);
                v.x /= sum; v.y /= sum; v.z /= sum; v.w /= sum;
                *reinterpret_cast<float4*>(ar + j) = v;
            }
        }
    }
    __syncthreads();
    stamp();
    // o and o_pt.  MF: on the matrix pipe -- per head one 16 x 16 x 32 tile per column block (16 channels of v, 24 point coordinates in two
    // blocks) and 32 keys: A = the attention rows of the Q queries (rows Q.. are zero), split under the scale 2^14; B = the f32
    // fragments k_ipa_prep left, split here under one scale per structure for v and one for the points (largest magnitude to
    // [2^13, 2^14)).  Wave w takes heads 3 (w >> 1) .. + 2 and the key half w & 1; the two partial sums per output meet in LDS
    // (osum) after o_pair.  As a per-thread dot product (below, MF = 0) this phase was the kernel's second largest: 2.5 k vector
    // instructions per thread.
    // Wave layout: Q = 4 (8 waves) -- heads 3 (w >> 1) .. + 2 and the key half w & 1, the two partial sums per output meet in LDS (osum)
    // after o_pair; Q = 8 (16 waves) -- wave w < 12 takes head w over all keys and writes its results itself.
    constexpr int JP = MF ? (NT / 64 >= H ? 1 : 2) : 1;      // key parts
    constexpr int PARTS = JP;
    float* osum = opt + Q * NPT;                // [JP][Q][HC + NPT]   (JP = 2 only)
    if constexpr (MF) {
        static_assert(Q <= 16 && Q % 4 == 0 && C == 16 && PV * 3 <= 32 && (JP == 1 || ((NT / 64) % JP == 0 && H % (NT / 64 / JP) == 0)), "matrix-pipe o / o_pt");
        constexpr int HG = JP == 1 ? H : NT / 64 / JP, HL = H / HG, NBK = 1 + (PV * 3 + 15) / 16;
        const int KS = (N + 31) >> 5, ksm = (KS + JP - 1) / JP;
        const int hg = wave / JP, jh = wave % JP, m = lane & 15, g = lane >> 4;
        const int ks0 = jh * ksm, nks = hg < HG ? max(min(KS - ks0, ksm), 0) : 0, nit = nks * HL;
        float mv = 0.f, mp = 0.f;                 // the structure's largest |v|, |v_pt|: every wave reduces k_ipa_prep_frag's group maxima itself
        for (int j = lane; j < (N + 7) >> 3; j += 64) {
            const float2 t = *reinterpret_cast<const float2*>(vmax + ((size_t)b * ((N + 7) >> 3) + j) * 2);
            mv = fmaxf(mv, t.x); mp = fmaxf(mp, t.y);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mv = fmaxf(mv, __shfl_xor(mv, o)); mp = fmaxf(mp, __shfl_xor(mp, o)); }
        const int exv = min(max((int)((__float_as_uint(mv) >> 23) & 0xff), 28), 254), exp_ = min(max((int)((__float_as_uint(mp) >> 23) & 0xff), 28), 254);
        const float sv = __uint_as_float((unsigned)(267 - exv) << 23), sp = __uint_as_float((unsigned)(267 - exp_) << 23);
        const float iv = __uint_as_float((unsigned)(exv - 27) << 23), ip = __uint_as_float((unsigned)(exp_ - 27) << 23);
        const float sa = (m < Q) ? 16384.0f : 0.0f;
        const float* ar = att + (size_t)min(m, Q - 1) * H * NPA + 8 * g;
        const float* vb = vf + ((size_t)b * H * NBK * KS) * 512 + lane * 8;
        // heads one after the other (12 accumulator registers), keys inside; the fragments of the next step are requested before this
        // one is worked on.  (Two steps ahead was slower: 1 KiB per load instruction and 64 B / clk into the CU, the phase runs at
        // three quarters of what the L1 can take -- its bytes, not their latency, are the limit.)
        f32x4 acc[NBK];
#pragma unroll
        for (int c = 0; c < NBK; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 bb[2][NBK][2];
        auto fetch = [&](int it, float4 (&dst)[NBK][2]) {
            const int hh = hg * HL + it / nks, ks = ks0 + it % nks;
#pragma unroll
            for (int c = 0; c < NBK; ++c) {
                const float* src = vb + (((size_t)hh * NBK + c) * KS + ks) * 512;
                dst[c][0] = *reinterpret_cast<const float4*>(src);
                dst[c][1] = *reinterpret_cast<const float4*>(src + 4);
            }
        };
        auto step = [&](int it, float4 (&cur)[NBK][2], float4 (&nxt)[NBK][2]) {
            if (it + 1 < nit) fetch(it + 1, nxt);
            const int hh = hg * HL + it / nks, ki = it % nks, ks = ks0 + ki;
            const bool live = ks * 32 + 8 * g < NP8;                              // att is zero padded up to NP8 only
            const float* ap = ar + hh * NPA + (live ? ks * 32 : 0);
            const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
            const float xa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
            h8 ah, al;
            hx_split8(xa, live ? sa : 0.0f, ah, al);
#pragma unroll
            for (int c = 0; c < NBK; ++c) {
                const float xb[8] = {cur[c][0].x, cur[c][0].y, cur[c][0].z, cur[c][0].w, cur[c][1].x, cur[c][1].y, cur[c][1].z, cur[c][1].w};
                h8 bh, bl;
                hx_split8(xb, c ? sp : sv, bh, bl);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[c], 0, 0, 0);
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[c], 0, 0, 0);
            }
            if (ki == nks - 1) {                                                  // the head is through
#pragma unroll
                for (int c = 0; c < NBK; ++c) {
                    const int w = (c - 1) * 16 + m;                                // point coordinate of this lane in blocks 1..
                    if (g < Q / 4 && (c == 0 || w < PV * 3)) {                     // D row 4 g + r = query, column m
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int q = 4 * g + r;
                            const float val = acc[c][r] * (c ? ip : iv);
                            if (JP > 1) osum[(jh * Q + q) * (HC + NPT) + (c == 0 ? hh * C + m : HC + hh * PV * 3 + w)] = val;
                            else if (q < nq) {
                                if (c == 0) cat[(size_t)(b * N + i0 + q) * NCAT + hh * C + m] = val;
                                else opt[q * NPT + hh * PV * 3 + w] = val;
                            }
                        }
                    }
                    acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        };
        if (nit > 0) fetch(0, bb[0]);
        for (int it = 0; it < nit; it += 2) {
            step(it, bb[0], bb[1]);
            if (it + 1 < nit) step(it + 1, bb[1], bb[0]);
        }
        if (JP > 1 && nit == 0 && g == 0 && hg < HG) {                             // no keys in this part (N <= 32): its partial sums are zero
            for (int hl = 0; hl < HL; ++hl)
                for (int c = m; c < C + PV * 3; c += 16)
                    for (int q = 0; q < Q; ++q) osum[(jh * Q + q) * (HC + NPT) + (c < C ? (hg * HL + hl) * C + c : HC + (hg * HL + hl) * PV * 3 + (c - C))] = 0.f;
        }
    } else if ((tid & 511) < HC + NPT) {
        const int col = tid & 511;
        const bool isv = col < HC;
        const int w = isv ? col : col - HC;
        const int hh = isv ? (col / C) : (w / (PV * 3));
        const float* vv = isv ? v + (size_t)b * N * HC + col : vp + (size_t)b * N * NPT + w;
        const int ld = isv ? HC : NPT;
        float acc[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = 0.f;
        // 16 rows of V in flight per thread (the loop is a chain of L2 round trips), then 8 at a time for the tail
        int j0 = 0;
        for (; j0 + 16 <= NP8; j0 += 16) {
            float x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = vv[(size_t)min(j0 + k, N - 1) * ld];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const float* ar = att + (q * H + hh) * NPA + j0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 a = *reinterpret_cast<const float4*>(ar + 4 * u);
                    acc[q] += a.x * x[4 * u]; acc[q] += a.y * x[4 * u + 1]; acc[q] += a.z * x[4 * u + 2]; acc[q] += a.w * x[4 * u + 3];
                }
            }
        }
        for (; j0 < NP8; j0 += 8) {
            float x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = vv[(size_t)min(j0 + k, N - 1) * ld];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const float* ar = att + (q * H + hh) * NPA + j0;
                const float4 a0 = *reinterpret_cast<const float4*>(ar), a1 = *reinterpret_cast<const float4*>(ar + 4);
                acc[q] += a0.x * x[0]; acc[q] += a0.y * x[1]; acc[q] += a0.z * x[2]; acc[q] += a0.w * x[3];
                acc[q] += a1.x * x[4]; acc[q] += a1.y * x[5]; acc[q] += a1.z * x[6]; acc[q] += a1.w * x[7];
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (q < nq) {
                if (isv) cat[(size_t)(b * N + i0 + q) * NCAT + col] = acc[q];
                else opt[q * NPT + w] = acc[q];
            }
        }
    }
    stamp();
    if constexpr (MF) {
        const int q = wave >> 1, hc = wave & 1, m = lane & 15, g = lane >> 4;
        if (q < nq) {
            const int row = b * N + i0 + q;
            // p S_p in [2^13, 2^14) at the tensor's largest magnitude; a 2^14 <= 2^14
            const int ex = min(max((int)((*pmax >> 23) & 0xff), 28), 254);
            const float sp = __uint_as_float((unsigned)(267 - ex) << 23);          // 2^(13 - (ex - 127))
            const float inv = __uint_as_float((unsigned)(ex - 27) << 23);          // 1 / (sp 2^14)
            const float sa = (m < H) ? 16384.0f : 0.0f;                            // rows 12..15 of the A tile are padding
            const float* zr = z + ((size_t)row * N) * CP + hc * 64 + m * 4;
            const float* ar = att + (q * H + min(m, H - 1)) * NPA + 8 * g;
            f32x4 acc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            float4 zz[2][8];
#pragma unroll
            for (int e = 0; e < 8; ++e) zz[0][e] = *reinterpret_cast<const float4*>(zr + (size_t)min(8 * g + e, N - 1) * CP);
            auto kstep = [&](int j0, float4 (&cur)[8], float4 (&nxt)[8]) {
                if (j0 + 32 < N) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        nxt[e] = *reinterpret_cast<const float4*>(zr + (size_t)min(j0 + 32 + 8 * g + e, N - 1) * CP);
                }
                const bool live = j0 + 8 * g < NP8;                               // att is zero padded up to NP8 only
                const float4 a0 = *reinterpret_cast<const float4*>(ar + (live ? j0 : 0));
                const float4 a1 = *reinterpret_cast<const float4*>(ar + (live ? j0 : 0) + 4);
                const float xa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                h8 ah, al;
                hx_split8(xa, live ? sa : 0.0f, ah, al);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float xb[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) xb[e] = r == 0 ? cur[e].x : r == 1 ? cur[e].y : r == 2 ? cur[e].z : cur[e].w;
                    h8 bh, bl;
                    hx_split8(xb, sp, bh, bl);
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[r], 0, 0, 0);
                    acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[r], 0, 0, 0);
                }
            };
            for (int j0 = 0; j0 < N; j0 += 64) {
                kstep(j0, zz[0], zz[1]);
                if (j0 + 32 < N) kstep(j0 + 32, zz[1], zz[0]);
            }
            float* op = cat + (size_t)row * NCAT + HC + H * PV * 4 + hc * 64 + m * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int hh = 4 * g + r;                                          // D row of register r
                if (hh < H)
                    *reinterpret_cast<float4*>(op + hh * CP) =
                        make_float4(acc[0][r] * inv, acc[1][r] * inv, acc[2][r] * inv, acc[3][r] * inv);
            }
        }
        __syncthreads();                                                           // opt (o_pt sums) is read below
    } else {
    // o_pair, one query after the other: thread = (channel quad c4, j-group jg of 16); 8 rows in flight
    const int c4 = tid & 31, jg = tid >> 5;
#pragma unroll 1
    for (int q = 0; q < nq; ++q) {
        const int row = b * N + i0 + q;
        float4 acc[H];
#pragma unroll
        for (int hh = 0; hh < H; ++hh) acc[hh] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* zr = z + ((size_t)row * N) * CP + c4 * 4;
        const float* aq = att + q * H * NPA;
        for (int jb = jg * 8; jb < NP8; jb += 128) {        // 8 consecutive rows per thread: attention weights as two b128 reads
            float4 zz[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) zz[k] = *reinterpret_cast<const float4*>(zr + (size_t)min(jb + k, N - 1) * CP);
#pragma unroll
            for (int hh = 0; hh < H; ++hh) {
                const float4 a0 = *reinterpret_cast<const float4*>(aq + hh * NPA + jb), a1 = *reinterpret_cast<const float4*>(aq + hh * NPA + jb + 4);
                const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};     // zero for j >= N (padding)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    acc[hh].x += a[k] * zz[k].x; acc[hh].y += a[k] * zz[k].y; acc[hh].z += a[k] * zz[k].z; acc[hh].w += a[k] * zz[k].w;
                }
            }
        }
#pragma unroll
        for (int hh = 0; hh < H; ++hh) {         // the two half-waves of a wave hold different j-groups
            acc[hh].x += __shfl_xor(acc[hh].x, 32); acc[hh].y += __shfl_xor(acc[hh].y, 32);
            acc[hh].z += __shfl_xor(acc[hh].z, 32); acc[hh].w += __shfl_xor(acc[hh].w, 32);
        }
        if (wave < 4 && lane < 32) {
#pragma unroll
            for (int hh = 0; hh < H; ++hh) *reinterpret_cast<float4*>(red + ((size_t)wave * H + hh) * CP + c4 * 4) = acc[hh];
        }
        __syncthreads();
        if (wave >= 4 && lane < 32) {
#pragma unroll
            for (int hh = 0; hh < H; ++hh) {
                float4* pr = reinterpret_cast<float4*>(red + ((size_t)(wave - 4) * H + hh) * CP + c4 * 4);
                float4 t = *pr;
                t.x += acc[hh].x; t.y += acc[hh].y; t.z += acc[hh].z; t.w += acc[hh].w;
                *pr = t;
            }
        }
        __syncthreads();
        {
            float* op = cat + (size_t)row * NCAT + HC + H * PV * 4;
            constexpr int tot = H * CP;
            for (int u = tid; u < tot; u += 512) op[u] = (red[u] + red[tot + u]) + (red[2 * tot + u] + red[3 * tot + u]);
        }
        __syncthreads();
    }
    }
    stamp();
    if constexpr (PARTS == 2) {                 // (after the barrier at the end of the o_pair branch)
        for (int u = tid; u < nq * (HC + NPT); u += NT) {
            const int q = u / (HC + NPT), col = u - q * (HC + NPT);
            const float sum = osum[q * (HC + NPT) + col] + osum[(Q + q) * (HC + NPT) + col];
            if (col < HC) cat[(size_t)(b * N + i0 + q) * NCAT + col] = sum;
            else opt[q * NPT + (col - HC)] = sum;
        }
        __syncthreads();
    }
    {
        constexpr int np = H * PV;
        for (int u = tid; u < nq * np; u += NT) {
            const int q = u / np, w = u - q * np;
            const int row = b * N + i0 + q;
            const float* R = rots + (size_t)row * 9;
            const float* t = trans + (size_t)row * 3;
            const float* o3 = opt + q * NPT + w * 3;
            const float x = o3[0] - t[0], y = o3[1] - t[1], zc = o3[2] - t[2];
            const float lx = R[0] * x + R[3] * y + R[6] * zc;
            const float ly = R[1] * x + R[4] * y + R[7] * zc;
            const float lz = R[2] * x + R[5] * y + R[8] * zc;
            float* crow = cat + (size_t)row * NCAT;
            crow[HC + w] = lx;
            crow[HC + np + w] = ly;
            crow[HC + 2 * np + w] = lz;
            crow[HC + 3 * np + w] = sqrtf(((lx * lx + ly * ly) + lz * lz) + 1e-8f);
        }
    }
}

// ---------------------------------------------------------------------------
// Training step, the two ends around the denoiser (diffusion/genie.py:77-105, utils/loss.py:4-36)
// ---------------------------------------------------------------------------
__global__ void k_q_sample(const float* __restrict__ x0, const float* __restrict__ z, const float* __restrict__ c0,
                           const float* __restrict__ c1, float* __restrict__ out, int n_per_b, int total) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < total) { const int b = u / n_per_b; out[u] = c0[b] * x0[u] + c1[b] * z[u]; }
}

// one work-group per structure: masked sums of err = sqrt(eps + |z_pred - z|^2) over the conditioned / infilled residues,
// and d weighted_loss / d z_pred = (w m_c + m_i) (z_pred - z) / err / (B (w n_c + n_i))
__global__ __launch_bounds__(256) void k_training_loss(const float* __restrict__ zp, const float* __restrict__ z,
                                                       const int32_t* __restrict__ rmask, const uint8_t* __restrict__ fsm, int B, int N,
                                                       float w, float* __restrict__ losses, float* __restrict__ stats, float* __restrict__ grad) {
    __shared__ float red[4][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v[4] = {0.f, 0.f, 0.f, 0.f};                    // condition loss, infill loss, n condition, n infill
    for (int n = tid; n < N; n += 256) {
        const size_t o = ((size_t)b * N + n) * 3;
        const float dx = zp[o] - z[o], dy = zp[o + 1] - z[o + 1], dz = zp[o + 2] - z[o + 2];
        const float err = sqrtf(1e-10f + ((dx * dx + dy * dy) + dz * dz));
        const float m = (float)rmask[b * N + n], f = fsm[b * N + n] ? 1.f : 0.f;
        const float mc = m * f, mi = m * (1.f - f);
        v[0] += err * mc; v[1] += err * mi; v[2] += mc; v[3] += mi;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
        if (lane == 0) red[wave][k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    if (tid == 0) {
        losses[2 + b] = v[0]; losses[2 + B + b] = v[1];
        stats[2 * b] = (v[0] + v[1]) / (v[2] + v[3]);                        // unweighted_losses[b] (num_residues = mask sum)
        stats[2 * b + 1] = (w * v[0] + v[1]) / (w * v[2] + v[3]);            // weighted_losses[b]
    }
    if (grad) {
        const float coef = 1.0f / ((float)B * (w * v[2] + v[3]));
        for (int n = tid; n < N; n += 256) {
            const size_t o = ((size_t)b * N + n) * 3;
            const float dx = zp[o] - z[o], dy = zp[o + 1] - z[o + 1], dz = zp[o + 2] - z[o + 2];
            const float err = sqrtf(1e-10f + ((dx * dx + dy * dy) + dz * dz));
            const float m = (float)rmask[b * N + n], f = fsm[b * N + n] ? 1.f : 0.f;
            const float g = coef * (w * m * f + m * (1.f - f)) / err;
            grad[o] = g * dx; grad[o + 1] = g * dy; grad[o + 2] = g * dz;
        }
    }
}
__global__ void k_training_loss_mean(const float* __restrict__ stats, int B, float* __restrict__ losses) {
    if (threadIdx.x == 0) {
        float u = 0.f, wl = 0.f;
        for (int b = 0; b < B; ++b) { u += stats[2 * b]; wl += stats[2 * b + 1]; }
        losses[0] = u / (float)B; losses[1] = wl / (float)B;
    }
}

// torch.optim.Adam's single-tensor update (ddpm.py:73-77), elementwise over a flat blob; bias corrections are computed on
// the host in double as torch does
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, size_t n, float omb1, float b2, float omb2, float eps,
                                              float step_size, float sqrt_bc2) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * omb1;                     // lerp_(grad, 1 - beta1), as torch writes exp_avg
        const float vi = v[i] * b2 + omb2 * (gi * gi);                  // mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / sqrt_bc2 + eps;                 // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        p[i] = p[i] - step_size * (mi / denom);                         // addcdiv_(exp_avg, denom, value=-step_size)
    }
}
void launch_adam(hipStream_t st, size_t n, float* p, const float* g, float* m, float* v, double lr, double b1, double b2, double eps,
                 int step) {
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, p, g, m, v, n,
                       (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, (float)(lr / bc1), (float)sqrt(bc2));
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
void launch_single_input(genie_ctx* h, hipStream_t st, const int32_t* timesteps) {
    ProfScope ps(h, st, KC_SINGLE_INPUT);
    const genie_dims_t& d = h->d;
    const int ldx = (d.c_pos_emb + d.c_chain_emb + d.c_timestep_emb + 23 + 7) / 8 * 8;
    hipLaunchKernelGGL(k_single_input, dim3(h->B * h->N), dim3(256), 0, st, h->xsingle, ldx, h->pos_tab, h->n_pos, d.c_pos_emb,
                       h->chain_tab, h->n_chain, d.c_chain_emb, h->t_tab, d.c_timestep_emb, timesteps, h->f_ridx, h->f_cidx,
                       h->f_aatype, h->f_fsm, h->f_ifm, h->N, d.n_timestep);
}

void launch_gemm_rows(genie_ctx* h, hipStream_t st, const float* A, int lda, int M, int K, const float* Wp, int Nout,
                      const float* bias, const float* res, int ldr, const float* rowmask, int relu, float* out, int ldo) {
    ProfScope ps(h, st, KC_GEMM_ROWS);
    dim3 grid((M + 31) / 32, (Nout + 127) / 128);
    if (h->hx) {
        for (int i = 0; i < h->n_hxg; ++i)
            if (h->hxg[i].w == Wp) {
                hipLaunchKernelGGL(k_gemm_rows_hx, grid, dim3(256), 0, st, A, lda, M, K, h->hxg[i].img, Nout, h->hxg[i].inv_s, bias, res,
                                   ldr, rowmask, relu, out, ldo, (size_t)0);
                return;
            }
    }
    hipLaunchKernelGGL(k_gemm_rows, grid, dim3(256), 0, st, A, lda, M, K, Wp, Nout, bias, res, ldr, rowmask, relu, out, ldo);
}

void launch_layernorm_rows(genie_ctx* h, hipStream_t st, const float* in, float* out, int M, int C, const float* g,
                           const float* b) {
    ProfScope ps(h, st, KC_LAYERNORM);
    hipLaunchKernelGGL(k_layernorm_rows, dim3((M + 3) / 4), dim3(256), 0, st, in, out, M, C, g, b);
}

static bool ipa_attn_mfma_av(const genie_ctx* h);      // the attention kernel in use wants V / v_pts as fragments
void launch_ipa_prep(genie_ctx* h, hipStream_t st, int b0, int nb) {
    ProfScope ps(h, st, KC_IPA_PREP);
    const genie_dims_t& d = h->d;
    if (nb < 0) { b0 = 0; nb = h->B; }
    const int ldp = d.n_head_ipa * (3 * d.c_hidden_ipa + 3 * d.n_qk_point + 3 * (d.n_qk_point + d.n_v_point));
    if (ipa_attn_mfma_av(h)) {
        hipLaunchKernelGGL(k_ipa_prep_frag, dim3(nb * ((h->N + 7) / 8)), dim3(256), 0, st, h->proj, ldp, h->rots_w, h->trans_w, h->kT, h->qp,
                           h->kpT, h->vf, h->vmax, h->N, d.n_head_ipa, d.c_hidden_ipa, d.n_qk_point, d.n_v_point, b0);
        return;
    }
    hipLaunchKernelGGL(k_ipa_prep, dim3(nb * h->N), dim3(256), 0, st, h->proj, ldp, h->rots_w, h->trans_w, h->kT, h->v, h->qp,
                       h->kpT, h->vp, h->N, d.n_head_ipa, d.c_hidden_ipa, d.n_qk_point, d.n_v_point, b0 * h->N);
}

static bool ipa_is_base(const genie_dims_t& d) {
    return d.n_head_ipa == 12 && d.c_hidden_ipa == 16 && d.n_qk_point == 4 && d.n_v_point == 8 && d.c_p == 128;
}
#define IPA_Q 4
static size_t ipa_attn_t1_lds(const genie_dims_t& d, int N) {      // k_ipa_attn_t<12, 16, 4, 8> (single query; long structures)
    return ((size_t)d.n_head_ipa * ((N + 7) & ~7) + 4 * d.n_head_ipa * d.c_p + d.n_head_ipa * d.n_v_point * 3) * sizeof(float);
}
#define IPA_Q8 8             // queries per work-group of the 1024-thread matrix-pipe form
static size_t ipa_attn_q_lds(const genie_dims_t& d, int N, bool mf = false, int Q = IPA_Q);
static bool ipa_use_q8(const genie_dims_t& d, int N) {
    // measured slower (80.5 vs 76.0 us per launch): opt-in; read per call so that a test can compare both forms
    return getenv("GENIE_IPA_Q8") != nullptr && ipa_attn_q_lds(d, N, true, IPA_Q8) <= 160 * 1024;
}
static bool ipa_use_q(const genie_dims_t& d, int N) { return ipa_attn_q_lds(d, N) <= 160 * 1024; }
static size_t ipa_attn_t_lds(const genie_dims_t& d, int N) { return ipa_use_q(d, N) ? ipa_attn_q_lds(d, N) : ipa_attn_t1_lds(d, N); }
static size_t ipa_attn_q_lds(const genie_dims_t& d, int N, bool mf, int Q) {   // k_ipa_attn_q<12, 16, 4, 8, Q, mf>: no reduction buffer with mf
    const size_t H = d.n_head_ipa;
    return ((size_t)Q * H * (((N + 7) & ~7) + 4) + Q * H * d.c_hidden_ipa + Q * H * d.n_qk_point * 3 + 16 +
            Q * H * d.n_v_point * 3 + (mf ? 0 : 4 * H * d.c_p) +
            (mf && Q == IPA_Q ? 2 * Q * H * (d.c_hidden_ipa + 3 * d.n_v_point) : 0)) * sizeof(float);       // mf, Q = 4: the two key halves' partial o / o_pt
}
size_t ipa_attn_lds(const genie_dims_t& d, int N) {
    if (ipa_is_base(d)) return ipa_attn_t_lds(d, N);
    return ((size_t)d.n_head_ipa * N + 8 * d.n_head_ipa * d.c_p + d.n_head_ipa * d.n_v_point * 3) * sizeof(float);
}

// hx arithmetic: a v and a v_pts run on the matrix pipe, from the fragment layout k_ipa_prep writes for it
static bool ipa_attn_mfma_av(const genie_ctx* h) { return h->hx && ipa_is_base(h->d) && ipa_use_q(h->d, h->N); }
size_t ipa_vf_floats(const genie_dims_t& d, int B, int N) {
    return ipa_is_base(d) ? (size_t)B * d.n_head_ipa * (1 + (3 * d.n_v_point + 15) / 16) * ((N + 31) / 32) * 512 : 64;
}
// the four-query kernel takes a batch range (the two halves of a batch run their structure layers on two streams)
bool ipa_attn_splits(const genie_ctx* h) { return ipa_is_base(h->d) && ipa_use_q(h->d, h->N); }
void launch_ipa_attn(genie_ctx* h, hipStream_t st, int layer, const float* head_w, int b0, int nb) {
    ProfScope ps(h, st, KC_IPA_ATTN);
    const genie_dims_t& d = h->d;
    if (nb < 0 || !ipa_attn_splits(h)) { b0 = 0; nb = h->B; }
    const int ldp = d.n_head_ipa * (3 * d.c_hidden_ipa + 3 * d.n_qk_point + 3 * (d.n_qk_point + d.n_v_point));
    if (ipa_is_base(d) && !ipa_use_q(d, h->N)) {
        hipLaunchKernelGGL((k_ipa_attn_t<12, 16, 4, 8>), dim3(h->B * h->N), dim3(256), ipa_attn_t1_lds(d, h->N), st, h->proj, ldp,
                           h->kT, h->v, h->qp, h->kpT, h->vp, h->ipa_bias, h->p, h->rots_w, h->trans_w, h->rmaskf, head_w, h->cat,
                           h->B, h->N, layer);
        return;
    }
    if (ipa_is_base(d)) {
        const dim3 grid(nb * ((h->N + IPA_Q - 1) / IPA_Q));
        const int rev = (int)((layer ^ h->hx_launches ^ 1) & 1);
        static unsigned long long* ts = nullptr;
        if (!ts && getenv("GENIE_SR_TS")) (void)hipMalloc((void**)&ts, 64 * sizeof(unsigned long long));
        if (h->hx && ipa_use_q8(d, h->N))
            hipLaunchKernelGGL((k_ipa_attn_q<12, 16, 4, 8, IPA_Q8, 1>), dim3(nb * ((h->N + IPA_Q8 - 1) / IPA_Q8)), dim3(1024),
                               ipa_attn_q_lds(d, h->N, true, IPA_Q8), st, h->proj, ldp, h->kT, h->v, h->qp, h->kpT, h->vp, h->ipa_bias, h->p,
                               h->rots_w, h->trans_w, h->rmaskf, head_w, h->cat, h->B, h->N, layer, rev, h->pmax, ts, b0, h->vf, h->vmax);
        else if (h->hx)
            hipLaunchKernelGGL((k_ipa_attn_q<12, 16, 4, 8, IPA_Q, 1>), grid, dim3(512), ipa_attn_q_lds(d, h->N, true), st, h->proj, ldp,
                               h->kT, h->v, h->qp, h->kpT, h->vp, h->ipa_bias, h->p, h->rots_w, h->trans_w, h->rmaskf, head_w, h->cat,
                               h->B, h->N, layer, rev, h->pmax, ts, b0, h->vf, h->vmax);
        else
            hipLaunchKernelGGL((k_ipa_attn_q<12, 16, 4, 8, IPA_Q, 0>), grid, dim3(512), ipa_attn_q_lds(d, h->N), st, h->proj, ldp,
                               h->kT, h->v, h->qp, h->kpT, h->vp, h->ipa_bias, h->p, h->rots_w, h->trans_w, h->rmaskf, head_w, h->cat,
                               h->B, h->N, layer, rev, h->pmax, ts, b0, nullptr, nullptr);
        if (ts) {
            unsigned long long v[24] = {0};
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(v, ts, sizeof(v), hipMemcpyDeviceToHost);
            fprintf(stderr, "ipa_attn wg0 (cycles): q-load %llu logits %llu softmax %llu o/o_pt %llu o_pair %llu total %llu; starts of wg 64k:", v[1] - v[0],
                    v[2] - v[1], v[3] - v[2], v[4] - v[3], v[5] - v[4], v[5] - v[0]);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %lld", (long long)(v[16 + k] - v[0]));
            fprintf(stderr, "\n");
        }
        return;
    }
    hipLaunchKernelGGL(k_ipa_attn, dim3(h->B * h->N), dim3(256), ipa_attn_lds(d, h->N), st, h->proj, ldp, h->kT, h->v, h->qp,
                       h->kpT, h->vp, h->ipa_bias, h->p, h->rots_w, h->trans_w, h->rmaskf, head_w, h->cat, h->B, h->N,
                       d.n_head_ipa, d.c_hidden_ipa, d.n_qk_point, d.n_v_point, d.c_p, layer);
}

void launch_bb_update(genie_ctx* h, hipStream_t st, const StructLayerW& w, const float* trans_in, float* z_out) {
    ProfScope ps(h, st, KC_BB_UPDATE);
    const int M = h->B * h->N;
    hipLaunchKernelGGL(k_bb_update, dim3((M + 3) / 4), dim3(256), 0, st, h->s, h->d.c_s, w.bb_w, w.bb_b, h->rots_w, h->trans_w,
                       M, trans_in, z_out, 1.0f / h->d.rescale);
}

static size_t struct_rows_lds() { return (size_t)2 * 32 * SR_LD * 4 + 6 * (2 * 32 * GH_ROWB + 32 * 4) + 32 * 12 * 4; }   // NKC = 6
static const HxGemmW* hx_image(const genie_ctx* h, const float* Wp) {
    for (int i = 0; i < h->n_hxg; ++i)
        if (h->hxg[i].w == Wp) return &h->hxg[i];
    return nullptr;
}
// IPA output projection (split-K, three slices) and the fused tail (their sum + bias + residual -> LayerNorm -> structure
// transition -> LayerNorm -> BackboneUpdate), two launches; false = this configuration keeps the separate launches (f32
// arithmetic, c_s != 384)
bool struct_tail_fused(const genie_ctx* h, const StructLayerW& S) {
    if (!h->hx || h->d.c_s != 384 || getenv("GENIE_NO_STRUCT_FUSE")) return false;
    return hx_image(h, S.out_w) && hx_image(h, S.t1_w) && hx_image(h, S.t2_w) && hx_image(h, S.t3_w);
}
// rows of batch entries b0 .. b0 + nb - 1 (nb < 0: all)
bool launch_struct_tail(genie_ctx* h, hipStream_t st, const StructLayerW& S, const float* trans_in, float* z_out, int b0, int nb) {
    if (!struct_tail_fused(h, S)) return false;
    const HxGemmW *w0 = hx_image(h, S.out_w), *w1 = hx_image(h, S.t1_w), *w2 = hx_image(h, S.t2_w), *w3 = hx_image(h, S.t3_w);
    if (nb < 0) { b0 = 0; nb = h->B; }
    const int Mall = h->B * h->N, M = nb * h->N, cs = h->d.c_s, ncat = h->d.n_head_ipa * (h->d.c_p + h->d.c_hidden_ipa + 4 * h->d.n_v_point);
    const size_t zs = (size_t)Mall * cs, r0 = (size_t)b0 * h->N;          // slice stride of the split-K partial sums; first row
    {
        ProfScope ps(h, st, KC_GEMM_ROWS);
        hipLaunchKernelGGL(k_gemm_rows_hx, dim3((M + 31) / 32, (cs + 127) / 128, SR_KSPLIT), dim3(256), 0, st, h->cat + r0 * ncat, ncat, M, ncat,
                           w0->img, cs, w0->inv_s, nullptr, nullptr, 0, nullptr, 0, h->spart + r0 * cs, cs, zs);
    }
    ProfScope ps(h, st, KC_STRUCT_ROWS);
    static unsigned long long* ts = nullptr;
    if (!ts && getenv("GENIE_SR_TS")) (void)hipMalloc((void**)&ts, 64 * sizeof(unsigned long long));
    const int nrb = (M + 31) / 32;                 // row work-groups; + 64 L2 prefetchers (8 per XCD) on CUs the rows leave idle
    hipLaunchKernelGGL((k_struct_rows_hx<12>), dim3(nrb + 64), dim3(768), struct_rows_lds(), st, h->spart + r0 * cs, SR_KSPLIT, zs, S.out_b, h->s + r0 * cs,
                       S.ln_ipa_g, S.ln_ipa_b, w1->img, w1->inv_s, S.t1_b, w2->img, w2->inv_s, S.t2_b, w3->img, w3->inv_s, S.t3_b,
                       S.ln_tr_g, S.ln_tr_b, S.bb_w, S.bb_b, h->s + r0 * cs, h->rots_w + r0 * 9, h->trans_w + r0 * 3, M, trans_in ? trans_in + r0 * 3 : nullptr,
                       z_out ? z_out + r0 * 3 : nullptr, 1.0f / h->d.rescale, nrb, ts);
    if (ts) {
        unsigned long long v[8] = {0};
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(v, ts, sizeof(v), hipMemcpyDeviceToHost);
        fprintf(stderr, "struct_rows wg0 (100 MHz ticks): load %llu ln %llu lin1 %llu lin2 %llu lin3 %llu ln %llu bb %llu total %llu\n", v[1] - v[0],
                v[2] - v[1], v[3] - v[2], v[4] - v[3], v[5] - v[4], v[6] - v[5], v[7] - v[6], v[7] - v[0]);
    }
    return true;
}

void launch_q_sample(genie_ctx* h, hipStream_t st, const float* x0, const float* z, const float* c0, const float* c1, float* trans_out) {
    ProfScope ps(h, st, KC_MISC);
    const int total = h->B * h->N * 3;
    hipLaunchKernelGGL(k_q_sample, dim3((total + 255) / 256), dim3(256), 0, st, x0, z, c0, c1, trans_out, h->N * 3, total);
}
void launch_training_loss(genie_ctx* h, hipStream_t st, const float* zp, const float* z, float w, float* losses, float* grad) {
    ProfScope ps(h, st, KC_MISC);
    float* stats = h->loop_z;                       // [B,N,3] scratch of the reverse loop: 2 B floats are used
    hipLaunchKernelGGL(k_training_loss, dim3(h->B), dim3(256), 0, st, zp, z, h->f_rmask, h->f_fsm, h->B, h->N, w, losses, stats, grad);
    hipLaunchKernelGGL(k_training_loss_mean, dim3(1), dim3(64), 0, st, stats, h->B, losses);
}

void launch_frenet(genie_ctx* h, hipStream_t st, int mode, int step, float scale, float* trans, float* rots, const float* z,
                   const float* eps) {
    ProfScope ps(h, st, KC_P_SAMPLE);
    float al = 1.f, sa = 1.f, so = 1.f, sb = 0.f;
    if (mode == 1) {
        const int T1 = h->d.n_timestep + 1;
        al = h->sched_host[0 * T1 + step];
        sa = h->sched_host[1 * T1 + step];
        so = h->sched_host[2 * T1 + step];
        sb = h->sched_host[3 * T1 + step];
    }
    const size_t lds = (size_t)(3 + 3 + 9 + 9) * h->N * sizeof(float);
    hipLaunchKernelGGL(k_p_sample_frenet, dim3(h->B), dim3(256), lds, st, mode, trans, rots, z, eps, h->f_rmask, h->f_cidx, h->N,
                       al, sa, so, sb, scale);
}

// handle-free form: compute_frenet_frames(coords, chains, mask) of genie/utils/geo_utils.py:21-85 as the reference calls it
int genie_frenet_frames(genie_stream_t stream, int B, int N, const float* coords, const int32_t* chains, const int32_t* mask,
                        float* rots_out) {
    if (!coords || !chains || !mask || !rots_out || B < 1 || N < 2) return GENIE_E_ARG;
    const size_t lds = (size_t)24 * N * sizeof(float);
    if (lds > 160 * 1024) return GENIE_E_ARG;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_p_sample_frenet), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_p_sample_frenet, dim3(B), dim3(256), lds, (hipStream_t)stream, 0, const_cast<float*>(coords), rots_out,
                       (const float*)nullptr, (const float*)nullptr, mask, chains, N, 1.f, 1.f, 1.f, 0.f, 0.f);
    return hipGetLastError() == hipSuccess ? GENIE_OK : GENIE_E_HIP;
}

void launch_fill_i32(genie_ctx* h, hipStream_t st, int32_t* p, int n, int v) {
    ProfScope ps(h, st, KC_MISC);
    hipLaunchKernelGGL(k_fill_i32, dim3((n + 255) / 256), dim3(256), 0, st, p, n, v);
}

void launch_scale_copy(genie_ctx* h, hipStream_t st, const float* in, float* out, int n, float s) {
    ProfScope ps(h, st, KC_MISC);
    hipLaunchKernelGGL(k_scale_copy, dim3((n + 255) / 256), dim3(256), 0, st, in, out, n, s);
}

void single_kernels_init(const genie_dims_t& d, int n_max) {
    // compute_frenet_frames keeps one structure in LDS (24 floats per residue): past 64 KiB the launch needs the attribute
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_p_sample_frenet), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)std::min<size_t>((size_t)24 * n_max * sizeof(float), 160 * 1024));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_struct_rows_hx<12>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)struct_rows_lds());
    if (ipa_is_base(d)) {
        if (ipa_use_q(d, n_max)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ipa_attn_q<12, 16, 4, 8, IPA_Q, 0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ipa_attn_q_lds(d, n_max));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ipa_attn_q<12, 16, 4, 8, IPA_Q, 1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ipa_attn_q_lds(d, n_max, true));
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ipa_attn_q<12, 16, 4, 8, IPA_Q8, 1>),       // (used up to the N that fits)
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::min<size_t>(ipa_attn_q_lds(d, n_max, true, IPA_Q8), 160 * 1024));
        } else
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ipa_attn_t<12, 16, 4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)ipa_attn_t1_lds(d, n_max));
    } else
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ipa_attn), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)ipa_attn_lds(d, n_max));
}
