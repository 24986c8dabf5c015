// Pair-stack GEMM kernels, wave-independent form ("WI").
//
// Profiling the first (LDS-tile + barrier) versions showed the MFMA pipes idle ~45 % of the
// time: all work-groups of a CU run their prologue (HBM tile load, LayerNorm), MFMA phase and
// epilogue in lockstep, so nothing overlaps.  Here every WAVE owns one 32-pair tile end to end:
//   * the activation tile is loaded straight into MFMA fragment registers (lane (p, h) holds
//     row p, k = 8kb + 4h .. +3 for kb = 0..15: half a row), LayerNorm runs in registers with one
//     cross-half shuffle -- no LDS, no barrier, no cross-wave dependency;
//   * weight fragments stream from L2 through an asm-issued register ring that stays WI_PD
//     fragments ahead of the MFMAs (common.h, "software-pipelined weight fragments");
//   * chained GEMMs (pair transition) feed the first GEMM's accumulator straight back as the next
//     MFMA operand: with D'[hidden][pair] = W1 zn^T, registers 4q..4q+3 of the accumulator ARE the
//     A fragment of k-block q of the second GEMM (same (lane&31, lane>>5) maps), so the 512-wide
//     hidden layer never leaves the register file.
// Waves progress independently and interleave at fragment granularity on each SIMD.
#include "common.h"

#ifndef WI_PD
#define WI_PD 8          // weight fragments in flight per wave (8 KiB)
#endif

__device__ __forceinline__ float fast_sigmoid(float x) {
    // 1 / (1 + 2^(-x log2 e)): v_exp_f32 + v_rcp_f32 (each <= 1 ulp); the precise expf/division
    // sequence costs ~5x the VALU issue slots and was a visible part of the epilogue.
    return __frcp_rn(1.0f + __expf(-x));
}

// Row pointer helpers -------------------------------------------------------------------------
struct WaveTile {
    int b, line, t0, nvalid;     // tile = pairs (b, line, t0 + 0..31) in the direction's orientation
};

// Load a 32-row x 128-channel tile as 16 k-block fragments (rows clamped to stay in bounds) and
// LayerNorm it in registers.  `rowp` = this lane's row pointer (row lane&31).
__device__ __forceinline__ void load_ln_tile(float4 (&zf)[16], const float* __restrict__ rowp, int h,
                                             const float* __restrict__ gamma, const float* __restrict__ beta) {
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) zf[kb] = *reinterpret_cast<const float4*>(rowp + kb * 8 + 4 * h);
    float s = 0.f;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) s += (zf[kb].x + zf[kb].y) + (zf[kb].z + zf[kb].w);
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / 128.0f);
    float ss = 0.f;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        const float a = zf[kb].x - mean, b = zf[kb].y - mean, c = zf[kb].z - mean, d = zf[kb].w - mean;
        ss += (a * a + b * b) + (c * c + d * d);
    }
    ss += __shfl_xor(ss, 32);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 128.0f) + GENIE_LN_EPS);
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + kb * 8 + 4 * h);
        const float4 bt = *reinterpret_cast<const float4*>(beta + kb * 8 + 4 * h);
        zf[kb].x = (zf[kb].x - mean) * rstd * g.x + bt.x;
        zf[kb].y = (zf[kb].y - mean) * rstd * g.y + bt.y;
        zf[kb].z = (zf[kb].z - mean) * rstd * g.z + bt.z;
        zf[kb].w = (zf[kb].w - mean) * rstd * g.w + bt.w;
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, projections (see pair_kernels.hip for the math and the layouts).
// 8 passes of (32 p-channels + 32 g-channels) x 32 pairs; D rows = channels, D cols = pairs.
// Weight stream per pass: kb = 0..15 x {p-block, g-block} = 32 fragments.
// ---------------------------------------------------------------------------------------------
template <bool OUTGOING>
__global__ __launch_bounds__(256, 2) void k_trimul_proj_wi(const float* __restrict__ z, const float* __restrict__ rmask,
                                                           const float* __restrict__ wp, const float* __restrict__ bias,
                                                           const float* __restrict__ lng, const float* __restrict__ lnb,
                                                           float* __restrict__ acm, float* __restrict__ bcm, int N, int NP,
                                                           int n_wtiles) {
    const int lane = threadIdx.x & 63;
    const int wt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wt >= n_wtiles) return;
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int st = wt % ntile;
    const int line = (wt / ntile) % N;
    const int b = wt / (ntile * N);
    const int t0 = st * 32;
    const int nvalid = min(32, N - t0);
    const int pr = min(pl, nvalid - 1);
    const float* rowp = OUTGOING ? z + (((size_t)b * N + line) * N + t0 + pr) * 128
                                 : z + (((size_t)b * N + t0 + pr) * N + line) * 128;

    // weight ring: start it before anything else so the first fragments land during the tile load
    v4f wq[WI_PD];
    auto frag_addr = [&](int t) {   // t = pass*32 + kb*2 + j  ->  block (j ? 8 + pass : pass), k-block kb
        const int pass = t >> 5, kb = (t >> 1) & 15, j = t & 1;
        return wfrag_ptr(wp, 16, j * 8 + pass, kb, lane);
    };
#pragma unroll
    for (int s = 0; s < WI_PD; ++s) wf_issue(wq[s], frag_addr(s));

    float4 zf[16];
    load_ln_tile(zf, rowp, h, lng, lnb);
    const float msk = (pl < nvalid) ? rmask[b * N + line] * rmask[b * N + t0 + pr] : 0.f;

#pragma unroll 1
    for (int pass = 0; pass < 8; ++pass) {
        f32x16 ap = zero16(), ag = zero16();
        float bpv[16], bgv[16];                 // issued now, consumed after the 128 MFMAs of the pass
#pragma unroll
        for (int r = 0; r < 16; ++r) { bpv[r] = bias[pass * 32 + acc_row(r, lane)]; bgv[r] = bias[256 + pass * 32 + acc_row(r, lane)]; }
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int slot = (kb * 2 + j) % WI_PD;
                wf_wait<WI_PD - 1>(wq[slot]);
                if (j == 0) ap = mfma_8k(wq[slot], zf[kb], ap);
                else        ag = mfma_8k(wq[slot], zf[kb], ag);
                __builtin_amdgcn_sched_barrier(0);
                const int tn = min(pass * 32 + kb * 2 + j + WI_PD, 8 * 32 - 1);   // tail: re-load of the last fragment
                wf_issue(wq[slot], frag_addr(tn));
            }
        }
        float* dst = (pass < 4) ? acm : bcm;
        const int chbase = (pass & 3) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lane);
            const float v = (ap[r] + bpv[r]) * fast_sigmoid(ag[r] + bgv[r]) * msk;
            if (pl < nvalid) dst[(((size_t)b * 128 + chbase + row) * NP + line) * NP + t0 + pl] = v;
        }
    }
#pragma unroll
    for (int s = 0; s < WI_PD; ++s) wf_wait<0>(wq[s]);     // retire the ring before its registers die
}

// ---------------------------------------------------------------------------------------------
// Pair transition + end-of-layer mask, hidden layer chained through the accumulator registers.
// Weight stream per hidden block hb (32 hidden units): 16 W1 fragments (k-blocks of the 128 input
// channels) then 16 W2 fragments (4 output blocks x 4 k-blocks of this hidden block).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_pair_transition_wi(float* __restrict__ z, const float* __restrict__ rmask,
                                                               const float* __restrict__ lng, const float* __restrict__ lnb,
                                                               const float* __restrict__ w1, const float* __restrict__ b1,
                                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                                               int N, long long M, int n_hb) {
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
    if (row0 >= M) return;
    const int h = lane >> 5, pl = lane & 31;
    const int nvalid = (int)min((long long)32, M - row0);
    const int pr = min(pl, nvalid - 1);
    float* zrow = z + row0 * 128;

    const int KB2 = n_hb * 4;                 // k-blocks of W2 (hidden / 8)
    v4f wq[WI_PD];
    auto frag_addr = [&](int t) {             // t = hb*32 + u; u < 16: W1[hb][kb=u]; else W2[ob=(u-16)>>2][kb = hb*4 + ((u-16)&3)]
        const int hb = t >> 5, u = t & 31;
        return (u < 16) ? wfrag_ptr(w1, 16, hb, u, lane) : wfrag_ptr(w2, KB2, (u - 16) >> 2, hb * 4 + ((u - 16) & 3), lane);
    };
    const int t_last = n_hb * 32 - 1;
#pragma unroll
    for (int s = 0; s < WI_PD; ++s) wf_issue(wq[s], frag_addr(s));

    float4 zn[16];
    load_ln_tile(zn, zrow + (size_t)pr * 128, h, lng, lnb);
    float m_own = 0.f;
    if (pl < nvalid) {
        const long long idx = row0 + pl;
        const int bb = (int)(idx / ((long long)N * N));
        const int rem = (int)(idx - (long long)bb * N * N);
        m_own = rmask[bb * N + rem / N] * rmask[bb * N + rem % N];
    }

    f32x16 o0 = zero16(), o1 = zero16(), o2 = zero16(), o3 = zero16();
#pragma unroll 1
    for (int hb = 0; hb < n_hb; ++hb) {
        f32x16 d = zero16();
        float bv[16];                           // issued now, consumed after the 64 MFMAs of GEMM 1
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = b1[hb * 32 + acc_row(r, lane)];
#pragma unroll
        for (int u = 0; u < 16; ++u) {        // hidden block = W1[hb] zn^T : D'[hidden][pair]
            const int slot = u % WI_PD;
            wf_wait<WI_PD - 1>(wq[slot]);
            d = mfma_8k(wq[slot], zn[u], d);
            __builtin_amdgcn_sched_barrier(0);
            wf_issue(wq[slot], frag_addr(min(hb * 32 + u + WI_PD, t_last)));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = fmaxf(d[r] + bv[r], 0.f);
#pragma unroll
        for (int u = 16; u < 32; ++u) {       // out[pair][o] += h[pair][hidden] W2[o][hidden]
            const int slot = u % WI_PD;
            const int q = (u - 16) & 3, ob = (u - 16) >> 2;
            const float4 hf = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
            wf_wait<WI_PD - 1>(wq[slot]);
            if (ob == 0) o0 = mfma_8k(hf, wq[slot], o0);
            else if (ob == 1) o1 = mfma_8k(hf, wq[slot], o1);
            else if (ob == 2) o2 = mfma_8k(hf, wq[slot], o2);
            else o3 = mfma_8k(hf, wq[slot], o3);
            __builtin_amdgcn_sched_barrier(0);
            wf_issue(wq[slot], frag_addr(min(hb * 32 + u + WI_PD, t_last)));
        }
    }
#pragma unroll
    for (int s = 0; s < WI_PD; ++s) wf_wait<0>(wq[s]);

#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = acc_row(r, lane);
        const float m = __shfl(m_own, t);
        if (t < nvalid) {
            float* q = zrow + (size_t)t * 128 + pl;
            q[0]  = ((o0[r] + b2[pl]) * m + q[0]) * m;
            q[32] = ((o1[r] + b2[32 + pl]) * m + q[32]) * m;
            q[64] = ((o2[r] + b2[64 + pl]) * m + q[64]) * m;
            q[96] = ((o3[r] + b2[96 + pl]) * m + q[96]) * m;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, output: z += (W_z LN_out(x) + b_z) * sigmoid(W_g LN_in(z) + b_g).
// The x tile arrives channel-major; lane (p, h) reads x[c][p] for its 64 channels directly in
// fragment order (each load instruction = two 128-B runs), LayerNorm over channels in registers.
// Weight stream: 4 blocks x 16 k-blocks of W_g, then the same of W_z.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_trimul_out_wi(float* __restrict__ z, const float* __restrict__ xcm,
                                                          const float* __restrict__ wg, const float* __restrict__ bg,
                                                          const float* __restrict__ wz, const float* __restrict__ bz,
                                                          const float* __restrict__ ln_in_g, const float* __restrict__ ln_in_b,
                                                          const float* __restrict__ ln_out_g, const float* __restrict__ ln_out_b,
                                                          int N, int NP, int n_wtiles) {
    const int lane = threadIdx.x & 63;
    const int wt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wt >= n_wtiles) return;
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int st = wt % ntile;
    const int i = (wt / ntile) % N;
    const int b = wt / (ntile * N);
    const int t0 = st * 32;
    const int nvalid = min(32, N - t0);
    const int pr = min(pl, nvalid - 1);
    float* zrow = z + (((size_t)b * N + i) * N + t0) * 128;

    v4f wq[WI_PD];
    auto frag_addr = [&](int t) {             // t = g*64 + ob*16 + kb, g = 0: W_g, 1: W_z
        return wfrag_ptr((t & 64) ? wz : wg, 16, (t >> 4) & 3, t & 15, lane);
    };
#pragma unroll
    for (int s = 0; s < WI_PD; ++s) wf_issue(wq[s], frag_addr(s));

    // x fragments first (longest latency: HBM), z tile next
    float4 xf[16];
    {
        const float* xp = xcm + (((size_t)b * 128 + 4 * h) * NP + i) * NP + t0 + pl;   // t0 + pl < NP always (NP = ceil32(N))
        const size_t cs = (size_t)NP * NP;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const float* p = xp + (size_t)(kb * 8) * cs;
            xf[kb] = make_float4(p[0], p[cs], p[2 * cs], p[3 * cs]);
        }
    }
    float4 zf[16];
    load_ln_tile(zf, zrow + (size_t)pr * 128, h, ln_in_g, ln_in_b);

    // gate: g[pair][ch] = sigmoid(zn W_g^T + b_g), A = zn fragments (i = pair), B = W_g (j = channel)
    f32x16 g0 = zero16(), g1 = zero16(), g2 = zero16(), g3 = zero16();
#pragma unroll
    for (int u = 0; u < 64; ++u) {
        const int slot = u % WI_PD, ob = u >> 4, kb = u & 15;
        wf_wait<WI_PD - 1>(wq[slot]);
        if (ob == 0) g0 = mfma_8k(zf[kb], wq[slot], g0);
        else if (ob == 1) g1 = mfma_8k(zf[kb], wq[slot], g1);
        else if (ob == 2) g2 = mfma_8k(zf[kb], wq[slot], g2);
        else g3 = mfma_8k(zf[kb], wq[slot], g3);
        __builtin_amdgcn_sched_barrier(0);
        wf_issue(wq[slot], frag_addr(u + WI_PD));           // runs on into the W_z stream
    }
    {
        const float c0 = bg[pl], c1 = bg[32 + pl], c2 = bg[64 + pl], c3 = bg[96 + pl];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            g0[r] = fast_sigmoid(g0[r] + c0); g1[r] = fast_sigmoid(g1[r] + c1);
            g2[r] = fast_sigmoid(g2[r] + c2); g3[r] = fast_sigmoid(g3[r] + c3);
        }
    }
    // LayerNorm of x over channels (this lane: 64 channels of pair pl; the other 64 in lane ^ 32)
    {
        float s = 0.f;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) s += (xf[kb].x + xf[kb].y) + (xf[kb].z + xf[kb].w);
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / 128.0f);
        float ss = 0.f;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const float a = xf[kb].x - mean, bq = xf[kb].y - mean, c = xf[kb].z - mean, d = xf[kb].w - mean;
            ss += (a * a + bq * bq) + (c * c + d * d);
        }
        ss += __shfl_xor(ss, 32);
        const float rstd = 1.0f / sqrtf(ss * (1.0f / 128.0f) + GENIE_LN_EPS);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const float4 gm = *reinterpret_cast<const float4*>(ln_out_g + kb * 8 + 4 * h);
            const float4 bt = *reinterpret_cast<const float4*>(ln_out_b + kb * 8 + 4 * h);
            xf[kb].x = (xf[kb].x - mean) * rstd * gm.x + bt.x;
            xf[kb].y = (xf[kb].y - mean) * rstd * gm.y + bt.y;
            xf[kb].z = (xf[kb].z - mean) * rstd * gm.z + bt.z;
            xf[kb].w = (xf[kb].w - mean) * rstd * gm.w + bt.w;
        }
    }
    // update, one 32-channel block at a time so only 16 accumulator registers are live
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
        f32x16 a = zero16();
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const int u = 64 + ob * 16 + kb, slot = u % WI_PD;
            wf_wait<WI_PD - 1>(wq[slot]);
            a = mfma_8k(xf[kb], wq[slot], a);
            __builtin_amdgcn_sched_barrier(0);
            wf_issue(wq[slot], frag_addr(min(u + WI_PD, 127)));
        }
        const float bzc = bz[ob * 32 + pl];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = acc_row(r, lane);
            const float gv = ob == 0 ? g0[r] : ob == 1 ? g1[r] : ob == 2 ? g2[r] : g3[r];
            if (t < nvalid) { float* q = zrow + (size_t)t * 128 + ob * 32 + pl; *q = (a[r] + bzc) * gv + *q; }
        }
    }
#pragma unroll
    for (int s = 0; s < WI_PD; ++s) wf_wait<0>(wq[s]);
}

// ---------------------------------------------------------------------------------------------
// launchers (same signatures as the tile versions; selected in genie_api.hip)
// ---------------------------------------------------------------------------------------------
void launch_trimul_proj_wi(genie_ctx* h, hipStream_t st, const TriMulW& w, bool outgoing) {
    const int N = h->N, ntile = (N + 31) / 32;
    const int n_wt = h->B * N * ntile;
    dim3 grid((n_wt + 3) / 4);
    if (outgoing)
        hipLaunchKernelGGL(k_trimul_proj_wi<true>, grid, dim3(256), 0, st, h->p, h->rmaskf, w.proj_w, w.proj_b, w.ln_in_g,
                           w.ln_in_b, h->acm, h->bcm, N, h->NP, n_wt);
    else
        hipLaunchKernelGGL(k_trimul_proj_wi<false>, grid, dim3(256), 0, st, h->p, h->rmaskf, w.proj_w, w.proj_b, w.ln_in_g,
                           w.ln_in_b, h->acm, h->bcm, N, h->NP, n_wt);
}

void launch_trimul_out_wi(genie_ctx* h, hipStream_t st, const TriMulW& w) {
    const int N = h->N, ntile = (N + 31) / 32;
    const int n_wt = h->B * N * ntile;
    hipLaunchKernelGGL(k_trimul_out_wi, dim3((n_wt + 3) / 4), dim3(256), 0, st, h->p, h->xcm, w.g_w, w.g_b, w.z_w, w.z_b,
                       w.ln_in_g, w.ln_in_b, w.ln_out_g, w.ln_out_b, N, h->NP, n_wt);
}

void launch_pair_transition_wi(genie_ctx* h, hipStream_t st, const PairLayerW& w) {
    const long long M = (long long)h->B * h->N * h->N;
    const long long n_wt = (M + 31) / 32;
    hipLaunchKernelGGL(k_pair_transition_wi, dim3((unsigned)((n_wt + 3) / 4)), dim3(256), 0, st, h->p, h->rmaskf, w.pt_ln_g,
                       w.pt_ln_b, w.pt_w1, w.pt_b1, w.pt_w2, w.pt_b2, h->N, M, h->d.pair_transition_n * 4);
}
