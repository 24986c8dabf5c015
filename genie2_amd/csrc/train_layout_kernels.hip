// Training path: the layout changes around the triangle multiplication's contraction, fused with the elementwise / LayerNorm work next to
// them (triangular_multiplicative_update.py:99-108 and their derivatives).  The contraction's operands are channel-major [B][C][N N],
// everything else is row-major [B][N N][C]; each kernel here does its arithmetic INSIDE the layout change, through an LDS tile, so that a
// tensor of B N N C floats is read once and written once where the separate passes (gate / LayerNorm kernel + k_transpose) read and
// wrote it two or three times.  C = 128 (every Genie 2 configuration); other widths keep the separate passes (train_cm_fusable).
#include "common.h"
#include "train.h"

bool train_cm_fusable(int C) { return C == 128; }

// a = ap m sigmoid(ag), b = bp m sigmoid(bg): row-major in, channel-major out; tile = 32 positions x 32 channels
__global__ __launch_bounds__(256) void k_gate_to_cm(const float* __restrict__ ap, const float* __restrict__ ag, const float* __restrict__ bp,
                                                    const float* __restrict__ bg, const float* __restrict__ rmask, float* __restrict__ acm,
                                                    float* __restrict__ bcm, int N, int C) {
    __shared__ float ta[32][33], tb[32][33];
    const int b = blockIdx.z, R = N * N;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t base = (size_t)b * R * C;
    for (int q = ty; q < 32; q += 8) {
        const int row = r0 + q;
        float a = 0.f, bb = 0.f;
        if (row < R) {
            const int i = row / N, j = row - i * N;
            const float m = rmask[b * N + i] * rmask[b * N + j];
            const size_t e = base + (size_t)row * C + c0 + tx;
            a = ap[e] * m / (1.0f + expf(-ag[e]));
            bb = bp[e] * m / (1.0f + expf(-bg[e]));
        }
        ta[q][tx] = a; tb[q][tx] = bb;
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8)
        if (r0 + tx < R) {
            const size_t e = base + (size_t)(c0 + q) * R + r0 + tx;
            acm[e] = ta[tx][q]; bcm[e] = tb[tx][q];
        }
}
void launch_gate_to_cm(hipStream_t st, const float* ap, const float* ag, const float* bp, const float* bg, const float* rmask, float* acm, float* bcm,
                       int B, int N, int C) {
    hipLaunchKernelGGL(k_gate_to_cm, dim3((N * N + 31) / 32, C / 32, B), dim3(256), 0, st, ap, ag, bp, bg, rmask, acm, bcm, N, C);
}

// x channel-major [B][128][R] -> LayerNorm over the channels of each position -> y = xhat g + b and xhat row-major [B][R][128], rstd.
// tile = 32 positions x 128 channels: 128 coalesced 128-B rows in, 32 rows of 512 B out
__global__ __launch_bounds__(256) void k_ln_from_cm(const float* __restrict__ xcm, const float* __restrict__ g, const float* __restrict__ bt,
                                                    float* __restrict__ y, float* __restrict__ xhat, float* __restrict__ rstd, int R) {
    constexpr int C = 128;
    __shared__ float t[C][33];
    __shared__ float smean[32], srs[32];
    const int b = blockIdx.y, r0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = xcm + (size_t)b * C * R;
    for (int c = ty; c < C; c += 8) t[c][tx] = r0 + tx < R ? src[(size_t)c * R + r0 + tx] : 0.f;
    __syncthreads();
    {   // statistics: 8 lanes per position (16 channels each)
        const int p = threadIdx.x >> 3, sub = threadIdx.x & 7;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += t[sub + 8 * k][p];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        const float mean = s * (1.0f / C);
        float ss = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float d = t[sub + 8 * k][p] - mean; ss += d * d; }
        ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
        if (sub == 0) { smean[p] = mean; srs[p] = 1.0f / sqrtf(ss * (1.0f / C) + GENIE_LN_EPS); }
    }
    __syncthreads();
    const float4 gv = *reinterpret_cast<const float4*>(g + 4 * tx), bv = *reinterpret_cast<const float4*>(bt + 4 * tx);
    for (int q = ty; q < 32; q += 8) {
        const int row = r0 + q;
        if (row >= R) continue;
        const float mean = smean[q], rs = srs[q];
        float4 xh;
        xh.x = (t[4 * tx][q] - mean) * rs; xh.y = (t[4 * tx + 1][q] - mean) * rs; xh.z = (t[4 * tx + 2][q] - mean) * rs; xh.w = (t[4 * tx + 3][q] - mean) * rs;
        const size_t e = ((size_t)b * R + row) * C + 4 * tx;
        *reinterpret_cast<float4*>(xhat + e) = xh;
        if (y) *reinterpret_cast<float4*>(y + e) = make_float4(xh.x * gv.x + bv.x, xh.y * gv.y + bv.y, xh.z * gv.z + bv.z, xh.w * gv.w + bv.w);
        if (tx == 0) rstd[(size_t)b * R + row] = rs;
    }
}
void launch_ln_from_cm(hipStream_t st, const float* xcm, const float* g, const float* b, float* y, float* xhat, float* rstd, int B, int R, int C) {
    (void)C;
    hipLaunchKernelGGL(k_ln_from_cm, dim3((R + 31) / 32, B), dim3(256), 0, st, xcm, g, b, y, xhat, rstd, R);
}

// LayerNorm backward of row-major dy [B][R][128], dx written channel-major [B][128][R]; gamma / beta gradients.  A work-group takes
// LBT_TILES tiles of 64 rows: every load of a tile is in flight at once, the channel-major stores are 256-byte runs, and the gamma /
// beta sums stay in registers across the tiles -- one atomic per column per 128 rows (the 32-row form spent 256 atomics per 32 rows
// and ran at 1.9 TB/s beside its atomic-free neighbours' 5.7)
#define LBT_TILES 2
__global__ __launch_bounds__(256) void k_ln_bwd_to_cm(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                      const float* __restrict__ g, float* __restrict__ dxcm, int R, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta) {
    constexpr int C = 128;
    __shared__ float t[C][65];
    const int b = blockIdx.y;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float4 gv = *reinterpret_cast<const float4*>(g + 4 * tx);
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = sg;
    float* dst = dxcm + (size_t)b * C * R;
    for (int tile = 0; tile < LBT_TILES; ++tile) {
        const int r0 = (blockIdx.x * LBT_TILES + tile) * 64;
        if (r0 >= R) break;                                 // uniform
        float4 d[8], xh[8];
        float rs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = r0 + ty + 8 * k;
            d[k] = make_float4(0.f, 0.f, 0.f, 0.f); xh[k] = d[k]; rs[k] = 0.f;
            if (row < R) {
                const size_t e = ((size_t)b * R + row) * C + 4 * tx;
                d[k] = *reinterpret_cast<const float4*>(dy + e);
                xh[k] = *reinterpret_cast<const float4*>(xhat + e);
                rs[k] = rstd[(size_t)b * R + row];
            }
        }
        if (tile) __syncthreads();                          // the previous tile has been stored
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int q = ty + 8 * k;
            sg.x += d[k].x * xh[k].x; sg.y += d[k].y * xh[k].y; sg.z += d[k].z * xh[k].z; sg.w += d[k].w * xh[k].w;
            sb.x += d[k].x; sb.y += d[k].y; sb.z += d[k].z; sb.w += d[k].w;
            const float4 tq = make_float4(d[k].x * gv.x, d[k].y * gv.y, d[k].z * gv.z, d[k].w * gv.w);
            float s1 = (tq.x + tq.y) + (tq.z + tq.w), s2 = (tq.x * xh[k].x + tq.y * xh[k].y) + (tq.z * xh[k].z + tq.w * xh[k].w);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }      // the 32 lanes of a row: one half-wave
            const float m1 = s1 * (1.0f / C), m2 = s2 * (1.0f / C);
            t[4 * tx][q] = rs[k] * (tq.x - m1 - xh[k].x * m2); t[4 * tx + 1][q] = rs[k] * (tq.y - m1 - xh[k].y * m2);
            t[4 * tx + 2][q] = rs[k] * (tq.z - m1 - xh[k].z * m2); t[4 * tx + 3][q] = rs[k] * (tq.w - m1 - xh[k].w * m2);
        }
        __syncthreads();
        const int rr = threadIdx.x & 63;
        for (int c = threadIdx.x >> 6; c < C; c += 4)
            if (r0 + rr < R) dst[(size_t)c * R + r0 + rr] = t[c][rr];
    }
    if (dgamma) {                                           // uniform
        __syncthreads();
        float* red = &t[0][0];                              // [2][8][C]
        *reinterpret_cast<float4*>(red + (0 * 8 + ty) * C + 4 * tx) = sg;
        *reinterpret_cast<float4*>(red + (1 * 8 + ty) * C + 4 * tx) = sb;
        __syncthreads();
        const int which = threadIdx.x >> 7, c = threadIdx.x & 127;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a += red[(which * 8 + k) * C + c];
        atomicAdd((which ? dbeta : dgamma) + c, a);
    }
}
void launch_ln_bwd_to_cm(hipStream_t st, const float* dy, const float* xhat, const float* rstd, const float* g, float* dxcm, int B, int R, int C,
                         float* dgamma, float* dbeta) {
    (void)C;
    hipLaunchKernelGGL(k_ln_bwd_to_cm, dim3((R + 64 * LBT_TILES - 1) / (64 * LBT_TILES), B), dim3(256), 0, st, dy, xhat, rstd, g, dxcm, R, dgamma, dbeta);
}

// backward of a = ap m sigmoid(ag) (and of b): with a itself at hand,  d ap = da m s,  d ag = da a (1 - s)   (s = sigmoid(ag); a = ap m s).
// da, db, a, b channel-major in; ag, bg and the four results row-major
__global__ __launch_bounds__(256) void k_gate_bwd_from_cm(const float* __restrict__ dacm, const float* __restrict__ dbcm, const float* __restrict__ acm,
                                                          const float* __restrict__ bcm, const float* __restrict__ ag, const float* __restrict__ bg,
                                                          const float* __restrict__ rmask, float* __restrict__ dap, float* __restrict__ dag,
                                                          float* __restrict__ dbp, float* __restrict__ dbg, int N, int C, int ldo) {
    __shared__ float t[4][32][33];
    const int b = blockIdx.z, R = N * N;
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t base = (size_t)b * R * C;
    for (int q = ty; q < 32; q += 8) {
        const bool ok = r0 + tx < R;
        const size_t e = base + (size_t)(c0 + q) * R + r0 + tx;
        t[0][q][tx] = ok ? dacm[e] : 0.f; t[1][q][tx] = ok ? dbcm[e] : 0.f;
        t[2][q][tx] = ok ? acm[e] : 0.f; t[3][q][tx] = ok ? bcm[e] : 0.f;
    }
    __syncthreads();
    for (int q = ty; q < 32; q += 8) {
        const int row = r0 + q;
        if (row >= R) continue;
        const int i = row / N, j = row - i * N;
        const float m = rmask[b * N + i] * rmask[b * N + j];
        const size_t e = base + (size_t)row * C + c0 + tx;
        const float sa = 1.0f / (1.0f + expf(-ag[e])), sb = 1.0f / (1.0f + expf(-bg[e]));
        const float da = t[0][tx][q], db = t[1][tx][q], a = t[2][tx][q], bb = t[3][tx][q];
        const size_t o = ((size_t)b * R + row) * ldo + c0 + tx;       // the results may be column blocks of a wider row-major matrix
        dap[o] = da * m * sa; dag[o] = da * a * (1.0f - sa);
        dbp[o] = db * m * sb; dbg[o] = db * bb * (1.0f - sb);
    }
}
void launch_gate_bwd_from_cm(hipStream_t st, const float* dacm, const float* dbcm, const float* acm, const float* bcm, const float* ag, const float* bg,
                             const float* rmask, float* dap, float* dag, float* dbp, float* dbg, int B, int N, int C, int ldo) {
    hipLaunchKernelGGL(k_gate_bwd_from_cm, dim3((N * N + 31) / 32, C / 32, B), dim3(256), 0, st, dacm, dbcm, acm, bcm, ag, bg, rmask, dap, dag, dbp, dbg,
                       N, C, ldo);
}

__global__ void k_rank1_add(float* __restrict__ dW, const float* __restrict__ beta, const float* __restrict__ db, int O, int C) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < O * C) dW[e] += beta[e % C] * db[e / C];
}
void launch_rank1_add(hipStream_t st, float* dW, const float* beta, const float* db, int O, int C) {
    hipLaunchKernelGGL(k_rank1_add, dim3((O * C + 255) / 256), dim3(256), 0, st, dW, beta, db, O, C);
}

// one wave per output row o: Wf[o][:] = W[o][:] * gamma, bf[o] = b[o] + <W[o][:], beta>
__global__ __launch_bounds__(256) void k_fold_ln(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float* __restrict__ Wf, float* __restrict__ bf, int O, int K) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= O) return;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(size_t)o * K + k];
        Wf[(size_t)o * K + k] = w * gamma[k];
        s += w * beta[k];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) bf[o] = (b ? b[o] : 0.f) + s;
}
void launch_fold_ln(hipStream_t st, const float* W, const float* b, const float* gamma, const float* beta, float* Wf, float* bf, int O, int K) {
    hipLaunchKernelGGL(k_fold_ln, dim3((O + 3) / 4), dim3(256), 0, st, W, b, gamma, beta, Wf, bf, O, K);
}

__global__ __launch_bounds__(256) void k_fold_ln_table(const float* __restrict__ wts, float* __restrict__ dst, const FoldEntry* __restrict__ tab) {
    const FoldEntry e = tab[blockIdx.y];
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= e.O) return;
    const float* W = wts + e.w + (size_t)o * e.K;
    float* Wf = dst + e.dst + (size_t)o * e.K;
    float s = 0.f;
    float* Wr = e.raw >= 0 ? dst + e.raw + (size_t)o * e.K : nullptr;
    for (int k = lane; k < e.K; k += 64) {
        const float w = W[k];
        Wf[k] = w * wts[e.g + k];
        if (Wr) Wr[k] = w;
        s += w * wts[e.beta + k];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) dst[e.bdst + o] = (e.b >= 0 ? wts[e.b + o] : 0.f) + s;
}
void launch_fold_ln_table(hipStream_t st, const float* wts, float* dst, const FoldEntry* table_dev, int n_entries, int max_O) {
    if (n_entries > 0) hipLaunchKernelGGL(k_fold_ln_table, dim3((max_O + 3) / 4, n_entries), dim3(256), 0, st, wts, dst, table_dev);
}

__global__ __launch_bounds__(256) void k_rank1_table(float* __restrict__ grads, const float* __restrict__ wts, const FoldEntry* __restrict__ tab) {
    const FoldEntry e = tab[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < e.O * e.K) grads[e.w + i] += wts[e.beta + i % e.K] * grads[e.b + i / e.K];
}
void launch_rank1_table(hipStream_t st, float* grads, const float* wts, const FoldEntry* table_dev, int n_entries, int max_OK) {
    if (n_entries > 0) hipLaunchKernelGGL(k_rank1_table, dim3((max_OK + 255) / 256, n_entries), dim3(256), 0, st, grads, wts, table_dev);
}
