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

// LayerNorm backward of row-major dy [B][R][128], dx written channel-major [B][128][R]; gamma / beta gradients: one atomic per column
// and block
__global__ __launch_bounds__(256) void k_ln_bwd_to_cm(const float* __restrict__ dy, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                      const float* __restrict__ g, float* __restrict__ dxcm, int R, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta) {
    constexpr int C = 128;
    __shared__ float t[C][33];
    __shared__ float red[2][8][C];
    const int b = blockIdx.y, r0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float4 gv = *reinterpret_cast<const float4*>(g + 4 * tx);
    float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sb = sg;
    for (int q = ty; q < 32; q += 8) {
        const int row = r0 + q;
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f), xh = d;
        float rs = 0.f;
        if (row < R) {
            const size_t e = ((size_t)b * R + row) * C + 4 * tx;
            d = *reinterpret_cast<const float4*>(dy + e);
            xh = *reinterpret_cast<const float4*>(xhat + e);
            rs = rstd[(size_t)b * R + row];
        }
        sg.x += d.x * xh.x; sg.y += d.y * xh.y; sg.z += d.z * xh.z; sg.w += d.w * xh.w;
        sb.x += d.x; sb.y += d.y; sb.z += d.z; sb.w += d.w;
        const float4 tq = make_float4(d.x * gv.x, d.y * gv.y, d.z * gv.z, d.w * gv.w);
        float s1 = (tq.x + tq.y) + (tq.z + tq.w), s2 = (tq.x * xh.x + tq.y * xh.y) + (tq.z * xh.z + tq.w * xh.w);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }      // the 32 lanes of a row: one half-wave
        const float m1 = s1 * (1.0f / C), m2 = s2 * (1.0f / C);
        t[4 * tx][q] = rs * (tq.x - m1 - xh.x * m2); t[4 * tx + 1][q] = rs * (tq.y - m1 - xh.y * m2);
        t[4 * tx + 2][q] = rs * (tq.z - m1 - xh.z * m2); t[4 * tx + 3][q] = rs * (tq.w - m1 - xh.w * m2);
    }
    if (dgamma) {
        red[0][ty][4 * tx] = sg.x; red[0][ty][4 * tx + 1] = sg.y; red[0][ty][4 * tx + 2] = sg.z; red[0][ty][4 * tx + 3] = sg.w;
        red[1][ty][4 * tx] = sb.x; red[1][ty][4 * tx + 1] = sb.y; red[1][ty][4 * tx + 2] = sb.z; red[1][ty][4 * tx + 3] = sb.w;
    }
    __syncthreads();
    float* dst = dxcm + (size_t)b * C * R;
    for (int c = ty; c < C; c += 8)
        if (r0 + tx < R) dst[(size_t)c * R + r0 + tx] = t[c][tx];
    if (dgamma && threadIdx.x < C) {
        const int c = threadIdx.x;
        float a = 0.f, bb = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a += red[0][k][c]; bb += red[1][k][c]; }
        atomicAdd(dgamma + c, a);
        atomicAdd(dbeta + c, bb);
    }
}
void launch_ln_bwd_to_cm(hipStream_t st, const float* dy, const float* xhat, const float* rstd, const float* g, float* dxcm, int B, int R, int C,
                         float* dgamma, float* dbeta) {
    (void)C;
    hipLaunchKernelGGL(k_ln_bwd_to_cm, dim3((R + 31) / 32, B), dim3(256), 0, st, dy, xhat, rstd, g, dxcm, R, dgamma, dbeta);
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
    for (int k = lane; k < e.K; k += 64) {
        const float w = W[k];
        Wf[k] = w * wts[e.g + k];
        s += w * wts[e.beta + k];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) dst[e.dst + (size_t)e.O * e.K + o] = (e.b >= 0 ? wts[e.b + o] : 0.f) + s;
}
void launch_fold_ln_table(hipStream_t st, const float* wts, float* dst, const FoldEntry* table_dev, int n_entries, int max_O) {
    if (n_entries > 0) hipLaunchKernelGGL(k_fold_ln_table, dim3((max_O + 3) / 4, n_entries), dim3(256), 0, st, wts, dst, table_dev);
}
