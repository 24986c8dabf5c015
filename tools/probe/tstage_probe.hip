// Probe: why does a transition stage of the fused chain (48 f16 MFMAs per wave: 24 into one accumulator, ReLU + split, 24 into four)
// take ~4.5 k cycles per 96 MFMAs of a SIMD instead of 3,072?  One work-group per CU, T threads; per iteration each wave runs the
// stage's instruction stream with parts knocked out.  Prints cycles per iteration (s_memtime, work-group 0), wall time and the clock.
//   hipcc -O3 --offload-arch=gfx950 -o tools/probe/tstage_probe tools/probe/tstage_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFH(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ h8 frag(const unsigned char* st, int u, int part, int lane) { return *reinterpret_cast<const h8*>(st + u * 2048 + part * 1024 + lane * 16); }
__device__ __forceinline__ void split2(float a, float b, float s, unsigned& hi, unsigned& lo) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(a), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(b), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(a), "v"(s), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(b), "v"(s), "v"(hi));
}
__device__ __forceinline__ void split8(const float (&x)[8], float s, h8& hi, h8& lo) {
    u32x4 h, l; unsigned a, b;
    split2(x[0], x[1], s, a, b); h.x = a; l.x = b; split2(x[2], x[3], s, a, b); h.y = a; l.y = b;
    split2(x[4], x[5], s, a, b); h.z = a; l.z = b; split2(x[6], x[7], s, a, b); h.w = a; l.w = b;
    hi = __builtin_bit_cast(h8, h); lo = __builtin_bit_cast(h8, l);
}
// LDSR: fragments re-read from LDS per k-chunk; VAL: ReLU + split between the GEMMs; BAR: s_barrier per stage;
// DEP: 1 = MFH3 into one accumulator (as the kernel), 0 = three accumulators round-robin; SKEW: waves >= T/128 run the two halves of the
// stage in the opposite order (W2-like first)
template <int T, bool LDSR, bool VAL, bool BAR, bool DEP, bool SKEW>
__global__ __launch_bounds__(T, 1) void probe(float* out, int iters, long long* cyc, const float* rnd) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 16384; i += T) reinterpret_cast<float*>(lds)[i] = rnd[(i * 7 + blockIdx.x) & 16383];
    __syncthreads();
    f32x16 d, v[4], x1, x2;
    h8 zh[8], zl[8];
    for (int i = 0; i < 16; ++i) { d[i] = 0; x1[i] = 0; x2[i] = 0; for (int k = 0; k < 4; ++k) v[k][i] = 0; }
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 8; ++i) { zh[k][i] = (_Float16)rnd[(lane * 64 + k * 8 + i) & 16383]; zl[k][i] = (_Float16)(0.001f * rnd[(lane * 64 + k * 8 + i + 5) & 16383]); }
    h8 ah[2] = {zh[0], zh[1]}, al[2] = {zl[0], zl[1]};
    const bool late = SKEW && wave >= T / 128;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned char* stage = lds + (it & 1) * 32768;
        auto gemm1 = [&]() {
            h8 wh = frag(stage, 0, 0, lane), wl = frag(stage, 0, 1, lane);
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const h8 nh = LDSR ? frag(stage, kc < 7 ? kc + 1 : 7, 0, lane) : wh, nl = LDSR ? frag(stage, kc < 7 ? kc + 1 : 7, 1, lane) : wl;
                FENCE();
#ifdef ORDER2      /* (lo,hi) (hi,hi) (hi,lo): each operand changes once per triple */
                if (DEP) { MFH(wl, zh[kc], d); MFH(wh, zh[kc], d); MFH(wh, zl[kc], d); }
#else
                if (DEP) { MFH(wl, zh[kc], d); MFH(wh, zl[kc], d); MFH(wh, zh[kc], d); }
#endif
                else { MFH(wl, zh[kc], d); MFH(wh, zl[kc], x1); MFH(wh, zh[kc], x2); }
                FENCE();
                wh = nh; wl = nl;
            }
        };
        auto relu = [&]() {
            if (!VAL) return;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = __builtin_fmaxf(d[8 * c + e], 0.f);
                split8(x, 0.37f, ah[c], al[c]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = reinterpret_cast<const float*>(lds)[(it * 32 + r * 2 + (lane >> 5)) & 1023];
        };
        auto gemm2 = [&]() {
            h8 bh = frag(stage, 8, 0, lane), bl = frag(stage, 8, 1, lane);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const h8 nh = LDSR ? frag(stage, 8 + (u < 7 ? u + 1 : 7), 0, lane) : bh, nl = LDSR ? frag(stage, 8 + (u < 7 ? u + 1 : 7), 1, lane) : bl;
                FENCE();
#ifdef ORDER2
                if (DEP) { MFH(bl, ah[u >> 2], v[u & 3]); MFH(bh, ah[u >> 2], v[u & 3]); MFH(bh, al[u >> 2], v[u & 3]); }
#else
                if (DEP) { MFH(bl, ah[u >> 2], v[u & 3]); MFH(bh, al[u >> 2], v[u & 3]); MFH(bh, ah[u >> 2], v[u & 3]); }
#endif
                else { MFH(bl, ah[u >> 2], v[u & 3]); MFH(bh, al[u >> 2], v[(u + 1) & 3]); MFH(bh, ah[u >> 2], v[(u + 2) & 3]); }
                FENCE();
                bh = nh; bl = nl;
            }
        };
        if (late) { gemm2(); FENCE(); gemm1(); FENCE(); relu(); }
        else { gemm1(); FENCE(); relu(); FENCE(); gemm2(); }
        if (BAR) { FENCE(); __builtin_amdgcn_s_barrier(); FENCE(); }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += d[i] + x1[i] + x2[i] + v[0][i] + v[1][i] + v[2][i] + v[3][i];
    out[blockIdx.x * T + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int T, bool LDSR, bool VAL, bool BAR, bool DEP, bool SKEW>
void run(const char* name, float* out, long long* cyc, const float* rnd) {
    const int iters = 2000, grid = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<T, LDSR, VAL, BAR, DEP, SKEW><<<grid, T>>>(out, 50, cyc, rnd);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<T, LDSR, VAL, BAR, DEP, SKEW><<<grid, T>>>(out, iters, cyc, rnd);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c[256]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)c[i] / 256;
    const double per_simd = 48.0 * (T / 256);                 // MFMAs per SIMD and iteration
    printf("%-44s T=%d  %7.0f cyc/iter = %5.1f cyc/MFMA/SIMD  %.3f ms  clock %.2f GHz  %.0f TFLOP/s\n", name, T, mean / iters, mean / iters / per_simd, ms,
           mean / (ms * 1e6), (double)grid * (T / 64) * 48.0 * iters * 32768.0 / ms / 1e9);
}

int main() {
    float *out, *rnd; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&rnd, 16384 * 4);
    float h[16384]; unsigned s = 12345; for (int i = 0; i < 16384; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    run<256, false, false, false, true, false>("1 wave/SIMD  MFMA only, dependent", out, cyc, rnd);
    run<256, false, false, false, false, false>("1 wave/SIMD  MFMA only, 3 accumulators", out, cyc, rnd);
    run<512, false, false, false, true, false>("2 waves/SIMD MFMA only, dependent", out, cyc, rnd);
    run<512, true, false, false, true, false>("2 waves/SIMD + LDS fragments", out, cyc, rnd);
    run<512, true, true, false, true, false>("2 waves/SIMD + LDS + ReLU/split", out, cyc, rnd);
    run<512, true, true, true, true, false>("2 waves/SIMD + LDS + ReLU/split + barrier", out, cyc, rnd);
    run<512, true, true, true, false, false>("... with 3 accumulators", out, cyc, rnd);
    run<512, true, true, true, true, true>("... waves 4-7 in the opposite order (skew)", out, cyc, rnd);
    run<512, true, true, false, true, true>("... skew, no barrier", out, cyc, rnd);
    run<256, true, true, true, true, false>("1 wave/SIMD  + LDS + ReLU/split + barrier", out, cyc, rnd);
    return 0;
}
