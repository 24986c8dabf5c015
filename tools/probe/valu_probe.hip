// Probe: do f32 MFMAs (v_mfma_f32_32x32x2_f32) overlap with f32 VALU work on the same SIMD?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int NV>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; c2[i] = 0; c3[i] = 0; }
    float a = 1.0f + lane * 1e-3f, b = 0.5f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane * 0.01f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            MF(a, b, c0); MF(a, b, c1); MF(a, b, c2); MF(a, b, c3);
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);   // NV independent-ish VALU per 4 MFMAs
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
void run(float* out, int wg) {
    const int iters = 4096, grid = 256 * wg;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<NV><<<grid, 256>>>(out, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<NV><<<grid, 256>>>(out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)grid * 4 * iters * 16 * 4096.0;
    printf("VALU per 4 MFMAs = %2d  wg/cu=%d  %.3f ms  %.1f MFMA-TFLOP/s\n", NV, wg, ms, flop / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int wg = 1; wg <= 2; ++wg) { run<0>(out, wg); run<8>(out, wg); run<16>(out, wg); run<32>(out, wg); run<64>(out, wg); }
    return 0;
}
