// Micro-probe: what MFMA rate does a wave stream reach under the access patterns of the pair kernels?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ w, float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 132];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 132; i += 256) lds[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x16 c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; c2[i] = 0; c3[i] = 0; }
    float4 a = make_float4(1.f, 2.f, 3.f, 4.f), b = make_float4(0.5f, 0.25f, 0.125f, 1.f);
    for (int it = 0; it < iters; ++it) {
        const int kb = it & 15;
        if (MODE >= 1) {   // operands from LDS like lfrag
            a = *reinterpret_cast<const float4*>(lds + (lane & 31) * 132 + kb * 8 + 4 * (lane >> 5));
            b = *reinterpret_cast<const float4*>(lds + (32 + (lane & 31)) * 132 + kb * 8 + 4 * (lane >> 5));
        }
        float4 wa = a, wb = b;
        if (MODE >= 2) {   // weight fragments from global (compiler-scheduled)
            wa = *reinterpret_cast<const float4*>(w + ((size_t)((blockIdx.x & 7) * 16 + kb) * 64 + lane) * 4);
            wb = *reinterpret_cast<const float4*>(w + ((size_t)((8 + (blockIdx.x & 7)) * 16 + kb) * 64 + lane) * 4);
        }
        if (MODE == 3) {          // one dependent chain
            MF(wa.x, a.x, c0); MF(wa.y, a.y, c0); MF(wa.z, a.z, c0); MF(wa.w, a.w, c0);
            MF(wb.x, b.x, c0); MF(wb.y, b.y, c0); MF(wb.z, b.z, c0); MF(wb.w, b.w, c0);
            MF(wa.x, b.x, c0); MF(wa.y, b.y, c0); MF(wa.z, b.z, c0); MF(wa.w, b.w, c0);
            MF(wb.x, a.x, c0); MF(wb.y, a.y, c0); MF(wb.z, a.z, c0); MF(wb.w, a.w, c0);
        } else if (MODE == 4) {   // chains of 4 dependent MFMAs, alternating two accumulators
            MF(wa.x, a.x, c0); MF(wa.y, a.y, c0); MF(wa.z, a.z, c0); MF(wa.w, a.w, c0);
            MF(wb.x, b.x, c1); MF(wb.y, b.y, c1); MF(wb.z, b.z, c1); MF(wb.w, b.w, c1);
            MF(wa.x, b.x, c0); MF(wa.y, b.y, c0); MF(wa.z, b.z, c0); MF(wa.w, b.w, c0);
            MF(wb.x, a.x, c1); MF(wb.y, a.y, c1); MF(wb.z, a.z, c1); MF(wb.w, a.w, c1);
        } else {
        MF(wa.x, a.x, c0); MF(wa.x, b.x, c1); MF(wb.x, a.x, c2); MF(wb.x, b.x, c3);
        MF(wa.y, a.y, c0); MF(wa.y, b.y, c1); MF(wb.y, a.y, c2); MF(wb.y, b.y, c3);
        MF(wa.z, a.z, c0); MF(wa.z, b.z, c1); MF(wb.z, a.z, c2); MF(wb.z, b.z, c3);
        MF(wa.w, a.w, c0); MF(wa.w, b.w, c1); MF(wb.w, a.w, c2); MF(wb.w, b.w, c3);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, const float* w, float* out, int wg_per_cu) {
    const int iters = 2048, grid = 256 * wg_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<grid, 256>>>(w, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<grid, 256>>>(w, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)grid * 4 * iters * 16 * 4096.0;
    printf("%-28s wg/cu=%d  %.3f ms  %.1f TFLOP/s\n", name, wg_per_cu, ms, flop / ms / 1e9);
}

int main() {
    float *w, *out;
    hipMalloc(&w, 16 * 16 * 64 * 4 * sizeof(float) * 2);
    hipMemset(w, 0, 16 * 16 * 64 * 4 * sizeof(float) * 2);
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int n = 1; n <= 4; ++n) run<0>("regs only", w, out, n);
    for (int n = 1; n <= 4; ++n) run<1>("lds operands", w, out, n);
    for (int n = 1; n <= 4; ++n) run<2>("lds + global W frags", w, out, n);
    for (int n = 1; n <= 3; ++n) run<3>("one dependent chain (regs)", w, out, n);
    for (int n = 1; n <= 3; ++n) run<4>("chains of 4, 2 accs (regs)", w, out, n);
    return 0;
}
