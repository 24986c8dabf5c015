// Probe: the WI weight-ring loop in isolation (asm ring vs plain loads), 32-pair register tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../genie2_amd/csrc/common.h"

template <int MODE, int PD>
__global__ __launch_bounds__(256, 2) void ring(const float* __restrict__ wp, float* out, int passes) {
    const int lane = threadIdx.x & 63;
    float4 zf[16];
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) zf[kb] = make_float4(0.001f * lane, 0.5f, 0.25f * kb, 1.f);
    v4f wq[PD];
    auto addr = [&](int t) { return wfrag_ptr(wp, 16, ((t >> 5) & 7) + 8 * (t & 1), (t >> 1) & 15, lane); };
    if (MODE == 0) {
#pragma unroll
        for (int s = 0; s < PD; ++s) wf_issue(wq[s], addr(s));
    }
    f32x16 acc_sum = zero16();
#pragma unroll 1
    for (int pass = 0; pass < passes; ++pass) {
        f32x16 ap = zero16(), ag = zero16();
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int slot = (kb * 2 + j) % PD;
                if (MODE == 0) {
                    wf_wait<PD - 1>(wq[slot]);
                    if (j == 0) ap = mfma_8k(wq[slot], zf[kb], ap); else ag = mfma_8k(wq[slot], zf[kb], ag);
                    __builtin_amdgcn_sched_barrier(0);
                    wf_issue(wq[slot], addr(min(pass * 32 + kb * 2 + j + PD, passes * 32 - 1)));
                } else if (MODE == 1) {
                    const float4 w = *reinterpret_cast<const float4*>(addr(pass * 32 + kb * 2 + j));
                    if (j == 0) ap = mfma_8k(w, zf[kb], ap); else ag = mfma_8k(w, zf[kb], ag);
                } else {   // MODE 2: no weight loads at all
                    if (j == 0) ap = mfma_8k(zf[(kb + 1) & 15], zf[kb], ap); else ag = mfma_8k(zf[(kb + 2) & 15], zf[kb], ag);
                }
            }
        }
        if (MODE == 3) {}
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_sum[r] += ap[r] * ag[r];
    }
    if (MODE == 0) {
#pragma unroll
        for (int s = 0; s < PD; ++s) wf_wait<0>(wq[s]);
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc_sum[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int PD>
void run(const char* name, const float* w, float* out, int wg_per_cu) {
    const int passes = 64, grid = 256 * wg_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    ring<MODE, PD><<<grid, 256>>>(w, out, 8);
    hipDeviceSynchronize();
    hipEventRecord(a);
    ring<MODE, PD><<<grid, 256>>>(w, out, passes);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)grid * 4 * passes * 128 * 4096.0;
    printf("%-36s wg/cu=%d  %.3f ms  %.1f TFLOP/s\n", name, wg_per_cu, ms, flop / ms / 1e9);
}

int main() {
    float *w, *out;
    hipMalloc(&w, 16 * 16 * 256 * sizeof(float));
    hipMemset(w, 0, 16 * 16 * 256 * sizeof(float));
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int n = 1; n <= 2; ++n) run<2, 8>("no weight loads", w, out, n);
    for (int n = 1; n <= 2; ++n) run<1, 8>("compiler loads", w, out, n);
    for (int n = 1; n <= 2; ++n) run<0, 8>("asm ring PD=8", w, out, n);
    for (int n = 1; n <= 2; ++n) run<0, 4>("asm ring PD=4", w, out, n);
    for (int n = 1; n <= 2; ++n) run<0, 16>("asm ring PD=16", w, out, n);
    return 0;
}
