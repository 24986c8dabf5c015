// Measurement aid (include/genie_hip.h, "measurement"): what dense f16 MFMA work this chip SUSTAINS, measured live.
// The roofline prices the pair-stack kernels against the 2.5 PFLOP/s dense f16 peak of MI355X_MICROARCH.md (2.4 GHz x 32 cycles per
// v_mfma_f32_32x32x16_f16); under matrix load the chip lowers its clock until it fits its power budget, and a loop of nothing but
// these MFMAs on random data holds 1.3 - 1.7 PFLOP/s whatever its cycle efficiency (tools/probe/tstage_probe.hip).
// genie_probe_mfma runs the instruction stream of one transition stage of the fused chain (48 MFMAs per wave: 24 chained into one
// accumulator from LDS weight fragments, ReLU + split, 24 into four accumulators; a barrier per stage; one 512-thread work-group per
// CU) for about `ms_target` milliseconds and returns the rate, so that bench.py can print the kernels' matrix rate next to what the
// device it ran on can sustain.  Nothing of the product path calls it.
#include "hx.h"

__global__ __launch_bounds__(512, 1) void k_probe_tstage(float* __restrict__ out, int iters, const float* __restrict__ rnd) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float*>(lds)[i] = rnd[(i * 7 + blockIdx.x) & 16383];
    __syncthreads();
    f32x16 d, v[4];
    h8 zh[8], zl[8];
    for (int i = 0; i < 16; ++i) { d[i] = 0; for (int k = 0; k < 4; ++k) v[k][i] = 0; }
    for (int k = 0; k < 8; ++k)
        for (int i = 0; i < 8; ++i) { zh[k][i] = (_Float16)rnd[(lane * 64 + k * 8 + i) & 16383]; zl[k][i] = (_Float16)(0.001f * rnd[(lane * 64 + k * 8 + i + 5) & 16383]); }
    h8 ah[2] = {zh[0], zh[1]}, al[2] = {zl[0], zl[1]};
    for (int it = 0; it < iters; ++it) {
        const unsigned char* stage = lds + (it & 1) * 32768;
        {
            h8 wh = hx_frag(stage, 0, 0, lane), wl = hx_frag(stage, 0, 1, lane);
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                const h8 nh = hx_frag(stage, kc < 7 ? kc + 1 : 7, 0, lane), nl = hx_frag(stage, kc < 7 ? kc + 1 : 7, 1, lane);
                __builtin_amdgcn_sched_barrier(0);
                MFH3(wh, wl, zh[kc], zl[kc], d);
                __builtin_amdgcn_sched_barrier(0);
                wh = nh; wl = nl;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = fmaxf(d[8 * c + e], 0.f);
            hx_split8(x, 0.37f, ah[c], al[c]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = reinterpret_cast<const float*>(lds)[(it * 32 + r * 2 + (lane >> 5)) & 1023];
        {
            h8 bh = hx_frag(stage, 8, 0, lane), bl = hx_frag(stage, 8, 1, lane);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const h8 nh = hx_frag(stage, 8 + (u < 7 ? u + 1 : 7), 0, lane), nl = hx_frag(stage, 8 + (u < 7 ? u + 1 : 7), 1, lane);
                __builtin_amdgcn_sched_barrier(0);
                MFH3(bh, bl, ah[u >> 2], al[u >> 2], v[u & 3]);
                __builtin_amdgcn_sched_barrier(0);
                bh = nh; bl = nl;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // keep the values bounded (the rate must not depend on overflowed / NaN operands)
        if ((it & 63) == 63)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[k][r] *= 1e-6f;
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += d[i] + v[0][i] + v[1][i] + v[2][i] + v[3][i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

extern "C" int genie_probe_mfma(void* stream, double ms_target, double* tflops_out, double* ms_out) {
    if (!tflops_out) return -1;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, ncu = 256;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ncu = pr.multiProcessorCount;
    float *out = nullptr, *rnd = nullptr;
    if (hipMalloc(&out, (size_t)ncu * 512 * 4) != hipSuccess || hipMalloc(&rnd, 16384 * 4) != hipSuccess) return -2;
    {
        float* hbuf = (float*)malloc(16384 * 4);
        unsigned s = 12345;
        for (int i = 0; i < 16384; ++i) { s = s * 1664525u + 1013904223u; hbuf[i] = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
        (void)hipMemcpy(rnd, hbuf, 16384 * 4, hipMemcpyHostToDevice);
        free(hbuf);
    }
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    // calibrate the iteration count on a short launch, then time the long one (the clock the chip settles at is part of the answer)
    int iters = 400;
    float ms = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        (void)hipEventRecord(a, st);
        hipLaunchKernelGGL(k_probe_tstage, dim3(ncu), dim3(512), 0, st, out, iters, rnd);
        (void)hipEventRecord(b, st);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
        if (pass == 0) { const double want = ms_target > 0 ? ms_target : 20.0; iters = (int)(iters * want / (ms > 1e-3f ? ms : 1e-3f)); if (iters < 400) iters = 400; if (iters > 4000000) iters = 4000000; }
    }
    const double flop = (double)ncu * 8 * 48.0 * (double)iters * 32768.0;
    *tflops_out = flop / (ms * 1e-3) / 1e12;
    if (ms_out) *ms_out = ms;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(out); (void)hipFree(rnd);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
