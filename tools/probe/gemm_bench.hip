// Developer probe: times launch_gemm (genie2_amd/csrc/train_kernels.hip) shape by shape on the GPU, outside Python.
// build: hipcc --offload-arch=gfx950 -O2 -I genie2_amd/csrc tools/probe/gemm_bench.hip -L genie2_amd/lib -lgenie_hip -Wl,-rpath,'$ORIGIN/../../genie2_amd/lib' -o tools/probe/gemm_bench
// usage: gemm_bench [terms] [R]  -- one line per shape class of the training step's GEMMs; R = pair rows (32768 = N 128, batch 2)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "train.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Case { const char* name; int M, N, K; long long am, ak, bk, bn, cm, cn; int split; int mode; };

int main(int argc, char** argv) {
    const int terms = argc > 1 ? atoi(argv[1]) : 3;
    const long long R = argc > 2 ? atoll(argv[2]) : 32768;
    const size_t big = (size_t)R * 512;
    float *a, *b, *c;
    CK(hipMalloc(&a, big * 4)); CK(hipMalloc(&b, big * 4)); CK(hipMalloc(&c, big * 4));
    std::vector<float> h(big);
    for (size_t i = 0; i < big; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(a, h.data(), big * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, h.data(), big * 4, hipMemcpyHostToDevice));
    CK(hipMemset(c, 0, big * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const Case cases[] = {
        {"fwd  X[R,128] W[128,128]^T      ", (int)R, 128, 128, 128, 1, 1, 128, 128, 1, 1, 0},
        {"dX   dY[R,128] W[128,128]       ", (int)R, 128, 128, 128, 1, 128, 1, 128, 1, 1, 0},
        {"fwd  X[R,128] W[512,128]^T      ", (int)R, 512, 128, 128, 1, 1, 128, 512, 1, 1, 0},
        {"fwd  X[R,512] W[128,512]^T      ", (int)R, 128, 512, 512, 1, 1, 512, 128, 1, 1, 0},
        {"dW   dY[R,128]^T X[R,128] auto  ", 128, 128, (int)R, 1, 128, 128, 1, 128, 1, -1, 2},
        {"dW   dY[R,128]^T X[R,128] s=48  ", 128, 128, (int)R, 1, 128, 128, 1, 128, 1, 48, 2},
        {"dW   dY[R,128]^T X[R,128] s=96  ", 128, 128, (int)R, 1, 128, 128, 1, 128, 1, 96, 2},
        {"dW   dY[R,128]^T X[R,128] s=256 ", 128, 128, (int)R, 1, 128, 128, 1, 128, 1, 256, 2},      // >= 256 tiles of 128: the big kernel
        {"dW   dY[R,128]^T X[R,128] s=512 ", 128, 128, (int)R, 1, 128, 128, 1, 128, 1, 512, 2},
        {"dW   dY[R,512]^T X[R,128] auto  ", 512, 128, (int)R, 1, 512, 128, 1, 128, 1, -1, 2},
        {"dW   dY[R,512]^T X[R,128] s=128 ", 512, 128, (int)R, 1, 512, 128, 1, 128, 1, 128, 2},
        {"s    X[256,384] W[384,384]^T    ", 256, 384, 384, 384, 1, 1, 384, 384, 1, 1, 0},
        {"s    X[256,2112] W[384,2112]^T  ", 256, 384, 2112, 2112, 1, 1, 2112, 384, 1, 1, 0},
    };
    for (const Case& q : cases) {
        GemmP p{a, b, c, nullptr, q.M, q.N, q.K, q.am, q.ak, q.bk, q.bn, q.cm, q.cn, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0f, q.mode};
        p.nsplit = q.split < 0 ? gemm_splits(q.M, q.N, q.K, 1) : q.split;
        for (int i = 0; i < 3; ++i) launch_gemm(st, p, terms);
        CK(hipStreamSynchronize(st));
        const int reps = 20;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch_gemm(st, p, terms);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, gf = 2.0 * q.M * q.N * q.K / 1e9;
        const double mb = ((double)q.M * q.K + (double)q.N * q.K + (double)q.M * q.N) * 4 / 1e6;
        printf("%s split %3d  %7.1f us  %6.1f TF/s alg  %6.2f TB/s  (%.2f GF, %.1f MB)\n", q.name, p.nsplit, us, gf / us * 1e3, mb / us, gf, mb);      // GF / us = PF/s;  MB / us = TB/s
    }
    return 0;
}
