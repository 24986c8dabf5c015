// Developer aid: what __builtin_amdgcn_permlane32_swap returns (gfx950).  hipcc --offload-arch=gfx950 -O2 swap_probe.hip -o swap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
}
int main() {
    unsigned* d; unsigned h[128];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("a = 100 + lane, b = 200 + lane\nr[0]: lane0 %u lane31 %u lane32 %u lane63 %u\nr[1]: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63],
           h[64], h[95], h[96], h[127]);
    return 0;
}
