// Probe: does running the memory phases of one row tile beside the matrix phases of another on the same CU pay, on a power-limited chip?
// Work per "tile" of 32 pair rows per wave, as chain B of the fused pair kernels: loads 32 KB (x + z, dword and b128), NST stages of
// MFMAs (LDS fragment reads, a barrier per stage), stores 48 KB (z, a, b).
//   form 8x1: one 512-thread work-group per CU, 48 MFMAs per wave and stage (what the kernels do: phases of all waves coincide)
//   form 4x2: two 256-thread work-groups per CU, 24 MFMAs per wave and stage, twice the stages, the second group started half a tile late
// Same bytes, same MFMAs, same fragment reads per MFMA.  Prints wall time per tile and CU.
//   hipcc -O3 --offload-arch=gfx950 -o tools/probe/overlap_probe tools/probe/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MFH(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ h8 frag(const unsigned char* st, int u, int part, int lane) { return *reinterpret_cast<const h8*>(st + u * 2048 + part * 1024 + lane * 16); }

// NW waves per group; UPS units (2 KiB of fragments, 3 MFMAs each) per stage; NST stages per tile; DELAY: groups with odd blockIdx.x >> 8 ... see launch
template <int NW, int UPS>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void probe(float* out, const float* src, float* dst, int tiles, int nst, unsigned src_bytes,
                                                                  unsigned dst_bytes, const float* rnd, int half_delay, int mem) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];        // 2 stages of UPS units
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < UPS * 1024; i += NW * 64) reinterpret_cast<float*>(lds)[i] = rnd[(i * 7 + blockIdx.x) & 16383];
    __syncthreads();
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, dst_bytes, 0x00020000);
    f32x16 v[4];
    h8 zh[8], zl[8];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) v[k][i] = 0;
    for (int k = 0; k < 8; ++k) for (int i = 0; i < 8; ++i) { zh[k][i] = (_Float16)rnd[(lane * 64 + k * 8 + i) & 16383]; zl[k][i] = (_Float16)(0.001f * rnd[(lane * 64 + k * 8 + i + 5) & 16383]); }
    if (half_delay > 0 && (blockIdx.x & 1)) {       // the second group of a CU starts half a tile late (blocks b, b + 1 land on one CU only by luck: measured both ways)
        const long long t0 = __builtin_amdgcn_s_memtime();
        while ((long long)__builtin_amdgcn_s_memtime() - t0 < half_delay) __builtin_amdgcn_s_sleep(16);
    }
    float keep = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const unsigned slot = (unsigned)((blockIdx.x * 131u + t) * NW + wave);
        if (mem) {      // ---- loads: 32 KB per wave (128 dword loads of 256 B), consumed
            const int lbase = (int)((slot * 32768u) % (src_bytes - 65536u)) & ~255;
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < 128; q += 16) {
                float r[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) r[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, lbase + (q + u) * 256, 0));
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += r[u];
            }
            keep += acc;
            zh[0][0] = (_Float16)(keep * 1e-9f);
        }
        // ---- stages
        for (int s = 0; s < nst; ++s) {
            const unsigned char* stage = lds + (s & 1) * UPS * 2048;
            h8 wh = frag(stage, 0, 0, lane), wl = frag(stage, 0, 1, lane);
#pragma unroll
            for (int u = 0; u < UPS; ++u) {
                const h8 nh = frag(stage, u + 1 < UPS ? u + 1 : u, 0, lane), nl = frag(stage, u + 1 < UPS ? u + 1 : u, 1, lane);
                FENCE();
                MFH(wl, zh[u & 7], v[u & 3]); MFH(wh, zl[u & 7], v[u & 3]); MFH(wh, zh[u & 7], v[u & 3]);
                FENCE();
                wh = nh; wl = nl;
            }
            FENCE(); __builtin_amdgcn_s_barrier(); FENCE();
        }
        if (mem) {      // ---- stores: 48 KB per wave
            const int sbase = (int)((slot * 49152u) % (dst_bytes - 131072u)) & ~255;
#pragma unroll
            for (int q = 0; q < 192; ++q)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[q & 3][q & 15]), rd, lane * 4, sbase + q * 256, 0);
        }
    }
    float s = keep;
    for (int i = 0; i < 16; ++i) s += v[0][i] + v[1][i] + v[2][i] + v[3][i];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

int main() {
    float *out, *rnd, *src, *dst;
    const unsigned SB = 1u << 30, DB = 1u << 30;
    hipMalloc(&out, 512 * 512 * 4); hipMalloc(&rnd, 16384 * 4); hipMalloc(&src, SB); hipMalloc(&dst, DB);
    hipMemset(src, 0, SB);
    static float h[16384]; unsigned s = 12345; for (int i = 0; i < 16384; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<8, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 65536);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int tiles = 160;          // >= 10 ms per configuration: the clock the chip settles at under each mix is part of the answer
  for (int rep = 0; rep < 2; ++rep)
    for (int nst : {28, 12}) {
        for (int mem : {1, 0}) {
            float ms;
            // 8x1: LDS sized so that only one group fits per CU
            probe<8, 16><<<256, 512, 131072>>>(out, src, dst, 40, nst, SB, DB, rnd, 0, mem);
            hipDeviceSynchronize(); hipEventRecord(a);
            probe<8, 16><<<256, 512, 131072>>>(out, src, dst, tiles, nst, SB, DB, rnd, 0, mem);
            hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
            printf("stages %2d mem %d  8 waves x 1 group            : %7.1f us per tile (256 pair rows) \n", nst, mem, ms * 1e3 / tiles);
            for (int delay : {0, 60000}) {
                probe<4, 8><<<512, 256, 70 * 1024>>>(out, src, dst, 40, 2 * nst, SB, DB, rnd, delay, mem);
                hipDeviceSynchronize(); hipEventRecord(a);
                probe<4, 8><<<512, 256, 70 * 1024>>>(out, src, dst, tiles, 2 * nst, SB, DB, rnd, delay, mem);
                hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
                printf("stages %2d mem %d  4 waves x 2 groups, delay %5d : %7.1f us per 2 tiles of 128 rows\n", nst, mem, delay, ms * 1e3 / tiles);
            }
        }
    }
    return 0;
}
