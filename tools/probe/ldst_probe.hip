// Probe: how do HBM-missing loads interact with a wave's MFMA + store stream on one CU?
// One 512-thread work-group per CU (2 waves per SIMD).  Per iteration every wave issues NM dependent
// f16 MFMAs with NS streaming stores (256 B each) interleaved, like the hx projection kernel.  Loads
// (1-KiB LDS-DMA or global_load_dwordx4 into VGPRs, streaming = HBM misses) are added in several ways:
//   mode 0: none
//   mode 1: every wave issues NL loads at the top of its iteration (what the kernels do)
//   mode 2: wave 7 alone issues 8 NL loads per iteration and does nothing else ("loader wave")
//   mode 3: like 1, but the iteration's stores are packed into its first half (store-free second half)
//   mode 4: like 1, with plain global loads into VGPRs instead of LDS-DMA
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MFH(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int MODE, int NL>
__global__ __launch_bounds__(512, 1) void probe(float* out, const float* src, float* dst, int iters, unsigned src_bytes,
                                                unsigned dst_bytes, long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 8192];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, dst_bytes, 0x00020000);
    f32x16 c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; }
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * lane + i); b[i] = (_Float16)(0.5f + 0.001f * i); }
    v4f keep[4] = {};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // distinct streaming addresses per (block, iteration, wave): wraps inside the big buffers
        const unsigned slot = (unsigned)((blockIdx.x * 977u + it) * 8u + wave);
        const int lbase = (int)((slot * (unsigned)(NL * 1024)) % (src_bytes - 65536u)) & ~1023;
        const int sbase = (int)((slot * 4096u) % (dst_bytes - 65536u)) & ~255;
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int q = 0; q < NL; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wave * 8192 + q * 1024), 16,
                                                         lane * 16, lbase + q * 1024, 0, 0);
        }
        if (MODE == 4) {
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, lbase + q * 1024, 0));
                keep[q & 3] = v;        // (consumed after the loop only)
            }
        }
        if (MODE == 2 && wave == 7) {
#pragma unroll
            for (int q = 0; q < 8 * NL; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (q & 7) * 8192 + (q >> 3) * 1024), 16,
                                                         lane * 16, lbase + q * 1024, 0, 0);
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                FENCE(); MFH(a, b, c0); FENCE();
                if (MODE == 3) {
                    if (g < 8) {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c1[0]), rd, lane * 4, sbase + (2 * g) * 256, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c1[1]), rd, lane * 4, sbase + (2 * g + 1) * 256, 0);
                    }
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c1[0]), rd, lane * 4, sbase + g * 256, 0);
                }
                FENCE(); MFH(a, b, c1); FENCE(); MFH(a, b, c0); FENCE();
            }
        }
        // per "stage": everything but the last 16 stores retired, then the work-group barrier (as in the kernels)
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
    for (int i = 0; i < 4; ++i) s += keep[i].x;
    out[blockIdx.x * 512 + threadIdx.x] = s + lds[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Second probe: what, beside the MFMAs, costs time in the projection stage?  Same skeleton (48 MFMAs + 16 stores per
// wave and iteration, barrier per iteration), plus per group of 6 MFMAs:  LDS = 4 ds_read_b128 operand fragments
// (used by the MFMAs of the NEXT group),  VAL = the epilogue's VALU (mul, exp2, add, rcp, mul, 2 x fma_mix per output,
// 2 outputs),  B32 = 4 ds_read_b32 (bias re-initialisation).
template <bool LDS, bool VAL, bool B32, int WDMA = 0, bool STRIDE = false, bool ZT = false, bool LATE = false, int LATEG = 4>
__global__ __launch_bounds__(512, 1) void probe2(float* out, float* dst, int iters, unsigned dst_bytes, long long* cyc, const float* wsrc = nullptr,
                                                const float* zsrc = nullptr) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, dst_bytes, 0x00020000);
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 c0, c1, e0, e1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; e0[i] = 0.1f * i; e1[i] = 0.2f * i; }
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * lane + i); b[i] = (_Float16)(0.5f + 0.001f * i); }
    h8 f0 = a, f1 = a, f2 = a, f3 = a;
    const float cg = 0.37f, pm = 1.1f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned slot = (unsigned)((blockIdx.x * 977u + it) * 8u + wave);
        const int sbase = (int)((slot * 4096u) % (dst_bytes - 65536u)) & ~255;
        const unsigned char* stage = lds + (it & 1) * 32768;
        auto zt_issue = [&]() {
            const rsrc_t rzt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zsrc), 0, 1u << 30, 0x00020000);
            const int zb = (int)(((slot * 2048u) % ((1u << 30) - (1u << 20))) & ~2047u);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rzt, (__attribute__((address_space(3))) void*)(lds + 49152 + wave * 2048 + q * 1024), 16,
                                                         (lane >> 4) * 512 + (((lane & 15) ^ (lane >> 4)) << 4), zb + q * 1024, 0, 0);
        };
        if (ZT && LATE) { /* issued at the end of the stage, below */ }
        else if (ZT) {       // two pieces of the next row tile per stage: 4 rows x 256 B each, streaming (HBM misses), per-lane addresses
            const rsrc_t rzt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zsrc), 0, 1u << 30, 0x00020000);
            const int zb = (int)(((slot * 2048u) % ((1u << 30) - (1u << 20))) & ~2047u);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rzt, (__attribute__((address_space(3))) void*)(lds + 49152 + wave * 2048 + q * 1024), 16,
                                                         (lane >> 4) * 512 + (((lane & 15) ^ (lane >> 4)) << 4), zb + q * 1024, 0, 0);
        }
        if (WDMA) {     // the kernels' weight stage: 4 x 1 KiB LDS-DMA per wave into the other buffer, from an L2-resident image
            const rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wsrc), 0, 1u << 20, 0x00020000);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rwt, (__attribute__((address_space(3))) void*)(lds + ((it + 1) & 1) * 32768 + (4 * wave + q) * 1024), 16,
                                                         lane * 16, ((it & 15) * 32 + 4 * wave + q) * 1024, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            h8 n0 = f0, n1 = f1, n2 = f2, n3 = f3;
            if (LDS) {
                n0 = *reinterpret_cast<const h8*>(stage + (4 * g + 0) * 1024 + lane * 16);
                n1 = *reinterpret_cast<const h8*>(stage + (4 * g + 1) * 1024 + lane * 16);
                n2 = *reinterpret_cast<const h8*>(stage + (4 * g + 2) * 1024 + lane * 16);
                n3 = *reinterpret_cast<const h8*>(stage + (4 * g + 3) * 1024 + lane * 16);
            }
            float t0v = 0.f, t1v = 0.f, u0 = 0.f, u1 = 0.f;
            unsigned w0 = 0, w1 = 0;
            FENCE(); MFH(f1, b, c0); FENCE(); if (VAL) t0v = __builtin_amdgcn_exp2f(e1[2 * g] * cg);
            FENCE(); MFH(f0, a, c0); FENCE(); if (VAL) { t0v = __builtin_amdgcn_rcpf(1.0f + t0v); u0 = e0[2 * g] * pm; }
            FENCE(); MFH(f0, b, c0); FENCE();
            if (VAL) { asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(w0) : "v"(u0), "v"(t0v)); asm("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel_hi:[0,0,1]" : "+v"(w0) : "v"(u0), "v"(t0v)); }
            else w0 = __builtin_bit_cast(unsigned, e0[2 * g]);
            if (LATE) {
                if (g >= LATEG) {
#pragma unroll
                    for (int qq = 0; qq < 8 / (8 - LATEG); ++qq)
                        __builtin_amdgcn_raw_buffer_store_b32(w0 + qq, rd, lane * 4, sbase + ((16 / (8 - LATEG)) * (g - LATEG) + qq) * 256, 0);
                }
            } else if (STRIDE) __builtin_amdgcn_raw_buffer_store_b32(w0, rd, (lane >> 5) * (4 << 18) + (lane & 31) * 4, (int)((slot * 128u + (2 * g) * (8u << 18)) % (dst_bytes - (64u << 18))) & ~127, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(w0, rd, lane * 4, sbase + (2 * g) * 256, 0);
            if (B32) { e0[2 * g] = reinterpret_cast<const float*>(lds)[(2 * g) * 8 + (lane >> 5)]; e1[2 * g] = reinterpret_cast<const float*>(lds)[256 + (2 * g) * 8 + (lane >> 5)]; }
            FENCE(); MFH(f3, b, c1); FENCE(); if (VAL) t1v = __builtin_amdgcn_exp2f(e1[2 * g + 1] * cg);
            FENCE(); MFH(f2, a, c1); FENCE(); if (VAL) { t1v = __builtin_amdgcn_rcpf(1.0f + t1v); u1 = e0[2 * g + 1] * pm; }
            FENCE(); MFH(f2, b, c1); FENCE();
            if (VAL) { asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(w1) : "v"(u1), "v"(t1v)); asm("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel_hi:[0,0,1]" : "+v"(w1) : "v"(u1), "v"(t1v)); }
            else w1 = __builtin_bit_cast(unsigned, e0[2 * g + 1]);
            if (LATE) {
                if (g >= LATEG) {
#pragma unroll
                    for (int qq = 0; qq < 8 / (8 - LATEG); ++qq)
                        __builtin_amdgcn_raw_buffer_store_b32(w1 + qq, rd, lane * 4, sbase + ((16 / (8 - LATEG)) * (g - LATEG) + 8 / (8 - LATEG) + qq) * 256, 0);
                }
            } else if (STRIDE) __builtin_amdgcn_raw_buffer_store_b32(w1, rd, (lane >> 5) * (4 << 18) + (lane & 31) * 4, (int)((slot * 128u + (2 * g + 1) * (8u << 18)) % (dst_bytes - (64u << 18))) & ~127, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(w1, rd, lane * 4, sbase + (2 * g + 1) * 256, 0);
            if (B32) { e0[2 * g + 1] = reinterpret_cast<const float*>(lds)[(2 * g + 1) * 8 + (lane >> 5)]; e1[2 * g + 1] = reinterpret_cast<const float*>(lds)[256 + (2 * g + 1) * 8 + (lane >> 5)]; }
            FENCE();
            f0 = n0; f1 = n1; f2 = n2; f3 = n3;
        }
        if (ZT && LATE) {
            asm volatile("s_waitcnt vmcnt(18)" ::: "memory");      // previous stage's row-tile pieces + this stage's weights have landed
            zt_issue();
        } else {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + e0[i] + e1[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// Third probe: MFMA shape.  The full projection-like stage (stores + LDS fragments + epilogue VALU + bias reads + weight DMA)
// with every v_mfma_f32_32x32x16_f16 replaced by two v_mfma_f32_16x16x32_f16 (same FLOPs, same operand bytes).
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MF16(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
__global__ __launch_bounds__(512, 1) void probe3(float* out, float* dst, int iters, unsigned dst_bytes, long long* cyc, const float* wsrc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, dst_bytes, 0x00020000);
    const rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wsrc), 0, 1u << 20, 0x00020000);
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x4v c[8];
    f32x16 e0, e1;
    for (int i = 0; i < 8; ++i) c[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) { e0[i] = 0.1f * i; e1[i] = 0.2f * i; }
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * lane + i); b[i] = (_Float16)(0.5f + 0.001f * i); }
    h8 f0 = a, f1 = a, f2 = a, f3 = a;
    const float cg = 0.37f, pm = 1.1f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned slot = (unsigned)((blockIdx.x * 977u + it) * 8u + wave);
        const int sbase = (int)((slot * 4096u) % (dst_bytes - 65536u)) & ~255;
        const unsigned char* stage = lds + (it & 1) * 32768;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rwt, (__attribute__((address_space(3))) void*)(lds + ((it + 1) & 1) * 32768 + (4 * wave + q) * 1024), 16,
                                                     lane * 16, ((it & 15) * 32 + 4 * wave + q) * 1024, 0, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const h8 n0 = *reinterpret_cast<const h8*>(stage + (4 * g + 0) * 1024 + lane * 16);
            const h8 n1 = *reinterpret_cast<const h8*>(stage + (4 * g + 1) * 1024 + lane * 16);
            const h8 n2 = *reinterpret_cast<const h8*>(stage + (4 * g + 2) * 1024 + lane * 16);
            const h8 n3 = *reinterpret_cast<const h8*>(stage + (4 * g + 3) * 1024 + lane * 16);
            float t0v, t1v, u0, u1;
            unsigned w0, w1;
            FENCE(); MF16(f1, b, c[0]); MF16(f1, a, c[1]); FENCE(); t0v = __builtin_amdgcn_exp2f(e1[2 * g] * cg);
            FENCE(); MF16(f0, a, c[2]); MF16(f0, b, c[3]); FENCE(); t0v = __builtin_amdgcn_rcpf(1.0f + t0v); u0 = e0[2 * g] * pm;
            FENCE(); MF16(f0, b, c[0]); MF16(f0, a, c[1]); FENCE();
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(w0) : "v"(u0), "v"(t0v)); asm("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel_hi:[0,0,1]" : "+v"(w0) : "v"(u0), "v"(t0v));
            __builtin_amdgcn_raw_buffer_store_b32(w0, rd, lane * 4, sbase + (2 * g) * 256, 0);
            e0[2 * g] = reinterpret_cast<const float*>(lds)[(2 * g) * 8 + (lane >> 5)]; e1[2 * g] = reinterpret_cast<const float*>(lds)[256 + (2 * g) * 8 + (lane >> 5)];
            FENCE(); MF16(f3, b, c[4]); MF16(f3, a, c[5]); FENCE(); t1v = __builtin_amdgcn_exp2f(e1[2 * g + 1] * cg);
            FENCE(); MF16(f2, a, c[6]); MF16(f2, b, c[7]); FENCE(); t1v = __builtin_amdgcn_rcpf(1.0f + t1v); u1 = e0[2 * g + 1] * pm;
            FENCE(); MF16(f2, b, c[4]); MF16(f2, a, c[5]); FENCE();
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(w1) : "v"(u1), "v"(t1v)); asm("v_fma_mixhi_f16 %0, %1, %2, -%0 op_sel_hi:[0,0,1]" : "+v"(w1) : "v"(u1), "v"(t1v));
            __builtin_amdgcn_raw_buffer_store_b32(w1, rd, lane * 4, sbase + (2 * g + 1) * 256, 0);
            e0[2 * g + 1] = reinterpret_cast<const float*>(lds)[(2 * g + 1) * 8 + (lane >> 5)]; e1[2 * g + 1] = reinterpret_cast<const float*>(lds)[256 + (2 * g + 1) * 8 + (lane >> 5)];
            FENCE();
            f0 = n0; f1 = n1; f2 = n2; f3 = n3;
        }
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w;
    for (int i = 0; i < 16; ++i) s += e0[i] + e1[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

void run3(float* out, float* dst, unsigned db, long long* cyc, const float* wsrc) {
    const int iters = 1500;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe3<<<256, 512>>>(out, dst, 8, db, cyc, wsrc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe3<<<256, 512>>>(out, dst, iters, db, cyc, wsrc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[2048]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 2048; ++i) m += h[i];
    m /= 2048.0 * iters;
    printf("%-44s %.3f ms  %.0f cycles/iteration (ideal MFMA 3072)  %.2f us/iteration\n", "full stage with 16x16x32 MFMAs (2 per 32x32x16)", ms, m, ms * 1e3 / iters);
}

template <bool LDS, bool VAL, bool B32, int WDMA = 0, bool STRIDE = false, bool ZT = false, bool LATE = false, int LATEG = 4>
void run2(float* out, float* dst, unsigned db, long long* cyc, const char* name, const float* wsrc = nullptr) {
    const int iters = 1500;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe2<LDS, VAL, B32, WDMA, STRIDE, ZT, LATE, LATEG><<<256, 512>>>(out, dst, 8, db, cyc, wsrc, wsrc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe2<LDS, VAL, B32, WDMA, STRIDE, ZT, LATE, LATEG><<<256, 512>>>(out, dst, iters, db, cyc, wsrc, wsrc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[2048]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 2048; ++i) m += h[i];
    m /= 2048.0 * iters;
    printf("%-44s %.3f ms  %.0f cycles/iteration (ideal MFMA 3072)  %.2f us/iteration\n", name, ms, m, ms * 1e3 / iters);
}

template <int MODE, int NL>
void run(float* out, float* src, float* dst, unsigned sb, unsigned db, long long* cyc, const char* name) {
    const int iters = 1500;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE, NL><<<256, 512>>>(out, src, dst, 8, sb, db, cyc);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE, NL><<<256, 512>>>(out, src, dst, iters, sb, db, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
    m /= 256.0 * iters;
    printf("%-34s NL=%d  %.3f ms  %.0f cycles/iteration (ideal MFMA 3072)  stores %.2f TB/s  loads %.2f TB/s\n", name, NL, ms, m,
           256.0 * 8 * 4096 * iters / ms / 1e9, (MODE ? 256.0 * 8 * NL * 1024 * iters / ms / 1e9 : 0.0));
}

int main() {
    float *out, *src, *dst; long long* cyc;
    const unsigned sb = 1u << 30, db = 1u << 30;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&src, sb); hipMalloc(&dst, db); hipMalloc(&cyc, 2048 * 8);
    run2<false, false, false>(out, dst, db, cyc, "MFMA + stores");
    run2<true, false, false>(out, dst, db, cyc, "MFMA + stores + LDS fragments");
    run2<false, true, false>(out, dst, db, cyc, "MFMA + stores + epilogue VALU");
    run2<true, true, false>(out, dst, db, cyc, "MFMA + stores + LDS fragments + VALU");
    run2<true, true, true>(out, dst, db, cyc, "MFMA + stores + LDS fragments + VALU + b32");
    run2<true, true, true, 1>(out, dst, db, cyc, "  ... + weight-stage LDS-DMA (L2 hits)", src);
    run2<false, false, false, 1>(out, dst, db, cyc, "MFMA + stores + weight-stage LDS-DMA", src);
    run2<true, true, true, 1>(out, dst, db, cyc, "full stage with 32x32x16 MFMAs (again)", src);
    run3(out, dst, db, cyc, src);
    run2<true, true, true, 1, true, false>(out, dst, db, cyc, "full stage, channel-strided stores", src);
    run2<true, true, true, 1, false, true>(out, dst, db, cyc, "full stage, + 2 row-tile DMAs from HBM", src);
    run2<true, true, true, 1, true, true>(out, dst, db, cyc, "full stage, strided stores + row-tile DMAs", src);
    run2<true, true, true, 1, false, true, true>(out, dst, db, cyc, "row-tile DMAs after the last store, stores in 2nd half", src);
    run2<true, true, true, 1, false, false, true>(out, dst, db, cyc, "(stores in 2nd half, no row-tile DMAs)", src);
    run2<true, true, true, 1, false, true, true, 6>(out, dst, db, cyc, "row-tile DMAs after the last store, stores in last quarter", src);
    run2<true, true, true, 1, false, true, true, 4>(out, dst, db, cyc, "row-tile DMAs after the last store, stores in 2nd half", src);
    hipMemset(src, 0, sb);
    run<0, 2>(out, src, dst, sb, db, cyc, "no loads");
    run<1, 2>(out, src, dst, sb, db, cyc, "every wave: LDS-DMA at top");
    run<1, 8>(out, src, dst, sb, db, cyc, "every wave: LDS-DMA at top");
    run<2, 2>(out, src, dst, sb, db, cyc, "loader wave 7 (7 compute waves)");
    run<2, 8>(out, src, dst, sb, db, cyc, "loader wave 7 (7 compute waves)");
    run<3, 2>(out, src, dst, sb, db, cyc, "stores in first half of the stage");
    run<3, 8>(out, src, dst, sb, db, cyc, "stores in first half of the stage");
    run<4, 2>(out, src, dst, sb, db, cyc, "every wave: global loads to VGPRs");
    run<4, 8>(out, src, dst, sb, db, cyc, "every wave: global loads to VGPRs");
    return 0;
}
