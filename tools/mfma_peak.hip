// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate for 1 / 2 waves per SIMD, with and
// without the ds_read_b128 fragment traffic of the NT GEMM.  (perf debugging aid)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 0: MFMA only, 1: + ds_read_b128 fragments, 2: + LDS-only barrier per 64 MFMAs,
// 3: + 8 ds_write_b128 per thread per tile (double-buffered), 4: + 8 streaming global loads
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* __restrict__ src, size_t stride, long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 x 256 x 36
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * 256 * 36; i += 256) lds[i] = 0.001f * (i & 15);
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float4 a4[2] = {{1.f, 2.f, 3.f, 4.f}, {1.f, 2.f, 3.f, 4.f}}, b4[2] = {{1.f, 1.f, 1.f, 1.f}, {2.f, 2.f, 2.f, 2.f}};
    const long long c0 = clock64(), w0 = wall_clock64();
    const float* As0 = lds + (lane & 31) * 36 + (lane >> 5) * 4;
    float4 st[8];
    for (int i = 0; i < 8; ++i) st[i] = make_float4(1.f, 2.f, 3.f, 4.f);
    const float* g = src + (size_t)blockIdx.x * stride + (threadIdx.x >> 3) * 2048 + (threadIdx.x & 7) * 4;
    for (int it = 0; it < iters; ++it) {
        const float* As = As0 + (MODE >= 3 ? (it & 1) * 256 * 36 : 0);
        if (MODE >= 3) {
            float* W = lds + ((it + 1) & 1) * 256 * 36 + (threadIdx.x >> 3) * 36 + (threadIdx.x & 7) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(W + i * 32 * 36) = st[i];
        }
        if (MODE >= 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE >= 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float* gi = (MODE == 6 && i >= 4) ? g - (size_t)blockIdx.x * stride : g;  // shared B panel
                const int itx = MODE == 7 ? it + (int)blockIdx.x * 5 : it;  // mode 7: per-block K rotation
                st[i] = *reinterpret_cast<const float4*>(gi + (MODE >= 5 && !(MODE == 6 && i >= 4) ? (size_t)((it >> 6) % 4) * 512 * stride : 0) + (size_t)i * 32 * 2048 + (size_t)(itx & 63) * 32);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (MODE == 1) {
                a4[0] = *reinterpret_cast<const float4*>(As + kk * 8);
                a4[1] = *reinterpret_cast<const float4*>(As + 32 * 36 + kk * 8);
                b4[0] = *reinterpret_cast<const float4*>(As + 128 * 36 + kk * 8);
                b4[1] = *reinterpret_cast<const float4*>(As + 160 * 36 + kk * 8);
            }
#define Q(q)                                                                          \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].q, b4[0].q, acc[0], 0, 0, 0); \
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].q, b4[1].q, acc[1], 0, 0, 0); \
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1].q, b4[0].q, acc[2], 0, 0, 0); \
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1].q, b4[1].q, acc[3], 0, 0, 0);
            Q(x) Q(y) Q(z) Q(w)
#undef Q
        }
    }
    float s = st[0].x + st[7].w;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) {
        clk[0] = clock64() - c0;
        clk[1] = wall_clock64() - w0;
    }
}

template <int MODE>
void run(int blocks_per_cu, const char* name) {
    const int blocks = 256 * blocks_per_cu, iters = MODE >= 5 ? 256 : 2000;
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    static float* src = nullptr;
    const size_t stride = (size_t)256 * 2048;  // 256 rows x 2048 floats per block
    if (!src) {
        hipMalloc(&src, (size_t)4 * 512 * stride * 4);  // 4 GiB: mode 5 streams fresh panels from HBM
        hipMemset(src, 0, (size_t)4 * 512 * stride * 4);
    }
    const size_t ldsb = 2 * 256 * 36 * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    static long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    k<MODE><<<blocks, 256, ldsb>>>(out, 10, src, stride, clk);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256, ldsb>>>(out, iters, src, stride, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /*waves*/ * iters * 64 /*mfma*/ * (32.0 * 32 * 2 * 2);
    long long h[2];
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-28s %d block(s)/CU: %.1f us  %.1f TFLOP/s  shader clock %.0f MHz\n", name, blocks_per_cu, ms * 1e3,
           flops / ms / 1e9, 100.0 * (double)h[0] / (double)h[1]);
    hipFree(out);
}

int main() {
    run<0>(1, "mfma only");
    run<0>(2, "mfma only");
    run<1>(1, "mfma + ds_read_b128");
    run<1>(2, "mfma + ds_read_b128");
    run<2>(1, "+ barrier / 64 mfma");
    run<2>(2, "+ barrier / 64 mfma");
    run<3>(1, "+ 8 ds_write_b128");
    run<3>(2, "+ 8 ds_write_b128");
    run<4>(1, "+ 8 global_load_dwordx4");
    run<4>(2, "+ 8 global_load_dwordx4");
    run<5>(1, "  ... streaming from HBM");
    run<5>(2, "  ... streaming from HBM");
    run<6>(2, "  ... + shared B panel");
    run<7>(2, "  ... streaming, K rotated");
    return 0;
}
