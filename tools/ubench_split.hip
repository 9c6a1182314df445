// Microbenchmarks behind the bf16x6 kernels' design: rate of v_mfma_f32_32x32x16_bf16, cost of the
// split arithmetic, and how far a second wave's VALU / LDS work overlaps with a matrix wave on the
// same SIMD.   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench_split.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16(float x, float y) {
    const f32x2_t v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void split_pair(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pack_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = pack_bf16(rx, ry);
    p2 = pack_bf16(rx - __uint_as_float(p1 << 16), ry - __uint_as_float(p1 & 0xffff0000u));
}
// truncating variant: a0 RN, a1 / a2 by masking (perm packs the high halves)
__device__ __forceinline__ void split_pair_t(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pack_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = __builtin_amdgcn_perm(__float_as_uint(ry), __float_as_uint(rx), 0x07060302u);
    const float sx = rx - __uint_as_float(__float_as_uint(rx) & 0xffff0000u);
    const float sy = ry - __uint_as_float(__float_as_uint(ry) & 0xffff0000u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(sy), __float_as_uint(sx), 0x07060302u);
}

// mode bit 0: waves 0-3 run MFMAs; bit 1: waves 4-7 run the split arithmetic; bit 2: waves 4-7 also
// write the result to LDS (ds_write_b64); bit 3: split variant t
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long t0 = clock64();
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (short)(lane + i); b[i] = (short)(lane * 3 + i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (lane == 0) cyc[blockIdx.x * 8 + wave] = clock64() - t0;
    } else {
        if (!(mode & 2)) return;
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = out[(threadIdx.x * 16 + i) & 1023] + i;
        uint32_t accu = 0;
        char* dst = lds + (threadIdx.x - 256) * 8;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {  // one float4 chunk -> 3 x 8 bytes
                uint32_t a0, a1, a2, b0, b1, b2;
                if (mode & 8) {
                    split_pair_t(v[4 * c], v[4 * c + 1], a0, a1, a2);
                    split_pair_t(v[4 * c + 2], v[4 * c + 3], b0, b1, b2);
                } else {
                    split_pair(v[4 * c], v[4 * c + 1], a0, a1, a2);
                    split_pair(v[4 * c + 2], v[4 * c + 3], b0, b1, b2);
                }
                if (mode & 4) {
                    *reinterpret_cast<uint2*>(dst + c * 2048) = make_uint2(a0, b0);
                    *reinterpret_cast<uint2*>(dst + c * 2048 + 10240) = make_uint2(a1, b1);
                    *reinterpret_cast<uint2*>(dst + c * 2048 + 20480) = make_uint2(a2, b2);
                } else {
                    accu += a0 ^ a1 ^ a2 ^ b0 ^ b1 ^ b2;
                }
                v[4 * c] += 1.0f;  // keep the loop from being hoisted
                v[4 * c + 2] += 1.0f;
            }
        }
        __syncthreads_count(0) ;
        out[blockIdx.x * 512 + threadIdx.x] = accu + lds[threadIdx.x];
        if (lane == 0) cyc[blockIdx.x * 8 + wave] = clock64() - t0;
    }
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 512 * 512 * 4);
    hipMalloc(&cyc, 512 * 8 * 8);
    hipMemset(out, 0, 512 * 512 * 4);
    const int iters = 2000;
    const int modes[] = {1, 2, 6, 10, 14, 3, 7, 11, 15};
    for (int m : modes) {
        hipMemset(cyc, 0, 512 * 8 * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k<<<256, 512>>>(m, 10, out, cyc);
        hipEventRecord(e0);
        k<<<256, 512>>>(m, iters, out, cyc);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[8];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %2d (%s%s%s%s): %.3f ms | per iter: %.1f ns | mfma wave %lld clk/iter (24 mfma), split wave %lld clk/iter (16 floats)\n",
               m, (m & 1) ? "mfma " : "", (m & 2) ? "split " : "", (m & 4) ? "+lds " : "", (m & 8) ? "trunc" : "",
               ms, ms * 1e6 / iters, h[0] / iters, h[4] / iters);
    }
    return 0;
}
