// How many single-issue instructions fit into the shadow of one v_mfma_f32_32x32x16_bf16 when
// ONE wave per SIMD issues both (the slot structure of gemm_split.hip)?
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/ubench_slot.hip -o /tmp/us && /tmp/us
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
// NV = VALU per slot (kind: 0 = v_sub_f32 chain-free, 1 = the split mix), DSW: ds_write_b64 every 2nd slot,
// DSR: ds_read_b128 every 2nd slot
template <int NV, int KIND, int DSW, int DSR>
__global__ __launch_bounds__(256) void k(int iters, float* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(lane + i); b[i] = (short)(lane * 3 + i); }
    float x[16];
    uint32_t p[8];
    for (int i = 0; i < 16; ++i) x[i] = out[(threadIdx.x * 16 + i) & 1023] + i;
    for (int i = 0; i < 8; ++i) p[i] = i;
    char* dst = lds + threadIdx.x * 8;
    const char* src = lds + threadIdx.x * 16;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 rs;
    {
        const uint64_t pa = reinterpret_cast<uint64_t>(out);
        rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)pa);
        rs.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(pa >> 32) & 0xffffu));
        rs.z = __builtin_amdgcn_readfirstlane(-1);
        rs.w = __builtin_amdgcn_readfirstlane(0x00020000);
    }
    asm volatile("" ::: "a255");
    bf16x8 fr = a;
    bf16x8 fr2[8];
    for (int u = 0; u < 8; ++u) fr2[u] = a;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, (DSR == 1 ? fr : (DSR == 3 ? fr2[((m >> 1) + 4) & 7] : b)), acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int e = (m * NV + v) & 15;
                if (KIND == 0) x[e] -= 1.5f;
                else if (((m * NV + v) & 3) == 0) p[e & 7] = pack_bf16(x[e], x[(e + 1) & 15]);
                else if (((m * NV + v) & 3) == 1) x[e] = __uint_as_float(p[e & 7] << 16);
                else if (((m * NV + v) & 3) == 2) x[e] = __uint_as_float(p[e & 7] & 0xffff0000u);
                else x[e] -= x[(e + 5) & 15];
            }
            if (DSW && (m & 1)) *reinterpret_cast<uint2*>(dst + (m & 6) * 1024) = make_uint2(p[m & 7], p[(m + 1) & 7]);
            if (DSR == 1 && (m & 1)) fr = *reinterpret_cast<const bf16x8*>(src + (m & 6) * 2048);
            if (DSR == 2 && (m & 1)) fr2[(m >> 1) & 7] = *reinterpret_cast<const bf16x8*>(src + (m & 6) * 2048);  // not an MFMA operand
            if (DSR == 3 && (m & 1)) fr2[(m >> 1) & 7] = *reinterpret_cast<const bf16x8*>(src + (m & 6) * 2048);  // consumed 8 slots later
            if (DSR == 5) {  // four reads of unrelated accumulation registers per slot (raw staging storage)
                float t0_, t1_, t2_, t3_;
                asm volatile("v_accvgpr_read_b32 %0, a[200]\n\tv_accvgpr_read_b32 %1, a[201]\n\tv_accvgpr_read_b32 %2, a[202]\n\tv_accvgpr_read_b32 %3, a[203]"
                             : "=v"(t0_), "=v"(t1_), "=v"(t2_), "=v"(t3_));
                x[m & 15] += t0_ + t1_ + t2_ + t3_;
            }
            if (DSR == 6 && (m & 1)) {  // one 16-byte LDS store straight from accumulation registers every 2nd slot
                asm volatile("ds_write_b128 %0, a[204:207]" : : "v"((uint32_t)(threadIdx.x * 16)) : "memory");
            }
            if (DSR == 7 && (m & 3) == 0) {  // one buffer load into accumulation registers every 4th slot
                asm volatile("buffer_load_dwordx4 a[208:211], %0, %1, 0 offen" : : "v"((uint32_t)(threadIdx.x * 16)), "s"(rs) : "memory");
            }
            if (DSR == 4) { if ((m & 7) == 7) { for (int u = 0; u < 4; ++u) fr2[u] = *reinterpret_cast<const bf16x8*>(src + u * 2048); } }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += p[i];
    for (int u = 0; u < 8; ++u) s += fr2[u][0];
    out[blockIdx.x * 256 + threadIdx.x] = s + fr[0];
    if (lane == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int NV, int KIND, int DSW, int DSR>
void run(float* out, long long* cyc) {
    const int iters = 2000;
    k<NV, KIND, DSW, DSR><<<256, 256>>>(10, out, cyc);
    k<NV, KIND, DSW, DSR><<<256, 256>>>(iters, out, cyc);
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("NV=%d kind=%s dsw=%d dsr=%d : %.1f clk per slot (MFMA alone = 32)\n", NV, KIND ? "split-mix" : "v_sub", DSW, DSR,
           (double)h[0] / iters / 16);
}
int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 64);
    hipMemset(out, 0, 256 * 256 * 4);
    run<0, 0, 0, 0>(out, cyc);
    run<2, 0, 0, 0>(out, cyc);
    run<4, 0, 0, 0>(out, cyc);
    run<5, 0, 0, 0>(out, cyc);
    run<6, 0, 0, 0>(out, cyc);
    run<8, 0, 0, 0>(out, cyc);
    run<2, 1, 0, 0>(out, cyc);
    run<4, 1, 0, 0>(out, cyc);
    run<5, 1, 0, 0>(out, cyc);
    run<4, 1, 1, 0>(out, cyc);
    run<4, 1, 0, 1>(out, cyc);
    run<4, 1, 1, 1>(out, cyc);
    run<2, 1, 1, 1>(out, cyc);
    run<0, 0, 0, 1>(out, cyc);
    run<0, 0, 0, 2>(out, cyc);
    run<0, 0, 0, 3>(out, cyc);
    run<0, 0, 0, 4>(out, cyc);
    run<4, 1, 1, 2>(out, cyc);
    run<4, 1, 1, 3>(out, cyc);
    run<0, 0, 0, 5>(out, cyc);
    run<4, 1, 0, 5>(out, cyc);
    run<0, 0, 0, 6>(out, cyc);
    run<0, 0, 0, 7>(out, cyc);
    return 0;
}
