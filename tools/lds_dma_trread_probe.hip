// Round 6 micro-test (hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S): after an LDS-DMA builtin, hipcc puts
// s_waitcnt vmcnt(0) in front of ds_read_b64_tr_b16 (k_tr) but not in front of ds_read_b128 (k_plain); an asm-issued DMA
// (k_tr_asm) is invisible to that pass.  See DESIGN.md 7.2.
#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr;

__global__ void k_plain(const char* g, float* out) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g), 0, 0x7fffffff, 0x00020000);
    int lane = threadIdx.x & 63;
    f32x16 acc = {};
    for (int s = 0; s < 64; ++s) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(sm + (s & 1) * 1024), 16, lane * 16, s * 1024, 0, 0);
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        bf16x8 a = *reinterpret_cast<const bf16x8*>(sm + ((s + 1) & 1) * 1024 + lane * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = acc[i];
}
__global__ void k_tr(const char* g, float* out) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g), 0, 0x7fffffff, 0x00020000);
    int lane = threadIdx.x & 63;
    f32x16 acc = {};
    for (int s = 0; s < 64; ++s) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(sm + (s & 1) * 1024), 16, lane * 16, s * 1024, 0, 0);
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* p = sm + ((s + 1) & 1) * 1024 + lane * 8;
        bf16x8 a = __builtin_shufflevector(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(p)),
                                           __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(p + 512)), 0, 1, 2, 3, 4, 5, 6, 7);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = acc[i];
}
// LDS-DMA by inline asm: the waitcnt pass does not know about it
__device__ __forceinline__ void dma16(i32x4 r, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(r), "s"(soff) : "memory", "m0");
}
__global__ void k_tr_asm(const char* g, float* out) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    unsigned long long ga = (unsigned long long)g;
    i32x4 r = {(int)__builtin_amdgcn_readfirstlane((unsigned)ga), (int)__builtin_amdgcn_readfirstlane((unsigned)(ga >> 32) & 0xffff), 0x7fffffff, 0x00020000};
    int lane = threadIdx.x & 63;
    f32x16 acc = {};
    const unsigned smb = (unsigned)(unsigned long long)(lds_ptr)sm;
    for (int s = 0; s < 64; ++s) {
        dma16(r, smb + (s & 1) * 1024, lane * 16, s * 1024);
        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char* p = sm + ((s + 1) & 1) * 1024 + lane * 8;
        bf16x8 a = __builtin_shufflevector(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(p)),
                                           __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(p + 512)), 0, 1, 2, 3, 4, 5, 6, 7);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = acc[i];
}
