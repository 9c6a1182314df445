// Probe of ds_read_b64_tr_b16 (gfx950): LDS holds u16 value = its element index; lane l reads at byte
// address base[l]; prints which elements every lane receives.  Used to lay out the TN image kernels.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const int* addr, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    uint32_t a = (uint32_t)(uintptr_t)(lds) + addr[threadIdx.x];
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = v.x & 0xffff;
    out[threadIdx.x * 4 + 1] = v.x >> 16;
    out[threadIdx.x * 4 + 2] = v.y & 0xffff;
    out[threadIdx.x * 4 + 3] = v.y >> 16;
}
int main() {
    int h_addr[64];
    int* d_addr;
    uint16_t *d_out, h_out[256];
    hipMalloc(&d_addr, sizeof(h_addr));
    hipMalloc(&d_out, sizeof(h_out));
    for (int mode = 0; mode < 2; ++mode) {
        // mode 0: lane l -> byte 8 l (linear).  mode 1: lane l -> row l / 4 (stride 64 B), 8-byte piece l % 4
        for (int l = 0; l < 64; ++l) h_addr[l] = mode == 0 ? 8 * l : (l / 4) * 64 + (l % 4) * 8;
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("mode %d (element index = byte / 2)\n", mode);
        for (int l = 0; l < 64; ++l)
            printf("lane %2d addr %4d: %4d %4d %4d %4d\n", l, h_addr[l], h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
    }
    return 0;
}
