// What an episode-persistent kernel would pay per dependent seam on this chip ("measure - not
// estimate"): the SAME chain of dependent phases (every workgroup reads a slice another workgroup wrote in the
// previous phase - an all-to-all seam like CNN -> LSTM -> panels -> sampling of one step) run
//   (a) as one kernel launch per phase on one stream (what the library does: kernel boundaries), and
//   (b) as ONE persistent launch, 256 workgroups (one per CU), with an in-kernel grid barrier per phase
//       (one monotonic counter, agent-scope release before the arrive, relaxed polling with s_sleep, one acquire
//       fence after - the cheapest correct form of MI355X_MICROARCH.md's price list).
// Prints microseconds per phase for both, at three amounts of work per phase.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gbp tools/grid_barrier_probe.hip && /tmp/gbp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int kSlice = 4096;  // floats per workgroup and phase (16 KB)

__device__ __forceinline__ void phase_body(const float* __restrict__ in, float* __restrict__ out, int phase, int spin) {
    const int G = gridDim.x, b = blockIdx.x;
    const int src = (b * 37 + phase * 101 + 1) % G;  // another workgroup's slice of the previous phase
    const float4* s = reinterpret_cast<const float4*>(in + (size_t)src * kSlice);
    float4* d = reinterpret_cast<float4*>(out + (size_t)b * kSlice);
    for (int i = threadIdx.x; i < kSlice / 4; i += blockDim.x) {
        float4 v = s[i];
        for (int k = 0; k < spin; ++k) {  // dependent FMA chain = the "work" of a latency-bound kernel
            v.x = v.x * 1.0000001f + 1e-7f;
            v.y = v.y * 1.0000001f + 1e-7f;
            v.z = v.z * 1.0000001f + 1e-7f;
            v.w = v.w * 1.0000001f + 1e-7f;
        }
        d[i] = v;
    }
}

__global__ __launch_bounds__(256) void one_phase(const float* in, float* out, int phase, int spin) { phase_body(in, out, phase, spin); }

__global__ __launch_bounds__(256) void persistent(float* a, float* b, int phases, int spin, unsigned* cnt, int* gave_up) {
    for (int p = 0; p < phases; ++p) {
        phase_body((p & 1) ? b : a, (p & 1) ? a : b, p, spin);
        // ---- grid barrier
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(p + 1) * gridDim.x;
            int tries = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++tries > (1 << 22)) { *gave_up = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
}

int main() {
    const int G = 256, phases = 64, reps = 20;
    float *a, *b;
    unsigned* cnt;
    int* gave_up;
    CHECK(hipMalloc(&a, (size_t)G * kSlice * 4));
    CHECK(hipMalloc(&b, (size_t)G * kSlice * 4));
    CHECK(hipMalloc(&cnt, 4));
    CHECK(hipMalloc(&gave_up, 4));
    CHECK(hipMemset(a, 0, (size_t)G * kSlice * 4));
    CHECK(hipMemset(gave_up, 0, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int spins[3] = {0, 400, 2000};
    for (int si = 0; si < 3; ++si) {
        const int spin = spins[si];
        float ms_l = 0, ms_p = 0;
        for (int w = 0; w < 2; ++w) {  // (first pass = warm-up)
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r)
                for (int p = 0; p < phases; ++p)
                    hipLaunchKernelGGL(one_phase, dim3(G), dim3(256), 0, 0, (p & 1) ? b : a, (p & 1) ? a : b, p, spin);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms_l, e0, e1));
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) {
                CHECK(hipMemsetAsync(cnt, 0, 4));
                hipLaunchKernelGGL(persistent, dim3(G), dim3(256), 0, 0, a, b, phases, spin, cnt, gave_up);
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms_p, e0, e1));
        }
        int gu = 0;
        CHECK(hipMemcpy(&gu, gave_up, 4, hipMemcpyDeviceToHost));
        printf("{\"work_fma_chain\": %d, \"us_per_phase_launches\": %.2f, \"us_per_phase_persistent_grid_barrier\": %.2f, \"barrier_gave_up\": %d}\n",
               spin, ms_l * 1e3 / (reps * phases), ms_p * 1e3 / (reps * phases), gu);
    }
    return 0;
}
