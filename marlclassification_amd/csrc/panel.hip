// Row-panel kernels for the small per-step networks (message decoder / encoder, policy
// hidden layer; reference networks/message.py:20-49, networks/policy.py:12-14).
//
// At R = Na*Nb rows and 64..384 features these layers are a few dozen MFLOP each: far too
// small for a tiled GEMM launch + a LayerNorm launch per layer (each launch is latency
// bound at ~10-20 us).  Here ONE workgroup owns a 32-row panel end to end:
//   stage the input rows in LDS (optionally computing the message mean over the other
//   agents on the fly) -> [ X * W^T on the matrix cores, one 32x32 tile per wave, weights
//   streamed straight from L2 -> bias -> LayerNorm + SiLU in the MFMA accumulator layout
//   (cross-lane + cross-wave row reductions) -> next layer's input panel in LDS ] x {1,2}.
// The backward kernel walks the same chain in reverse (LayerNorm/SiLU backward in the tile
// layout, dX GEMMs against the transposed weight copies) and emits per-panel partial sums
// for the LayerNorm affine gradients (fixed order -> deterministic).
#include <stdlib.h>

#include "common.h"
#include "sample.h"

namespace marl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that only waits for LDS traffic: the global stores of saved activations
// issued before it are never read back by this kernel, so there is no reason to drain vmcnt
// (a plain __syncthreads() would wait for every outstanding store: ~1-2 us each time).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ float silu_p(float y) { return silu_fast(y); }
__device__ __forceinline__ float silu_grad_p(float y) { return silu_grad_fast(y); }

// 16-row panels: R = Na*Nb = 4096 rows give 256 workgroups (one per CU); the f32 matrix-core
// time of the widest layer (256 -> 384) is ~5 us per panel, half of what a 32-row panel on
// half the CUs would need.
constexpr int kPanelRows = 16;
constexpr int kPanelMaxWaves = 12;  // 768 threads: 170 VGPRs per lane, no spills
constexpr int kPanelMaxCols = 6;    // LayerNorm columns per lane: widths up to 384

__host__ __device__ inline int panel_stride(int k) { return ((k + 15) & ~15) + 4; }

// One wave = one (32-column tile j, K slice s) pair, computed as two 16x16x4 MFMA sub-tiles:
// acc[t] = in[16 x Kslice] (LDS) * W[16 tile rows, Kslice]^T.  Weight fragments come straight
// from global memory (L2), 8 x 16-byte loads (a 64-deep chunk of both sub-tiles) in flight
// per wave; splitting K over waves puts ALL of a layer's weight loads in flight at once.
// C/D layout of the 16x16 tile: col = lane & 15, row = 4 * (lane >> 4) + reg.
__device__ __forceinline__ void panel_gemm(f32x4 (&acc)[2], const float* in, int stride, int K,
                                           const float* __restrict__ w, int ldw, int n, int j,
                                           int kbeg, int kend, int lane) {
    const int K4 = (K + 3) & ~3;
    const int quad = lane >> 4, l16 = lane & 15;
    const float* arow = in + l16 * stride + 4 * quad;
    int r0 = j * 32 + l16, r1 = r0 + 16;
    r0 = r0 < n ? r0 : n - 1;
    r1 = r1 < n ? r1 : n - 1;
    const float* w0 = w + (size_t)r0 * ldw + 4 * quad;
    const float* w1 = w + (size_t)r1 * ldw + 4 * quad;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[0][r] = acc[1][r] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        float4 b0[4], b1[4];
        // columns [K, round16(K)) of the LDS panel are zero: a clamped (finite) read is exact
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = k0 + 16 * i;
            const int off = kk + 4 * quad < K4 ? kk : -4 * quad;
            b0[i] = *reinterpret_cast<const float4*>(w0 + off);
            b1[i] = *reinterpret_cast<const float4*>(w1 + off);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (k0 + 16 * i < kend) {
                const float4 a = *reinterpret_cast<const float4*>(arow + k0 + 16 * i);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0[i].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1[i].x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0[i].y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1[i].y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0[i].z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1[i].z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0[i].w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1[i].w, acc[1], 0, 0, 0);
            }
        }
    }
}

// K-slice partial tiles -> slice 0's accumulator (fixed order).  part: [(ks-1) * nt][64][8]
__device__ __forceinline__ void panel_ksum(f32x4 (&acc)[2], float* part, int nt, int ks, int j, int s,
                                           int lane, bool active) {
    if (ks > 1) {
        if (active && s > 0) {
            float* p = part + ((size_t)((s - 1) * nt + j) * 64 + lane) * 8;
            *reinterpret_cast<float4*>(p) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
        }
        lds_barrier();
        if (active && s == 0) {
            for (int q = 1; q < ks; ++q) {
                const float* p = part + ((size_t)((q - 1) * nt + j) * 64 + lane) * 8;
                const float4 u = *reinterpret_cast<const float4*>(p);
                const float4 v = *reinterpret_cast<const float4*>(p + 4);
                acc[0][0] += u.x;
                acc[0][1] += u.y;
                acc[0][2] += u.z;
                acc[0][3] += u.w;
                acc[1][0] += v.x;
                acc[1][1] += v.y;
                acc[1][2] += v.z;
                acc[1][3] += v.w;
            }
        }
    }
}

__host__ __device__ inline int panel_ksplit(int K, int nt, int nwaves) {
    const int K16 = (K + 15) & ~15;
    int ks = nwaves / nt;
    const int want = (K16 + 63) / 64;
    ks = ks < want ? ks : want;
    return ks < 1 ? 1 : ks;
}

// LayerNorm + SiLU row pass of the forward panel kernel: wave w owns rows w, w + nwaves, ...; a row lives
// in registers (NC columns per lane: widths up to 64 * NC), two-pass statistics with VALU wave reductions,
// result written over the LDS panel (next layer's input) and to global.  NC = 2 for the message chain's
// widths (<= 128) is a third of the six-slot form's instructions.
template <int NC>
__device__ __forceinline__ void panel_ln_rows(const PanelLayer& Lr, float* outp, int ys, int n, const float* lgamma,
                                              const float* lbeta, const int* rowmap, int wave, int nwaves, int lane)
{
    constexpr int RPW = 4;  // launcher guarantees >= 4 waves for the 16 rows
    float g[NC], bt[NC];
#pragma unroll
    for (int u = 0; u < NC; ++u) {
        const int c = lane + 64 * u;
        g[u] = c < n ? lgamma[c] : 0.f;
        bt[u] = c < n ? lbeta[c] : 0.f;
    }
    const float inv_n = 1.0f / (float)n;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int lr = wave + i * nwaves;
        if (lr < kPanelRows) {  // wave-uniform
            float* zr = outp + lr * ys;
            const int row = rowmap[lr];
            float v[NC];
            float sm = 0.f;
#pragma unroll
            for (int u = 0; u < NC; ++u) {
                const int c = lane + 64 * u;
                v[u] = c < n ? zr[c] : 0.f;
                sm += v[u];
                if (Lr.z && row >= 0 && c < n) Lr.z[(size_t)row * Lr.ldz + c] = v[u];
            }
            const float mean = wave_sum(sm) * inv_n;
            float q = 0.f;
#pragma unroll
            for (int u = 0; u < NC; ++u) {
                const int c = lane + 64 * u;
                const float d = c < n ? v[u] - mean : 0.f;
                v[u] = d;
                q += d * d;
            }
            const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_n + 1e-5f);
            float* arow = Lr.a + (size_t)(row < 0 ? 0 : row) * Lr.lda;
#pragma unroll
            for (int u = 0; u < NC; ++u) {
                const int c = lane + 64 * u;
                if (c < n) {
                    const float av = silu_p(v[u] * rstd * g[u] + bt[u]);
                    zr[c] = av;
                    if (row >= 0) arow[c] = av;
                }
            }
            if (Lr.stats && lane == 0 && row >= 0) {
                Lr.stats[(size_t)row * 2] = mean;
                Lr.stats[(size_t)row * 2 + 1] = rstd;
            }
            if (Lr.a3 && row >= 0) {
                // the row's activations are in the LDS panel (written by this wave just above): four
                // consecutive columns per lane -> 8-byte pieces of the three image planes
                for (int c4 = 4 * lane; c4 < n; c4 += 256) {
                    const float4 t = *reinterpret_cast<const float4*>(zr + c4);
                    const int col = Lr.a3_col0 + c4;
                    img_store4(Lr.a3 + img_off((int64_t)Lr.a3_row0 + row, col >> 4, Lr.a3_steps), col, t.x, t.y, t.z, t.w);
                }
            }
        }
    }
}

// ===========================================================================
// forward
// ===========================================================================
// SAMPLE: 0 = panels only; 4 / MARL_MAX_ACTIONS = sampling workgroups ride along behind the panel ones, with
// the action loops of sample.h bounded at that many actions
template <int SAMPLE>
__global__ __launch_bounds__(768, 6) void panel_fwd_kernel(const PanelFwdBatch B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (SAMPLE > 0 && (int)blockIdx.x >= B.panel_blocks) {
        // ride-along sampling workgroup: one wave per row of the policy activations
        const int r = ((int)blockIdx.x - B.panel_blocks) * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
        if (r >= B.sample.R) return;
        constexpr int MAXA = SAMPLE > 0 ? SAMPLE : 4;
        float p[MAXA];
        SamplePre<MAXA> S;
        sample_prefetch<MAXA, false>(B.sample, r, threadIdx.x & 63, S);
        sample_row_logits<MAXA>(B.sample, r, p, threadIdx.x & 63);
        sample_finish<MAXA>(B.sample, r, p, threadIdx.x & 63, S);
        return;
    }
    const PanelFwdProb& P = B.p[blockIdx.y];
    const int m0 = blockIdx.x * kPanelRows;
    const int bpb = P.by_batch;  // > 0: this workgroup owns all agents of bpb batch elements
    if (bpb ? (int)blockIdx.x * bpb >= P.g_nb : m0 >= P.m) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    // global row of local row lr, or -1 (padding): a small table in LDS
    int* rowmap = reinterpret_cast<int*>(lds + B.off_red);
    if (tid < kPanelRows) {
        int r = m0 + tid < P.m ? m0 + tid : -1;
        if (bpb) {
            const int a = tid / bpb, b = (int)blockIdx.x * bpb + (tid - a * bpb);
            r = (a < P.g_na && b < P.g_nb) ? a * P.g_nb + b : -1;
        }
        rowmap[tid] = r;
    }
    lds_barrier();
    const int quad = lane >> 4, l16 = lane & 15;
#ifdef MARL_KERNEL_TS
    MARL_TS_DECL(B.ts);
#endif
    MARL_TS();
    float* X = lds;
    float* Y = lds + B.off_panel1;
    float* part = lds + B.off_part;
    float* prm = lds + B.off_prm;  // [layer][bias | gamma | beta][n] staged once
    // The per-channel vectors of EVERY layer and the input rows are requested before anything is
    // waited for (one round trip to L2 at the head of the kernel instead of one per layer plus
    // one per staging pass); the LDS stores follow the staging below.  n <= 32 * waves <= threads.
    float pv[kPanelMaxLayers][3];
#pragma unroll
    for (int l = 0; l < kPanelMaxLayers; ++l) {
        pv[l][0] = pv[l][1] = pv[l][2] = 0.f;
        if (l < P.nlayers && tid < P.layer[l].n) {
            pv[l][0] = P.layer[l].bias[tid];
            pv[l][1] = P.layer[l].gamma[tid];
            pv[l][2] = P.layer[l].beta[tid];
        }
    }

    // ---- stage the 32-row input panel (zero-filled past k0 and past M)
    {
        const int k0 = P.k0, xs = panel_stride(k0), K16 = (k0 + 15) & ~15;
        const int c4 = K16 >> 2;
        const int K4 = (k0 + 3) & ~3;
        if (P.agg_na > 0) {
            // message mean over the OTHER agents (networks/message.py:5-17), 4 loads in flight
            const int na = P.agg_na, nb = P.agg_nb;
            const float den = (float)(na - 1);
            for (int e = tid; e < kPanelRows * c4; e += blockDim.x) {
                const int lr = e / c4, k = (e % c4) * 4;
                const int r = rowmap[lr];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r >= 0 && k < K4 && na > 1) {
                    const int b = r % nb;
                    const float* base = P.x + (size_t)b * P.ldx + k;
                    const size_t astr = (size_t)nb * P.ldx;
                    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int a0 = 0; a0 < na; a0 += 8) {  // 8 independent loads in flight
                        float4 qv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int a = a0 + u < na ? a0 + u : na - 1;
                            qv[u] = *reinterpret_cast<const float4*>(base + (size_t)a * astr);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (a0 + u < na) {  // sequential over agents, as the reference's sum(dim=0)
                                s.x += qv[u].x;
                                s.y += qv[u].y;
                                s.z += qv[u].z;
                                s.w += qv[u].w;
                            }
                        }
                    }
                    const float4 me = *reinterpret_cast<const float4*>(P.x + (size_t)r * P.ldx + k);
                    v = make_float4((s.x - me.x) / den, (s.y - me.y) / den, (s.z - me.z) / den,
                                    (s.w - me.w) / den);
                    // pad columns [k0, K4) of the message rows are zero, so v is zero there too
                    if (P.xbar) *reinterpret_cast<float4*>(P.xbar + (size_t)r * P.ldx + k) = v;
                }
                *reinterpret_cast<float4*>(X + lr * xs + k) = v;
            }
        } else {
            float4 xv[2];  // the first two passes' loads fly together
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = tid + it * (int)blockDim.x;
                xv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < kPanelRows * c4) {
                    const int lr = e / c4, k = (e % c4) * 4;
                    const int r = rowmap[lr];
                    if (r >= 0 && k < K4) xv[it] = *reinterpret_cast<const float4*>(P.x + (size_t)r * P.ldx + k);
                }
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = tid + it * (int)blockDim.x;
                if (e < kPanelRows * c4) {
                    const int lr = e / c4, k = (e % c4) * 4;
                    *reinterpret_cast<float4*>(X + lr * xs + k) = xv[it];
                }
            }
            for (int e = tid + 2 * (int)blockDim.x; e < kPanelRows * c4; e += blockDim.x) {
                const int lr = e / c4, k = (e % c4) * 4;
                const int r = rowmap[lr];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r >= 0 && k < K4) v = *reinterpret_cast<const float4*>(P.x + (size_t)r * P.ldx + k);
                *reinterpret_cast<float4*>(X + lr * xs + k) = v;
            }
        }
    }
    {
        int off = 0;
#pragma unroll
        for (int l = 0; l < kPanelMaxLayers; ++l) {
            if (l < P.nlayers) {
                const int n = P.layer[l].n;
                if (tid < n) {
                    prm[off + tid] = pv[l][0];
                    prm[off + n + tid] = pv[l][1];
                    prm[off + 2 * n + tid] = pv[l][2];
                }
                off += 3 * n;
            }
        }
    }
    MARL_TS();
    lds_barrier();
    MARL_TS();

    int prm_off = 0, K = P.k0;
    for (int l = 0; l < P.nlayers; ++l) {
        const PanelLayer Lr = P.layer[l];  // by value: no scalar re-loads inside the loops
        const float* lbias = prm + prm_off;
        const float* lgamma = lbias + Lr.n;
        const float* lbeta = lgamma + Lr.n;
        prm_off += 3 * Lr.n;
        float* in = (l & 1) ? Y : X;
        float* outp = (l & 1) ? X : Y;  // this layer's output panel (next layer's input)
        const int K16 = (K + 15) & ~15;
        const int stride = panel_stride(K);
        if (l > 0 && l == P.agg_at) {
            // the panel holds the messages of every agent of this workgroup's batch elements:
            // message mean over the OTHER agents in place (sequential over agents, as the
            // reference's sum(dim=0)), kept in xbar for the weight gradients
            const int c4 = K16 >> 2, K4 = (K + 3) & ~3, na = P.g_na;
            const float den = (float)(na - 1);
            float4 v[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {  // launcher: kPanelRows * c4 <= 2 * blockDim.x
                const int e = tid + it * (int)blockDim.x;
                v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < kPanelRows * c4) {
                    const int lr = e / c4, k = (e - lr * c4) * 4;
                    const int a = lr / bpb, i = lr - a * bpb;
                    if (a < na && na > 1) {
                        float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (int a2 = 0; a2 < na; ++a2) {
                            const float4 q = *reinterpret_cast<const float4*>(in + (a2 * bpb + i) * stride + k);
                            sm.x += q.x;
                            sm.y += q.y;
                            sm.z += q.z;
                            sm.w += q.w;
                        }
                        const float4 me = *reinterpret_cast<const float4*>(in + lr * stride + k);
                        v[it] = make_float4((sm.x - me.x) / den, (sm.y - me.y) / den, (sm.z - me.z) / den,
                                            (sm.w - me.w) / den);
                    }
                }
            }
            lds_barrier();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = tid + it * (int)blockDim.x;
                if (e < kPanelRows * c4) {
                    const int lr = e / c4, k = (e - lr * c4) * 4;
                    *reinterpret_cast<float4*>(in + lr * stride + k) = v[it];
                    const int r = rowmap[lr];
                    if (r >= 0 && k < K4 && P.xbar)
                        *reinterpret_cast<float4*>(P.xbar + (size_t)r * P.ld_xbar + k) = v[it];
                }
            }
            lds_barrier();
        }
        const int n = Lr.n, nt = (n + 31) >> 5;
        const int ks = panel_ksplit(K, nt, nwaves);
        const int j = wave % nt, s = wave / nt;
        const bool active = s < ks;
        const int kper = (((K16 + ks - 1) / ks) + 15) & ~15;
        const int kbeg = s * kper;
        const int kend = kbeg + kper < K16 ? kbeg + kper : K16;
        f32x4 acc[2];
        if (active) panel_gemm(acc, in, stride, K, Lr.w, Lr.ldw, n, j, kbeg, kend, lane);
        MARL_TS();
        panel_ksum(acc, part, nt, ks, j, s, lane, active);
        MARL_TS();

        // z = acc + bias -> output panel in LDS (its copy kept for backward is written row-wise,
        // coalesced, by the LayerNorm pass below instead of as 64-byte pieces from this layout)
        const int ys = panel_stride(n), n16 = (n + 15) & ~15;
        if (active && s == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int col = j * 32 + 16 * t + l16;
                const bool cv = col < n;
                const float bv = cv ? lbias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = 4 * quad + r;
                    const float zv = cv ? acc[t][r] + bv : 0.f;
                    if (col < n16) outp[lr * ys + col] = zv;
                }
            }
        }
        MARL_TS();
        lds_barrier();
        MARL_TS();
        // LayerNorm + SiLU row pass (panel_ln_rows above), unrolled for this layer's width
        if (n <= B.ln_narrow_max)
            panel_ln_rows<2>(Lr, outp, ys, n, lgamma, lbeta, rowmap, wave, nwaves, lane);
        else
            panel_ln_rows<kPanelMaxCols>(Lr, outp, ys, n, lgamma, lbeta, rowmap, wave, nwaves, lane);
        MARL_TS();
        lds_barrier();
        MARL_TS();
        K = n;
    }
}

constexpr size_t kPanelMaxLds = 144 * 1024;

static int panel_waves_for(int k, int n) {
    const int nt = (n + 31) / 32;
    const int want = (((k + 15) & ~15) + 63) / 64;
    int w = nt * want;
    if (w < nt) w = nt;
    return w > kPanelMaxWaves ? (nt > kPanelMaxWaves ? -1 : (kPanelMaxWaves / nt) * nt) : w;
}

int panel_chain_supported(int na, int n_msg, int threads) {
    const int K16 = (n_msg + 15) & ~15;  // (two passes of the workgroup cover the message panel)
    return na >= 1 && na <= kPanelRows && kPanelRows * (K16 / 4) <= 2 * threads;
}

int panel_supported(int k0, int n0, int n1) {
    if (n0 > 32 * kPanelMaxWaves || n1 > 32 * kPanelMaxWaves) return 0;  // one column tile per wave
    if (n0 > 64 * kPanelMaxCols || n1 > 64 * kPanelMaxCols) return 0;
    const int kx = k0 > n1 ? k0 : n1;
    const size_t lds = (size_t)(kPanelRows * panel_stride(kx) + kPanelRows * panel_stride(n0) +
                                (kPanelMaxWaves + 1) * 32 + (kPanelMaxWaves - 1) * 512 +
                                3 * (n0 + n1) + 16) * 4;
    return lds <= kPanelMaxLds;
}

int launch_panel_fwd(PanelFwdBatch& b, hipStream_t st) {
    int waves = 1, x0 = 0, x1 = 0;
    for (int i = 0; i < b.count; ++i) {
        const PanelFwdProb& p = b.p[i];
        for (int l = 0; l < p.nlayers; ++l) {
            const int w = panel_waves_for(l == 0 ? p.k0 : p.layer[l - 1].n, p.layer[l].n);
            if (w < 0) {
                set_error("panel kernel: layer width %d too large", p.layer[l].n);
                return MARL_ELIMIT;
            }
            waves = w > waves ? w : waves;
        }
        // X: the input and the outputs of the odd layers; Y: the outputs of the even layers
        int s0 = kPanelRows * panel_stride(p.k0), s1 = 0;
        for (int l = 0; l < p.nlayers; ++l) {
            const int v = kPanelRows * panel_stride(p.layer[l].n);
            if (l & 1)
                s0 = v > s0 ? v : s0;
            else
                s1 = v > s1 ? v : s1;
        }
        if (p.agg_at > 0 && (p.by_batch < 1 || p.g_na * p.by_batch > kPanelRows || p.agg_at >= p.nlayers ||
                             !panel_chain_supported(p.g_na, p.layer[p.agg_at - 1].n, 256))) {
            set_error("panel kernel: in-panel message mean outside its range");
            return MARL_ELIMIT;
        }
        x0 = s0 > x0 ? s0 : x0;
        x1 = s1 > x1 ? s1 : x1;
    }
    if (waves < 4) waves = 4;  // the LayerNorm row loop covers <= 4 rows per wave
    b.off_panel1 = x0;
    b.off_red = x0 + x1;
    b.off_part = b.off_red + (kPanelMaxWaves + 1) * 32;
    b.off_prm = b.off_part + (waves - 1) * 512;
    int prm = 0;
    for (int i = 0; i < b.count; ++i) {
        int q = 0;
        for (int l = 0; l < b.p[i].nlayers; ++l) q += 3 * b.p[i].layer[l].n;
        prm = q > prm ? q : prm;
    }
    const size_t lds = (size_t)(b.off_prm + prm + 16) * sizeof(float);
    if (lds > kPanelMaxLds) {
        set_error("panel kernel: shape outside its range");
        return MARL_ELIMIT;
    }
    static bool raised = false;
    if (!raised) {
        const void* kerns[3] = {reinterpret_cast<const void*>(panel_fwd_kernel<0>),
                                reinterpret_cast<const void*>(panel_fwd_kernel<4>),
                                reinterpret_cast<const void*>(panel_fwd_kernel<MARL_MAX_ACTIONS>)};
        for (const void* kf : kerns)
            MARL_HIP_CHECK(hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPanelMaxLds));
        raised = true;
    }
    b.ln_narrow_max = 128;  // widths that take the two-slot row pass
    unsigned pblocks = 0;
    for (int i = 0; i < b.count; ++i) {
        const PanelFwdProb& p = b.p[i];
        const unsigned nblk = p.by_batch > 0 ? (unsigned)cdiv(p.g_nb, p.by_batch) : (unsigned)cdiv(p.m, kPanelRows);
        pblocks = nblk > pblocks ? nblk : pblocks;
    }
#ifdef MARL_KERNEL_TS
    static long long* d_ts = nullptr;
    static int calls = 0;
    const int rec = ts_begin(&d_ts, calls++);
    b.ts = rec ? d_ts : nullptr;
#endif
    prof_before(4, st);
    if (b.has_sample && b.count == 1) {
        b.panel_blocks = (int)pblocks;
        const unsigned sblocks = (unsigned)cdiv(b.sample.R, waves);
        if (b.sample.nA <= 4)
            hipLaunchKernelGGL(panel_fwd_kernel<4>, dim3(pblocks + sblocks, 1), dim3(64 * waves), lds, st, b);
        else
            hipLaunchKernelGGL(panel_fwd_kernel<MARL_MAX_ACTIONS>, dim3(pblocks + sblocks, 1), dim3(64 * waves), lds, st, b);
    } else {
        b.has_sample = 0;
        hipLaunchKernelGGL(panel_fwd_kernel<0>, dim3(pblocks, (unsigned)b.count), dim3(64 * waves),
                           lds, st, b);
    }
    prof_after(4, st);
    MARL_LAUNCH_CHECK();
#ifdef MARL_KERNEL_TS
    if (rec) {
        fprintf(stderr, "[ts] panel_fwd count %d sample %d waves %d lds %zu\n", b.count, b.has_sample, waves, lds);
        ts_report("panel_fwd", d_ts, waves);
    }
#endif
    return MARL_OK;
}

// ===========================================================================
// backward: d(SiLU out) -> [LayerNorm/SiLU backward on the LDS panel -> dz (kept) ->
//           dX = dz * W on the matrix cores] per layer, last layer first
// ===========================================================================
constexpr int kBwdMaxCols = kPanelMaxCols;  // columns per lane in the row pass: widths up to 384

// MAXC: LayerNorm columns per lane the row pass is unrolled for (widths up to 64 * MAXC).  The chained
// encoder / decoder backward of the README dimensions has widths <= 128: two column slots instead of six
// are a third of the row pass's instructions and 24 fewer registers.
template <bool CELL, int MAXC>
__global__ __launch_bounds__(768, 4) void panel_bwd_kernel(const PanelBwdProb P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (CELL && (int)blockIdx.x >= P.panel_blocks) {
        const int64_t idx = (int64_t)((int)blockIdx.x - P.panel_blocks) * blockDim.x + threadIdx.x;
        if (P.cell_vec4)
            lstm_cell_bwd_elem4(P.cell, P.cell_rows, idx);
        else
            lstm_cell_bwd_elem(P.cell, P.cell_rows, idx);
        return;
    }
    const int m0 = blockIdx.x * kPanelRows;
    const int bpb = P.by_batch;  // > 0: this workgroup owns all agents of bpb batch elements
    if (bpb ? (int)blockIdx.x * bpb >= P.g_nb : m0 >= P.m) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int quad = lane >> 4, l16 = lane & 15;
    float* D = lds;            // current gradient panel
    float* E = lds + P.off_e;  // next one
    float* prm = lds + P.off_prm;
    float* colp = lds + P.off_colp;  // [nwaves][2][n]
    float* part = lds + P.off_part;
    // global row of local row lr, or -1 (padding): a small table in LDS
    int* rowmap = reinterpret_cast<int*>(lds + P.off_rowmap);
    if (tid < kPanelRows) {
        int r = m0 + tid < P.m ? m0 + tid : -1;
        if (bpb) {
            const int a = tid / bpb, b = (int)blockIdx.x * bpb + (tid - a * bpb);
            r = (a < P.g_na && b < P.g_nb) ? a * P.g_nb + b : -1;
        }
        rowmap[tid] = r;
    }
    lds_barrier();
#ifdef MARL_KERNEL_TS
    MARL_TS_DECL(P.ts);
#endif
    MARL_TS();

    // The saved pre-LayerNorm rows (and statistics) of a layer are requested one layer ahead: the
    // two rows of this wave for layer l + 1 fly while layer l's dX product runs.  Unconditional
    // loads (padding rows read row 0) so that no wait is widened by a branch.
    float zpre[2][MAXC], mpre[2], rpre[2];
    auto zfetch = [&](const PanelBwdLayer& L) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = wave + i * nwaves;
            const int row = lr < kPanelRows ? rowmap[lr] : -1;
            const size_t rr = row < 0 ? 0 : (size_t)row;
            mpre[i] = L.stats[rr * 2];
            rpre[i] = L.stats[rr * 2 + 1];
#pragma unroll
            for (int u = 0; u < MAXC; ++u) {
                const int c = lane + 64 * u;
                zpre[i][u] = c < L.n ? L.z[rr * L.ldz + c] : 0.f;
            }
        }
    };
    zfetch(P.layer[0]);  // first: in flight behind the parameter and gradient-panel loads below
    // LayerNorm affine parameters of every layer -> LDS; d(a_last) panel -> D
    {
        // (all layers' vectors requested before anything is waited for; n <= 384 <= threads)
        float pv[kPanelMaxLayers][2];
#pragma unroll
        for (int l = 0; l < kPanelMaxLayers; ++l) {
            pv[l][0] = pv[l][1] = 0.f;
            if (l < P.nlayers && tid < P.layer[l].n) {
                pv[l][0] = P.layer[l].gamma[tid];
                pv[l][1] = P.layer[l].beta[tid];
            }
        }
        int off = 0;
#pragma unroll
        for (int l = 0; l < kPanelMaxLayers; ++l) {
            if (l < P.nlayers) {
                const int n = P.layer[l].n;
                if (tid < n) {
                    prm[off + tid] = pv[l][0];
                    prm[off + n + tid] = pv[l][1];
                }
                off += 2 * n;
            }
        }
        const int n = P.layer[0].n, ds = panel_stride(n), n16 = (n + 15) & ~15, c4 = n16 >> 2;
        const int n4 = (n + 3) & ~3;
        for (int e = tid; e < kPanelRows * c4; e += blockDim.x) {
            const int lr = e / c4, k = (e % c4) * 4;
            const int r = rowmap[lr];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r >= 0 && k < n4 && P.agg_na > 1) {
                const int na = P.agg_na, nb = P.agg_nb;
                const float den = (float)(na - 1);
                const float* base = P.da + (size_t)(r % nb) * P.ldda + k;
                const size_t astr = (size_t)nb * P.ldda;
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int a0 = 0; a0 < na; a0 += 8) {  // 8 independent loads in flight
                    float4 qv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int a = a0 + u < na ? a0 + u : na - 1;
                        qv[u] = *reinterpret_cast<const float4*>(base + (size_t)a * astr);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (a0 + u < na) {  // sequential over agents, as in the forward mean
                            s.x += qv[u].x;
                            s.y += qv[u].y;
                            s.z += qv[u].z;
                            s.w += qv[u].w;
                        }
                    }
                }
                const float4 me = *reinterpret_cast<const float4*>(P.da + (size_t)r * P.ldda + k);
                v = make_float4((s.x - me.x) / den, (s.y - me.y) / den, (s.z - me.z) / den,
                                (s.w - me.w) / den);
            } else if (r >= 0 && k < n4 && P.agg_na == 0) {
                v = *reinterpret_cast<const float4*>(P.da + (size_t)r * P.ldda + k);
                if (P.da2) {
                    const float4 w = *reinterpret_cast<const float4*>(P.da2 + (size_t)r * P.ldda2 + k);
                    v = make_float4(v.x + w.x, v.y + w.y, v.z + w.z, v.w + w.w);
                }
            }
            *reinterpret_cast<float4*>(D + lr * ds + k) = v;
        }
    }
    MARL_TS();
    lds_barrier();
    MARL_TS();

    int prm_off = 0;
    for (int l = 0; l < P.nlayers; ++l) {
        const PanelBwdLayer Lr = P.layer[l];  // by value: no scalar re-loads inside the loops
        const int n = Lr.n, ds = panel_stride(n), n16 = (n + 15) & ~15;
        const float* lgamma = prm + prm_off;
        const float* lbeta = lgamma + n;
        prm_off += 2 * n;
        if (l > 0 && l == P.agg_at) {
            // D holds the gradient of the message MEAN for every agent of this workgroup's batch
            // elements: the mean over the other agents is self-adjoint - the same sum (sequential
            // over agents, as the staging version) gives the gradient of the messages
            const int c4 = n16 >> 2, na = P.g_na;
            const float den = (float)(na - 1);
            float4 v[2];
#pragma unroll
            for (int it = 0; it < 2; ++it) {  // launcher: kPanelRows * c4 <= 2 * blockDim.x
                const int e = tid + it * (int)blockDim.x;
                v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < kPanelRows * c4) {
                    const int lr = e / c4, k = (e - lr * c4) * 4;
                    const int a = lr / bpb, i = lr - a * bpb;
                    if (a < na && na > 1) {
                        float4 sm = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (int a2 = 0; a2 < na; ++a2) {
                            const float4 q = *reinterpret_cast<const float4*>(D + (a2 * bpb + i) * ds + k);
                            sm.x += q.x;
                            sm.y += q.y;
                            sm.z += q.z;
                            sm.w += q.w;
                        }
                        const float4 me = *reinterpret_cast<const float4*>(D + lr * ds + k);
                        v[it] = make_float4((sm.x - me.x) / den, (sm.y - me.y) / den, (sm.z - me.z) / den,
                                            (sm.w - me.w) / den);
                    }
                }
            }
            lds_barrier();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int e = tid + it * (int)blockDim.x;
                if (e < kPanelRows * c4) {
                    const int lr = e / c4, k = (e - lr * c4) * 4;
                    *reinterpret_cast<float4*>(D + lr * ds + k) = v[it];
                }
            }
            lds_barrier();
        }
        // ---- row pass: D holds d(SiLU out); D <- dz, dgamma/dbeta column partials.  A row
        // lives in registers between the statistics and the dz pass.
        {
            constexpr int RPW = 2;  // launcher guarantees >= 8 waves for the 16 rows
            float pg[MAXC], pb[MAXC], g[MAXC], bt[MAXC];
#pragma unroll
            for (int u = 0; u < MAXC; ++u) {
                const int c = lane + 64 * u;
                pg[u] = pb[u] = 0.f;
                g[u] = c < n ? lgamma[c] : 0.f;
                bt[u] = c < n ? lbeta[c] : 0.f;
            }
            const float inv_n = 1.0f / (float)n;
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int lr = wave + i * nwaves;
                if (lr < kPanelRows) {  // wave-uniform
                    const int row = rowmap[lr];
                    const bool rv = row >= 0;
                    float xh[MAXC], dxh[MAXC];
                    float s1 = 0.f, s2 = 0.f, rstd = 0.f;
                    if (rv) {
                        const float mean = mpre[i];
                        rstd = rpre[i];
#pragma unroll
                        for (int u = 0; u < MAXC; ++u) {
                            const int c = lane + 64 * u;
                            xh[u] = dxh[u] = 0.f;
                            if (c < n) {
                                xh[u] = (zpre[i][u] - mean) * rstd;
                                const float dy = D[lr * ds + c] * silu_grad_p(g[u] * xh[u] + bt[u]);
                                dxh[u] = dy * g[u];
                                s1 += dxh[u];
                                s2 += dxh[u] * xh[u];
                                pg[u] += dy * xh[u];
                                pb[u] += dy;
                            }
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < MAXC; ++u) xh[u] = dxh[u] = 0.f;
                    }
                    const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
#pragma unroll
                    for (int u = 0; u < MAXC; ++u) {
                        const int c = lane + 64 * u;
                        if (c < n16) {
                            float dzv = 0.f;
                            if (rv && c < n) {
                                dzv = rstd * (dxh[u] - m1 - xh[u] * m2);
                                Lr.dz[(size_t)row * Lr.lddz + c] = dzv;
                            }
                            D[lr * ds + c] = dzv;
                        }
                    }
                }
            }
            float* cw = colp + (size_t)wave * 2 * n;
#pragma unroll
            for (int u = 0; u < MAXC; ++u) {
                const int c = lane + 64 * u;
                if (c < n) {
                    cw[c] = pg[u];
                    cw[n + c] = pb[u];
                }
            }
        }
        MARL_TS();
        lds_barrier();
        MARL_TS();
        for (int c = tid; c < 2 * n; c += blockDim.x) {
            float t = 0.f;
            for (int w = 0; w < nwaves; ++w) t += colp[(size_t)w * 2 * n + c];
            Lr.part[(size_t)blockIdx.x * 2 * n + c] = t;
        }
        if (l + 1 < P.nlayers) zfetch(P.layer[l + 1]);
        MARL_TS();
        // ---- dX = dz * W: out tiles over k_in columns, contraction over n
        const int nout = Lr.k_in, nt = (nout + 31) >> 5;
        const int ks = panel_ksplit(n, nt, nwaves);
        const int kper = (((n16 + ks - 1) / ks) + 15) & ~15;
        const bool last = l + 1 == P.nlayers;
        const int es = panel_stride(nout), o16 = (nout + 15) & ~15;
        // more column tiles than waves: walk them in rounds (first layer's input can be wide)
        for (int t0 = 0; t0 < nt; t0 += nwaves / ks) {
            const int tiles_round = nwaves / ks;
            const int j = t0 + wave % tiles_round, s = wave / tiles_round;
            const bool active = s < ks && j < nt;
            const int kbeg = s * kper;
            const int kend = kbeg + kper < n16 ? kbeg + kper : n16;
            f32x4 acc[2];
            if (active) panel_gemm(acc, D, ds, n, Lr.wt, Lr.ldwt, nout, j, kbeg, kend, lane);
            // slice partials are indexed by the tile's slot in this round
            panel_ksum(acc, part, tiles_round, ks, wave % tiles_round, s, lane, active);
            if (active && s == 0) {
                int rw[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) rw[r] = last ? rowmap[4 * quad + r] : -1;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int col = j * 32 + 16 * t + l16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int lr = 4 * quad + r;
                        const int row = rw[r];
                        const float v = col < nout ? acc[t][r] : 0.f;
                        if (last && P.tail_lds) {  // final dX through LDS: finished row-wise below
                            if (col < o16) E[lr * es + col] = v;
                        } else if (last) {
                            if (col < nout && row >= 0) {
                                float* o = P.dx + (size_t)row * P.lddx + col;
                                const float tot = P.accumulate ? *o + v : v;
                                *o = tot;
                                if (P.has_cellb) lstm_cell_bwd_at(P.cellb, row, col, tot);
                            }
                        } else if (col < o16) {
                            E[lr * es + col] = v;
                        }
                    }
                }
            }
            lds_barrier();
        }
        MARL_TS();
        if (last && P.tail_lds) {
            // Final dX (+ the belief cell's elementwise backward) ROW-wise from the LDS panel: four
            // consecutive columns per thread, 16-byte loads of everything two items need before
            // the first store.  In the accumulator layout the same work is eight scattered
            // elements per lane whose stores keep the next element's loads from being issued
            // (in-kernel timestamps: 20 us of the 46 us chain).
            const LstmBwdArgs& Cb = P.cellb;
            const int c4n = nout >> 2, items = kPanelRows * c4n, cn = Cb.n;
            // (one item per pass: two in flight need 162 registers and then the riding-along cell
            // workgroups no longer fit beside a panel workgroup)
            constexpr int kIt = 1;
            for (int base = 0; base < items; base += kIt * (int)blockDim.x) {
                float4 ov[kIt], gi[kIt], gf[kIt], gg[kIt], go[kIt], cnew[kIt], cprev[kIt], dcv[kIt];
                int rowi[kIt], coli[kIt], lri[kIt];
#pragma unroll
                for (int k = 0; k < kIt; ++k) {
                    const int i = base + tid + k * (int)blockDim.x;
                    const int lr = i < items ? i / c4n : 0;
                    const int c = i < items ? (i - lr * c4n) * 4 : 0;
                    const int row = i < items ? rowmap[lr] : -1;
                    const size_t rr = row < 0 ? 0 : (size_t)row;
                    rowi[k] = row;
                    coli[k] = c;
                    lri[k] = lr;
                    ov[k] = P.accumulate ? *reinterpret_cast<const float4*>(P.dx + rr * P.lddx + c)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (P.has_cellb) {
                        const float* g = Cb.gates + rr * Cb.ldg + c;
                        gi[k] = *reinterpret_cast<const float4*>(g);
                        gf[k] = *reinterpret_cast<const float4*>(g + cn);
                        gg[k] = *reinterpret_cast<const float4*>(g + 2 * cn);
                        go[k] = *reinterpret_cast<const float4*>(g + 3 * cn);
                        cnew[k] = *reinterpret_cast<const float4*>(Cb.c_new + rr * Cb.ldc + c);
                        cprev[k] = *reinterpret_cast<const float4*>(Cb.c_prev + rr * Cb.ldc + c);
                        dcv[k] = *reinterpret_cast<const float4*>(Cb.dc + rr * Cb.lddc + c);
                    }
                }
#pragma unroll
                for (int k = 0; k < kIt; ++k) {
                    if (rowi[k] < 0) continue;
                    const float4 v = *reinterpret_cast<const float4*>(E + lri[k] * es + coli[k]);
                    const float tot[4] = {ov[k].x + v.x, ov[k].y + v.y, ov[k].z + v.z, ov[k].w + v.w};
                    *reinterpret_cast<float4*>(P.dx + (size_t)rowi[k] * P.lddx + coli[k]) =
                        make_float4(tot[0], tot[1], tot[2], tot[3]);
                    if (P.has_cellb) {  // lstm_cell_bwd_at() on four consecutive units
                        const float gi_[4] = {gi[k].x, gi[k].y, gi[k].z, gi[k].w};
                        const float gf_[4] = {gf[k].x, gf[k].y, gf[k].z, gf[k].w};
                        const float gg_[4] = {gg[k].x, gg[k].y, gg[k].z, gg[k].w};
                        const float go_[4] = {go[k].x, go[k].y, go[k].z, go[k].w};
                        const float cn_[4] = {cnew[k].x, cnew[k].y, cnew[k].z, cnew[k].w};
                        const float cp_[4] = {cprev[k].x, cprev[k].y, cprev[k].z, cprev[k].w};
                        const float dc_[4] = {dcv[k].x, dcv[k].y, dcv[k].z, dcv[k].w};
                        float o0[4], o1[4], o2[4], o3[4], o4[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float tc = tanh_fast(cn_[q]);
                            const float d = tot[q] * go_[q] * (1.0f - tc * tc) + dc_[q];
                            o0[q] = d * gg_[q] * gi_[q] * (1.0f - gi_[q]);
                            o1[q] = d * cp_[q] * gf_[q] * (1.0f - gf_[q]);
                            o2[q] = d * gi_[q] * (1.0f - gg_[q] * gg_[q]);
                            o3[q] = tot[q] * tc * go_[q] * (1.0f - go_[q]);
                            o4[q] = d * gf_[q];
                        }
                        if (!(Cb.g3 && Cb.skip_f32)) {
                            float* g = Cb.gates + (size_t)rowi[k] * Cb.ldg + coli[k];
                            *reinterpret_cast<float4*>(g) = make_float4(o0[0], o0[1], o0[2], o0[3]);
                            *reinterpret_cast<float4*>(g + cn) = make_float4(o1[0], o1[1], o1[2], o1[3]);
                            *reinterpret_cast<float4*>(g + 2 * cn) = make_float4(o2[0], o2[1], o2[2], o2[3]);
                            *reinterpret_cast<float4*>(g + 3 * cn) = make_float4(o3[0], o3[1], o3[2], o3[3]);
                        }
                        *reinterpret_cast<float4*>(Cb.dc + (size_t)rowi[k] * Cb.lddc + coli[k]) =
                            make_float4(o4[0], o4[1], o4[2], o4[3]);
                        if (Cb.g3) {  // the k16 image of the gate gradients (A operand of the image GEMMs)
                            const int64_t ir = (int64_t)Cb.g3_row0 + rowi[k];
                            const int c0 = coli[k];
                            img_store4(Cb.g3 + img_off(ir, c0 >> 4, Cb.g3_steps), c0, o0[0], o0[1], o0[2], o0[3]);
                            img_store4(Cb.g3 + img_off(ir, (cn + c0) >> 4, Cb.g3_steps), cn + c0, o1[0], o1[1], o1[2], o1[3]);
                            img_store4(Cb.g3 + img_off(ir, (2 * cn + c0) >> 4, Cb.g3_steps), 2 * cn + c0, o2[0], o2[1], o2[2], o2[3]);
                            img_store4(Cb.g3 + img_off(ir, (3 * cn + c0) >> 4, Cb.g3_steps), 3 * cn + c0, o3[0], o3[1], o3[2], o3[3]);
                        }
                    }
                }
            }
        }
        MARL_TS();
        float* tmp = D;
        D = E;
        E = tmp;
    }
}

int panel_bwd_blocks(int m) { return (int)cdiv(m, kPanelRows); }
int panel_chain_blocks(int na, int nb) { return na >= 1 && na <= kPanelRows ? (int)cdiv(nb, kPanelRows / na) : 0; }
int panel_chain_by_batch(int na) { return na >= 1 && na <= kPanelRows ? kPanelRows / na : 0; }

int launch_panel_bwd(PanelBwdProb& p, hipStream_t st) {
    int waves = 8, pmax = 0, nmax = 0, prm = 0;
    // the last layer's dX goes through LDS and is finished row-wise (16-byte accesses) when its
    // rows keep float4 alignment
    const PanelBwdLayer& Ll = p.layer[p.nlayers - 1];
    bool tail_lds = (Ll.k_in & 3) == 0 && (p.lddx & 3) == 0 &&
                    (!p.has_cellb || ((p.cellb.n & 3) == 0 && (p.cellb.ldg & 3) == 0 && (p.cellb.ldc & 3) == 0 &&
                                      (p.cellb.lddc & 3) == 0 && p.cellb.n == Ll.k_in)) &&
                    true;
plan:  // (second pass without the LDS tail when the extra output panel does not fit)
    waves = 8;
    pmax = nmax = prm = 0;
    p.tail_lds = tail_lds ? 1 : 0;
    for (int l = 0; l < p.nlayers; ++l) {
        const PanelBwdLayer& L = p.layer[l];
        if (L.n > 64 * kBwdMaxCols) {
            set_error("panel backward: LayerNorm width %d > %d", L.n, 64 * kBwdMaxCols);
            return MARL_ELIMIT;
        }
        int w = panel_waves_for(L.n, L.k_in);
        if (w < 0) w = kPanelMaxWaves;  // wide input: tiles are walked in rounds
        waves = w > waves ? w : waves;
        nmax = L.n > nmax ? L.n : nmax;
        pmax = panel_stride(L.n) > pmax ? panel_stride(L.n) : pmax;
        if ((l + 1 < p.nlayers || tail_lds) && panel_stride(L.k_in) > pmax) pmax = panel_stride(L.k_in);
        prm += 2 * L.n;
    }
    if (waves > kPanelMaxWaves) waves = kPanelMaxWaves;
    p.off_e = kPanelRows * pmax;
    p.off_prm = 2 * kPanelRows * pmax;
    p.off_colp = p.off_prm + prm + 16;
    p.off_part = p.off_colp + waves * 2 * nmax;
    p.off_rowmap = p.off_part + (waves - 1) * 512;
    const size_t lds = (size_t)(p.off_rowmap + kPanelRows) * sizeof(float);
    if (p.agg_at > 0 && (p.by_batch < 1 || p.g_na * p.by_batch > kPanelRows || p.agg_at >= p.nlayers ||
                         !panel_chain_supported(p.g_na, p.layer[p.agg_at].n, 256))) {
        set_error("panel backward: in-panel message mean outside its range");
        return MARL_ELIMIT;
    }
    if (lds > kPanelMaxLds && tail_lds) {
        tail_lds = false;
        goto plan;
    }
    if (lds > kPanelMaxLds) {
        set_error("panel backward: shape outside its range");
        return MARL_ELIMIT;
    }
    static bool raised = false;
    if (!raised) {
        const void* kerns[4] = {reinterpret_cast<const void*>(panel_bwd_kernel<false, 2>),
                                reinterpret_cast<const void*>(panel_bwd_kernel<true, 2>),
                                reinterpret_cast<const void*>(panel_bwd_kernel<false, kBwdMaxCols>),
                                reinterpret_cast<const void*>(panel_bwd_kernel<true, kBwdMaxCols>)};
        for (const void* kf : kerns)
            MARL_HIP_CHECK(hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPanelMaxLds));
        raised = true;
    }
    const unsigned pblocks = p.by_batch > 0 ? (unsigned)cdiv(p.g_nb, p.by_batch) : (unsigned)cdiv(p.m, kPanelRows);
    prof_before(4, st);
    p.cellb_img_done = p.has_cellb && p.cellb.g3 && tail_lds;
    p.cell_vec4 = p.has_cell && lstm_bwd_vec4_ok(p.cell);
    p.cell_img_done = p.has_cell && p.cell.g3 && p.cell_vec4;
    if (!p.cellb_img_done) p.cellb.skip_f32 = 0;  // (the fallback builds the image from the fp32 gradients)
    if (!p.cell_img_done) p.cell.skip_f32 = 0;
#ifdef MARL_KERNEL_TS
    static long long* d_ts = nullptr;
    static int calls = 0;
    const int rec = ts_begin(&d_ts, calls++);
    p.ts = rec ? d_ts : nullptr;
#endif
    const int maxc = nmax <= 128 ? 2 : kBwdMaxCols;
    if (p.has_cell) {
        p.panel_blocks = (int)pblocks;
        const unsigned cblocks = (unsigned)cdiv(p.cell_rows * (p.cell_vec4 ? p.cell.n / 4 : p.cell.n), 64 * waves);
        if (maxc == 2)
            hipLaunchKernelGGL((panel_bwd_kernel<true, 2>), dim3(pblocks + cblocks), dim3(64 * waves), lds, st, p);
        else
            hipLaunchKernelGGL((panel_bwd_kernel<true, kBwdMaxCols>), dim3(pblocks + cblocks), dim3(64 * waves), lds, st, p);
    } else if (maxc == 2) {
        hipLaunchKernelGGL((panel_bwd_kernel<false, 2>), dim3(pblocks), dim3(64 * waves), lds, st, p);
    } else {
        hipLaunchKernelGGL((panel_bwd_kernel<false, kBwdMaxCols>), dim3(pblocks), dim3(64 * waves), lds, st, p);
    }
    prof_after(4, st);
    MARL_LAUNCH_CHECK();
#ifdef MARL_KERNEL_TS
    if (rec) {
        fprintf(stderr, "[ts] panel_bwd layers %d waves %d lds %zu cell %d cellb %d tail_lds %d\n", p.nlayers, waves, lds,
                p.has_cell, p.has_cellb, p.tail_lds);
        ts_report("panel_bwd", d_ts, waves);
    }
#endif
    return MARL_OK;
}

}  // namespace marl
