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

namespace marl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float half_sum(float v) {  // over the 32 lanes of a half wave
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}
// Workgroup barrier that only waits for LDS traffic: the global stores of saved activations
// issued before it are never read back by this kernel, so there is no reason to drain vmcnt
// (a plain __syncthreads() would wait for every outstanding store: ~1-2 us each time).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ float silu_p(float y) { return y / (1.0f + expf(-y)); }
__device__ __forceinline__ float silu_grad_p(float y) {
    const float s = 1.0f / (1.0f + expf(-y));
    return s * (1.0f + y * (1.0f - s));
}
__device__ __forceinline__ int tile_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

constexpr int kPanelRows = 32;
constexpr int kPanelMaxWaves = 12;  // 768 threads: 170 VGPRs per lane, no spills

__host__ __device__ inline int panel_stride(int k) { return ((k + 7) & ~7) + 4; }

// One wave = one (column tile j, K slice s) pair: acc = in[32 x Kslice] (LDS) * W[tile rows,
// Kslice]^T.  Weight fragments come straight from global memory (L2), 8 x 16-byte loads (a
// 64-deep chunk) in flight per wave; splitting K over waves puts ALL of a layer's weight
// loads in flight at once, so a layer costs ~one L2 round trip + <= 32 MFMAs per wave.
__device__ __forceinline__ void panel_gemm(f32x16& acc, const float* in, int stride, int K,
                                           const float* __restrict__ w, int ldw, int n, int j,
                                           int kbeg, int kend, int lane) {
    const int K4 = (K + 3) & ~3;
    const int half = lane >> 5;
    const float* arow = in + (lane & 31) * stride + 4 * half;
    int row = j * 32 + (lane & 31);
    row = row < n ? row : n - 1;
    const float* wrow = w + (size_t)row * ldw + 4 * half;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += 64) {
        float4 bq[8];
        // columns [K, round8(K)) of the LDS panel are zero: a clamped (finite) read is exact
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kk = k0 + 8 * i;
            bq[i] = *reinterpret_cast<const float4*>(wrow + (kk + 4 * half < K4 ? kk : -4 * half));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (k0 + 8 * i < kend) {
                const float4 a = *reinterpret_cast<const float4*>(arow + k0 + 8 * i);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[i].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[i].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[i].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[i].w, acc, 0, 0, 0);
            }
        }
    }
}

// K-slice partial tiles -> slice 0's accumulator (fixed order).  part: [(ks-1) * nt][64][16]
__device__ __forceinline__ void panel_ksum(f32x16& acc, float* part, int nt, int ks, int j, int s,
                                           int lane, bool active) {
    if (ks > 1) {
        if (active && s > 0) {
            float* p = part + ((size_t)((s - 1) * nt + j) * 64 + lane) * 16;
#pragma unroll
            for (int r = 0; r < 16; r += 4)
                *reinterpret_cast<float4*>(p + r) = make_float4(acc[r], acc[r + 1], acc[r + 2], acc[r + 3]);
        }
        __syncthreads();
        if (active && s == 0) {
            for (int q = 1; q < ks; ++q) {
                const float* p = part + ((size_t)((q - 1) * nt + j) * 64 + lane) * 16;
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(p + r);
                    acc[r] += v.x;
                    acc[r + 1] += v.y;
                    acc[r + 2] += v.z;
                    acc[r + 3] += v.w;
                }
            }
        }
    }
}

// Row sums over all columns of the panel: per-lane val[r] (this wave's tile, 0 if it has
// none) -> res[32] * scale.  red: [nt][32] floats.
__device__ __forceinline__ void panel_row_sum(const float (&val)[16], bool owner, int j, int nt,
                                              float* red, float* res, int lane, float scale) {
    const int half = lane >> 5;
    if (owner) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s = half_sum(val[r]);
            if ((lane & 31) == 0) red[j * 32 + tile_row(r, half)] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        float s = 0.f;
        for (int w = 0; w < nt; ++w) s += red[w * 32 + threadIdx.x];
        res[threadIdx.x] = s * scale;
    }
    __syncthreads();
}

__host__ __device__ inline int panel_ksplit(int K, int nt, int nwaves) {
    const int K8 = (K + 7) & ~7;
    int ks = nwaves / nt;
    const int want = (K8 + 63) / 64;
    ks = ks < want ? ks : want;
    return ks < 1 ? 1 : ks;
}

// ===========================================================================
// forward
// ===========================================================================
__global__ __launch_bounds__(768) void panel_fwd_kernel(const PanelFwdBatch B) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const PanelFwdProb& P = B.p[blockIdx.y];
    const int m0 = blockIdx.x * kPanelRows;
    if (m0 >= P.m) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int half = lane >> 5;
    float* X = lds;
    float* Y = lds + B.off_panel1;
    float* part = lds + B.off_part;
    float* prm = lds + B.off_prm;  // [layer][bias | gamma | beta][n] staged once
    {
        int off = 0;
        for (int l = 0; l < P.nlayers; ++l) {
            const int n = P.layer[l].n;
            for (int c = tid; c < n; c += blockDim.x) {
                prm[off + c] = P.layer[l].bias[c];
                prm[off + n + c] = P.layer[l].gamma[c];
                prm[off + 2 * n + c] = P.layer[l].beta[c];
            }
            off += 3 * n;
        }
    }

    // ---- stage the 32-row input panel (zero-filled past k0 and past M)
    {
        const int k0 = P.k0, xs = panel_stride(k0), K8 = (k0 + 7) & ~7;
        const int c4 = K8 >> 2;
        const int K4 = (k0 + 3) & ~3;
        if (P.agg_na > 0) {
            // message mean over the OTHER agents (networks/message.py:5-17), 4 loads in flight
            const int na = P.agg_na, nb = P.agg_nb;
            const float den = (float)(na - 1);
            for (int e = tid; e < kPanelRows * c4; e += blockDim.x) {
                const int lr = e / c4, k = (e % c4) * 4;
                const int r = m0 + lr;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < P.m && k < K4 && na > 1) {
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
            for (int e = tid; e < kPanelRows * c4; e += blockDim.x) {
                const int lr = e / c4, k = (e % c4) * 4;
                const int r = m0 + lr;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < P.m && k < K4) v = *reinterpret_cast<const float4*>(P.x + (size_t)r * P.ldx + k);
                *reinterpret_cast<float4*>(X + lr * xs + k) = v;
            }
        }
    }
    lds_barrier();

    for (int l = 0; l < P.nlayers; ++l) {
        const PanelLayer& Lr = P.layer[l];
        const float* lbias = prm + (l == 0 ? 0 : 3 * P.layer[0].n);
        const float* lgamma = lbias + Lr.n;
        const float* lbeta = lgamma + Lr.n;
        const float* in = (l & 1) ? Y : X;
        float* outp = (l & 1) ? X : Y;  // this layer's output panel (next layer's input)
        const int K = l == 0 ? P.k0 : P.layer[0].n;
        const int K8 = (K + 7) & ~7;
        const int stride = panel_stride(K);
        const int n = Lr.n, nt = (n + 31) >> 5;
        const int ks = panel_ksplit(K, nt, nwaves);
        const int j = wave % nt, s = wave / nt;
        const bool active = s < ks;
        const int kper = (((K8 + ks - 1) / ks) + 7) & ~7;
        const int kbeg = s * kper;
        const int kend = kbeg + kper < K8 ? kbeg + kper : K8;
        f32x16 acc;
        if (active) panel_gemm(acc, in, stride, K, Lr.w, Lr.ldw, n, j, kbeg, kend, lane);
        panel_ksum(acc, part, nt, ks, j, s, lane, active);

        // z = acc + bias -> output panel in LDS (and global, kept for backward)
        const int ys = panel_stride(n), n8 = (n + 7) & ~7;
        if (active && s == 0) {
            const int col = j * 32 + (lane & 31);
            const bool cv = col < n;
            const float bv = cv ? lbias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = tile_row(r, half);
                const float zv = cv ? acc[r] + bv : 0.f;
                if (col < n8) outp[lr * ys + col] = zv;
                if (cv && Lr.z && m0 + lr < P.m) Lr.z[(size_t)(m0 + lr) * Lr.ldz + col] = zv;
            }
        }
        lds_barrier();
        // LayerNorm + SiLU from the LDS panel: wave w owns rows w, w + nwaves, ... and walks
        // them TOGETHER (independent shuffle chains interleave), two-pass statistics, result
        // written in place (next layer's input) and to global.
        {
            constexpr int RPW = 8;  // >= 32 rows / 4 waves
            float sm[RPW], q[RPW], mean[RPW], rstd[RPW];
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                sm[i] = 0.f;
                const int lr = wave + i * nwaves;
                if (lr < kPanelRows)
                    for (int c = lane; c < n; c += 64) sm[i] += outp[lr * ys + c];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int i = 0; i < RPW; ++i) sm[i] += __shfl_xor(sm[i], o);
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                mean[i] = sm[i] / (float)n;
                q[i] = 0.f;
                const int lr = wave + i * nwaves;
                if (lr < kPanelRows)
                    for (int c = lane; c < n; c += 64) {
                        const float d = outp[lr * ys + c] - mean[i];
                        q[i] += d * d;
                    }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int i = 0; i < RPW; ++i) q[i] += __shfl_xor(q[i], o);
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                rstd[i] = 1.0f / sqrtf(q[i] / (float)n + 1e-5f);
                const int lr = wave + i * nwaves;
                if (lr < kPanelRows) {
                    float* zr = outp + lr * ys;
                    const int row = m0 + lr;
                    float* arow = Lr.a + (size_t)row * Lr.lda;
                    for (int c = lane; c < n; c += 64) {
                        const float av = silu_p((zr[c] - mean[i]) * rstd[i] * lgamma[c] + lbeta[c]);
                        zr[c] = av;
                        if (row < P.m) arow[c] = av;
                    }
                    if (Lr.stats && lane == 0 && row < P.m) {
                        Lr.stats[(size_t)row * 2] = mean[i];
                        Lr.stats[(size_t)row * 2 + 1] = rstd[i];
                    }
                }
            }
        }
        lds_barrier();
    }
}

constexpr size_t kPanelMaxLds = 144 * 1024;

static int panel_waves_for(int k, int n) {
    const int nt = (n + 31) / 32;
    const int want = (((k + 7) & ~7) + 63) / 64;
    int w = nt * want;
    if (w < nt) w = nt;
    return w > kPanelMaxWaves ? (nt > kPanelMaxWaves ? -1 : (kPanelMaxWaves / nt) * nt) : w;
}

int panel_supported(int k0, int n0, int n1) {
    if (n0 > 32 * kPanelMaxWaves || n1 > 32 * kPanelMaxWaves) return 0;  // one column tile per wave
    const int kx = k0 > n1 ? k0 : n1;
    const size_t lds = (size_t)(kPanelRows * panel_stride(kx) + kPanelRows * panel_stride(n0) +
                                (kPanelMaxWaves + 1) * 32 + (kPanelMaxWaves - 1) * 1024 +
                                3 * (n0 + n1) + 16) * 4;
    return lds <= kPanelMaxLds;
}

int launch_panel_fwd(PanelFwdBatch& b, hipStream_t st) {
    int mmax = 0, waves = 1, x0 = 0, x1 = 0;
    for (int i = 0; i < b.count; ++i) {
        const PanelFwdProb& p = b.p[i];
        mmax = p.m > mmax ? p.m : mmax;
        for (int l = 0; l < p.nlayers; ++l) {
            const int w = panel_waves_for(l == 0 ? p.k0 : p.layer[0].n, p.layer[l].n);
            if (w < 0) {
                set_error("panel kernel: layer width %d too large", p.layer[l].n);
                return MARL_ELIMIT;
            }
            waves = w > waves ? w : waves;
        }
        int s0 = kPanelRows * panel_stride(p.k0);
        if (p.nlayers > 1 && kPanelRows * panel_stride(p.layer[1].n) > s0)
            s0 = kPanelRows * panel_stride(p.layer[1].n);  // layer 1 writes its output panel here
        const int s1 = kPanelRows * panel_stride(p.layer[0].n);
        x0 = s0 > x0 ? s0 : x0;
        x1 = s1 > x1 ? s1 : x1;
    }
    if (waves < 4) waves = 4;  // the LayerNorm row loop keeps <= 8 rows per wave in flight
    b.off_panel1 = x0;
    b.off_red = x0 + x1;
    b.off_part = b.off_red + (kPanelMaxWaves + 1) * 32;
    b.off_prm = b.off_part + (waves - 1) * 1024;
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
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kPanelMaxLds));
        raised = true;
    }
    hipLaunchKernelGGL(panel_fwd_kernel, dim3((unsigned)cdiv(mmax, kPanelRows), (unsigned)b.count),
                       dim3(64 * waves), lds, st, b);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ===========================================================================
// backward: d(SiLU out) -> [LayerNorm/SiLU backward on the LDS panel -> dz (kept) ->
//           dX = dz * W on the matrix cores] per layer, last layer first
// ===========================================================================
constexpr int kBwdMaxCols = 6;  // columns per lane in the row pass: widths up to 384

__global__ __launch_bounds__(768) void panel_bwd_kernel(const PanelBwdProb P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m0 = blockIdx.x * kPanelRows;
    if (m0 >= P.m) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int half = lane >> 5;
    float* D = lds;            // current gradient panel
    float* E = lds + P.off_e;  // next one
    float* prm = lds + P.off_prm;
    float* colp = lds + P.off_colp;  // [nwaves][2][n]
    float* part = lds + P.off_part;

    // LayerNorm affine parameters of every layer -> LDS; d(a_last) panel -> D
    {
        int off = 0;
        for (int l = 0; l < P.nlayers; ++l) {
            const int n = P.layer[l].n;
            for (int c = tid; c < n; c += blockDim.x) {
                prm[off + c] = P.layer[l].gamma[c];
                prm[off + n + c] = P.layer[l].beta[c];
            }
            off += 2 * n;
        }
        const int n = P.layer[0].n, ds = panel_stride(n), n8 = (n + 7) & ~7, c4 = n8 >> 2;
        const int n4 = (n + 3) & ~3;
        for (int e = tid; e < kPanelRows * c4; e += blockDim.x) {
            const int lr = e / c4, k = (e % c4) * 4;
            const int r = m0 + lr;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < P.m && k < n4) v = *reinterpret_cast<const float4*>(P.da + (size_t)r * P.ldda + k);
            *reinterpret_cast<float4*>(D + lr * ds + k) = v;
        }
    }
    lds_barrier();

    int prm_off = 0;
    for (int l = 0; l < P.nlayers; ++l) {
        const PanelBwdLayer& Lr = P.layer[l];
        const int n = Lr.n, ds = panel_stride(n), n8 = (n + 7) & ~7;
        const float* lgamma = prm + prm_off;
        const float* lbeta = lgamma + n;
        prm_off += 2 * n;
        // ---- row pass: D holds d(SiLU out); D <- dz, dgamma/dbeta column partials
        {
            constexpr int RPW = 4;  // launcher guarantees >= 8 waves
            float pg[kBwdMaxCols], pb[kBwdMaxCols];
#pragma unroll
            for (int u = 0; u < kBwdMaxCols; ++u) pg[u] = pb[u] = 0.f;
            float s1[RPW], s2[RPW], mean[RPW], rstd[RPW];
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                s1[i] = s2[i] = 0.f;
                mean[i] = 0.f;
                rstd[i] = 0.f;
                const int lr = wave + i * nwaves;
                const int row = m0 + lr;
                if (lr < kPanelRows && row < P.m) {
                    mean[i] = Lr.stats[(size_t)row * 2];
                    rstd[i] = Lr.stats[(size_t)row * 2 + 1];
                    const float* zr = Lr.z + (size_t)row * Lr.ldz;
#pragma unroll
                    for (int u = 0; u < kBwdMaxCols; ++u) {
                        const int c = lane + 64 * u;
                        if (c < n) {
                            const float xh = (zr[c] - mean[i]) * rstd[i];
                            const float g = lgamma[c];
                            const float dy = D[lr * ds + c] * silu_grad_p(g * xh + lbeta[c]);
                            const float dxh = dy * g;
                            s1[i] += dxh;
                            s2[i] += dxh * xh;
                            pg[u] += dy * xh;
                            pb[u] += dy;
                        }
                    }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    s1[i] += __shfl_xor(s1[i], o);
                    s2[i] += __shfl_xor(s2[i], o);
                }
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int lr = wave + i * nwaves;
                const int row = m0 + lr;
                if (lr < kPanelRows) {
                    const bool rv = row < P.m;
                    const float m1 = s1[i] / (float)n, m2 = s2[i] / (float)n;
                    const float* zr = Lr.z + (size_t)(rv ? row : 0) * Lr.ldz;
#pragma unroll
                    for (int u = 0; u < kBwdMaxCols; ++u) {
                        const int c = lane + 64 * u;
                        if (c < n8) {
                            float dzv = 0.f;
                            if (rv && c < n) {
                                const float xh = (zr[c] - mean[i]) * rstd[i];
                                const float g = lgamma[c];
                                const float dxh = D[lr * ds + c] * silu_grad_p(g * xh + lbeta[c]) * g;
                                dzv = rstd[i] * (dxh - m1 - xh * m2);
                                Lr.dz[(size_t)row * Lr.lddz + c] = dzv;
                            }
                            D[lr * ds + c] = dzv;
                        }
                    }
                }
            }
            float* cw = colp + (size_t)wave * 2 * n;
#pragma unroll
            for (int u = 0; u < kBwdMaxCols; ++u) {
                const int c = lane + 64 * u;
                if (c < n) {
                    cw[c] = pg[u];
                    cw[n + c] = pb[u];
                }
            }
        }
        lds_barrier();
        for (int c = tid; c < 2 * n; c += blockDim.x) {
            float t = 0.f;
            for (int w = 0; w < nwaves; ++w) t += colp[(size_t)w * 2 * n + c];
            Lr.part[(size_t)blockIdx.x * 2 * n + c] = t;
        }
        // ---- dX = dz * W: out tiles over k_in columns, contraction over n
        const int nout = Lr.k_in, nt = (nout + 31) >> 5;
        const int ks = panel_ksplit(n, nt, nwaves);
        const int kper = (((n8 + ks - 1) / ks) + 7) & ~7;
        const bool last = l + 1 == P.nlayers;
        const int es = panel_stride(nout), o8 = (nout + 7) & ~7;
        // more column tiles than waves: walk them in rounds (first layer's input can be wide)
        for (int t0 = 0; t0 < nt; t0 += nwaves / ks) {
            const int tiles_round = nwaves / ks;
            const int j = t0 + wave % tiles_round, s = wave / tiles_round;
            const bool active = s < ks && j < nt;
            const int kbeg = s * kper;
            const int kend = kbeg + kper < n8 ? kbeg + kper : n8;
            f32x16 acc;
            if (active) panel_gemm(acc, D, ds, n, Lr.wt, Lr.ldwt, nout, j, kbeg, kend, lane);
            // slice partials are indexed by the tile's slot in this round
            panel_ksum(acc, part, tiles_round, ks, wave % tiles_round, s, lane, active);
            if (active && s == 0) {
                const int col = j * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = tile_row(r, half);
                    const int row = m0 + lr;
                    const float v = col < nout ? acc[r] : 0.f;
                    if (last) {
                        if (col < nout && row < P.m) {
                            float* o = P.dx + (size_t)row * P.lddx + col;
                            *o = P.accumulate ? *o + v : v;
                        }
                    } else if (col < o8) {
                        E[lr * es + col] = v;
                    }
                }
            }
            lds_barrier();
        }
        float* tmp = D;
        D = E;
        E = tmp;
    }
}

int panel_bwd_blocks(int m) { return (int)cdiv(m, kPanelRows); }

int launch_panel_bwd(PanelBwdProb& p, hipStream_t st) {
    int waves = 8, pmax = 0, nmax = 0, prm = 0;
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
        if (l + 1 < p.nlayers && panel_stride(L.k_in) > pmax) pmax = panel_stride(L.k_in);
        prm += 2 * L.n;
    }
    if (waves > kPanelMaxWaves) waves = kPanelMaxWaves;
    p.off_e = kPanelRows * pmax;
    p.off_prm = 2 * kPanelRows * pmax;
    p.off_colp = p.off_prm + prm + 16;
    p.off_part = p.off_colp + waves * 2 * nmax;
    const size_t lds = (size_t)(p.off_part + (waves - 1) * 1024) * sizeof(float);
    if (lds > kPanelMaxLds) {
        set_error("panel backward: shape outside its range");
        return MARL_ELIMIT;
    }
    static bool raised = false;
    if (!raised) {
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(panel_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kPanelMaxLds));
        raised = true;
    }
    hipLaunchKernelGGL(panel_bwd_kernel, dim3((unsigned)cdiv(p.m, kPanelRows)), dim3(64 * waves), lds,
                       st, p);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

}  // namespace marl
