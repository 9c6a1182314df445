// Row-wise and elementwise kernels of the episode: LayerNorm/GroupNorm + SiLU (forward and
// backward), message mean, position embedding, LSTM-cell backward, policy sampling +
// bounded move, deterministic reductions, weight (un)packing and Adam.
// One 64-lane wave owns a row wherever a row reduction is needed.
#include <stdarg.h>
#include <stdlib.h>

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "sample.h"

namespace marl {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

// ---- tuning knobs ----------------------------------------------------------------------
namespace {
struct Knob {
    char key[32];
    int value;
    bool set;
};
constexpr int kMaxKnobs = 128;
Knob g_knobs[kMaxKnobs];
int g_nknobs = 0;
Knob* knob(const char* key, bool create) {
    for (int i = 0; i < g_nknobs; ++i)
        if (!strcmp(g_knobs[i].key, key)) return &g_knobs[i];
    if (!create || g_nknobs >= kMaxKnobs || strlen(key) >= sizeof(g_knobs[0].key)) return nullptr;
    Knob* k = &g_knobs[g_nknobs++];
    strcpy(k->key, key);
    k->value = 0;
    k->set = false;
    return k;
}
}  // namespace

int tune_set(const char* key, int value) {
    Knob* k = knob(key, true);
    if (!k) {
        set_error("marl_tune: knob table full or key too long (%s)", key);
        return MARL_EINVAL;
    }
    k->value = value;
    k->set = true;
    return MARL_OK;
}

int tune_get(const char* key, int dflt) {
    Knob* k = knob(key, true);
    if (k && k->set) return k->value;
    // first use: take the environment's value (MARL_<KEY>), remember the outcome
    char env[48] = "MARL_";
    size_t n = strlen(env);
    for (const char* c = key; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)toupper((unsigned char)*c);
    env[n] = 0;
    const char* e = getenv(env);
    const int v = e ? atoi(e) : dflt;
    if (k && e) {
        k->value = v;
        k->set = true;
    }
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float silu_f(float y) { return silu_fast(y); }
__device__ __forceinline__ float silu_grad(float y) { return silu_grad_fast(y); }

// ---------------------------------------------------------------------------
// LayerNorm (eps 1e-5, biased variance, affine) + SiLU     (Linear-LN-SiLU blocks of
// networks/message.py, policy.py, prediction.py, state.py)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_silu_fwd_kernel(const float* __restrict__ z, int ldz,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ out, int ldo,
                                                          float* __restrict__ stats, int64_t m,
                                                          int n) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const float* zr = z + row * ldz;
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += zr[c];
    const float mean = wave_sum(s) / (float)n;
    float q = 0.f;
    for (int c = lane; c < n; c += 64) {
        const float d = zr[c] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)n + 1e-5f);
    float* orow = out + row * ldo;
    for (int c = lane; c < n; c += 64) {
        const float y = (zr[c] - mean) * rstd * gamma[c] + beta[c];
        orow[c] = silu_f(y);
    }
    if (stats && lane == 0) {
        stats[row * 2] = mean;
        stats[row * 2 + 1] = rstd;
    }
}

// widths <= 64 * U: the row is read once and stays in registers for both statistics passes.
// A wave owns TWO rows (row and row + half): both rows' loads are in flight together and the
// affine parameters are fetched once.
template <int U>
__global__ __launch_bounds__(256) void ln_silu_fwd_reg_kernel(const float* __restrict__ z, int ldz,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              float* __restrict__ out, int ldo,
                                                              float* __restrict__ stats, int64_t m,
                                                              int n, const float* __restrict__ wdot,
                                                              const float* __restrict__ bdot,
                                                              float* __restrict__ dot_out, int64_t half) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row0 >= half) return;
    const int64_t rows[2] = {row0, row0 + half};
    const bool ok1 = rows[1] < m;
    float v[2][U], gm[U], bt[U], wd[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = lane + 64 * u;
        v[0][u] = c < n ? z[rows[0] * ldz + c] : 0.f;
        v[1][u] = (c < n && ok1) ? z[rows[1] * ldz + c] : 0.f;
        gm[u] = c < n ? gamma[c] : 0.f;
        bt[u] = c < n ? beta[c] : 0.f;
        wd[u] = (c < n && wdot) ? wdot[c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k == 1 && !ok1) break;
        const int64_t row = rows[k];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[k][u];
        const float mean = wave_sum(s) / (float)n;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float d = (lane + 64 * u < n) ? v[k][u] - mean : 0.f;
            v[k][u] = d;
            q += d * d;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)n + 1e-5f);
        float* orow = out + row * ldo;
        float dot = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = lane + 64 * u;
            if (c < n) {
                const float a = silu_f(v[k][u] * rstd * gm[u] + bt[u]);
                orow[c] = a;
                dot += a * wd[u];
            }
        }
        if (stats && lane == 0) {
            stats[row * 2] = mean;
            stats[row * 2 + 1] = rstd;
        }
        if (wdot) {  // a one-output layer on top (critic value) comes for free from the registers
            dot = wave_sum(dot);
            if (lane == 0) dot_out[row] = dot + bdot[0];
        }
    }
}

int launch_ln_silu_fwd(const float* z, int ldz, const float* gamma, const float* beta, float* out,
                       int ldo, float* stats, int64_t m, int n, hipStream_t st, const float* wdot,
                       const float* bdot, float* dot_out) {
    if (m <= 0) return MARL_OK;
    const int64_t half = (m + 1) / 2;  // the register kernels: two rows per wave
    dim3 grid((unsigned)cdiv(n <= 384 ? half : m, 4)), blk(256);
    if (wdot && n > 384) {
        set_error("fused row dot needs a LayerNorm width <= 384");
        return MARL_ELIMIT;
    }
    if (n <= 128)
        hipLaunchKernelGGL(ln_silu_fwd_reg_kernel<2>, grid, blk, 0, st, z, ldz, gamma, beta, out, ldo,
                           stats, m, n, wdot, bdot, dot_out, half);
    else if (n <= 384)
        hipLaunchKernelGGL(ln_silu_fwd_reg_kernel<6>, grid, blk, 0, st, z, ldz, gamma, beta, out, ldo,
                           stats, m, n, wdot, bdot, dot_out, half);
    else
        hipLaunchKernelGGL(ln_silu_fwd_kernel, grid, blk, 0, st, z, ldz, gamma, beta, out, ldo, stats,
                           m, n);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// Backward launches use wide workgroups (up to 16 waves, one row at a time per wave) so that
// thousands of waves are in flight while only ~256 partial dgamma/dbeta slabs are produced.
static int bwd_waves(int n) { return n <= 512 ? 16 : (n <= 1024 ? 8 : 4); }
static int bwd_rows_per_wave(int64_t m, int waves) {
    int64_t r = cdiv(m, (int64_t)waves * 256);
    if (r < 1) r = 1;
    if (r > 64) r = 64;
    return (int)r;
}
int ln_bwd_blocks(int64_t m, int n) {
    const int w = bwd_waves(n);
    return (int)cdiv(m, (int64_t)w * bwd_rows_per_wave(m, w));
}

// part[blk][0][n] = sum_rows dy * xhat (dgamma), part[blk][1][n] = sum_rows dy (dbeta)
__global__ __launch_bounds__(1024) void ln_silu_bwd_kernel(
    const float* __restrict__ da, int ldda, const float* __restrict__ z, int ldz,
    const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ dz, int lddz, float* __restrict__ part,
    int64_t m, int n, int rpw) {
    extern __shared__ __attribute__((aligned(16))) float sacc[];  // [waves][2][n]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    float* ga = sacc + (size_t)wave * 2 * n;
    float* gb = ga + n;
    for (int c = lane; c < n; c += 64) {
        ga[c] = 0.f;
        gb[c] = 0.f;
    }
    for (int rr = 0; rr < rpw; ++rr) {
        const int64_t row = ((int64_t)blockIdx.x * nwaves + wave) * rpw + rr;
        if (row >= m) break;
        const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
        const float* zr = z + row * ldz;
        const float* dar = da + row * ldda;
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < n; c += 64) {
            const float xh = (zr[c] - mean) * rstd;
            const float g = gamma[c];
            const float dy = dar[c] * silu_grad(g * xh + beta[c]);
            const float dxh = dy * g;
            s1 += dxh;
            s2 += dxh * xh;
            ga[c] += dy * xh;
            gb[c] += dy;
        }
        const float m1 = wave_sum(s1) / (float)n;
        const float m2 = wave_sum(s2) / (float)n;
        float* dzr = dz + row * lddz;
        for (int c = lane; c < n; c += 64) {
            const float xh = (zr[c] - mean) * rstd;
            const float g = gamma[c];
            const float dxh = dar[c] * silu_grad(g * xh + beta[c]) * g;
            dzr[c] = rstd * (dxh - m1 - xh * m2);
        }
    }
    __syncthreads();
    float* p = part + (size_t)blockIdx.x * 2 * n;
    for (int c = threadIdx.x; c < 2 * n; c += blockDim.x) {
        float t = 0.f;
        for (int w = 0; w < nwaves; ++w) t += sacc[(size_t)w * 2 * n + c];
        p[c] = t;
    }
}

// Register-resident variant for widths <= 64 * U: z and da are read ONCE (the row lives in
// registers between the statistics and the dz pass), the affine partial sums stay in registers
// over the wave's rows.  With kin > 0 the incoming gradient is not read at all but formed on
// the fly as da[r][c] = sum_{j < kin} g[r][j] * bt[c][j] (the layer above has only kin <= 4
// outputs - critic value, policy logits - so its dX GEMM is a rank-kin update).
template <int U>
__global__ __launch_bounds__(1024) void ln_silu_bwd_reg_kernel(
    const float* __restrict__ da, int ldda, const float* __restrict__ g, int ldg, int kin,
    const float* __restrict__ bt, int ldbt, const float* __restrict__ z, int ldz,
    const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ dz, int lddz, float* __restrict__ part,
    int64_t m, int n, int rpw) {
    extern __shared__ __attribute__((aligned(16))) float sacc[];  // [waves][2][n]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    float gam[U], bet[U], pga[U], pgb[U], bw[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = lane + 64 * u;
        gam[u] = c < n ? gamma[c] : 0.f;
        bet[u] = c < n ? beta[c] : 0.f;
        pga[u] = pgb[u] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) bw[u][j] = (c < n && j < kin) ? bt[(size_t)c * ldbt + j] : 0.f;
    }
    const float inv_n = 1.0f / (float)n;
    // The loads of row rr + 1 are issued before row rr is processed: a wave keeps two rows in
    // flight (the pass is HBM-latency bound - one row at a time leaves the memory pipe idle during
    // the statistics and the dependent stores).
    const int64_t row_base = ((int64_t)blockIdx.x * nwaves + wave) * rpw;
    float zc[U], dc[U], gvc[4], mean_c = 0.f, rstd_c = 0.f;
    auto fetch = [&](int64_t row, float (&zv)[U], float (&dv)[U], float (&gv)[4], float& mean, float& rstd) {
        const bool ok = row < m;
        const int64_t r = ok ? row : 0;
        mean = stats[r * 2];
        rstd = stats[r * 2 + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) gv[j] = j < kin ? g[r * ldg + j] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = lane + 64 * u;
            zv[u] = c < n ? z[r * ldz + c] : 0.f;
            dv[u] = (kin == 0 && c < n) ? da[r * ldda + c] : 0.f;
        }
    };
    constexpr bool PF = U >= 2;  // (narrow rows: measured slower with the prefetch)
    if (PF && row_base < m) fetch(row_base, zc, dc, gvc, mean_c, rstd_c);
    for (int rr = 0; rr < rpw; ++rr) {
        const int64_t row = row_base + rr;
        if (row >= m) break;
        if (!PF) fetch(row, zc, dc, gvc, mean_c, rstd_c);
        float zn[U], dn[U], gvn[4], mean_n = 0.f, rstd_n = 0.f;
        const bool more = PF && rr + 1 < rpw && row + 1 < m;
        if (more) fetch(row + 1, zn, dn, gvn, mean_n, rstd_n);
        const float mean = mean_c, rstd = rstd_c;
        float xh[U], dxh[U];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = lane + 64 * u;
            xh[u] = dxh[u] = 0.f;
            if (c < n) {
                float dav;
                if (kin > 0)
                    dav = ((gvc[0] * bw[u][0] + gvc[1] * bw[u][1]) + gvc[2] * bw[u][2]) + gvc[3] * bw[u][3];
                else
                    dav = dc[u];
                xh[u] = (zc[u] - mean) * rstd;
                const float dy = dav * silu_grad(gam[u] * xh[u] + bet[u]);
                dxh[u] = dy * gam[u];
                s1 += dxh[u];
                s2 += dxh[u] * xh[u];
                pga[u] += dy * xh[u];
                pgb[u] += dy;
            }
        }
        const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
        float* dzr = dz + row * lddz;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = lane + 64 * u;
            if (c < n) dzr[c] = rstd * (dxh[u] - m1 - xh[u] * m2);
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                zc[u] = zn[u];
                dc[u] = dn[u];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) gvc[j] = gvn[j];
            mean_c = mean_n;
            rstd_c = rstd_n;
        }
    }
    float* ga = sacc + (size_t)wave * 2 * n;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = lane + 64 * u;
        if (c < n) {
            ga[c] = pga[u];
            ga[n + c] = pgb[u];
        }
    }
    __syncthreads();
    float* p = part + (size_t)blockIdx.x * 2 * n;
    for (int c = threadIdx.x; c < 2 * n; c += blockDim.x) {
        float t = 0.f;
        for (int w = 0; w < nwaves; ++w) t += sacc[(size_t)w * 2 * n + c];
        p[c] = t;
    }
}

// da == nullptr: rank-kin source (g, bt), see the kernel
static int launch_ln_silu_bwd_any(const float* da, int ldda, const float* g, int ldg, int kin,
                                  const float* bt, int ldbt, const float* z, int ldz,
                                  const float* stats, const float* gamma, const float* beta,
                                  float* dz, int lddz, float* part, int64_t m, int n,
                                  hipStream_t st) {
    const int w = bwd_waves(n);
    const int rpw = bwd_rows_per_wave(m, w);
    const dim3 grid((unsigned)ln_bwd_blocks(m, n)), blk(64 * w);
    const size_t lds = (size_t)w * 2 * n * sizeof(float);
#define MARL_LN_REG(U_)                                                                         \
    hipLaunchKernelGGL(ln_silu_bwd_reg_kernel<U_>, grid, blk, lds, st, da, ldda, g, ldg, kin, bt, \
                       ldbt, z, ldz, stats, gamma, beta, dz, lddz, part, m, n, rpw)
    if (n <= 64)
        MARL_LN_REG(1);
    else if (n <= 128)
        MARL_LN_REG(2);
    else if (n <= 256)
        MARL_LN_REG(4);
    else
        MARL_LN_REG(6);
#undef MARL_LN_REG
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

int launch_ln_silu_bwd_rank(const float* g, int ldg, int kin, const float* bt, int ldbt,
                            const float* z, int ldz, const float* stats, const float* gamma,
                            const float* beta, float* dz, int lddz, float* part, int64_t m, int n,
                            hipStream_t st) {
    if (m <= 0) return MARL_OK;
    if (kin < 1 || kin > 4 || n > 384) {
        set_error("rank-k LayerNorm backward: kin=%d n=%d outside its range", kin, n);
        return MARL_ELIMIT;
    }
    return launch_ln_silu_bwd_any(nullptr, 0, g, ldg, kin, bt, ldbt, z, ldz, stats, gamma, beta, dz,
                                  lddz, part, m, n, st);
}

int launch_ln_silu_bwd(const float* da, int ldda, const float* z, int ldz, const float* stats,
                       const float* gamma, const float* beta, float* dz, int lddz, float* part,
                       int64_t m, int n, hipStream_t st) {
    if (m <= 0) return MARL_OK;
    if (n > 2048) {
        set_error("LayerNorm width %d > 2048 unsupported", n);
        return MARL_ELIMIT;
    }
    if (n <= 384)
        return launch_ln_silu_bwd_any(da, ldda, nullptr, 0, 0, nullptr, 0, z, ldz, stats, gamma, beta,
                                      dz, lddz, part, m, n, st);
    const int w = bwd_waves(n);
    const int rpw = bwd_rows_per_wave(m, w);
    hipLaunchKernelGGL(ln_silu_bwd_kernel, dim3((unsigned)ln_bwd_blocks(m, n)), dim3(64 * w),
                       (size_t)w * 2 * n * sizeof(float), st, da, ldda, z, ldz, stats, gamma, beta,
                       dz, lddz, part, m, n, rpw);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// stage 1 of a long partial reduction: parts [y * chunk, (y + 1) * chunk) -> row y * chunk, in
// place (each block only touches its own 64 columns of its own chunk; fixed order).
__global__ __launch_bounds__(256) void reduce_chunks_kernel(float* __restrict__ part, int64_t nparts,
                                                            int64_t stride, int ncols, int chunk) {
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int64_t p0 = (int64_t)blockIdx.y * chunk;
    int64_t p1 = p0 + chunk;
    if (p1 > nparts) p1 = nparts;
    float s = 0.f;
    if (c < ncols) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int64_t p = p0 + grp;
        for (; p + 12 < p1; p += 16) {
            s0 += part[p * stride + c];
            s1 += part[(p + 4) * stride + c];
            s2 += part[(p + 8) * stride + c];
            s3 += part[(p + 12) * stride + c];
        }
        for (; p < p1; p += 4) s0 += part[p * stride + c];
        s = (s0 + s1) + (s2 + s3);
    }
    sh[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && c < ncols)
        part[p0 * stride + c] = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
}

// dgamma / dbeta (+)= sum over parts p * pstep, p < nparts, of part[..][2n]
__global__ __launch_bounds__(256) void reduce_affine_kernel(const float* __restrict__ part,
                                                            int64_t nparts, int64_t pstep, int n,
                                                            float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta,
                                                            int accumulate) {
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;  // column in [0, 2n): gamma then beta
    const int64_t rs = pstep * 2 * n;
    float s = 0.f;
    if (c < 2 * n) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int64_t p = grp;
        for (; p + 12 < nparts; p += 16) {
            s0 += part[p * rs + c];
            s1 += part[(p + 4) * rs + c];
            s2 += part[(p + 8) * rs + c];
            s3 += part[(p + 12) * rs + c];
        }
        for (; p < nparts; p += 4) s0 += part[p * rs + c];
        s = (s0 + s1) + (s2 + s3);
    }
    sh[grp][lane] = s;
    __syncthreads();
    if (grp == 0 && c < 2 * n) {
        float t = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
        float* o = c < n ? dgamma + c : dbeta + (c - n);
        if (accumulate) t += *o;
        *o = t;
    }
}

// NOTE: long reductions (>= 256 parts) fold the partial buffer in place first - `part` is
// consumed by this call.
int launch_reduce_affine(float* part, int64_t nparts, int n, float* dgamma, float* dbeta,
                         int accumulate, hipStream_t st, RedQueue* q) {
    if (n <= 0) return MARL_OK;
    if (q && !accumulate) {  // queued: `part` must stay untouched until the queue is flushed
        q->push(part, (int64_t)2 * n, (int)nparts, 2 * n, dgamma, n, n, n, dbeta, 0);
        return q->rc;
    }
    int64_t pstep = 1;
    if (nparts >= 256) {
        int chunk = 16;
        while ((int64_t)chunk * chunk < nparts) chunk <<= 1;  // ~sqrt split: 4096 -> 64 x 64
        hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)cdiv(2 * n, 64), (unsigned)cdiv(nparts, chunk)),
                           dim3(256), 0, st, part, nparts, (int64_t)2 * n, 2 * n, chunk);
        MARL_LAUNCH_CHECK();
        pstep = chunk;
        nparts = cdiv(nparts, chunk);
    }
    hipLaunchKernelGGL(reduce_affine_kernel, dim3((unsigned)cdiv(2 * n, 64)), dim3(256), 0, st,
                       part, nparts, pstep, n, dgamma, dbeta, accumulate);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// GroupNorm (eps 1e-5, biased variance over (C/G) * P) + SiLU on NHWC rows
// (networks/vision.py:33-38)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_silu_fwd_kernel(const float* __restrict__ z,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ out, int64_t ldo,
                                                          int out_chw, float* __restrict__ stats,
                                                          int64_t rows, int P, int C, int G) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int Cg = C / G;
    const int cnt = P * Cg;
    const float* zr = z + row * (int64_t)P * C;
    float* orow = out + row * ldo;
    for (int g = 0; g < G; ++g) {
        float s = 0.f;
        for (int i = lane; i < cnt; i += 64) s += zr[(i / Cg) * C + g * Cg + (i % Cg)];
        const float mean = wave_sum(s) / (float)cnt;
        float q = 0.f;
        for (int i = lane; i < cnt; i += 64) {
            const float d = zr[(i / Cg) * C + g * Cg + (i % Cg)] - mean;
            q += d * d;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)cnt + 1e-5f);
        for (int i = lane; i < cnt; i += 64) {
            const int pos = i / Cg, c = g * Cg + (i % Cg);
            const float y = (zr[pos * C + c] - mean) * rstd * gamma[c] + beta[c];
            orow[out_chw ? (int64_t)c * P + pos : (int64_t)pos * C + c] = silu_f(y);
        }
        if (stats && lane == 0) {
            stats[(row * G + g) * 2] = mean;
            stats[(row * G + g) * 2 + 1] = rstd;
        }
    }
}

// Flat-order forward for power-of-two channel counts <= 64 and rows of <= 64 * EPL
// elements: the row is read ONCE into registers (coalesced), group statistics come from
// xor-shuffles (lane bits inside the group x position bits), two-pass variance.
template <int EPL>
__global__ __launch_bounds__(256) void gn_silu_fwd_flat_kernel(
    const float* __restrict__ z, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ out, int64_t ldo, int out_chw, float* __restrict__ stats, int64_t rows,
    int P, int C, int G, float* __restrict__ cols, int ldk, int hin) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int Cg = C / G, E = P * C;
    const int c = lane % C, g = c / Cg;
    const float* zr = z + row * (int64_t)E;
    float v[EPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int e = lane + 64 * i;
        v[i] = e < E ? zr[e] : 0.f;
        s += v[i];
    }
    for (int o = 1; o < Cg; o <<= 1) s += __shfl_xor(s, o);
    for (int o = C; o < 64; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)(P * Cg);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const float d = v[i] - mean;
        q += (lane + 64 * i < E) ? d * d : 0.f;
    }
    for (int o = 1; o < Cg; o <<= 1) q += __shfl_xor(q, o);
    for (int o = C; o < 64; o <<= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)(P * Cg) + 1e-5f);
    const float gm = gamma[c], bt = beta[c];
    float* orow = out + row * ldo;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int e = lane + 64 * i;
        if (e < E) {
            const float y = (v[i] - mean) * rstd * gm + bt;
            const float av = silu_f(y);
            if (out) orow[out_chw ? (int64_t)c * P + e / C : (int64_t)e] = av;
            if (cols) {
                // the next layer's im2col rows (3x3, stride 2, pad 1) straight from registers:
                // input pixel (py, px) feeds output (oy, ox) through tap (kh, kw) when
                // 2*oy - 1 + kh == py.  Padding taps are never written (the buffer is zero).
                const int pos = e / C, py = pos / hin, px = pos % hin;
                const int hout = (hin - 1) / 2 + 1;
                float* crow = cols + row * (int64_t)hout * hout * ldk + c;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int ty = py + 1 - kh;
                    if (ty < 0 || (ty & 1) || (ty >> 1) >= hout) continue;
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int tx = px + 1 - kw;
                        if (tx < 0 || (tx & 1) || (tx >> 1) >= hout) continue;
                        crow[(int64_t)((ty >> 1) * hout + (tx >> 1)) * ldk + (kh * 3 + kw) * C] = av;
                    }
                }
            }
        }
    }
    if (stats && lane < C && (lane % Cg) == 0) {
        stats[(row * G + g) * 2] = mean;
        stats[(row * G + g) * 2 + 1] = rstd;
    }
}

static bool gn_flat_ok(int P, int C) {
    return (C & (C - 1)) == 0 && C >= 4 && C <= 64 && P * C <= 64 * 16;
}
int gn_fwd_im2col_supported(int P, int C) { return gn_flat_ok(P, C) ? 1 : 0; }

int launch_gn_silu_fwd(const float* z, const float* gamma, const float* beta, float* out,
                       int64_t ldo, int out_chw, float* stats, int64_t rows, int P, int C, int G,
                       hipStream_t st, float* cols, int ldk, int hin) {
    if (rows <= 0) return MARL_OK;
    const int E = P * C;
    if (cols && !gn_flat_ok(P, C)) {
        set_error("fused GroupNorm + im2col needs a power-of-two channel count <= 64");
        return MARL_ELIMIT;
    }
    if (gn_flat_ok(P, C)) {
        const dim3 grid((unsigned)cdiv(rows, 4)), blk(256);
        if (E <= 64 * 4)
            hipLaunchKernelGGL(gn_silu_fwd_flat_kernel<4>, grid, blk, 0, st, z, gamma, beta, out, ldo,
                               out_chw, stats, rows, P, C, G, cols, ldk, hin);
        else if (E <= 64 * 9)
            hipLaunchKernelGGL(gn_silu_fwd_flat_kernel<9>, grid, blk, 0, st, z, gamma, beta, out, ldo,
                               out_chw, stats, rows, P, C, G, cols, ldk, hin);
        else
            hipLaunchKernelGGL(gn_silu_fwd_flat_kernel<16>, grid, blk, 0, st, z, gamma, beta, out,
                               ldo, out_chw, stats, rows, P, C, G, cols, ldk, hin);
        MARL_LAUNCH_CHECK();
        return MARL_OK;
    }
    hipLaunchKernelGGL(gn_silu_fwd_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st, z,
                       gamma, beta, out, ldo, out_chw, stats, rows, P, C, G);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

int gn_bwd_blocks(int64_t rows, int C) {
    const int w = bwd_waves(C);
    return (int)cdiv(rows, (int64_t)w * bwd_rows_per_wave(rows, w));
}

// Lane (cc, pslot) owns channel g*Cg+cc and walks positions pslot, pslot+64/Cg, ... so the
// per-channel dgamma/dbeta sums have a single owner (no LDS races, fixed order).
__global__ __launch_bounds__(1024) void gn_silu_bwd_kernel(
    const float* __restrict__ da, int64_t ldda, int da_chw, const float* __restrict__ z,
    const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ dz, float* __restrict__ part,
    int64_t rows, int P, int C, int G, int rpw) {
    extern __shared__ __attribute__((aligned(16))) float sacc[];  // [waves][2][C]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    float* ga = sacc + (size_t)wave * 2 * C;
    float* gb = ga + C;
    for (int c = lane; c < C; c += 64) {
        ga[c] = 0.f;
        gb[c] = 0.f;
    }
    const int Cg = C / G;
    const int cc = lane % Cg, pslot = lane / Cg, pstep = 64 / Cg;
    const float inv_cnt = 1.0f / (float)(P * Cg);
    for (int rr = 0; rr < rpw; ++rr) {
        const int64_t row = ((int64_t)blockIdx.x * nwaves + wave) * rpw + rr;
        if (row >= rows) break;
        const float* zr = z + row * (int64_t)P * C;
        const float* dar = da + row * ldda;
        float* dzr = dz + row * (int64_t)P * C;
        for (int g = 0; g < G; ++g) {
            const float mean = stats[(row * G + g) * 2], rstd = stats[(row * G + g) * 2 + 1];
            const int c = g * Cg + cc;
            const float gm = gamma[c], bt = beta[c];
            float s1 = 0.f, s2 = 0.f, pg = 0.f, pb = 0.f;
            for (int pos = pslot; pos < P; pos += pstep) {
                const float xh = (zr[pos * C + c] - mean) * rstd;
                const float dav = dar[da_chw ? (int64_t)c * P + pos : (int64_t)pos * C + c];
                const float dy = dav * silu_grad(gm * xh + bt);
                const float dxh = dy * gm;
                s1 += dxh;
                s2 += dxh * xh;
                pg += dy * xh;
                pb += dy;
            }
            const float m1 = wave_sum(s1) * inv_cnt;
            const float m2 = wave_sum(s2) * inv_cnt;
            for (int o = Cg; o < 64; o <<= 1) {
                pg += __shfl_xor(pg, o);
                pb += __shfl_xor(pb, o);
            }
            if (pslot == 0) {
                ga[c] += pg;
                gb[c] += pb;
            }
            for (int pos = pslot; pos < P; pos += pstep) {
                const float xh = (zr[pos * C + c] - mean) * rstd;
                const float dav = dar[da_chw ? (int64_t)c * P + pos : (int64_t)pos * C + c];
                const float dxh = dav * silu_grad(gm * xh + bt) * gm;
                dzr[pos * C + c] = rstd * (dxh - m1 - xh * m2);
            }
        }
    }
    __syncthreads();
    float* p = part + (size_t)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
        float t = 0.f;
        for (int w = 0; w < nwaves; ++w) t += sacc[(size_t)w * 2 * C + c];
        p[c] = t;
    }
}

// Coalesced variant for power-of-two channel counts: a wave walks the NHWC row in flat
// order (64 consecutive floats per load), so every lane keeps ONE channel (C <= 64) or the
// channels lane + 64u (C = 128, 256).  Group sums come from xor-shuffles over the lane bits
// that stay inside a group, channel sums (dgamma / dbeta) over the position bits.
template <int U>  // channels per lane = max(1, C / 64)
__global__ __launch_bounds__(1024) void gn_silu_bwd_flat_kernel(
    const float* __restrict__ da, int64_t ldda, int da_chw, const float* __restrict__ z,
    const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ dz, float* __restrict__ part,
    int64_t rows, int P, int C, int G, int rpw) {
    extern __shared__ __attribute__((aligned(16))) float sacc[];  // [waves][2][C]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    float* ga = sacc + (size_t)wave * 2 * C;
    float* gb = ga + C;
    for (int c = lane; c < C; c += 64) {
        ga[c] = 0.f;
        gb[c] = 0.f;
    }
    const int Cg = C / G;
    const int E = P * C;
    const int lc = C < 64 ? C : 64;  // lanes per position
    const float inv_cnt = 1.0f / (float)(P * Cg);
    float gm[U], bt[U];
    int ch[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        ch[u] = (lane % lc) + 64 * u;
        gm[u] = gamma[ch[u]];
        bt[u] = beta[ch[u]];
    }
    for (int rr = 0; rr < rpw; ++rr) {
        const int64_t row = ((int64_t)blockIdx.x * nwaves + wave) * rpw + rr;
        if (row >= rows) break;
        const float* zr = z + row * (int64_t)E;
        const float* dar = da + row * ldda;
        float* dzr = dz + row * (int64_t)E;
        float mean[U], rstd[U], s1[U], s2[U], pg[U], pb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = ch[u] / Cg;
            mean[u] = stats[(row * G + g) * 2];
            rstd[u] = stats[(row * G + g) * 2 + 1];
            s1[u] = s2[u] = pg[u] = pb[u] = 0.f;
        }
        // element e = lane + 64 * i: position e / C, channel e % C (fixed per lane and u)
        for (int e0 = 0; e0 < E; e0 += 64 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + lane + 64 * u;
                if (e < E) {
                    const int pos = e / C;
                    const float xh = (zr[e] - mean[u]) * rstd[u];
                    const float dav = dar[da_chw ? (int64_t)ch[u] * P + pos : (int64_t)e];
                    const float dy = dav * silu_grad(gm[u] * xh + bt[u]);
                    const float dxh = dy * gm[u];
                    s1[u] += dxh;
                    s2[u] += dxh * xh;
                    pg[u] += dy * xh;
                    pb[u] += dy;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // group sums: lanes of the same group = same bits above log2(Cg) within the
            // position (when Cg <= 64) x all position bits
            for (int o = 1; o < Cg && o < 64; o <<= 1) {
                s1[u] += __shfl_xor(s1[u], o);
                s2[u] += __shfl_xor(s2[u], o);
            }
            for (int o = lc; o < 64; o <<= 1) {
                s1[u] += __shfl_xor(s1[u], o);
                s2[u] += __shfl_xor(s2[u], o);
                pg[u] += __shfl_xor(pg[u], o);
                pb[u] += __shfl_xor(pb[u], o);
            }
            if (lane < lc) {  // single owner per channel
                ga[ch[u]] += pg[u];
                gb[ch[u]] += pb[u];
            }
        }
        for (int e0 = 0; e0 < E; e0 += 64 * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + lane + 64 * u;
                if (e < E) {
                    const int pos = e / C;
                    const float xh = (zr[e] - mean[u]) * rstd[u];
                    const float dav = dar[da_chw ? (int64_t)ch[u] * P + pos : (int64_t)e];
                    const float dxh = dav * silu_grad(gm[u] * xh + bt[u]) * gm[u];
                    dzr[e] = rstd[u] * (dxh - s1[u] * inv_cnt - xh * s2[u] * inv_cnt);
                }
            }
        }
    }
    __syncthreads();
    float* p = part + (size_t)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
        float t = 0.f;
        for (int w = 0; w < nwaves; ++w) t += sacc[(size_t)w * 2 * C + c];
        p[c] = t;
    }
}

// Register-resident form of the flat kernel for rows of exactly 64 * U * NE elements (NE <= 4):
// z and da are read ONCE (the row lives in registers between the statistics and the dz pass),
// the next row's loads are issued before the current row is processed, and the per-channel
// dgamma / dbeta sums stay in registers over all rows of the wave (one cross-lane reduction at
// the end instead of one per row).
template <int U, int NE>
__global__ __launch_bounds__(1024) void gn_silu_bwd_rowreg_kernel(
    const float* __restrict__ da, int64_t ldda, int da_chw, const float* __restrict__ z,
    const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ dz, float* __restrict__ part,
    int64_t rows, int P, int C, int G, int rpw) {
    extern __shared__ __attribute__((aligned(16))) float sacc[];  // [waves][2][C]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int Cg = C / G;
    const int E = P * C;
    const int lc = C < 64 ? C : 64;  // lanes per position
    const float inv_cnt = 1.0f / (float)(P * Cg);
    float gm[U], bt[U], pga[U], pgb[U];
    int ch[U], grp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        ch[u] = (lane % lc) + 64 * u;
        grp[u] = ch[u] / Cg;
        gm[u] = gamma[ch[u]];
        bt[u] = beta[ch[u]];
        pga[u] = pgb[u] = 0.f;
    }
    // element (i, u) of a lane: e = 64 * U * i + lane + 64 * u -> position e / C, channel ch[u]
    struct Row {
        float zv[NE][U], dv[NE][U], mean[U], rstd[U];
    };
    auto fetch = [&](int64_t row, Row& r) {
        const int64_t rr = row < rows ? row : 0;
        const float* zr = z + rr * (int64_t)E;
        const float* dar = da + rr * ldda;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r.mean[u] = stats[(rr * G + grp[u]) * 2];
            r.rstd[u] = stats[(rr * G + grp[u]) * 2 + 1];
        }
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = 64 * U * i + lane + 64 * u;
                r.zv[i][u] = zr[e];
                r.dv[i][u] = dar[da_chw ? (int64_t)ch[u] * P + e / C : (int64_t)e];
            }
    };
    const int64_t row_base = ((int64_t)blockIdx.x * nwaves + wave) * rpw;
    Row cur, nxt;
    if (row_base < rows) fetch(row_base, cur);
    for (int rr = 0; rr < rpw; ++rr) {
        const int64_t row = row_base + rr;
        if (row >= rows) break;
        const bool more = rr + 1 < rpw && row + 1 < rows;
        if (more) fetch(row + 1, nxt);
        float xh[NE][U], dxh[NE][U], s1[U], s2[U], pg[U], pb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) s1[u] = s2[u] = pg[u] = pb[u] = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xh[i][u] = (cur.zv[i][u] - cur.mean[u]) * cur.rstd[u];
                const float dy = cur.dv[i][u] * silu_grad(gm[u] * xh[i][u] + bt[u]);
                dxh[i][u] = dy * gm[u];
                s1[u] += dxh[i][u];
                s2[u] += dxh[i][u] * xh[i][u];
                pg[u] += dy * xh[i][u];
                pb[u] += dy;
            }
        float* dzr = dz + row * (int64_t)E;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // group sums: lanes of the same group (bits below log2(Cg)) x all position bits
            for (int o = 1; o < Cg && o < 64; o <<= 1) {
                s1[u] += __shfl_xor(s1[u], o);
                s2[u] += __shfl_xor(s2[u], o);
            }
            for (int o = lc; o < 64; o <<= 1) {
                s1[u] += __shfl_xor(s1[u], o);
                s2[u] += __shfl_xor(s2[u], o);
            }
            pga[u] += pg[u];
            pgb[u] += pb[u];
#pragma unroll
            for (int i = 0; i < NE; ++i)
                dzr[64 * U * i + lane + 64 * u] =
                    cur.rstd[u] * (dxh[i][u] - s1[u] * inv_cnt - xh[i][u] * s2[u] * inv_cnt);
        }
        if (more) cur = nxt;
    }
    float* ga = sacc + (size_t)wave * 2 * C;
    float* gb = ga + C;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        for (int o = lc; o < 64; o <<= 1) {  // lanes of the same channel at other positions
            pga[u] += __shfl_xor(pga[u], o);
            pgb[u] += __shfl_xor(pgb[u], o);
        }
        if (lane < lc) {  // single owner per channel
            ga[ch[u]] = pga[u];
            gb[ch[u]] = pgb[u];
        }
    }
    __syncthreads();
    float* p = part + (size_t)blockIdx.x * 2 * C;
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
        float t = 0.f;
        for (int w = 0; w < nwaves; ++w) t += sacc[(size_t)w * 2 * C + c];
        p[c] = t;
    }
}

int launch_gn_silu_bwd(const float* da, int64_t ldda, int da_chw, const float* z,
                       const float* stats, const float* gamma, const float* beta, float* dz,
                       float* part, int64_t rows, int P, int C, int G, hipStream_t st) {
    if (rows <= 0) return MARL_OK;
    const int Cg = C / G;
    if (Cg > 64 || (Cg & (Cg - 1)) != 0 || C > 2048) {
        set_error("GroupNorm backward needs channels/group a power of two <= 64 (got %d)", Cg);
        return MARL_ELIMIT;
    }
    const int w = bwd_waves(C);
    const int rpw = bwd_rows_per_wave(rows, w);
    const bool pow2 = (C & (C - 1)) == 0 && C <= 256 && C >= 4;
    if (pow2) {
        const dim3 grid((unsigned)gn_bwd_blocks(rows, C)), blk(64 * w);
        const size_t lds = (size_t)w * 2 * C * sizeof(float);
        const int U = C <= 64 ? 1 : C / 64, E = P * C;
        if (E % (64 * U) == 0 && E / (64 * U) <= 4) {
            const int ne = E / (64 * U);
#define MARL_GN_RR(U_, NE_)                                                                        \
    hipLaunchKernelGGL((gn_silu_bwd_rowreg_kernel<U_, NE_>), grid, blk, lds, st, da, ldda, da_chw, z, \
                       stats, gamma, beta, dz, part, rows, P, C, G, rpw)
#define MARL_GN_RRU(U_)        \
    if (ne == 1)               \
        MARL_GN_RR(U_, 1);     \
    else if (ne == 2)          \
        MARL_GN_RR(U_, 2);     \
    else if (ne == 3)          \
        MARL_GN_RR(U_, 3);     \
    else                       \
        MARL_GN_RR(U_, 4)
            if (U == 1) {
                MARL_GN_RRU(1);
            } else if (U == 2) {
                MARL_GN_RRU(2);
            } else {
                MARL_GN_RRU(4);
            }
#undef MARL_GN_RRU
#undef MARL_GN_RR
            MARL_LAUNCH_CHECK();
            return MARL_OK;
        }
        if (C <= 64)
            hipLaunchKernelGGL(gn_silu_bwd_flat_kernel<1>, grid, blk, lds, st, da, ldda, da_chw, z,
                               stats, gamma, beta, dz, part, rows, P, C, G, rpw);
        else if (C == 128)
            hipLaunchKernelGGL(gn_silu_bwd_flat_kernel<2>, grid, blk, lds, st, da, ldda, da_chw, z,
                               stats, gamma, beta, dz, part, rows, P, C, G, rpw);
        else
            hipLaunchKernelGGL(gn_silu_bwd_flat_kernel<4>, grid, blk, lds, st, da, ldda, da_chw, z,
                               stats, gamma, beta, dz, part, rows, P, C, G, rpw);
        MARL_LAUNCH_CHECK();
        return MARL_OK;
    }
    hipLaunchKernelGGL(gn_silu_bwd_kernel, dim3((unsigned)gn_bwd_blocks(rows, C)), dim3(64 * w),
                       (size_t)w * 2 * C * sizeof(float), st, da, ldda, da_chw, z, stats, gamma,
                       beta, dz, part, rows, P, C, G, rpw);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// aggregate_messages (networks/message.py:5-17): (sum_a m - m) / (Na - 1)
// ---------------------------------------------------------------------------
__global__ void agg_msg_kernel(const float* __restrict__ m, float* __restrict__ out, int ld,
                               int na, int nb, int n) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)nb * n) return;
    const int b = (int)(idx / n), k = (int)(idx % n);
    if (na == 1) {
        out[(size_t)b * ld + k] = 0.f;
        return;
    }
    float s = 0.f;
    for (int a = 0; a < na; ++a) s += m[((size_t)a * nb + b) * ld + k];
    const float den = (float)(na - 1);
    for (int a = 0; a < na; ++a) {
        const size_t o = ((size_t)a * nb + b) * ld + k;
        out[o] = (s - m[o]) / den;
    }
}

int launch_agg_msg(const float* m, float* out, int ld, int na, int nb, int n, hipStream_t st) {
    const int64_t tot = (int64_t)nb * n;
    hipLaunchKernelGGL(agg_msg_kernel, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, st, m, out, ld,
                       na, nb, n);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// map_pos (networks/state.py:14-16) on normalised positions (core/environment.py:74-81)
// ---------------------------------------------------------------------------
__global__ void pos_embed_fwd_kernel(const int32_t* __restrict__ pos,
                                     const float* __restrict__ npos_in, int H, int W_,
                                     const float* __restrict__ W, const float* __restrict__ b,
                                     const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float* __restrict__ npos4,
                                     float* __restrict__ z, int ldz, float* __restrict__ stats,
                                     float* __restrict__ out, int ldo, int64_t rows, int nd) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float p0 = npos_in ? npos_in[r * 2] : (float)pos[r * 2] / (float)H;
    const float p1 = npos_in ? npos_in[r * 2 + 1] : (float)pos[r * 2 + 1] / (float)W_;
    if (npos4) {
        npos4[r * 4] = p0;
        npos4[r * 4 + 1] = p1;
    }
    float* zr = z + r * ldz;
    float s = 0.f;
    for (int j = 0; j < nd; ++j) {
        const float v = b[j] + p0 * W[4 * j] + p1 * W[4 * j + 1];  // W packed [nd, 4]
        zr[j] = v;
        s += v;
    }
    const float mean = s / (float)nd;
    float q = 0.f;
    for (int j = 0; j < nd; ++j) {
        const float d = zr[j] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(q / (float)nd + 1e-5f);
    if (stats) {
        stats[r * 2] = mean;
        stats[r * 2 + 1] = rstd;
    }
    float* orow = out + r * ldo;
    for (int j = 0; j < nd; ++j) orow[j] = silu_f((zr[j] - mean) * rstd * gamma[j] + beta[j]);
}

int launch_pos_embed_fwd(const int32_t* pos, const float* npos_in, int h, int w, const float* W,
                         const float* b, const float* gamma, const float* beta, float* npos4,
                         float* z, int ldz, float* stats, float* out, int ldo, int64_t rows, int nd,
                         hipStream_t st) {
    hipLaunchKernelGGL(pos_embed_fwd_kernel, dim3((unsigned)cdiv(rows, 128)), dim3(128), 0, st, pos,
                       npos_in, h, w, W, b, gamma, beta, npos4, z, ldz, stats, out, ldo, rows, nd);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// LSTM cell backward (pointwise part). gates holds activated i|f|g|o on entry and the
// pre-activation gradients on exit; dc holds dL/dc_t on entry and dL/dc_{t-1} on exit.
// ---------------------------------------------------------------------------
__global__ void lstm_cell_bwd_kernel(const LstmBwdBatch B) {
    if (B.vec4)
        lstm_cell_bwd_elem4(B.a[blockIdx.y], B.rows, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    else
        lstm_cell_bwd_elem(B.a[blockIdx.y], B.rows, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

int launch_lstm_cell_bwd_batch(LstmBwdBatch& b, int count, hipStream_t st) {
    int nmax = 0;
    b.vec4 = 1;
    for (int i = 0; i < count; ++i) {
        nmax = b.a[i].n > nmax ? b.a[i].n : nmax;
        if (!lstm_bwd_vec4_ok(b.a[i])) b.vec4 = 0;
    }
    if (!b.vec4)
        for (int i = 0; i < count; ++i)
            if (b.a[i].g3) {
                set_error("lstm_cell_bwd: the gate-gradient image needs the 4-wide form");
                return MARL_EINVAL;
            }
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((unsigned)cdiv(b.rows * (b.vec4 ? nmax / 4 : nmax), 256), (unsigned)count),
                       dim3(256), 0, st, b);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

int launch_lstm_cell_bwd(const float* dh, int lddh, float* dc, int lddc, float* gates, int ldg,
                         const float* c_prev, const float* c_new, int ldc, int64_t rows, int n,
                         hipStream_t st) {
    LstmBwdBatch b{};
    b.a[0] = LstmBwdArgs{dh, dc, gates, c_prev, c_new, lddh, lddc, ldg, ldc, n, nullptr, 0, 0, 0};
    b.rows = rows;
    return launch_lstm_cell_bwd_batch(b, 1, st);
}

// belief and action cells in one launch
int launch_lstm_cell_bwd2(const float* dh0, int lddh0, float* dc0, int lddc0, float* gates0,
                          int ldg0, const float* cp0, const float* cn0, int ldc0, int n0,
                          const float* dh1, int lddh1, float* dc1, int lddc1, float* gates1,
                          int ldg1, const float* cp1, const float* cn1, int ldc1, int n1,
                          int64_t rows, hipStream_t st) {
    LstmBwdBatch b{};
    b.a[0] = LstmBwdArgs{dh0, dc0, gates0, cp0, cn0, lddh0, lddc0, ldg0, ldc0, n0, nullptr, 0, 0, 0};
    b.a[1] = LstmBwdArgs{dh1, dc1, gates1, cp1, cn1, lddh1, lddc1, ldg1, ldc1, n1, nullptr, 0, 0, 0};
    b.rows = rows;
    return launch_lstm_cell_bwd_batch(b, 2, st);
}

// ---------------------------------------------------------------------------
// Policy output layer + softmax + multinomial (argmax p/q) + log-prob + bounded move
// (networks/policy.py:15-16, core/agent.py:53-61, core/environment.py:56-66,128-150)
// ---------------------------------------------------------------------------
template <int MAXA>
__global__ __launch_bounds__(256) void sample_kernel(const SampleArgs A) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= A.R) return;
    float p[MAXA];
    SamplePre<MAXA> S;
    sample_prefetch<MAXA>(A, r, lane, S);
    sample_row_logits<MAXA>(A, r, p, lane);
    sample_finish<MAXA>(A, r, p, lane, S);
}

int launch_sample(const SampleArgs& a, hipStream_t st) {
    if (a.nA > MARL_MAX_ACTIONS) return MARL_ELIMIT;
    if (a.nA <= 4)
        hipLaunchKernelGGL(sample_kernel<4>, dim3((unsigned)cdiv(a.R, 4)), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(sample_kernel<MARL_MAX_ACTIONS>, dim3((unsigned)cdiv(a.R, 4)), dim3(256), 0, st, a);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// Perf-mode episode draws: one launch replaces the reference's seven draw calls
// (environment.py:33-43: randint per dimension; models.py:148-159: four randn) and, when asked,
// the per-step exponential_() of th.multinomial.  Element e of stream s uses Philox counter
// (offset, s, e / 4): independent of the grid, reproducible for a (seed, offset) pair.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void draw_episode_kernel(
    uint64_t seed, uint64_t offset, const uint64_t* __restrict__ offset_dev,
    int64_t* __restrict__ pos0, int R, int H, int W, int f,
    float* __restrict__ h0, float* __restrict__ c0, int64_t nb_el, float* __restrict__ hc0,
    float* __restrict__ cc0, int64_t na_el, float* __restrict__ noise, int64_t n_noise,
    int64_t q_pos, int64_t q_b, int64_t q_a, int64_t q_n) {
    // quads (4 values each) are laid out stream after stream: pos | h0 | c0 | hc0 | cc0 | noise
    if (offset_dev) offset += *offset_dev;
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t rel = q;
    if (rel < q_pos) {  // two rows (4 coordinates) per quad
        const Philox4 u = philox4x32_10(seed, offset, (uint32_t)rel, 0u);
        const uint32_t uv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t e = rel * 4 + i;
            if (e < (int64_t)R * 2) {
                const uint32_t span = (uint32_t)(((e & 1) ? W : H) - f);
                pos0[e] = (int64_t)(((uint64_t)uv[i] * span) >> 32);  // [0, span)
            }
        }
        return;
    }
    rel -= q_pos;
    float* dst = nullptr;
    int64_t n = 0;
    uint32_t stream = 1;
    bool normal = true;
    if (rel < q_b) {
        dst = h0, n = nb_el, stream = 1;
    } else if ((rel -= q_b) < q_b) {
        dst = c0, n = nb_el, stream = 2;
    } else if ((rel -= q_b) < q_a) {
        dst = hc0, n = na_el, stream = 3;
    } else if ((rel -= q_a) < q_a) {
        dst = cc0, n = na_el, stream = 4;
    } else if ((rel -= q_a) < q_n) {
        dst = noise, n = n_noise, stream = 5, normal = false;
    } else {
        return;
    }
    const Philox4 u = philox4x32_10(seed, offset, (uint32_t)rel, stream | ((uint32_t)(rel >> 32) << 8));
    float v[4];
    if (normal) {  // Box-Muller on two pairs
        const float r0 = sqrtf(-2.0f * logf(philox_u01(u.x))), r1 = sqrtf(-2.0f * logf(philox_u01(u.z)));
        const float a0 = 6.283185307179586f * philox_u01(u.y), a1 = 6.283185307179586f * philox_u01(u.w);
        v[0] = r0 * cosf(a0);
        v[1] = r0 * sinf(a0);
        v[2] = r1 * cosf(a1);
        v[3] = r1 * sinf(a1);
    } else {  // Exp(1)
        v[0] = -logf(philox_u01(u.x));
        v[1] = -logf(philox_u01(u.y));
        v[2] = -logf(philox_u01(u.z));
        v[3] = -logf(philox_u01(u.w));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (rel * 4 + i < n) dst[rel * 4 + i] = v[i];
}

int launch_draw_episode(uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int64_t* pos0,
                        int R, int H, int W, int f, float* h0, float* c0, int n_b, float* hc0,
                        float* cc0, int n_a, float* noise, int64_t n_noise, hipStream_t st) {
    const int64_t nb_el = (int64_t)R * n_b, na_el = (int64_t)R * n_a;
    const int64_t q_pos = cdiv((int64_t)R * 2, 4), q_b = cdiv(nb_el, 4), q_a = cdiv(na_el, 4),
                  q_n = noise ? cdiv(n_noise, 4) : 0;
    const int64_t total = q_pos + 2 * q_b + 2 * q_a + q_n;
    hipLaunchKernelGGL(draw_episode_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, seed,
                       offset, offset_dev, pos0, R, H, W, f, h0, c0, nb_el, hc0, cc0, na_el, noise, n_noise, q_pos,
                       q_b, q_a, q_n);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// permuted copies (weight packing / gradient unpacking)
// ---------------------------------------------------------------------------
__global__ void permute_kernel(const PermBatch B) {
    const PermDesc& d = B.d[blockIdx.y];
    const int64_t tot = (int64_t)d.rows * d.cols;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / d.cols), c = (int)(idx % d.cols);
        const int64_t so = (int64_t)(r / d.rd) * d.rs1 + (int64_t)(r % d.rd) * d.rs2 +
                           (int64_t)(c / d.cd) * d.cs1 + (int64_t)(c % d.cd) * d.cs2;
        float v = d.src ? d.src[so] : 0.f;
        if (d.src2) v += d.src2[so];
        d.dst[(int64_t)r * d.dst_ld + c] = v;
    }
}

int launch_permute(const PermBatch& b, hipStream_t st) {
    if (b.count <= 0) return MARL_OK;
    if (b.count > kMaxPerm) return MARL_EINVAL;
    int64_t mx = 0;
    for (int i = 0; i < b.count; ++i) {
        const int64_t t = (int64_t)b.d[i].rows * b.d[i].cols;
        mx = t > mx ? t : mx;
    }
    int64_t gx = cdiv(mx, 256);
    if (gx > 1024) gx = 1024;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(permute_kernel, dim3((unsigned)gx, (unsigned)b.count), dim3(256), 0, st, b);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// Adam (th.optim.Adam single-tensor update order, training/trainer.py:33,116)
// ---------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, int64_t n,
                            float step_size, float inv_sqrt_bc2, float beta1, float beta2,
                            float eps, float grad_scale, const Counters* __restrict__ cnt) {
    if (cnt) {  // graph replay: the step-dependent scalars live on the device
        step_size = cnt->lr_over_bc1;
        inv_sqrt_bc2 = cnt->inv_sqrt_bc2;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float gr = g[i] * grad_scale;
        const float mi = m[i] + (1.0f - beta1) * (gr - m[i]);
        const float vi = v[i] * beta2 + (1.0f - beta2) * (gr * gr);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_over_bc1,
                float inv_sqrt_bc2, float beta1, float beta2, float eps, float grad_scale,
                hipStream_t st, const Counters* cnt) {
    if (n <= 0) return MARL_OK;
    int64_t gx = cdiv(n, 256);
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)gx), dim3(256), 0, st, p, g, m, v, n,
                       lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, grad_scale, cnt);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// counter block: set (tick = 0) or advance by one iteration (tick = 1); one thread
__global__ void counters_kernel(Counters* c, uint64_t rng_offset, int64_t step, float lr, float beta1,
                                float beta2, int tick) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint64_t off = rng_offset, st = (uint64_t)step;
    if (tick) {
        off = c->rng_offset + 1;
        st = c->step + 1;
    }
    c->rng_offset = off;
    c->step = st;
    const double bc1 = 1.0 - pow((double)beta1, (double)st);
    const double bc2 = 1.0 - pow((double)beta2, (double)st);
    c->lr_over_bc1 = (float)((double)lr / bc1);
    c->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
}

int launch_counters_set(Counters* c, uint64_t rng_offset, int64_t step, float lr, float beta1,
                        float beta2, int tick, hipStream_t st) {
    hipLaunchKernelGGL(counters_kernel, dim3(1), dim3(64), 0, st, c, rng_offset, step, lr, beta1, beta2, tick);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// small elementwise helpers
// ---------------------------------------------------------------------------
__global__ void fill_kernel(float* p, int64_t n, float v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        p[i] = v;
}
int launch_fill(float* p, int64_t n, float v, hipStream_t st) {
    if (n <= 0) return MARL_OK;
    int64_t gx = cdiv(n, 256);
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)gx), dim3(256), 0, st, p, n, v);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

__global__ void copy2d_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst,
                              int64_t ldd, int64_t rows, int cols) {
    const int64_t tot = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i % cols);
        dst[r * ldd + c] = src[r * lds + c];
    }
}
__global__ void add2d_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b,
                             int64_t ldb, float* __restrict__ out, int64_t ldo, int64_t rows, int cols) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int64_t r = idx / cols;
    const int c = (int)(idx % cols);
    out[r * ldo + c] = a[r * lda + c] + b[r * ldb + c];
}
int launch_add2d(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo,
                 int64_t rows, int cols, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return MARL_OK;
    hipLaunchKernelGGL(add2d_kernel, dim3((unsigned)cdiv(rows * cols, 256)), dim3(256), 0, st, a, lda, b,
                       ldb, out, ldo, rows, cols);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

int launch_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int cols,
                  hipStream_t st) {
    if (rows <= 0 || cols <= 0) return MARL_OK;
    int64_t gx = cdiv(rows * cols, 256);
    if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)gx), dim3(256), 0, st, src, lds, dst, ldd, rows,
                       cols);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

__global__ void i64_to_i32_kernel(const int64_t* __restrict__ s, int32_t* __restrict__ d,
                                  int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = (int32_t)s[i];
}
int launch_i64_to_i32(const int64_t* src, int32_t* dst, int64_t n, hipStream_t st) {
    if (n <= 0) return MARL_OK;
    hipLaunchKernelGGL(i64_to_i32_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, src, dst,
                       n);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

__global__ void policy_dlogits_kernel(const float* __restrict__ dlogp,
                                      const float* __restrict__ probs,
                                      const int32_t* __restrict__ actions, float* __restrict__ out,
                                      int ld, int64_t rows, int nA) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * nA) return;
    const int64_t r = idx / nA;
    const int j = (int)(idx % nA);
    const float g = dlogp ? dlogp[r] : 0.f;
    out[r * ld + j] = g * ((j == actions[r] ? 1.0f : 0.0f) - probs[r * nA + j]);
}
int launch_policy_dlogits(const float* dlogp, const float* probs, const int32_t* actions,
                          float* out, int ld, int64_t rows, int nA, hipStream_t st) {
    hipLaunchKernelGGL(policy_dlogits_kernel, dim3((unsigned)cdiv(rows * nA, 256)), dim3(256), 0,
                       st, dlogp, probs, actions, out, ld, rows, nA);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ a, int lda,
                                                     const float* __restrict__ w,
                                                     const float* __restrict__ b,
                                                     float* __restrict__ out, int64_t rows,
                                                     int n) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* ar = a + r * lda;
    float s = 0.f;
    for (int k = lane; k < n; k += 64) s += ar[k] * w[k];
    s = wave_sum(s);
    if (lane == 0) out[r] = s + b[0];
}
int launch_rowdot(const float* a, int lda, const float* w, const float* b, float* out,
                  int64_t rows, int n, hipStream_t st) {
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st, a, lda, w, b,
                       out, rows, n);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

}  // namespace marl

extern "C" const char* marl_last_error(void) { return marl::last_error(); }
