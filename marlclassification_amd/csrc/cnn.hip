// Data movement of the observation path: the patch gather
// (Environment.__observation, core/environment.py:95-126) fused with the first
// convolution's im2col, and im2col / col2im for the deeper 3x3 stride-2 pad-1 layers
// (networks/vision.py:33-35).  Convolutions themselves run on the matrix cores as
// cols[rows*P, 9*Cin] x W[Cout, 9*Cin]^T (gemm.hip).  K order is (kh, kw, ci) so that
// NHWC activations give 16-byte contiguous chunks.
#include "common.h"

namespace marl {

// ---------------------------------------------------------------------------
// gather + im2col of layer 0.  A workgroup stages RB patches in LDS with coalesced row
// segment reads (f contiguous floats per (channel, line)), then emits the im2col rows.
// ---------------------------------------------------------------------------
template <bool FROM_OBS, typename PIX>
__global__ __launch_bounds__(256) void gather_im2col_kernel(
    const PIX* __restrict__ img, const int32_t* __restrict__ pos, float* __restrict__ cols,
    int ldk, int64_t rows, int nb, int c_img, int cin, int H, int W, int f, int rb) {
    extern __shared__ __attribute__((aligned(16))) float patch[];  // [rb][cin][f][f]
    const int ff = f * f;
    const int pe = cin * ff;
    const int64_t row0 = (int64_t)blockIdx.x * rb;
    int nrow = (int)(rows - row0 < rb ? rows - row0 : rb);
    for (int idx = threadIdx.x; idx < nrow * pe; idx += 256) {
        const int lr = idx / pe, e = idx % pe;
        const int ci = e / ff, y = (e % ff) / f, x = e % f;
        const int64_t r = row0 + lr;
        float v;
        if (FROM_OBS) {
            v = (float)img[((r * c_img + ci) * f + y) * f + x];
        } else {
            const int b = (int)(r % nb);
            const int p0 = pos[r * 2], p1 = pos[r * 2 + 1];
            v = (float)img[(((int64_t)b * c_img + ci) * H + (p0 + y)) * W + (p1 + x)];
        }
        if (sizeof(PIX) == 1) v = v / 255.0f;  // ToTensor on the fly (uint8 images)
        patch[idx] = v;
    }
    __syncthreads();
    const int oh = (f - 1) / 2 + 1;
    const int P = oh * oh;
    const int K = 9 * cin;
    const int per_row = P * K;
    for (int idx = threadIdx.x; idx < nrow * per_row; idx += 256) {
        const int lr = idx / per_row, e = idx % per_row;
        const int opos = e / K, k = e % K;
        const int tap = k / cin, ci = k % cin;
        const int iy = 2 * (opos / oh) - 1 + tap / 3;
        const int ix = 2 * (opos % oh) - 1 + tap % 3;
        float v = 0.f;
        if (iy >= 0 && iy < f && ix >= 0 && ix < f) v = patch[lr * pe + (ci * f + iy) * f + ix];
        cols[((row0 + lr) * P + opos) * ldk + k] = v;
    }
}

static int gather_rb(int cin, int f) {
    const int oh = (f - 1) / 2 + 1;
    const int work = oh * oh * 9 * cin;
    int rb = 1024 / work;
    if (rb < 1) rb = 1;
    if (rb > 16) rb = 16;
    return rb;
}

int launch_gather_im2col(const void* img, int img_u8, const int32_t* pos, float* cols, int ldk,
                         int na, int nb, int c_img, int cin, int H, int W, int f, hipStream_t st) {
    const int64_t rows = (int64_t)na * nb;
    const int rb = gather_rb(cin, f);
    const size_t lds = (size_t)rb * cin * f * f * sizeof(float);
    if (lds > 64 * 1024) {
        set_error("window %d too large for the gather kernel", f);
        return MARL_ELIMIT;
    }
    if (img_u8)
        hipLaunchKernelGGL((gather_im2col_kernel<false, unsigned char>),
                           dim3((unsigned)cdiv(rows, rb)), dim3(256), lds, st,
                           static_cast<const unsigned char*>(img), pos, cols, ldk, rows, nb, c_img,
                           cin, H, W, f, rb);
    else
        hipLaunchKernelGGL((gather_im2col_kernel<false, float>), dim3((unsigned)cdiv(rows, rb)),
                           dim3(256), lds, st, static_cast<const float*>(img), pos, cols, ldk, rows,
                           nb, c_img, cin, H, W, f, rb);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

int launch_obs_im2col(const float* obs, float* cols, int ldk, int64_t rows, int c_img, int cin,
                      int f, hipStream_t st) {
    const int rb = gather_rb(cin, f);
    const size_t lds = (size_t)rb * cin * f * f * sizeof(float);
    if (lds > 64 * 1024) return MARL_ELIMIT;
    hipLaunchKernelGGL((gather_im2col_kernel<true, float>), dim3((unsigned)cdiv(rows, rb)), dim3(256),
                       lds, st, obs, (const int32_t*)nullptr, cols, ldk, rows, 1, c_img, cin, 0, 0,
                       f, rb);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// Fused CNN forward for one step (networks/vision.py:13-53 applied to the patches of
// core/environment.py:95-126).  One workgroup owns `rb` patches end to end:
//   layer-0 im2col tile straight from the image (LDS) -> for every layer:
//   [tile -> global (kept for the weight gradients)] -> conv as 16x16x4 f32 MFMA tiles, weights
//   streamed from L2 -> + bias -> Z (LDS + global) -> GroupNorm statistics (wave per
//   (patch, group)) -> normalise + SiLU while gathering the NEXT layer's im2col tile in LDS
//   (or the flattened feature row of U after the last layer).
// Replaces gather + L x (GEMM launch + GroupNorm/im2col launch): the intermediates never
// leave the CU, the only HBM traffic is the buffers backward needs.
// ---------------------------------------------------------------------------
typedef float cf32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float cnn_silu(float y) { return y / (1.0f + expf(-y)); }

__global__ __launch_bounds__(512) void cnn_fwd_kernel(const CnnFwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* T = lds;               // im2col tile of the current layer [rows16 + 1][s]
    float* Zb = lds + A.off_z;    // conv output [rb * P][cout + 4]
    float* gstat = lds + A.off_stat;  // [rb * G][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int quad = lane >> 4, l16 = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * A.rb;
    const int nrow = (int)(A.rows - row0 < A.rb ? A.rows - row0 : A.rb);

    // ---- layer-0 im2col tile from the image (zero padding taps, zero tail rows)
    {
        const CnnFwdLayer& L0 = A.layer[0];
        const int s0 = A.s[0], P = L0.P, cin = L0.cin, K = L0.K, hout = L0.hout, f = A.f;
        const int M = nrow * P;
        const int tot = (((M + 15) & ~15) + 1) * s0;
        const float* imgf = static_cast<const float*>(A.img);
        const unsigned char* imgb = static_cast<const unsigned char*>(A.img);
#pragma unroll 4
        for (int idx = tid; idx < tot; idx += nthreads) {
            const int mo = idx / s0, k = idx - mo * s0;
            float v = 0.f;
            if (mo < M && k < K) {
                const int lr = mo / P, opos = mo - lr * P;
                const int tap = k / cin, ci = k - tap * cin;
                const int kh = tap / 3, kw = tap - 3 * kh;
                const int oy = opos / hout, ox = opos - oy * hout;
                const int iy = 2 * oy - 1 + kh, ix = 2 * ox - 1 + kw;
                if (iy >= 0 && iy < f && ix >= 0 && ix < f) {
                    const int64_t r = row0 + lr;
                    if (A.obs) {
                        v = A.obs[((r * A.c_img + ci) * f + iy) * f + ix];
                    } else {
                        const int b = (int)(r % A.nb);
                        const int p0 = A.pos[r * 2], p1 = A.pos[r * 2 + 1];
                        const int64_t off = (((int64_t)b * A.c_img + ci) * A.H + (p0 + iy)) * A.W + (p1 + ix);
                        v = A.img_u8 ? (float)imgb[off] / 255.0f : imgf[off];  // ToTensor on the fly
                    }
                }
            }
            T[idx] = v;
        }
    }
    __syncthreads();

    for (int l = 0; l < A.L; ++l) {
        const CnnFwdLayer& Ly = A.layer[l];
        const int s = A.s[l], P = Ly.P, cout = Ly.cout, K = Ly.K, ldk = Ly.ldk;
        const int M = nrow * P, MT = (M + 15) >> 4, NT = (cout + 15) >> 4;
        const int zs = cout + 4;
        // ---- this layer's im2col rows -> global (the weight-gradient GEMM reads them)
        if (Ly.cols) {
            const int c4 = ldk >> 2;
            float* dst = Ly.cols + row0 * P * (int64_t)ldk;
            for (int idx = tid; idx < M * c4; idx += nthreads) {
                const int m = idx / c4, k = (idx - m * c4) * 4;
                *reinterpret_cast<float4*>(dst + (int64_t)m * ldk + k) =
                    *reinterpret_cast<const float4*>(T + m * s + k);
            }
        }
        // ---- conv: Z[m][n] = sum_k T[m][k] * W[n][k] + bias[n]; one 16x16 tile per wave turn
        const int steps = (K + 15) >> 4;
        for (int ti = wave; ti < MT * NT; ti += nwaves) {
            const int nt = ti % NT, mt = ti / NT;
            const float* arow = T + (mt * 16 + l16) * s + 4 * quad;
            int wr = nt * 16 + l16;
            wr = wr < cout ? wr : cout - 1;
            const float* wrow = Ly.w + (int64_t)wr * ldk + 4 * quad;
            cf32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int st0 = 0; st0 < steps; st0 += 8) {
                float4 bq[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kk = (st0 + i) * 16;
                    const bool ok = st0 + i < steps && kk + 4 * quad < ldk;
                    const float4 v = *reinterpret_cast<const float4*>(wrow + (ok ? kk : -4 * quad));
                    bq[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (st0 + i < steps) {
                        const float4 a = *reinterpret_cast<const float4*>(arow + (st0 + i) * 16);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[i].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[i].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[i].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[i].w, acc, 0, 0, 0);
                    }
                }
            }
            const int n = nt * 16 + l16;
            if (n < cout) {
                const float bv = Ly.bias[n];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mt * 16 + 4 * quad + r;
                    if (m < M) {
                        const float zv = acc[r] + bv;
                        Zb[m * zs + n] = zv;
                        if (Ly.z) Ly.z[(row0 * P + m) * (int64_t)cout + n] = zv;
                    }
                }
            }
        }
        __syncthreads();
        // ---- GroupNorm statistics (eps 1e-5, biased variance, two passes): wave per (patch, group)
        const int G = Ly.G, cpg = cout / G, cnt = P * cpg;
        for (int pi = wave; pi < nrow * G; pi += nwaves) {
            const int lr = pi / G, g = pi - lr * G;
            const float* base = Zb + lr * P * zs + g * cpg;
            float sm = 0.f;
            for (int e = lane; e < cnt; e += 64) {
                const int pos = e / cpg;
                sm += base[pos * zs + (e - pos * cpg)];
            }
            const float mean = wave_sum(sm) / (float)cnt;
            float q = 0.f;
            for (int e = lane; e < cnt; e += 64) {
                const int pos = e / cpg;
                const float d = base[pos * zs + (e - pos * cpg)] - mean;
                q += d * d;
            }
            const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)cnt + 1e-5f);
            if (lane == 0) {
                gstat[pi * 2] = mean;
                gstat[pi * 2 + 1] = rstd;
                if (Ly.gst) {
                    Ly.gst[((row0 + lr) * G + g) * 2] = mean;
                    Ly.gst[((row0 + lr) * G + g) * 2 + 1] = rstd;
                }
            }
        }
        __syncthreads();
        if (l + 1 < A.L) {
            // ---- normalise + SiLU while gathering the next layer's im2col tile (3x3, stride 2,
            // pad 1; k = tap * cin + ci, float4 along ci)
            const CnnFwdLayer& Nx = A.layer[l + 1];
            const int s1 = A.s[l + 1], c4 = s1 >> 2, P1 = Nx.P, hin = Ly.hout, hout = Nx.hout;
            const int M1 = nrow * P1;
            const int tot = (((M1 + 15) & ~15) + 1) * c4;
            for (int idx = tid; idx < tot; idx += nthreads) {
                const int mo = idx / c4, k = (idx - mo * c4) * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mo < M1 && k < Nx.K) {
                    const int lr = mo / P1, opos = mo - lr * P1;
                    const int tap = k / cout, ci = k - tap * cout;
                    const int kh = tap / 3, kw = tap - 3 * kh;
                    const int oy = opos / hout, ox = opos - oy * hout;
                    const int iy = 2 * oy - 1 + kh, ix = 2 * ox - 1 + kw;
                    if (iy >= 0 && iy < hin && ix >= 0 && ix < hin) {
                        const float4 z = *reinterpret_cast<const float4*>(Zb + (lr * P + iy * hin + ix) * zs + ci);
                        const int g = ci / cpg;
                        const float mean = gstat[(lr * G + g) * 2], rstd = gstat[(lr * G + g) * 2 + 1];
                        const float4 gm = *reinterpret_cast<const float4*>(Ly.gamma + ci);
                        const float4 bt = *reinterpret_cast<const float4*>(Ly.beta + ci);
                        v.x = cnn_silu((z.x - mean) * rstd * gm.x + bt.x);
                        v.y = cnn_silu((z.y - mean) * rstd * gm.y + bt.y);
                        v.z = cnn_silu((z.z - mean) * rstd * gm.z + bt.z);
                        v.w = cnn_silu((z.w - mean) * rstd * gm.w + bt.w);
                    }
                }
                *reinterpret_cast<float4*>(T + mo * s1 + k) = v;
            }
        } else {
            // ---- last layer: features in the reference's (C, H, W) flatten order -> U
            const int E = P * cout;
            for (int idx = tid; idx < nrow * E; idx += nthreads) {
                const int lr = idx / E, e = idx - lr * E;
                const int c = e / P, pos = e - c * P;
                const int g = c / cpg;
                const float mean = gstat[(lr * G + g) * 2], rstd = gstat[(lr * G + g) * 2 + 1];
                const float zv = Zb[(lr * P + pos) * zs + c];
                A.u[(row0 + lr) * (int64_t)A.ldu + e] = cnn_silu((zv - mean) * rstd * Ly.gamma[c] + Ly.beta[c]);
            }
        }
        __syncthreads();
    }
}

static int cnn_lds_stride(int ldk) { return (ldk & 7) == 4 ? ldk : ldk + 4; }

// LDS floats for rb patches per workgroup; fills a.s / a.off_*
static size_t cnn_fwd_plan(CnnFwdArgs& a, int rb) {
    size_t tile = 0, zb = 0, st = 0;
    for (int l = 0; l < a.L; ++l) {
        const CnnFwdLayer& L = a.layer[l];
        a.s[l] = cnn_lds_stride(L.ldk);
        const size_t rows16 = (((size_t)rb * L.P + 15) & ~(size_t)15) + 1;
        // + 16 floats: the last k-step of the last row may read past the row stride
        const size_t t = rows16 * a.s[l] + 16;
        tile = t > tile ? t : tile;
        const size_t z = (size_t)rb * L.P * (L.cout + 4);
        zb = z > zb ? z : zb;
        const size_t g = (size_t)rb * L.G * 2;
        st = g > st ? g : st;
    }
    tile = (tile + 3) & ~(size_t)3;
    zb = (zb + 3) & ~(size_t)3;
    a.rb = rb;
    a.off_z = (int)tile;
    a.off_stat = (int)(tile + zb);
    return tile + zb + st;
}

int cnn_fwd_supported(const CnnFwdArgs& a0) {
    if (getenv("MARL_CNN_FUSED") && getenv("MARL_CNN_FUSED")[0] == '0') return 0;
    CnnFwdArgs a = a0;
    for (int l = 0; l < a.L; ++l) {
        const CnnFwdLayer& L = a.layer[l];
        if (L.cout % L.G != 0 || ((L.cout / L.G) & 3) || (L.cout & 3)) return 0;  // float4 stays inside a group
        if (l > 0 && (L.cin & 3)) return 0;
    }
    return cnn_fwd_plan(a, 1) * sizeof(float) <= 144 * 1024;
}

int launch_cnn_fwd(CnnFwdArgs& a, hipStream_t st) {
    if (a.rows <= 0) return MARL_OK;
    // as many patches per workgroup as fit two workgroups per CU (the conv weights are re-read
    // from L2 by every workgroup), but keep >= 256 workgroups
    int rb = 8;
    while (rb > 1 && (cnn_fwd_plan(a, rb) * sizeof(float) > 72 * 1024 || cdiv(a.rows, rb) < 256)) --rb;
    const size_t lds = cnn_fwd_plan(a, rb) * sizeof(float);
    if (lds > 144 * 1024) {
        set_error("fused CNN forward: window %d outside its range", a.f);
        return MARL_ELIMIT;
    }
    static bool raised = false;
    if (!raised) {
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        raised = true;
    }
    hipLaunchKernelGGL(cnn_fwd_kernel, dim3((unsigned)cdiv(a.rows, rb)), dim3(512), lds, st, a);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// Environment.observe(): obs[r, c, y, x] = img[b, c, p0 + y, p1 + x]
__global__ void patch_gather_kernel(const float* __restrict__ img, const int64_t* __restrict__ pos,
                                    float* __restrict__ obs, int64_t rows, int nb, int c, int H,
                                    int W, int f) {
    const int64_t per = (int64_t)c * f * f;
    const int64_t tot = rows * per;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / per;
        const int e = (int)(idx % per);
        const int ci = e / (f * f), y = (e / f) % f, x = e % f;
        const int b = (int)(r % nb);
        const int64_t p0 = pos[r * 2], p1 = pos[r * 2 + 1];
        obs[idx] = img[(((int64_t)b * c + ci) * H + (p0 + y)) * W + (p1 + x)];
    }
}

int launch_patch_gather(const float* img, const int64_t* pos, float* obs, int na, int nb, int c,
                        int H, int W, int f, hipStream_t st) {
    const int64_t rows = (int64_t)na * nb;
    const int64_t tot = rows * c * f * f;
    if (tot <= 0) return MARL_OK;
    int64_t gx = cdiv(tot, 256);
    if (gx > 16384) gx = 16384;
    hipLaunchKernelGGL(patch_gather_kernel, dim3((unsigned)gx), dim3(256), 0, st, img, pos, obs,
                       rows, nb, c, H, W, f);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// im2col / col2im for layers >= 1 on NHWC activations
// ---------------------------------------------------------------------------
template <int VEC>
__global__ void im2col_kernel(const float* __restrict__ a, float* __restrict__ cols, int ldk,
                              int64_t rows, int hin, int cin) {
    const int hout = (hin - 1) / 2 + 1;
    const int P = hout * hout;
    const int cv = cin / VEC;
    const int64_t tot = rows * P * 9 * cv;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % cv);
        int64_t t = idx / cv;
        const int tap = (int)(t % 9);
        t /= 9;
        const int opos = (int)(t % P);
        const int64_t row = t / P;
        const int iy = 2 * (opos / hout) - 1 + tap / 3;
        const int ix = 2 * (opos % hout) - 1 + tap % 3;
        const bool in = iy >= 0 && iy < hin && ix >= 0 && ix < hin;
        const float* src = a + ((row * hin + iy) * hin + ix) * cin + c4 * VEC;
        float* dst = cols + (row * P + opos) * ldk + tap * cin + c4 * VEC;
        if (VEC == 4) {
            *reinterpret_cast<float4*>(dst) =
                in ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            *dst = in ? *src : 0.f;
        }
    }
}

int launch_im2col(const float* a, float* cols, int ldk, int64_t rows, int hin, int cin,
                  hipStream_t st) {
    const int hout = (hin - 1) / 2 + 1;
    const bool vec = (cin % 4 == 0) && (ldk % 4 == 0);
    const int64_t tot = rows * hout * hout * 9 * (vec ? cin / 4 : cin);
    if (tot <= 0) return MARL_OK;
    int64_t gx = cdiv(tot, 256);
    if (gx > 65536) gx = 65536;
    if (vec)
        hipLaunchKernelGGL((im2col_kernel<4>), dim3((unsigned)gx), dim3(256), 0, st, a, cols, ldk,
                           rows, hin, cin);
    else
        hipLaunchKernelGGL((im2col_kernel<1>), dim3((unsigned)gx), dim3(256), 0, st, a, cols, ldk,
                           rows, hin, cin);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// da[row, iy, ix, ci] = sum over taps (kh, kw) with oy = (iy + 1 - kh) / 2 integral and in
// range of dcols[(row, oy, ox), (kh, kw, ci)]   (fixed summation order)
template <int VEC>
__global__ void col2im_kernel(const float* __restrict__ dcols, int ldk, float* __restrict__ da,
                              int64_t rows, int hin, int cin) {
    const int hout = (hin - 1) / 2 + 1;
    const int P = hout * hout;
    const int cv = cin / VEC;
    const int64_t tot = rows * hin * hin * cv;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % cv);
        int64_t t = idx / cv;
        const int ipos = (int)(t % (hin * hin));
        const int64_t row = t / (hin * hin);
        const int iy = ipos / hin, ix = ipos % hin;
        float acc[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ty = iy + 1 - kh;
            if (ty < 0 || (ty & 1)) continue;
            const int oy = ty >> 1;
            if (oy >= hout) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int tx = ix + 1 - kw;
                if (tx < 0 || (tx & 1)) continue;
                const int ox = tx >> 1;
                if (ox >= hout) continue;
                const float* src =
                    dcols + (row * P + oy * hout + ox) * ldk + (kh * 3 + kw) * cin + c4 * VEC;
                if (VEC == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    acc[0] += v.x;
                    acc[1 % VEC] += v.y;
                    acc[2 % VEC] += v.z;
                    acc[3 % VEC] += v.w;
                } else {
                    acc[0] += *src;
                }
            }
        }
        float* dst = da + (row * hin * hin + ipos) * cin + c4 * VEC;
#pragma unroll
        for (int v = 0; v < VEC; ++v) dst[v] = acc[v];
    }
}

int launch_col2im(const float* dcols, int ldk, float* da, int64_t rows, int hin, int cin,
                  hipStream_t st) {
    const bool vec = (cin % 4 == 0) && (ldk % 4 == 0);
    const int64_t tot = rows * hin * hin * (vec ? cin / 4 : cin);
    if (tot <= 0) return MARL_OK;
    int64_t gx = cdiv(tot, 256);
    if (gx > 65536) gx = 65536;
    if (vec)
        hipLaunchKernelGGL((col2im_kernel<4>), dim3((unsigned)gx), dim3(256), 0, st, dcols, ldk, da,
                           rows, hin, cin);
    else
        hipLaunchKernelGGL((col2im_kernel<1>), dim3((unsigned)gx), dim3(256), 0, st, dcols, ldk, da,
                           rows, hin, cin);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

}  // namespace marl
