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
