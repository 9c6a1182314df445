// Data movement of the observation path: the patch gather
// (Environment.__observation, core/environment.py:95-126) fused with the first
// convolution's im2col, and im2col / col2im for the deeper 3x3 stride-2 pad-1 layers
// (networks/vision.py:33-35).  Convolutions themselves run on the matrix cores as
// cols[rows*P, 9*Cin] x W[Cout, 9*Cin]^T (gemm.hip).  K order is (kh, kw, ci) so that
// NHWC activations give 16-byte contiguous chunks.
#include <stdlib.h>

#include <type_traits>

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
//   raw patches -> LDS; for every layer: conv as 16x16x4 f32 MFMA tiles whose A fragments are
//   gathered straight from the previous activation in LDS (implicit im2col: 3x3, stride 2,
//   pad 1, k = tap * cin + ci) and whose weights stream from L2; the gathered fragments are
//   also the im2col rows backward needs, so they go to global from registers -> + bias -> Z
//   (LDS + global) -> GroupNorm statistics (wave per (patch, group)) -> normalise + SiLU in
//   place (the next layer's input), or the flattened feature row of U after the last layer.
// Replaces gather + L x (GEMM launch + GroupNorm/im2col launch): the intermediates never
// leave the CU, the only HBM traffic is the buffers backward needs.
// ---------------------------------------------------------------------------
#ifdef MARL_KERNEL_TS
int ts_begin(long long** dev, int call) {
    if (!*dev) (void)hipMalloc(dev, 16 * 48 * sizeof(long long));
    const char* e = getenv("MARL_TS_CALL");
    const int want = e ? atoi(e) : 100;
    if (call != want) return 0;
    (void)hipMemset(*dev, 0, 16 * 48 * sizeof(long long));
    return 1;
}
void ts_report(const char* tag, long long* dev, int waves) {
    static long long h[16 * 48];
    (void)hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < waves; w += (waves > 1 ? waves - 1 : 1)) {
        const long long* t = h + w * 48;
        fprintf(stderr, "[ts] %s wave %d/%d:", tag, w, waves);
        long long tot = 0;
        for (int i = 1; i < 48 && t[i]; ++i) {
            fprintf(stderr, " %lld", t[i] - t[i - 1]);
            tot += t[i] - t[i - 1];
        }
        fprintf(stderr, " | total %.2f us\n", tot / 100.0);
    }
}
#endif

typedef float cf32x4 __attribute__((ext_vector_type(4)));

// v_exp_f32 / v_rcp_f32 based (~1e-7 relative error, inside the 1e-5 parity budget)
__device__ __forceinline__ float cnn_silu(float y) { return silu_fast(y); }

__global__ __launch_bounds__(512) void cnn_fwd_kernel(const CnnFwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* patch = lds;                  // [rb][cin0][f][f]
    float* gstat = lds + A.off_stat;     // [rb * G][2]
    __shared__ int64_t simg[16];         // per patch: offset of its image (or observation)
    __shared__ int spos[16][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int quad = lane >> 4, l16 = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * A.rb;
    const int nrow = (int)(A.rows - row0 < A.rb ? A.rows - row0 : A.rb);

#ifdef MARL_KERNEL_TS
    MARL_TS_DECL(A.ts);
#endif
    MARL_TS();
    if (tid < nrow) {
        const int64_t r = row0 + tid;
        if (A.obs) {
            simg[tid] = r * A.c_img * A.f * A.f;
            spos[tid][0] = spos[tid][1] = 0;
        } else {
            simg[tid] = (r % A.nb) * A.c_img * (int64_t)A.H * A.W;
            spos[tid][0] = A.pos[r * 2];
            spos[tid][1] = A.pos[r * 2 + 1];
        }
    }
    __syncthreads();
    // ---- raw patches -> LDS
    {
        const int f = A.f, ff = f * f, pe = A.layer[0].cin * ff;
        const float* imgf = static_cast<const float*>(A.img);
        const unsigned char* imgb = static_cast<const unsigned char*>(A.img);
#pragma unroll 4
        for (int idx = tid; idx < nrow * pe; idx += nthreads) {
            const int lr = fdiv(idx, A.dpe), e = idx - lr * pe;
            float v;
            if (A.obs) {
                v = A.obs[simg[lr] + e];
            } else {
                const int ci = fdiv(e, A.dff), e2 = e - ci * ff;
                const int iy = fdiv(e2, A.df), ix = e2 - iy * f;
                const int64_t off = simg[lr] + ((int64_t)ci * A.H + (spos[lr][0] + iy)) * A.W + (spos[lr][1] + ix);
                v = A.img_u8 ? (float)imgb[off] / 255.0f : imgf[off];  // ToTensor on the fly
            }
            patch[idx] = v;
        }
    }
    MARL_TS();
    __syncthreads();
    MARL_TS();

    for (int l = 0; l < A.L; ++l) {
        // everything layer-specific is copied out of the (dynamically indexed) kernel-argument
        // arrays ONCE per layer: a scalar load inside the tile loops costs an s_waitcnt lgkmcnt(0)
        // that also drains the LDS queue
        const CnnFwdLayer Ly = A.layer[l];
        const FDiv fP = A.dP[l], fhout = A.dhout[l], fcin = A.dcin[l], fcpg = A.dcpg[l], fG = A.dG[l],
                   fNT = A.dNT[l], fc4o = A.dc4o[l];
        const int P = Ly.P, cin = Ly.cin, cout = Ly.cout, K = Ly.K, ldk = Ly.ldk, hin = Ly.hin,
                  hout = Ly.hout;
        const int M = nrow * P, MT = (M + 15) >> 4, NT = (cout + 15) >> 4;
        const int zs = cout + 4;
        const float* in = l == 0 ? patch : lds + ((l & 1) ? A.off_b0 : A.off_b1);
        float* Zb = lds + ((l & 1) ? A.off_b1 : A.off_b0);
        const bool vec = l > 0;            // cin % 4 == 0: a fragment is one float4 along ci
        const int cs = cin + 4;            // channel stride of the NHWC input (l > 0)
        const int in_per = l == 0 ? cin * hin * hin : hin * hin * cs;
        // ---- conv: Z[m][n] = sum_k im2col(in)[m][k] * W[n][k] + bias[n]; a 16x16 tile per turn
        const int steps = (K + 15) >> 4;
        for (int ti = wave; ti < MT * NT; ti += nwaves) {
            const int mt = fdiv(ti, fNT), nt = ti - mt * NT;
            const int m = mt * 16 + l16;
            const bool mv = m < M;
            const int lr = fdiv(m, fP), opos = m - lr * P;
            const int oy = fdiv(opos, fhout), ox = opos - oy * hout;
            const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
            const float* src = in + lr * in_per;
            const bool wcols = Ly.cols != nullptr && nt == 0;
            float* crow = Ly.cols + (row0 * P + m) * (int64_t)ldk;
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
                        const int k = (st0 + i) * 16 + 4 * quad;
                        float4 a;
                        if (vec) {
                            const int tap = fdiv(k, fcin), ci = k - tap * cin;
                            const int kh = tap / 3, kw = tap - 3 * kh;
                            const int iy = iy0 + kh, ix = ix0 + kw;
                            const bool ok = mv && k < K && (unsigned)iy < (unsigned)hin &&
                                            (unsigned)ix < (unsigned)hin;
                            a = *reinterpret_cast<const float4*>(src + (ok ? (iy * hin + ix) * cs + ci : 0));
                            if (!ok) a = make_float4(0.f, 0.f, 0.f, 0.f);
                        } else {
                            float av[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int kj = k + j;
                                const int tap = fdiv(kj, fcin), ci = kj - tap * cin;
                                const int kh = tap / 3, kw = tap - 3 * kh;
                                const int iy = iy0 + kh, ix = ix0 + kw;
                                const bool ok = mv && kj < K && (unsigned)iy < (unsigned)hin &&
                                                (unsigned)ix < (unsigned)hin;
                                const float t = src[ok ? (ci * hin + iy) * hin + ix : 0];
                                av[j] = ok ? t : 0.f;
                            }
                            a = make_float4(av[0], av[1], av[2], av[3]);
                        }
                        // the im2col row kept for the weight-gradient GEMM
                        if (wcols && mv && k < ldk) *reinterpret_cast<float4*>(crow + k) = a;
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
                    const int mr = mt * 16 + 4 * quad + r;
                    if (mr < M) {
                        const float zv = acc[r] + bv;
                        Zb[mr * zs + n] = zv;
                        if (Ly.z) Ly.z[(row0 * P + mr) * (int64_t)cout + n] = zv;
                    }
                }
            }
        }
        MARL_TS();
        __syncthreads();
        MARL_TS();
        // ---- GroupNorm statistics (eps 1e-5, biased variance, two passes): wave per (patch,
        // group), four pairs walked together so their reduction chains overlap
        const int G = Ly.G, cpg = (int)fcpg.d, cnt = P * cpg, npairs = nrow * G;
        for (int pi0 = wave * 4; pi0 < npairs; pi0 += nwaves * 4) {
            const float* base[4];
            float sm[4], q[4], mean[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pi = pi0 + u < npairs ? pi0 + u : npairs - 1;
                const int lr = fdiv(pi, fG), g = pi - lr * G;
                base[u] = Zb + lr * P * zs + g * cpg;
                sm[u] = q[u] = 0.f;
            }
            for (int e = lane; e < cnt; e += 64) {
                const int pos = fdiv(e, fcpg);
                const int off = pos * zs + (e - pos * cpg);
#pragma unroll
                for (int u = 0; u < 4; ++u) sm[u] += base[u][off];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) mean[u] = wave_sum(sm[u]) / (float)cnt;
            for (int e = lane; e < cnt; e += 64) {
                const int pos = fdiv(e, fcpg);
                const int off = pos * zs + (e - pos * cpg);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float d = base[u][off] - mean[u];
                    q[u] += d * d;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float rstd = 1.0f / sqrtf(wave_sum(q[u]) / (float)cnt + 1e-5f);
                const int pi = pi0 + u;
                if (lane == 0 && pi < npairs) {
                    gstat[pi * 2] = mean[u];
                    gstat[pi * 2 + 1] = rstd;
                    if (Ly.gst) {
                        Ly.gst[(row0 * G + pi) * 2] = mean[u];
                        Ly.gst[(row0 * G + pi) * 2 + 1] = rstd;
                    }
                }
            }
        }
        MARL_TS();
        __syncthreads();
        MARL_TS();
        if (l + 1 < A.L) {
            // ---- normalise + SiLU in place: the next layer's NHWC input
            const int c4 = cout >> 2;
            for (int idx = tid; idx < M * c4; idx += nthreads) {
                const int m = fdiv(idx, fc4o), c = (idx - m * c4) * 4;
                const int lr = fdiv(m, fP);
                const int g = fdiv(c, fcpg);
                const float mean = gstat[(lr * G + g) * 2], rstd = gstat[(lr * G + g) * 2 + 1];
                float4* zp = reinterpret_cast<float4*>(Zb + m * zs + c);
                const float4 z = *zp;
                const float4 gm = *reinterpret_cast<const float4*>(Ly.gamma + c);
                const float4 bt = *reinterpret_cast<const float4*>(Ly.beta + c);
                float4 v;
                v.x = cnn_silu((z.x - mean) * rstd * gm.x + bt.x);
                v.y = cnn_silu((z.y - mean) * rstd * gm.y + bt.y);
                v.z = cnn_silu((z.z - mean) * rstd * gm.z + bt.z);
                v.w = cnn_silu((z.w - mean) * rstd * gm.w + bt.w);
                *zp = v;
            }
        } else {
            // ---- last layer: features in the reference's (C, H, W) flatten order -> U
            const int E = P * cout;
            for (int idx = tid; idx < nrow * E; idx += nthreads) {
                const int lr = fdiv(idx, A.dE), e = idx - lr * E;
                const int c = fdiv(e, fP), pos = e - c * P;
                const int g = fdiv(c, fcpg);
                const float mean = gstat[(lr * G + g) * 2], rstd = gstat[(lr * G + g) * 2 + 1];
                const float zv = Zb[(lr * P + pos) * zs + c];
                A.u[(row0 + lr) * (int64_t)A.ldu + e] = cnn_silu((zv - mean) * rstd * Ly.gamma[c] + Ly.beta[c]);
            }
        }
        MARL_TS();
        __syncthreads();
        MARL_TS();
    }
}

// ---------------------------------------------------------------------------
// Second-generation fused forward for small extractors (every layer's weights fit LDS together).
// What the first kernel spends its time on is latency, not work: every 16x16 tile re-streams
// its weights from L2, every layer has three workgroup barriers, GroupNorm walks LDS twice.
// Here
//   * 256 persistent workgroups copy ALL conv weights into LDS once, then walk chunks of 8 patches;
//   * in the layers whose output has few positions a WAVE owns a patch end to end (gather ->
//     conv tiles -> GroupNorm statistics by cross-lane sums over the accumulator registers ->
//     normalise + SiLU -> next layer's zero-bordered LDS image): no workgroup barrier at all;
//   * the last layer (4 positions per patch) runs as 16-row tiles over 4 patches, again with the
//     statistics taken from the registers; two barriers per chunk in total;
//   * zero borders around every LDS image make every tap address valid (no masks in the loops).
// Same math, same summation order inside a tile as cnn_fwd_kernel (k ascending), so the results
// agree with it to the last bits the parity tests look at.
// ---------------------------------------------------------------------------
// all-reduce over aligned blocks of `n` lanes (n = 1, 2, 4, 8, 16) inside a 16-lane row: DPP
// quad permutes and row mirrors on the VALU, no LDS-crossbar shuffles
__device__ __forceinline__ float row_block_sum(float v, int n) {
#define MARL_DPP_XADD(ctrl) \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
    if (n >= 2) MARL_DPP_XADD(0xB1);   // quad_perm [1,0,3,2]
    if (n >= 4) MARL_DPP_XADD(0x4E);   // quad_perm [2,3,0,1]
    if (n >= 8) MARL_DPP_XADD(0x141);  // row_half_mirror: the other quad of the 8-lane block
    if (n >= 16) MARL_DPP_XADD(0x140); // row_mirror: the other half of the row
#undef MARL_DPP_XADD
    return v;
}
// all-reduce over the four 16-lane rows of a wave (same position inside the row)
__device__ __forceinline__ float cross_row_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// The network shape is a TEMPLATE parameter: with run-time layer dimensions the tile loops are
// full of scalar branches, divisions and kernel-argument reloads and the kernel is bound by
// instruction issue (measured: 3x slower than this form).  One instantiation per extractor of
// the reference (networks/vision.py:55-127) at its README window; other shapes use cnn_fwd_kernel.
// RP1..RP3 / PP / CP1: LDS padding (floats) of the image rows of layers 1..3, of a patch's region
// and of layer 1's channel stride.  They only move addresses: chosen (tools/lds_conflicts.py, the
// bank model of MI355X_MICROARCH.md) so that the 16-lane groups of every ds_read_b128 fragment
// read - lanes = output positions two image columns / rows apart, quads = adjacent 16-byte
// pieces - fall on distinct banks.
template <int F_, int L_, int C0, int C1, int C2, int C3, int C4, int G0, int G1, int G2, int G3,
          int RP1 = 0, int RP2 = 0, int RP3 = 0, int PP = 0, int CP1 = 4>
struct Fwd2Net {
    static constexpr int F = F_, L = L_;
    static constexpr int ch(int l) { return l == 0 ? C0 : l == 1 ? C1 : l == 2 ? C2 : l == 3 ? C3 : C4; }
    static constexpr int grp(int l) { return l == 0 ? G0 : l == 1 ? G1 : l == 2 ? G2 : G3; }
    static constexpr int hin(int l) {
        int h = F_;
        for (int i = 0; i < l; ++i) h = (h - 1) / 2 + 1;
        return h;
    }
    static constexpr int hout(int l) { return (hin(l) - 1) / 2 + 1; }
    static constexpr int P(int l) { return hout(l) * hout(l); }
    static constexpr int cin(int l) { return ch(l); }
    static constexpr int cout(int l) { return ch(l + 1); }
    static constexpr int K(int l) { return 9 * cin(l); }
    static constexpr int ldk(int l) { return (K(l) + 3) & ~3; }
    static constexpr int cpg(int l) { return cout(l) / grp(l); }
    static constexpr int mode(int l) { return (l + 1 == L_ && L_ > 1 && P(l) == 4) ? 1 : 0; }
    static constexpr int mt(int l) { return (P(l) + 15) / 16; }
    static constexpr int nt(int l) { return (cout(l) + 15) / 16; }
    static constexpr int hp(int l) { return hin(l) + 2; }
    static constexpr int cs(int l) { return l == 0 ? cin(0) : cin(l) + (l == 1 ? CP1 : 4); }
    static constexpr int rs(int l) { return hp(l) * cs(l) + (l == 1 ? RP1 : l == 2 ? RP2 : l == 3 ? RP3 : 0); }
    static constexpr int in_per(int l) { return (hp(l) * rs(l) + 3) & ~3; }
    static constexpr int steps(int l) { return (K(l) + 15) / 16; }
    // + 8: the row stride in 16-byte pieces is 2 mod 4 - rows on even pieces, the odd quads beside them
    static constexpr int ldw(int l) { return steps(l) * 16 + 8; }
    static constexpr int w_floats(int l) { return nt(l) * 16 * ldw(l) + 3 * nt(l) * 16; }
    static constexpr int w_off(int l) {
        int o = 0;
        for (int i = 0; i < l; ++i) o += w_floats(i);
        return o;
    }
    static constexpr int p_off(int l) { return w_off(l) + nt(l) * 16 * ldw(l); }  // bias | gamma | beta
    static constexpr int patch_base() { return w_off(L_); }
    static constexpr int amax() {
        int m = 0;
        for (int l = 1; l < L_; ++l) m = in_per(l) > m ? in_per(l) : m;
        return m;
    }
    static constexpr int in_off(int l) { return l == 0 ? 0 : in_per(0); }  // deeper images overlay
    static constexpr int ppad() { return PP; }
    static constexpr int per_patch() { return in_per(0) + amax() + PP; }
    static constexpr int lds_floats() { return patch_base() + 8 * per_patch(); }
    static constexpr bool ok() {
        for (int l = 0; l < L_; ++l) {
            const int c = cpg(l);
            if (cout(l) % grp(l) != 0 || (c & (c - 1)) || c > 16 || (cout(l) & 3)) return false;
            if (l > 0 && (cin(l) & 3)) return false;
            if (mode(l) == 0 && (mt(l) > 3 || nt(l) > 2)) return false;
        }
        return cin(0) * F_ * F_ <= 512 && lds_floats() * 4 <= 160 * 1024;
    }
};

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter,
// i.e. every prefetch (next group's pixels, next layer's weight fragments) would be waited for at the
// next barrier (measured: 8.8 us per group).  Global stores need no ordering inside this kernel.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// one layer of cnn_fwd2_kernel for this wave (mode 0) or this workgroup (mode 1)
template <class N, int l>
__device__ __forceinline__ void fwd2_layer(const CnnFwdArgs& A, float* lds, float* region, int wave, int lane,
                                           int64_t row0, int nrow) {
    constexpr int P = N::P(l), cin = N::cin(l), cout = N::cout(l), K = N::K(l), hout = N::hout(l), G = N::grp(l);
    constexpr int cpg = N::cpg(l), rs = N::rs(l), cs = N::cs(l), ldw = N::ldw(l), steps = N::steps(l);
    constexpr int MT = N::mt(l), NT = N::nt(l), cp = NT * 16;
    constexpr bool last = l + 1 == N::L;
    constexpr int ln = last ? l : l + 1;
    constexpr int rs_n = N::rs(ln), cs_n = N::cs(ln), in_per_n = N::in_per(ln);
    constexpr float inv_cnt = 1.0f / (float)(P * cpg);
    const int quad = lane >> 4, l16 = lane & 15;
    const CnnFwdLayer& Ly = A.layer[l];
    const float* W = lds + N::w_off(l);
    const float* pvec = lds + N::p_off(l);
    const bool mine = wave < nrow;
    const int64_t prow = row0 + wave;
    if constexpr (N::mode(l) == 0) {
        // ================= a wave owns its patch: MT x NT tiles, k ascending
        if (!mine) return;
        const float* in = region + N::in_off(l);
        int rbase[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int m = mt * 16 + l16;
            m = m < P ? m : 0;
            rbase[mt] = 2 * (m / hout) * rs + 2 * (m % hout) * cs;
        }
        cf32x4 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = cf32x4{0.f, 0.f, 0.f, 0.f};
        // fragments of step kk + 1 are read while the matrix instructions of step kk issue
        float4 a[MT], b[NT], an[MT], bn[NT];
        auto frag = [&](int kk, float4 (&fa)[MT], float4 (&fb)[NT]) {
            const int k0 = kk * 16 + 4 * quad;
            if constexpr (l > 0) {  // cin % 4 == 0: the four k of a lane are one float4 along ci
                int tap = k0 / cin;
                const int ci = k0 - tap * cin;
                tap = tap < 9 ? tap : 8;
                const int off = (tap / 3) * rs + (tap % 3) * cs + ci;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const float4*>(in + rbase[mt] + off);
            } else {
                int off[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int k = k0 + j;
                    k = k < K ? k : K - 1;  // the weight column is zero there
                    const int tap = k / cin, ci = k - tap * cin;
                    off[j] = (tap / 3) * rs + (tap % 3) * cs + ci;
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    fa[mt] = make_float4(in[rbase[mt] + off[0]], in[rbase[mt] + off[1]], in[rbase[mt] + off[2]],
                                         in[rbase[mt] + off[3]]);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[nt] = *reinterpret_cast<const float4*>(W + (nt * 16 + l16) * ldw + k0);
        };
        frag(0, a, b);
#pragma unroll 2
        for (int kk = 0; kk < steps; ++kk) {
            frag(kk + 1 < steps ? kk + 1 : kk, an, bn);
            // k-major: consecutive matrix instructions never share an accumulator
#define MARL_F2_MFMA(q_)                                                                          \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt].q_, b[nt].q_, acc[mt][nt], 0, 0, 0);
            MARL_F2_MFMA(x) MARL_F2_MFMA(y) MARL_F2_MFMA(z) MARL_F2_MFMA(w)
#undef MARL_F2_MFMA
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = an[mt];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
        }
        // ---- bias, GroupNorm statistics from the registers (two passes), SiLU
        float* nxt = region + N::in_off(ln);
        if constexpr (!last)  // fresh zero border for the next layer's image (it may overlay this input)
            for (int i = lane; i < (in_per_n >> 2); i += 64)
                *reinterpret_cast<float4*>(nxt + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ch = nt * 16 + l16;
            const bool cv = ch < cout;
            const float bv = pvec[ch], gm = pvec[cp + ch], bt = pvec[2 * cp + ch];
            float s = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[mt][nt][r] += bv;
                    if (mt * 16 + 4 * quad + r < P) s += acc[mt][nt][r];
                }
            s = row_block_sum(cross_row_sum(s), cpg);
            const float mean = s * inv_cnt;
            float q = 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (mt * 16 + 4 * quad + r < P) {
                        const float d = acc[mt][nt][r] - mean;
                        q += d * d;
                    }
            q = row_block_sum(cross_row_sum(q), cpg);
            const float rstd = 1.0f / sqrtf(q * inv_cnt + 1e-5f);
            if (Ly.gst && cv && quad == 0 && (l16 & (cpg - 1)) == 0) {
                float* gs = Ly.gst + (prow * G + ch / cpg) * 2;
                gs[0] = mean;
                gs[1] = rstd;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mt * 16 + 4 * quad + r;
                    if (row < P && cv) {
                        const float zv = acc[mt][nt][r];
                        if (Ly.z) Ly.z[(prow * P + row) * (int64_t)cout + ch] = zv;
                        const float av = cnn_silu((zv - mean) * rstd * gm + bt);
                        if constexpr (last)
                            A.u[prow * (int64_t)A.ldu + ch * P + row] = av;
                        else
                            nxt[(row / hout + 1) * rs_n + (row % hout + 1) * cs_n + ch] = av;
                    }
                }
        }
    } else {
        // ================= 16-row tiles over 4 patches (P == 4), one (row tile, column tile) per turn
        lds_barrier();  // the other waves' images of this layer's input are complete
        constexpr int ntask = 2 * NT;  // 8 patches * 4 positions = 2 row tiles
        for (int task = wave; task < ntask; task += 8) {
            const int gi = task / NT, nt = task - gi * NT;
            const int lr_a = gi * 4 + (l16 >> 2), pos_a = l16 & 3;
            const float* in = lds + N::patch_base() + lr_a * N::per_patch() + N::in_off(l) +
                              2 * (pos_a / hout) * rs + 2 * (pos_a % hout) * cs;
            const float* wrow = W + (nt * 16 + l16) * ldw + 4 * quad;
            // two accumulators (even / odd 16-deep k steps): the single tile of this wave would
            // otherwise be one dependent chain of matrix instructions
            cf32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            auto fragA = [&](int kk) {
                const int k0 = kk * 16 + 4 * quad;
                int tap = k0 / cin;
                const int ci = k0 - tap * cin;
                tap = tap < 9 ? tap : 8;
                return *reinterpret_cast<const float4*>(in + (tap / 3) * rs + (tap % 3) * cs + ci);
            };
            float4 a0 = fragA(0), b0 = *reinterpret_cast<const float4*>(wrow);
            float4 a1 = fragA(1 < steps ? 1 : 0), b1 = *reinterpret_cast<const float4*>(wrow + (1 < steps ? 16 : 0));
#pragma unroll 2
            for (int kk = 0; kk < steps; kk += 2) {
                const int k2 = kk + 2 < steps ? kk + 2 : kk, k3 = kk + 3 < steps ? kk + 3 : kk;
                const float4 a2 = fragA(k2), b2 = *reinterpret_cast<const float4*>(wrow + k2 * 16);
                const float4 a3 = fragA(k3), b3 = *reinterpret_cast<const float4*>(wrow + k3 * 16);
                const bool odd = kk + 1 < steps;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
                if (odd) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
                if (odd) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
                if (odd) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
                if (odd) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
                a0 = a2;
                b0 = b2;
                a1 = a3;
                b1 = b3;
            }
            acc0 += acc1;
            // lane (quad, l16) holds patch gi * 4 + quad, positions r = 0..3, channel nt * 16 + l16
            const int lr_o = gi * 4 + quad;
            const int ch = nt * 16 + l16;
            const bool cv = ch < cout, pv = lr_o < nrow;
            const float bv = pvec[ch], gm = pvec[cp + ch], bt = pvec[2 * cp + ch];
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0[r] += bv;
                s += acc0[r];
            }
            s = row_block_sum(s, cpg);
            const float mean = s * inv_cnt;
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc0[r] - mean;
                q += d * d;
            }
            q = row_block_sum(q, cpg);
            const float rstd = 1.0f / sqrtf(q * inv_cnt + 1e-5f);
            const int64_t orow = row0 + lr_o;
            if (pv && cv) {
                if (Ly.gst && (l16 & (cpg - 1)) == 0) {
                    float* gs = Ly.gst + (orow * G + ch / cpg) * 2;
                    gs[0] = mean;
                    gs[1] = rstd;
                }
                float av[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float zv = acc0[r];
                    if (Ly.z) Ly.z[(orow * P + r) * (int64_t)cout + ch] = zv;
                    av[r] = cnn_silu((zv - mean) * rstd * gm + bt);
                    A.u[orow * (int64_t)A.ldu + ch * P + r] = av[r];
                }
                // (P == 4: the lane's four positions are four consecutive feature columns of U)
                if (A.u3)
                    img_store4(A.u3 + img_off(A.u3_row0 + orow, (ch * 4) >> 4, A.u3_steps), ch * 4, av[0], av[1], av[2], av[3]);
            }
        }
        lds_barrier();  // before the next chunk's layers overwrite the images read above
    }
}

template <class N>
__global__ __launch_bounds__(512) void cnn_fwd2_kernel(const CnnFwdArgs A, const int nchunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int f = N::F, ff = f * f, pe = N::cin(0) * ff;
    const float* imgf = static_cast<const float*>(A.img);
    const unsigned char* imgb = static_cast<const unsigned char*>(A.img);
    float* region = lds + N::patch_base() + wave * N::per_patch();  // this wave's patch
    constexpr int kPF = (pe + 63) / 64;  // gathered pixels per lane
    float pf[kPF];
#ifdef MARL_KERNEL_TS
    MARL_TS_DECL(A.ts);
#endif
    MARL_TS();
    int np0 = 0, np1 = 0;  // position of the patch whose pixels are fetched next
    auto load_pos = [&](int chunk) {
        const int64_t r = (int64_t)chunk * 8 + wave;
        if (chunk < nchunks && r < A.rows && !A.obs) {
            np0 = A.pos[r * 2];
            np1 = A.pos[r * 2 + 1];
        }
    };
    auto prefetch = [&](int chunk) {  // pixels of this wave's patch of `chunk` (position in np0 / np1)
        const int64_t r = (int64_t)chunk * 8 + wave;
        const bool have = chunk < nchunks && r < A.rows;
        const int64_t base = !have ? 0 : A.obs ? r * A.c_img * ff : (r % A.nb) * A.c_img * (int64_t)A.H * A.W;
#pragma unroll
        for (int i = 0; i < kPF; ++i) {
            const int e = lane + 64 * i;
            pf[i] = 0.f;
            if (have && e < pe) {
                if (A.obs) {
                    pf[i] = A.obs[base + e];
                } else {
                    const int ci = e / ff, e2 = e - ci * ff, iy = e2 / f, ix = e2 - iy * f;
                    const int64_t off = base + ((int64_t)ci * A.H + (np0 + iy)) * A.W + (np1 + ix);
                    pf[i] = A.img_u8 ? (float)imgb[off] / 255.0f : imgf[off];  // ToTensor on the fly
                }
            }
        }
    };
    load_pos(blockIdx.x);  // in flight while LDS is set up

    // ---- one-time: zero LDS (borders, weight padding), then all conv weights and the per-channel
    // vectors (bias | gamma | beta) -> LDS
    for (int i = tid; i < (N::lds_floats() >> 2); i += 512)
        *reinterpret_cast<float4*>(lds + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    auto stage_w = [&](auto lc) {
        constexpr int l = decltype(lc)::value;
        if constexpr (l < N::L) {
            const CnnFwdLayer& Ly = A.layer[l];
            constexpr int k4 = N::ldk(l) >> 2, cout = N::cout(l), nld = (cout * k4 + 511) / 512;
            // all loads of a layer in flight before the first LDS store (one L2 round trip, not nld)
            float4 wv[nld];
#pragma unroll
            for (int i = 0; i < nld; ++i) {
                const int idx = tid + 512 * i;
                wv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (idx < cout * k4) wv[i] = *reinterpret_cast<const float4*>(Ly.w + 4 * idx);
            }
#pragma unroll
            for (int i = 0; i < nld; ++i) {
                const int idx = tid + 512 * i;
                const int n = idx / k4, c = (idx - n * k4) * 4;
                if (idx < cout * k4) *reinterpret_cast<float4*>(lds + N::w_off(l) + n * N::ldw(l) + c) = wv[i];
            }
            float* pv = lds + N::p_off(l);
            constexpr int cp = N::nt(l) * 16;
            for (int c = tid; c < cout; c += 512) {
                pv[c] = Ly.bias[c];
                pv[cp + c] = Ly.gamma[c];
                pv[2 * cp + c] = Ly.beta[c];
            }
        }
    };
    stage_w(std::integral_constant<int, 0>{});
    stage_w(std::integral_constant<int, 1>{});
    stage_w(std::integral_constant<int, 2>{});
    prefetch(blockIdx.x);
    load_pos(blockIdx.x + gridDim.x);
    MARL_TS();
    __syncthreads();
    MARL_TS();

    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        MARL_TS();
        const int64_t row0 = (int64_t)chunk * 8;
        const int nrow = (int)(A.rows - row0 < 8 ? A.rows - row0 : 8);
        // ---- raw patch -> this wave's zero-bordered input image (HWC)
        {
            constexpr int rs = N::rs(0), cs = N::cs(0);
            float* in0 = region + N::in_off(0);
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                const int e = lane + 64 * i;
                if (e < pe) {
                    const int ci = e / ff, e2 = e - ci * ff, iy = e2 / f, ix = e2 - iy * f;
                    in0[(iy + 1) * rs + (ix + 1) * cs + ci] = pf[i];
                }
            }
        }
        prefetch(chunk + gridDim.x);           // the next chunk's pixels fly during this chunk's layers
        load_pos(chunk + 2 * (int)gridDim.x);  // and the position after that
        MARL_TS();
        fwd2_layer<N, 0>(A, lds, region, wave, lane, row0, nrow);
        MARL_TS();
        if constexpr (N::L > 1) fwd2_layer<N, 1>(A, lds, region, wave, lane, row0, nrow);
        MARL_TS();
        if constexpr (N::L > 2) fwd2_layer<N, 2>(A, lds, region, wave, lane, row0, nrow);
        MARL_TS();
    }
}

// the extractor shapes cnn_fwd2_kernel is built for
using Fwd2Resisc = Fwd2Net<12, 3, 3, 16, 32, 64, 0, 2, 4, 8, 0, 4, 28, 0, 4>;  // Resisc45Cnn / SkinCancerCnn, f = 12
using Fwd2Mnist6 = Fwd2Net<6, 2, 1, 8, 16, 0, 0, 2, 4, 1, 0>;           // MnistCnn, f = 6 (README)
using Fwd2Mnist12 = Fwd2Net<12, 2, 1, 8, 16, 0, 0, 2, 4, 1, 0, 28>;     // MnistCnn, f = 12 (the reference's tests)
static_assert(Fwd2Resisc::ok() && Fwd2Mnist6::ok() && Fwd2Mnist12::ok(), "fwd2 nets");

template <class N>
static bool fwd2_matches(const CnnFwdArgs& a) {
    if (a.L != N::L || a.f != N::F) return false;
    for (int l = 0; l < N::L; ++l) {
        const CnnFwdLayer& Ly = a.layer[l];
        if (Ly.cin != N::cin(l) || Ly.cout != N::cout(l) || Ly.G != N::grp(l) || Ly.ldk != N::ldk(l)) return false;
        if (Ly.cols) return false;  // im2col rows wanted (weight gradient outside the fused kernel's range)
    }
    return true;
}

template <class N>
static int fwd2_launch(CnnFwdArgs& a, hipStream_t st) {
    static bool raised = false;  // per process; one process drives one GPU
    auto kern = cnn_fwd2_kernel<N>;
    constexpr size_t lds = (size_t)N::lds_floats() * sizeof(float);
    if (!raised) {
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    const int nchunks = (int)cdiv(a.rows, 8);
    const int blocks = nchunks < 256 ? nchunks : 256;
#ifdef MARL_KERNEL_TS
    static long long* d_ts2 = nullptr;
    static int calls2 = 0;
    const int rec2 = ts_begin(&d_ts2, calls2++);
    a.ts = rec2 ? d_ts2 : nullptr;
#endif
    prof_before(3, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, st, a, nchunks);
    prof_after(3, st);
    MARL_LAUNCH_CHECK();
#ifdef MARL_KERNEL_TS
    if (rec2) ts_report("cnn_fwd2", d_ts2, 8);
#endif
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// cnn_fwd3_kernel: the same fused extractor for the DEEP shapes (AidCnn: four layers, window 24 /
// 32, 128 output channels; networks/vision.py:73-77).  A wave cannot own such a patch - its
// images alone are 25-40 KB and the last two layers' weights 370 KB - so here a WORKGROUP owns
// GP = 4 patches and every layer is cut into tasks (patch block x 16-channel tile) for its eight
// waves:
//   * a task holds all positions of its patches, so the GroupNorm statistics still come from the
//     accumulator registers (cross-lane sums), never through memory;
//   * layers >= 1 read their weight fragments straight from global memory (L2-resident), a block
//     of K steps ahead in registers; a task covers PB patches so that a fragment is used
//     PB x MT times; only layer 0's weights (2.5 KB) sit in LDS;
//   * the last layer (4 positions per patch) packs the 4 patches into one 16-row tile per
//     channel tile, as cnn_fwd2 does;
//   * images ping-pong between two LDS regions per patch; the next image's zero border is
//     written by all threads while the tasks' K loops run (barrier before the epilogues).
// Tasks per layer are 4 or 8: one per SIMD or two - the matrix pipe, which bounds this kernel
// (v_mfma_f32_16x16x4f32: 32 cycles per 2 KFLOP), stays evenly loaded.
// ---------------------------------------------------------------------------
template <class N, int GP_>
struct Fwd3Plan {
    static constexpr int GP = GP_, NWV = 8;
    // measured on C4 / C5: two patches per task at most (eight tasks keep two waves per SIMD busy: 45.7 vs
    // 49.7 us at C4) and ONE early weight block (two: the fragments held across the epilogue spill, 55 us)
    static constexpr int EARLY_BLOCKS = 1, PB_CAP = 2;
    static constexpr int max2(int a, int b) { return a > b ? a : b; }
    static constexpr int even_max() {
        int m = 0;
        for (int l = 0; l < N::L; l += 2) m = max2(m, N::in_per(l));
        return m;
    }
    static constexpr int odd_max() {
        int m = 0;
        for (int l = 1; l < N::L; l += 2) m = max2(m, N::in_per(l));
        return m;
    }
    static constexpr int reg_off(int l) { return (l & 1) ? even_max() : 0; }
    static constexpr int per_patch() { return even_max() + odd_max() + N::ppad(); }
    static constexpr int img_base() { return N::w_floats(0); }
    static constexpr int lds_floats() { return img_base() + GP * per_patch(); }
    // patches per task: <= 8 accumulator tiles, a power of two, <= PB_CAP
    static constexpr int pb(int l) {
        if (N::mode(l) == 1) return GP;
        int b = 8 / N::mt(l);
        b = b >= 4 ? 4 : b >= 2 ? 2 : 1;
        b = b < PB_CAP ? b : PB_CAP;
        return b < GP ? b : GP;
    }
    static constexpr int ntask(int l) { return N::mode(l) == 1 ? N::nt(l) : (GP / pb(l)) * N::nt(l); }
    // K steps per weight-fragment block (registers: 4 per step): the largest divisor of steps <= 12
    static constexpr int kb(int l) {
        if (l <= 0 || l >= N::L) return 1;
        int b = 1;
        for (int d = 1; d <= 12; ++d)
            if (N::steps(l) % d == 0) b = d;
        return b;
    }
    static constexpr int nblk(int l) { return N::steps(l) / kb(l); }
    // K steps of layer l whose weight fragments the PREVIOUS layer requests (two blocks at most)
    static constexpr int early(int l) {
        if (l <= 0 || l >= N::L) return 1;
        return (nblk(l) < EARLY_BLOCKS ? nblk(l) : EARLY_BLOCKS) * kb(l);
    }
    static constexpr int bsteps(int l) { return (l <= 0 || l >= N::L) ? 1 : N::steps(l); }
    static constexpr bool ok() {
        if (N::L != 4 || N::mode(N::L - 1) != 1 || GP != 4) return false;
        for (int l = 0; l < N::L; ++l) {
            const int c = N::cpg(l);
            if (N::cout(l) % N::grp(l) != 0 || (c & (c - 1)) || c > 16 || (N::cout(l) & 15)) return false;
            if (l > 0 && (N::cin(l) & 15)) return false;  // a 16-deep K step stays inside one tap
            if (ntask(l) > NWV || N::mt(l) > 16) return false;
        }
        return lds_floats() * 4 <= 160 * 1024;
    }
};

#ifdef MARL_KERNEL_TS
#define MARL_F3_TS() if (ts_ && (threadIdx.x & 63) == 0 && blockIdx.x == 37 && tsi_ < 48) ts_[(threadIdx.x >> 6) * 48 + tsi_++] = wall_clock64()
#define MARL_F3_TSARGS , long long* ts_, int& tsi_
#define MARL_F3_TSPASS , ts_, tsi_
#else
#define MARL_F3_TS()
#define MARL_F3_TSARGS
#define MARL_F3_TSPASS
#endif

// this wave's weight-fragment pointer in layer l (global memory, fragment order - CnnFwdLayer::wfrag: a K
// step of a 16-channel tile is 1 KB contiguous, one float4 per lane; from the row-major copy a wave's load
// touched 16 rows x 64 B, half of every cache line): null when the wave has no task there
template <class N, class PL, int l>
__device__ __forceinline__ const float* fwd3_wrow(const CnnFwdArgs& A, int wave, int lane) {
    if constexpr (l <= 0 || l >= N::L) {
        return nullptr;
    } else {
        if (wave >= PL::ntask(l)) return nullptr;
        const int nt = N::mode(l) == 1 ? wave : wave % N::nt(l);
        return A.layer[l].wfrag + ((size_t)nt * N::steps(l) * 64 + lane) * 4;  // + 256 floats per K step
    }
}
// K steps [lo, hi) of a task's weight fragments -> registers
template <int LO, int HI, int NB>
__device__ __forceinline__ void fwd3_loadb(float4 (&b)[NB], const float* wrow) {
#pragma unroll
    for (int j = LO; j < HI; ++j) b[j] = *reinterpret_cast<const float4*>(wrow + j * 256);
}

// bw: this layer's weight fragments, one float4 per K step - the first PL::early(l) steps were requested
// by the previous layer, the rest is requested at the head of the K loop; bwn: the next layer's, whose
// early steps are requested here between the K loop and the epilogue (in flight across the epilogue
// and two barriers).  (hipcc moves a one-block-ahead prefetch written as a rotating pair of register
// sets down to its first use; every block therefore has its own registers.)
template <class N, class PL, int l, int KB0, int KBN>
__device__ __forceinline__ void fwd3_layer(const CnnFwdArgs& A, float* lds, int tid, int64_t row0, int nrow,
                                           float4 (&bw)[KB0], float4 (&bwn)[KBN] MARL_F3_TSARGS) {
    constexpr int P = N::P(l), cin = N::cin(l), cout = N::cout(l), K = N::K(l), hout = N::hout(l), G = N::grp(l);
    constexpr int cpg = N::cpg(l), rs = N::rs(l), cs = N::cs(l), steps = N::steps(l);
    constexpr int MT = N::mt(l), NT = N::nt(l), PB = PL::pb(l), NTASK = PL::ntask(l);
    static_assert(l == 0 || KB0 == steps, "one fragment per K step");
    constexpr bool last = l + 1 == N::L;
    constexpr int ln = last ? 0 : l + 1;  // the region zeroed during this layer: next image / next group's input
    constexpr int rs_n = N::rs(ln), cs_n = N::cs(ln);
    constexpr float inv_cnt = 1.0f / (float)(P * cpg);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = lane >> 4, l16 = lane & 15;
    const CnnFwdLayer& Ly = A.layer[l];
    float* const img = lds + PL::img_base();
    const bool has_task = wave < NTASK;
    const float* const wrow = fwd3_wrow<N, PL, l>(A, wave, lane);
    const float* const wrow_n = fwd3_wrow<N, PL, l + 1>(A, wave, lane);

    // ---- every thread: zero border (whole image) of the region this layer's epilogues write into
    {
        constexpr int n4 = N::in_per(ln) >> 2;
        for (int i = tid; i < PL::GP * n4; i += 512) {
            const int p = i / n4, r = i - p * n4;
            *reinterpret_cast<float4*>(img + p * PL::per_patch() + PL::reg_off(ln) + 4 * r) =
                make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if constexpr (N::mode(l) == 0) {
        const int nt = has_task ? wave % NT : 0, pg = has_task ? wave / NT : 0;
        const int ch = nt * 16 + l16;
        constexpr int KA = (MT == 1 && l > 0) ? 2 : 1;  // a single tile: even / odd K steps in two chains
        cf32x4 acc[KA][PB][MT];
        if (has_task) {
#pragma unroll
            for (int a = 0; a < KA; ++a)
#pragma unroll
                for (int p = 0; p < PB; ++p)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[a][p][mt] = cf32x4{0.f, 0.f, 0.f, 0.f};
            int rbase[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                int m = mt * 16 + l16;
                m = m < P ? m : 0;
                rbase[mt] = 2 * (m / hout) * rs + 2 * (m % hout) * cs;
            }
            const float* in0 = img + (pg * PB) * PL::per_patch() + PL::reg_off(l);
            if constexpr (l == 0) {
                // K = 9 * cin (27): scalar taps, weights in LDS
                const float* W = lds + (nt * 16 + l16) * N::ldw(0);
#pragma unroll
                for (int kk = 0; kk < steps; ++kk) {
                    const int k0 = kk * 16 + 4 * quad;
                    int off[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        int k = k0 + j;
                        k = k < K ? k : K - 1;  // the weight column is zero there
                        const int tap = k / cin, ci = k - tap * cin;
                        off[j] = (tap / 3) * rs + (tap % 3) * cs + ci;
                    }
                    const float4 b = *reinterpret_cast<const float4*>(W + k0);
                    constexpr int MB = MT < 4 ? MT : 4;
#pragma unroll
                    for (int p = 0; p < PB; ++p)
#pragma unroll
                        for (int m0 = 0; m0 < MT; m0 += MB) {
                            float4 a[MB];
#pragma unroll
                            for (int i = 0; i < MB; ++i) {
                                const float* q = in0 + p * PL::per_patch() + rbase[m0 + i < MT ? m0 + i : MT - 1];
                                a[i] = make_float4(q[off[0]], q[off[1]], q[off[2]], q[off[3]]);
                            }
#define MARL_F3_MFMA(q_)                                                                      \
    _Pragma("unroll") for (int i = 0; i < MB; ++i) if (m0 + i < MT)                           \
        acc[0][p][m0 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].q_, b.q_, acc[0][p][m0 + i], 0, 0, 0);
                            MARL_F3_MFMA(x) MARL_F3_MFMA(y) MARL_F3_MFMA(z) MARL_F3_MFMA(w)
#undef MARL_F3_MFMA
                        }
                }
            } else {
                fwd3_loadb<PL::early(l), steps>(bw, wrow);
                {
#pragma unroll
                    for (int kk = 0; kk < steps; ++kk) {
                        const int k0 = kk * 16;
                        const int tap = k0 / cin, ci = k0 - tap * cin;
                        const int off = (tap / 3) * rs + (tap % 3) * cs + ci + 4 * quad;
#pragma unroll
                        for (int p = 0; p < PB; ++p) {
                            float4 a[MT];
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                a[mt] = *reinterpret_cast<const float4*>(in0 + p * PL::per_patch() + rbase[mt] + off);
#define MARL_F3_MFMA(q_)                                                                      \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                         \
        acc[KA == 2 ? (kk & 1) : 0][p][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(            \
            a[mt].q_, bw[kk].q_, acc[KA == 2 ? (kk & 1) : 0][p][mt], 0, 0, 0);
                            MARL_F3_MFMA(x) MARL_F3_MFMA(y) MARL_F3_MFMA(z) MARL_F3_MFMA(w)
#undef MARL_F3_MFMA
                        }
                    }
                }
                if constexpr (KA == 2)
#pragma unroll
                    for (int p = 0; p < PB; ++p) acc[0][p][0] += acc[KA - 1][p][0];
            }
        }
        if (wrow_n) fwd3_loadb<0, PL::early(l + 1)>(bwn, wrow_n);
        // (the per-channel vectors are requested ahead of the barrier too)
        const float bv = l == 0 ? lds[N::p_off(0) + ch] : Ly.bias[ch];
        const float gm = l == 0 ? lds[N::p_off(0) + NT * 16 + ch] : Ly.gamma[ch];
        const float bt = l == 0 ? lds[N::p_off(0) + 2 * NT * 16 + ch] : Ly.beta[ch];
        MARL_F3_TS();
        lds_barrier();  // the zero borders are complete before any epilogue writes an interior
        MARL_F3_TS();
        if (has_task) {
            // ---- bias, GroupNorm statistics from the registers (two passes), SiLU  (as cnn_fwd2)
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const int lp = pg * PB + p;
                const int64_t prow = row0 + lp;
                float s = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[0][p][mt][r] += bv;
                        if (mt * 16 + 4 * quad + r < P) s += acc[0][p][mt][r];
                    }
                s = row_block_sum(cross_row_sum(s), cpg);
                const float mean = s * inv_cnt;
                float q = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (mt * 16 + 4 * quad + r < P) {
                            const float d = acc[0][p][mt][r] - mean;
                            q += d * d;
                        }
                q = row_block_sum(cross_row_sum(q), cpg);
                const float rstd = 1.0f / sqrtf(q * inv_cnt + 1e-5f);
                const bool mine = lp < nrow;
                if (mine && Ly.gst && quad == 0 && (l16 & (cpg - 1)) == 0) {
                    float* gs = Ly.gst + (prow * G + ch / cpg) * 2;
                    gs[0] = mean;
                    gs[1] = rstd;
                }
                float* nxt = img + lp * PL::per_patch() + PL::reg_off(ln);
                float* zp = Ly.z ? Ly.z + prow * (int64_t)(P * cout) + ch : nullptr;  // + row * cout
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = mt * 16 + 4 * quad + r;
                        if (row < P) {
                            const float zv = acc[0][p][mt][r];
                            if (mine && zp) zp[row * cout] = zv;
                            const float av = cnn_silu((zv - mean) * rstd * gm + bt);
                            if constexpr (last) {
                                if (mine) A.u[prow * (int64_t)A.ldu + ch * P + row] = av;
                            } else {
                                nxt[(row / hout + 1) * rs_n + (row % hout + 1) * cs_n + ch] = av;
                            }
                        }
                    }
            }
        }
        MARL_F3_TS();
        lds_barrier();  // next layer reads the images
        MARL_F3_TS();
    } else {
        // ================= last layer, P == 4: one 16-row tile = the 4 patches x 4 positions
        static_assert(PL::GP == 4 && last, "packed last layer");
        if (has_task) {
            const int nt = wave;
            const int ch = nt * 16 + l16;
            const float bv = Ly.bias[ch], gm = Ly.gamma[ch], bt = Ly.beta[ch];
            const int lr_a = l16 >> 2, pos_a = l16 & 3;
            const float* in = img + lr_a * PL::per_patch() + PL::reg_off(l) +
                              2 * (pos_a / hout) * rs + 2 * (pos_a % hout) * cs + 4 * quad;
            cf32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            fwd3_loadb<PL::early(l), steps>(bw, wrow);
            constexpr int AB = 4;  // image fragments read per batch
            static_assert(steps % AB == 0, "K steps per batch");
#pragma unroll
            for (int kb = 0; kb < steps; kb += AB) {
                float4 a[AB];
#pragma unroll
                for (int j = 0; j < AB; ++j) {
                    const int k0 = (kb + j) * 16;
                    const int tap = k0 / cin, ci = k0 - tap * cin;
                    a[j] = *reinterpret_cast<const float4*>(in + (tap / 3) * rs + (tap % 3) * cs + ci);
                }
#define MARL_F3_MFMA(q_)                                                                      \
    _Pragma("unroll") for (int j = 0; j < AB; ++j) {                                          \
        if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].q_, bw[kb + j].q_, acc1, 0, 0, 0); \
        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].q_, bw[kb + j].q_, acc0, 0, 0, 0); \
    }
                MARL_F3_MFMA(x) MARL_F3_MFMA(y) MARL_F3_MFMA(z) MARL_F3_MFMA(w)
#undef MARL_F3_MFMA
            }
            acc0 += acc1;
            // lane (quad, l16) holds patch quad, positions r = 0..3, channel nt * 16 + l16
            const bool pv = quad < nrow;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc0[r] += bv;
                s += acc0[r];
            }
            s = row_block_sum(s, cpg);
            const float mean = s * inv_cnt;
            float q = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc0[r] - mean;
                q += d * d;
            }
            q = row_block_sum(q, cpg);
            const float rstd = 1.0f / sqrtf(q * inv_cnt + 1e-5f);
            const int64_t orow = row0 + quad;
            if (pv) {
                if (Ly.gst && (l16 & (cpg - 1)) == 0) {
                    float* gs = Ly.gst + (orow * G + ch / cpg) * 2;
                    gs[0] = mean;
                    gs[1] = rstd;
                }
                float av[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float zv = acc0[r];
                    if (Ly.z) Ly.z[(orow * P + r) * (int64_t)cout + ch] = zv;
                    av[r] = cnn_silu((zv - mean) * rstd * gm + bt);
                    A.u[orow * (int64_t)A.ldu + ch * P + r] = av[r];
                }
                if (A.u3)  // (as in cnn_fwd2: four positions = four consecutive feature columns of U)
                    img_store4(A.u3 + img_off(A.u3_row0 + orow, (ch * 4) >> 4, A.u3_steps), ch * 4, av[0], av[1], av[2], av[3]);
            }
        }
        MARL_F3_TS();
        lds_barrier();  // region 0 is zeroed and free: the next group's pixels may land
        MARL_F3_TS();
    }
}

template <class N, class PL>
__global__ __launch_bounds__(512) void cnn_fwd3_kernel(const CnnFwdArgs A, const int ngroups) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    constexpr int GP = PL::GP, f = N::F, ff = f * f, pe = N::cin(0) * ff;
    constexpr int kPF = (GP * pe + 511) / 512;  // gathered pixels per thread and group
    const float* imgf = static_cast<const float*>(A.img);
    const unsigned char* imgb = static_cast<const unsigned char*>(A.img);
    float pf[kPF];
#ifdef MARL_KERNEL_TS
    MARL_TS_DECL(A.ts);
#endif
    MARL_F3_TS();
    // Pixels of the GP patches of group `grp`.  Every load is UNCONDITIONAL with clamped indices (a load
    // behind a branch makes hipcc wait for each one in turn - measured: 12 us for the first group's 24
    // loads per lane); rows past the end re-read the last patch, their results are never stored.
    auto prefetch = [&](int grp) {
        const int64_t r0 = (int64_t)(grp < ngroups ? grp : ngroups - 1) * GP;
        int py[GP], px[GP], rr[GP], ib[GP];  // position, patch row, image of the patch (one division per patch)
#pragma unroll
        for (int p = 0; p < GP; ++p) {
            const int64_t r = r0 + p < A.rows ? r0 + p : A.rows - 1;
            rr[p] = (int)r;
            ib[p] = (int)((uint32_t)r % (uint32_t)A.nb);
            py[p] = A.obs ? 0 : A.pos[r * 2];
            px[p] = A.obs ? 0 : A.pos[r * 2 + 1];
        }
#define MARL_F3_INDEX(i_)                                                                     \
    int e = tid + 512 * (i_);                                                                 \
    asm volatile("" : "+v"(e)); /* recomputed per group: the index chains are not kept in registers */ \
    e = e < GP * pe ? e : GP * pe - 1;                                                        \
    const int p = e / pe, e1 = e - p * pe;                                                    \
    int y0 = py[0], x0 = px[0], r = rr[0], b = ib[0];                                         \
    _Pragma("unroll") for (int q = 1; q < GP; ++q) if (p == q) {                              \
        y0 = py[q];                                                                           \
        x0 = px[q];                                                                           \
        r = rr[q];                                                                            \
        b = ib[q];                                                                            \
    }                                                                                         \
    const int ci = e1 / ff, e2 = e1 - ci * ff, iy = e2 / f, ix = e2 - iy * f;                 \
    const int64_t oi = (int64_t)(b * A.c_img + ci) * ((int64_t)A.H * A.W) + ((y0 + iy) * A.W + (x0 + ix)); \
    const int64_t oo = (int64_t)r * (A.c_img * ff) + e1;                                      \
    (void)oi;                                                                                 \
    (void)oo;
        if (A.obs) {
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                MARL_F3_INDEX(i)
                pf[i] = A.obs[oo];
            }
        } else if (A.img_u8) {
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                MARL_F3_INDEX(i)
                pf[i] = (float)imgb[oi];  // (/ 255 when the pixel goes to LDS: ToTensor on the fly)
            }
        } else {
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                MARL_F3_INDEX(i)
                pf[i] = imgf[oi];
            }
        }
#undef MARL_F3_INDEX
    };
    prefetch(blockIdx.x);  // in flight while LDS is set up
    // ---- one-time: zero layer 0's weight area and the patches' first region (borders), then layer 0's
    // weights and per-channel vectors -> LDS
    for (int i = tid; i < (PL::img_base() >> 2); i += 512)
        *reinterpret_cast<float4*>(lds + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        constexpr int n4 = N::in_per(0) >> 2;
        for (int i = tid; i < GP * n4; i += 512) {
            const int p = i / n4, r = i - p * n4;
            *reinterpret_cast<float4*>(lds + PL::img_base() + p * PL::per_patch() + 4 * r) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    lds_barrier();
    {
        const CnnFwdLayer& Ly = A.layer[0];
        constexpr int k4 = N::ldk(0) >> 2, cout = N::cout(0), cp = N::nt(0) * 16;
        for (int idx = tid; idx < cout * k4; idx += 512) {
            const int n = idx / k4, c = (idx - n * k4) * 4;
            *reinterpret_cast<float4*>(lds + n * N::ldw(0) + c) = *reinterpret_cast<const float4*>(Ly.w + 4 * idx);
        }
        float* pv = lds + N::p_off(0);
        for (int c = tid; c < cout; c += 512) {
            pv[c] = Ly.bias[c];
            pv[cp + c] = Ly.gamma[c];
            pv[2 * cp + c] = Ly.beta[c];
        }
    }
    MARL_F3_TS();

    float4 bw1[PL::bsteps(1)], bw2[PL::bsteps(2)], bw3[PL::bsteps(3)], bnone[1];
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t row0 = (int64_t)grp * GP;
        const int nrow = (int)(A.rows - row0 < GP ? A.rows - row0 : GP);
        // ---- raw patches -> zero-bordered input images (HWC) in region 0
        {
            constexpr int rs = N::rs(0), cs = N::cs(0);
#pragma unroll
            for (int i = 0; i < kPF; ++i) {
                int e = tid + 512 * i;
                asm volatile("" : "+v"(e));
                const int p = e / pe, e1 = e - p * pe;
                if (p < GP) {
                    const int ci = e1 / ff, e2 = e1 - ci * ff, iy = e2 / f, ix = e2 - iy * f;
                    lds[PL::img_base() + p * PL::per_patch() + (iy + 1) * rs + (ix + 1) * cs + ci] =
                        (A.img_u8 && !A.obs) ? pf[i] / 255.0f : pf[i];
                }
            }
        }
        MARL_F3_TS();
        prefetch(grp + gridDim.x);  // the next group's pixels fly during this group's layers
        lds_barrier();
        MARL_F3_TS();
        // an opaque copy of the thread index per layer: everything derived from it (tile bases, zeroing
        // addresses, weight pointers of FOUR layers) would otherwise be hoisted out of the group loop
        // and spilled (measured: 146 scratch stores ahead of the loop, 75 reloads in one epilogue)
        int t0 = tid, t1 = tid, t2 = tid, t3 = tid;
        asm volatile("" : "+v"(t0));
        fwd3_layer<N, PL, 0>(A, lds, t0, row0, nrow, bnone, bw1 MARL_F3_TSPASS);
        asm volatile("" : "+v"(t1));
        fwd3_layer<N, PL, 1>(A, lds, t1, row0, nrow, bw1, bw2 MARL_F3_TSPASS);
        asm volatile("" : "+v"(t2));
        fwd3_layer<N, PL, 2>(A, lds, t2, row0, nrow, bw2, bw3 MARL_F3_TSPASS);
        asm volatile("" : "+v"(t3));
        fwd3_layer<N, PL, 3>(A, lds, t3, row0, nrow, bw3, bnone MARL_F3_TSPASS);
    }
}

using Fwd3Aid24 = Fwd2Net<24, 4, 3, 16, 32, 64, 128, 2, 4, 8, 16, 0, 20, 28, 0, 4>;  // AidCnn, f = 24 (configs[3])
using Fwd3Aid32 = Fwd2Net<32, 4, 3, 16, 32, 64, 128, 2, 4, 8, 16, 4, 8, 24, 8, 0>;   // AidCnn, f = 32 (configs[4])
using Plan3Aid24 = Fwd3Plan<Fwd3Aid24, 4>;
using Plan3Aid32 = Fwd3Plan<Fwd3Aid32, 4>;
static_assert(Plan3Aid24::ok() && Plan3Aid32::ok(), "fwd3 nets");

template <class N, class PL>
static int fwd3_launch(CnnFwdArgs& a, hipStream_t st) {
    static bool raised = false;  // per process; one process drives one GPU
    auto kern = cnn_fwd3_kernel<N, PL>;
    constexpr size_t lds = (size_t)PL::lds_floats() * sizeof(float);
    if (!raised) {
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised = true;
    }
    const int ngroups = (int)cdiv(a.rows, PL::GP);
    const int blocks = ngroups < 256 ? ngroups : 256;
#ifdef MARL_KERNEL_TS
    static long long* d_ts3 = nullptr;
    static int calls3 = 0;
    const int rec3 = ts_begin(&d_ts3, calls3++);
    a.ts = rec3 ? d_ts3 : nullptr;
#endif
    prof_before(3, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, st, a, ngroups);
    prof_after(3, st);
    MARL_LAUNCH_CHECK();
#ifdef MARL_KERNEL_TS
    if (rec3) ts_report("cnn_fwd3", d_ts3, 8);
#endif
    return MARL_OK;
}

// 0 = not covered, else 1 + index of the instantiation
static int cnn_fwd2_which(const CnnFwdArgs& a) {
    if (!tune_get("cnn_fwd2", 1)) return 0;
    if (fwd2_matches<Fwd2Resisc>(a)) return 1;
    if (fwd2_matches<Fwd2Mnist6>(a)) return 2;
    if (fwd2_matches<Fwd2Mnist12>(a)) return 3;
    if (tune_get("cnn_fwd3", 1)) {
        for (int l = 1; l < a.L; ++l)
            if (!a.layer[l].wfrag) return 0;  // (no weights workspace behind this call)
        if (fwd2_matches<Fwd3Aid24>(a)) return 4;
        if (fwd2_matches<Fwd3Aid32>(a)) return 5;
    }
    return 0;
}

// LDS floats for rb patches per workgroup; fills the launcher-owned fields of `a`
static size_t cnn_fwd_plan(CnnFwdArgs& a, int rb) {
    size_t b0 = 0, b1 = 0, st = 0;
    for (int l = 0; l < a.L; ++l) {
        const CnnFwdLayer& L = a.layer[l];
        a.dP[l] = make_fdiv(L.P);
        a.dhout[l] = make_fdiv(L.hout);
        a.dcin[l] = make_fdiv(L.cin);
        a.dcpg[l] = make_fdiv(L.cout / L.G);
        a.dG[l] = make_fdiv(L.G);
        a.dNT[l] = make_fdiv((L.cout + 15) / 16);
        a.dc4o[l] = make_fdiv(L.cout / 4);
        const size_t z = (size_t)rb * L.P * (L.cout + 4);
        if (l & 1)
            b1 = z > b1 ? z : b1;
        else
            b0 = z > b0 ? z : b0;
        const size_t g = (size_t)rb * L.G * 2;
        st = g > st ? g : st;
    }
    const int ff = a.f * a.f;
    size_t patch = (size_t)rb * a.layer[0].cin * ff;
    patch = (patch + 3) & ~(size_t)3;
    a.rb = rb;
    a.dpe = make_fdiv(a.layer[0].cin * ff);
    a.dff = make_fdiv(ff);
    a.df = make_fdiv(a.f);
    a.dE = make_fdiv(a.layer[a.L - 1].P * a.layer[a.L - 1].cout);
    a.off_b0 = (int)patch;
    a.off_b1 = (int)(patch + b0);
    a.off_stat = (int)(patch + b0 + b1);
    return patch + b0 + b1 + st;
}

int cnn_fwd_writes_image(const CnnFwdArgs& a) {
    // Fwd2Resisc / Fwd2Mnist6 (mode-1 last layer) and the two AidCnn plans (packed last layer): the last
    // layer has four positions per patch, a lane's four outputs are four consecutive columns of U
    const int w = cnn_fwd2_which(a);
    return w == 1 || w == 2 || w == 4 || w == 5;
}

int cnn_fwd_supported(const CnnFwdArgs& a0) {
    if (getenv("MARL_CNN_FUSED") && getenv("MARL_CNN_FUSED")[0] == '0') return 0;
    CnnFwdArgs a = a0;
    if (cnn_fwd2_which(a)) return 1;
    for (int l = 0; l < a.L; ++l) {
        const CnnFwdLayer& L = a.layer[l];
        if (L.cout % L.G != 0 || ((L.cout / L.G) & 3) || (L.cout & 3)) return 0;  // float4 stays inside a group
        if (l > 0 && (L.cin & 3)) return 0;
    }
    return cnn_fwd_plan(a, 1) * sizeof(float) <= 144 * 1024;
}

int launch_cnn_fwd(CnnFwdArgs& a, hipStream_t st) {
    if (a.rows <= 0) return MARL_OK;
    switch (cnn_fwd2_which(a)) {
        case 1: return fwd2_launch<Fwd2Resisc>(a, st);
        case 2: return fwd2_launch<Fwd2Mnist6>(a, st);
        case 3: return fwd2_launch<Fwd2Mnist12>(a, st);
        case 4: return fwd3_launch<Fwd3Aid24, Plan3Aid24>(a, st);
        case 5: return fwd3_launch<Fwd3Aid32, Plan3Aid32>(a, st);
        default: break;
    }
    // as many patches per workgroup as fit three workgroups per CU (the conv weights are
    // re-read from L2 by every workgroup), but keep >= 256 workgroups
    constexpr int rb_max = 8, lds_cap = 52;  // patches per workgroup (<= 16: per-patch LDS tables), LDS cap in KB
    int rb = rb_max;
    while (rb > 1 && (cnn_fwd_plan(a, rb) * sizeof(float) > (size_t)lds_cap * 1024 || cdiv(a.rows, rb) < 256)) --rb;
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
#ifdef MARL_KERNEL_TS
    static long long* d_ts = nullptr;
    static int calls = 0;
    const int rec = ts_begin(&d_ts, calls++);
    a.ts = rec ? d_ts : nullptr;
#endif
    prof_before(3, st);
    hipLaunchKernelGGL(cnn_fwd_kernel, dim3((unsigned)cdiv(a.rows, rb)), dim3(512), lds, st, a);
    prof_after(3, st);
    MARL_LAUNCH_CHECK();
#ifdef MARL_KERNEL_TS
    if (rec) ts_report("cnn_fwd", d_ts, 8);
#endif
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// Fused layer backward: for rb patches per workgroup,
//   dA_{l-1}[p][ci] = sum over taps, c of dZ_l[o(p, tap)][c] * W_l[c][tap][ci]
// as 16x16x4 f32 MFMA tiles (the transposed convolution in gather form: input positions are
// sorted by (row parity, column parity) so that a 16-row tile shares its set of valid taps and
// the other taps are skipped), followed in LDS by the GroupNorm + SiLU backward of layer l-1.
// Replaces the dCOLS GEMM, col2im and the GroupNorm-backward launch: dCOLS and dA never exist
// in HBM.  (Backward of networks/vision.py:33-38 for Conv2d(3, stride 2, pad 1) + GroupNorm.)
// ---------------------------------------------------------------------------
constexpr int kDgradTiles = 4;  // row tiles per wave

// v_exp_f32 / v_rcp_f32 based (~1e-7 relative error, inside the 1e-5 parity budget)
__device__ __forceinline__ float cnn_silu_grad(float y) {
    return silu_grad_fast(y);
}

// W0 (the launch that produces dZ_0): the FIRST layer's weight gradient is formed right here from the dZ_0 panel in
// LDS and the raw image patch (gathered at the saved positions into a zero-bordered LDS image), as in
// cnn_wgrad_kernel<.., FIRST>: dZ_0 - the largest activation gradient of the network (1 GB at BASELINE configs[4]) -
// is neither written nor read back, and the separate first-layer launch disappears.
template <bool W0>
__global__ __launch_bounds__(512, W0 ? 4 : 1) void cnn_dgrad_kernel(const CnnDgradArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Dz = lds;                    // [rb * P][cout + 4]
    float* Zin = lds + A.off_zin;       // [rb * Pin][cin + 4]
    float* Da = lds + A.off_da;         // [rb * Pin][cin + 4]   dA, then dZ in place
    int* perm = reinterpret_cast<int*>(lds + A.off_perm);  // [MT * 16] lr << 16 | py << 8 | px, or -1
    float* gstat = lds + A.off_stat;    // [rb * G][2]
    float* gsum = lds + A.off_gsum;     // [nwaves][2][cpg]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int quad = lane >> 4, l16 = lane & 15;
    const int cin = A.cin, cout = A.cout, hin = A.hin, hout = A.hout, P = A.P, Pin = A.Pin, G = A.G;
    const int zs = cout + 4, cs = cin + 4, cpg = (int)A.dcpg.d;
    // PERSISTENT over chunks of rb patches (round 4): the row order of the tiles, the tap sets and the
    // lane roles are the same for every full chunk - building them was a third of the kernel's VALU
    // instructions when every chunk was its own workgroup - the next chunk is staged while this one is
    // written out, and the affine partial sums of a workgroup stay in registers across its chunks
    // (gridDim.x partial rows instead of one per chunk; the order of the sums is fixed by the grid).
    const int nchunks = (int)((A.rows + A.rb - 1) / A.rb);
    const int full_rows = A.rb;

#ifdef MARL_KERNEL_TS
    MARL_TS_DECL(A.ts);
#endif
    MARL_TS();
    // ---- dZ_l, Z_{l-1} and the statistics of one chunk -> LDS
    auto stage = [&](int chunk) {
        const int64_t row0 = (int64_t)chunk * full_rows;
        const int nrow = (int)(A.rows - row0 < full_rows ? A.rows - row0 : full_rows);
        const int M = nrow * P, c4 = cout >> 2;
        const float* src = A.dz + row0 * P * (int64_t)cout;
        for (int idx = tid; idx < M * c4; idx += nthreads) {
            const int m = fdiv(idx, A.dc4o), k = (idx - m * c4) * 4;
            *reinterpret_cast<float4*>(Dz + m * zs + k) = *reinterpret_cast<const float4*>(src + (int64_t)m * cout + k);
        }
        const int Mi = nrow * Pin, i4 = cin >> 2;
        const float* zsrc = A.zin + row0 * Pin * (int64_t)cin;
        for (int idx = tid; idx < Mi * i4; idx += nthreads) {
            const int m = fdiv(idx, A.dc4i), k = (idx - m * i4) * 4;
            *reinterpret_cast<float4*>(Zin + m * cs + k) = *reinterpret_cast<const float4*>(zsrc + (int64_t)m * cin + k);
        }
        for (int idx = tid; idx < nrow * G * 2; idx += nthreads) gstat[idx] = A.gst[row0 * G * 2 + idx];
        if constexpr (W0) {  // raw patches of the chunk -> interior of the zero-bordered images [lr][y + 1][x + 1][ci]
            // Batches of kU elements per thread with every load of a batch in flight at once (positions first, then
            // pixels; clamped indices, predicated LDS writes) instead of two dependent global round trips per element.
            // (Measured: the fused form still loses to the separate first-layer launch - C3 +0.08...0.16 ms, C5 +0.16,
            // C4 +0.06 - so the gather was not what costs; see cnn_dgrad_w0_ok.)
            float* Pix = lds + A.off_pix;
            const int f0 = A.f0, ff = f0 * f0, pe = A.cin0 * ff, tot = nrow * pe;
            const int64_t plane = (int64_t)A.c_img * A.H * A.W;
            const float* imgf = static_cast<const float*>(A.img);
            const unsigned char* imgb = static_cast<const unsigned char*>(A.img);
            constexpr int kU = 4;
            for (int base = tid; base < tot; base += kU * nthreads) {
                int lo[kU], p0[kU], p1[kU], go[kU], img_i[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    int idx = base + u * nthreads;
                    idx = idx < tot ? idx : tot - 1;
                    const int lr = fdiv(idx, A.dpe0), e = idx - lr * pe;
                    const int ci = fdiv(e, A.dff0), e2 = e - ci * ff;
                    const int iy = fdiv(e2, A.df0), ix = e2 - iy * f0;
                    const int64_t r = row0 + lr;
                    p0[u] = A.pos[r * 2];
                    p1[u] = A.pos[r * 2 + 1];
                    img_i[u] = (int)(r % A.nb);
                    go[u] = (ci * A.H + iy) * A.W + ix;
                    lo[u] = lr * A.pix_per + ((iy + 1) * (f0 + 2) + ix + 1) * A.cs0 + ci;
                }
                float pv[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    const int64_t off = img_i[u] * plane + (int64_t)p0[u] * A.W + p1[u] + go[u];
                    pv[u] = A.img_u8 ? (float)imgb[off] / 255.0f : imgf[off];
                }
#pragma unroll
                for (int u = 0; u < kU; ++u)
                    if (base + u * nthreads < tot) Pix[lo[u]] = pv[u];
            }
        }
    };
    // first-layer weight gradient (W0): per-lane window offsets of the lane's im2col columns, accumulators that
    // live across all chunks of the workgroup (two 16-wide k tiles cover K0 = 9 * cin0 <= 32)
    constexpr int NK0 = 2;
    int toff0[NK0];
    cf32x4 acc0[NK0];
    float bsum0 = 0.f;
    if constexpr (W0) {
#pragma unroll
        for (int j = 0; j < NK0; ++j) {
            int kc = j * 16 + l16;
            kc = kc < A.K0 ? kc : A.K0 - 1;
            const int tap = kc / A.cin0, ci = kc - tap * A.cin0;
            toff0[j] = ((tap / 3) * (A.f0 + 2) + (tap % 3)) * A.cs0 + ci;
            acc0[j] = cf32x4{0.f, 0.f, 0.f, 0.f};
        }
        float* Pix = lds + A.off_pix;  // zero once: the borders are never written again
        for (int i = tid; i < A.rb * A.pix_per; i += nthreads) Pix[i] = 0.f;
        __syncthreads();
    }

    // transposed convolution: wave w owns column tile nt = w % NT and a CONTIGUOUS range of row tiles
    // (same parity class -> same taps, so one weight fragment feeds all of them)
    const int NT = A.NT, wpn = nwaves / NT;
    const int nt = wave % NT, wslot = wave / NT;
    const bool wactive = wslot < wpn;  // nwaves need not be a multiple of NT
    // row-tile range of this wave slot (equal-cost split made by the launcher)
    const int tb = wactive ? A.tbeg[wslot] : 0, tpw = wactive ? A.tbeg[wslot + 1] - tb : 0;
    // per row tile: this lane's row (patch base offset in Dz, input position) and the 9-bit set of
    // taps that reach a valid output position; the wave-wide union of the sets is kept on the scalar
    // unit so that skipped taps cost one scalar branch
    int rbase[kDgradTiles], rpy[kDgradTiles], rpx[kDgradTiles], tmask[kDgradTiles];
    unsigned wmask[kDgradTiles], wave_mask = 0;
    int nrow_built = -1;  // the chunk height the tables were built for
    int wr = nt * 16 + l16;
    wr = wr < cin ? wr : cin - 1;
    const int steps = (cout + 15) >> 4;
    const float* wbase = A.wt + (int64_t)wr * A.ldwt + 4 * quad;
    const int64_t tap_stride = (int64_t)cin * A.ldwt;

    // GroupNorm + SiLU backward: wave w always works on group w % G (nwaves % G == 0), lane (cc, pslot)
    // owns channel g * cpg + cc, so the affine partial sums have a single owner and a fixed order
    const int g = wave % G, wpg = nwaves / G;
    const int cc = lane % cpg, pslot = lane / cpg, pstep = 64 / cpg;
    const int c = g * cpg + cc;
    const float gm = A.gamma[c], bt = A.beta[c];
    const float inv_cnt = 1.0f / (float)(Pin * cpg);
    float pg = 0.f, pb = 0.f;

    stage(blockIdx.x);
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int64_t row0 = (int64_t)chunk * full_rows;
        const int nrow = (int)(A.rows - row0 < full_rows ? A.rows - row0 : full_rows);
        const bool rebuild = nrow != nrow_built;
        // ---- row order of the dA tiles: parity class major, then patch, then position
        if (rebuild) {
            const int ne = (hin + 1) >> 1, no = hin >> 1;
            int start = 0;
            int cstart[5];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cstart[q] = start;
                start += nrow * ((q >> 1) ? no : ne) * ((q & 1) ? no : ne);
            }
            cstart[4] = start;
            for (int m = tid; m < A.MT * 16; m += nthreads) {
                int v = -1;
                if (m < cstart[4]) {
                    int q = 0;
                    if (m >= cstart[1]) q = 1;
                    if (m >= cstart[2]) q = 2;
                    if (m >= cstart[3]) q = 3;
                    const int ny = (q >> 1) ? no : ne, nx = (q & 1) ? no : ne;
                    const int idx = m - cstart[q];
                    const int lr = idx / (ny * nx), r = idx - lr * ny * nx;
                    const int yi = r / nx, xi = r - yi * nx;
                    v = (lr << 16) | ((2 * yi + (q >> 1)) << 8) | (2 * xi + (q & 1));
                }
                perm[m] = v;
            }
            nrow_built = nrow;
        }
        MARL_TS();
        __syncthreads();
        MARL_TS();
        if (rebuild) {
            wave_mask = 0;
#pragma unroll
            for (int i = 0; i < kDgradTiles; ++i) {
                const int mt = tb + i;
                const int pk = (wactive && i < tpw && mt < A.MT) ? perm[mt * 16 + l16] : -1;
                const int py = (pk >> 8) & 255, px = pk & 255;
                rbase[i] = (pk >> 16) * P * zs + 4 * quad;
                rpy[i] = py;
                rpx[i] = px;
                int my = 0, mx = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int ty = py + 1 - k, tx = px + 1 - k;
                    if (ty >= 0 && !(ty & 1) && (ty >> 1) < hout) my |= 1 << k;
                    if (tx >= 0 && !(tx & 1) && (tx >> 1) < hout) mx |= 1 << k;
                }
                int tm = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (my & (1 << k)) tm |= mx << (3 * k);
                tmask[i] = pk < 0 ? 0 : tm;
                unsigned wm_ = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if (__ballot((tmask[i] >> t) & 1) != 0ull) wm_ |= 1u << t;
                wmask[i] = wm_;
                wave_mask |= wm_;
            }
        }
        // ---- the accumulators stay in registers across the nine taps
        {
            cf32x4 acc[kDgradTiles];
#pragma unroll
            for (int i = 0; i < kDgradTiles; ++i) acc[i] = cf32x4{0.f, 0.f, 0.f, 0.f};
            for (int tap = 0; tap < 9; ++tap) {
                if (!((wave_mask >> tap) & 1u)) continue;  // none of this wave's rows sees this tap
                const int kh = tap / 3, kw = tap - 3 * kh;
                for (int st0 = 0; st0 < steps; st0 += 4) {
                    float4 bq[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kk = (st0 + j) * 16;
                        const bool ok = st0 + j < steps && kk + 4 * quad < cout;
                        const float4 v = *reinterpret_cast<const float4*>(wbase + tap * tap_stride + (ok ? kk : -4 * quad));
                        bq[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int i = 0; i < kDgradTiles; ++i) {
                        if (!((wmask[i] >> tap) & 1u)) continue;
                        const bool valid = (tmask[i] >> tap) & 1;
                        const int off = rbase[i] + (((rpy[i] + 1 - kh) >> 1) * hout + ((rpx[i] + 1 - kw) >> 1)) * zs;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (st0 + j < steps) {
                                float4 a = *reinterpret_cast<const float4*>(Dz + (valid ? off + (st0 + j) * 16 : 4 * quad));
                                if (!valid || (st0 + j) * 16 + 4 * quad >= cout) a = make_float4(0.f, 0.f, 0.f, 0.f);
                                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[j].x, acc[i], 0, 0, 0);
                                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[j].y, acc[i], 0, 0, 0);
                                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[j].z, acc[i], 0, 0, 0);
                                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[j].w, acc[i], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            MARL_TS();
            const int n = nt * 16 + l16;
#pragma unroll
            for (int i = 0; i < kDgradTiles; ++i) {
                const int mt = tb + i;
                if (wactive && i < tpw && mt < A.MT && n < cin) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int pk = perm[mt * 16 + 4 * quad + r];
                        if (pk >= 0)
                            Da[((pk >> 16) * Pin + ((pk >> 8) & 255) * hin + (pk & 255)) * cs + n] = acc[i][r];
                    }
                }
            }
        }
        MARL_TS();
        __syncthreads();
        MARL_TS();

        // ---- GroupNorm + SiLU backward of layer l-1 on the LDS panels
        {
            constexpr int kKeep = 8;  // positions per lane kept in registers between the two passes
            const bool keep = Pin <= kKeep * pstep;
            // Fewer patches in the chunk than waves per group (one-patch chunks of the wide first layers: half of
            // the waves had no row): `share` waves split the positions of one (patch, group) and exchange their
            // two statistics sums through LDS - every wave then has exactly one row, so the barriers stay uniform
            const int share = (!keep && nrow < wpg && wpg % nrow == 0) ? wpg / nrow : 1;
            if (share > 1) {
                const int slot = wave / G, lr = slot / share, part = slot - lr * share;
                const float mean = gstat[(lr * G + g) * 2], rstd = gstat[(lr * G + g) * 2 + 1];
                const float* zr = Zin + lr * Pin * cs + c;
                float* dr = Da + lr * Pin * cs + c;
                float s1 = 0.f, s2 = 0.f;
                for (int pos = pslot + part * pstep; pos < Pin; pos += pstep * share) {
                    const float xh = (zr[pos * cs] - mean) * rstd;
                    const float dy = dr[pos * cs] * cnn_silu_grad(gm * xh + bt);
                    const float dxh = dy * gm;
                    s1 += dxh;
                    s2 += dxh * xh;
                    pg += dy * xh;
                    pb += dy;
                }
                s1 = wave_sum(s1);
                s2 = wave_sum(s2);
                if (lane == 0) {
                    gsum[wave * 2] = s1;
                    gsum[wave * 2 + 1] = s2;
                }
                lds_barrier();
                float t1 = 0.f, t2 = 0.f;
                for (int q = 0; q < share; ++q) {  // fixed order: the same sums in every wave of the row
                    const int w2 = g + G * (lr * share + q);
                    t1 += gsum[w2 * 2];
                    t2 += gsum[w2 * 2 + 1];
                }
                const float m1 = t1 * inv_cnt, m2 = t2 * inv_cnt;
                for (int pos = pslot + part * pstep; pos < Pin; pos += pstep * share) {
                    const float xh = (zr[pos * cs] - mean) * rstd;
                    const float dxh = dr[pos * cs] * cnn_silu_grad(gm * xh + bt) * gm;
                    dr[pos * cs] = rstd * (dxh - m1 - xh * m2);
                }
            } else {
            for (int lr = wave / G; lr < nrow; lr += wpg) {
                const float mean = gstat[(lr * G + g) * 2], rstd = gstat[(lr * G + g) * 2 + 1];
                const float* zr = Zin + lr * Pin * cs + c;
                float* dr = Da + lr * Pin * cs + c;
                float s1 = 0.f, s2 = 0.f;
                if (keep) {
                    float xk[kKeep], dk[kKeep];
#pragma unroll
                    for (int u = 0; u < kKeep; ++u) {
                        const int pos = pslot + u * pstep;
                        xk[u] = dk[u] = 0.f;
                        if (pos < Pin) {
                            const float xh = (zr[pos * cs] - mean) * rstd;
                            const float dy = dr[pos * cs] * cnn_silu_grad(gm * xh + bt);
                            const float dxh = dy * gm;
                            xk[u] = xh;
                            dk[u] = dxh;
                            s1 += dxh;
                            s2 += dxh * xh;
                            pg += dy * xh;
                            pb += dy;
                        }
                    }
                    const float m1 = wave_sum(s1) * inv_cnt, m2 = wave_sum(s2) * inv_cnt;
#pragma unroll
                    for (int u = 0; u < kKeep; ++u) {
                        const int pos = pslot + u * pstep;
                        if (pos < Pin) dr[pos * cs] = rstd * (dk[u] - m1 - xk[u] * m2);
                    }
                    continue;
                }
                for (int pos = pslot; pos < Pin; pos += pstep) {
                    const float xh = (zr[pos * cs] - mean) * rstd;
                    const float dy = dr[pos * cs] * cnn_silu_grad(gm * xh + bt);
                    const float dxh = dy * gm;
                    s1 += dxh;
                    s2 += dxh * xh;
                    pg += dy * xh;
                    pb += dy;
                }
                const float m1 = wave_sum(s1) * inv_cnt, m2 = wave_sum(s2) * inv_cnt;
                for (int pos = pslot; pos < Pin; pos += pstep) {
                    const float xh = (zr[pos * cs] - mean) * rstd;
                    const float dxh = dr[pos * cs] * cnn_silu_grad(gm * xh + bt) * gm;
                    dr[pos * cs] = rstd * (dxh - m1 - xh * m2);
                }
            }
            }
        }
        MARL_TS();
        __syncthreads();
        MARL_TS();
        if constexpr (W0) {
            // ---- dW_0[co][k] += sum_m dZ_0[m][co] * im2col(patch)[m][k] on 16x16x4 f32 MFMA tiles: A = the dZ_0 panel
            // (rows m = patch-major layer-0 output positions), B gathered from the pixel image; wave w takes the row
            // steps s = w, w + nwaves, ...
            const float* Pix = lds + A.off_pix;
            const int M0 = nrow * Pin, msteps = (M0 + 3) >> 2;
            for (int s0 = wave; s0 < msteps; s0 += nwaves) {
                const int m = s0 * 4 + quad;
                const bool mv = m < M0;
                const int mm = mv ? m : 0;
                const int lr = fdiv(mm, A.dPin), ipos = mm - lr * Pin;
                const int oy = fdiv(ipos, A.dhin0), ox = ipos - oy * hin;
                const float a = (mv && l16 < cin) ? Da[mm * cs + l16] : 0.f;
                const float* win = Pix + lr * A.pix_per + (2 * oy * (A.f0 + 2) + 2 * ox) * A.cs0;
                bsum0 += a;
#pragma unroll
                for (int j = 0; j < NK0; ++j) acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, win[toff0[j]], acc0[j], 0, 0, 0);
            }
            __syncthreads();  // the pixel images are free: the next chunk's may land
        }
        // ---- the next chunk's loads go out first (Dz, Zin and the statistics are free from here on),
        // then dZ_{l-1} of this chunk -> global (coalesced)
        if (chunk + (int)gridDim.x < nchunks) stage(chunk + (int)gridDim.x);
        if (A.dzin) {
            const int Mi = nrow * Pin, i4 = cin >> 2;
            float* dst = A.dzin + row0 * Pin * (int64_t)cin;
            for (int idx = tid; idx < Mi * i4; idx += nthreads) {
                const int m = fdiv(idx, A.dc4i), k = (idx - m * i4) * 4;
                *reinterpret_cast<float4*>(dst + (int64_t)m * cin + k) = *reinterpret_cast<const float4*>(Da + m * cs + k);
            }
        }
        MARL_TS();
    }
    // ---- affine partials of this workgroup
    for (int o = cpg; o < 64; o <<= 1) {
        pg += __shfl_xor(pg, o);
        pb += __shfl_xor(pb, o);
    }
    if (pslot == 0) {
        gsum[(wave * 2) * cpg + cc] = pg;
        gsum[(wave * 2 + 1) * cpg + cc] = pb;
    }
    __syncthreads();
    for (int e = tid; e < 2 * cin; e += nthreads) {
        const int which = e >= cin, ch = e - which * cin;
        const int gg = fdiv(ch, A.dcpg), c2 = ch - gg * cpg;
        float t = 0.f;
        for (int w = gg; w < nwaves; w += G) t += gsum[(w * 2 + which) * cpg + c2];
        A.part[(size_t)blockIdx.x * 2 * cin + e] = t;
    }
    if constexpr (W0) {
        // ---- the waves' partial tiles (different row steps) are summed in a fixed order; one slab per workgroup
        __syncthreads();
        float* red = lds;  // [nwaves][NK0][64 lanes][4] + [nwaves][64] bias partials
        float* redb = lds + nwaves * NK0 * 256;
#pragma unroll
        for (int j = 0; j < NK0; ++j) *reinterpret_cast<cf32x4*>(red + ((wave * NK0 + j) * 64 + lane) * 4) = acc0[j];
        redb[wave * 64 + lane] = bsum0;
        __syncthreads();
        if (wave == 0) {
            float* pw = A.w0_part + (size_t)blockIdx.x * cin * A.K0;
#pragma unroll
            for (int j = 0; j < NK0; ++j) {
                cf32x4 v = acc0[j];
                for (int q = 1; q < nwaves; ++q) v += *reinterpret_cast<const cf32x4*>(red + ((q * NK0 + j) * 64 + lane) * 4);
                const int kcol = j * 16 + l16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 4 * quad + r;
                    if (co < cin && kcol < A.K0) pw[(size_t)co * A.K0 + kcol] = v[r];
                }
            }
            // bias gradient of layer 0 = column sums of dZ_0: lane (quad, l16) summed channel l16 over its rows
            float t = 0.f;
            for (int q = 0; q < nwaves; ++q)
                t += ((redb[q * 64 + l16] + redb[q * 64 + 16 + l16]) + redb[q * 64 + 32 + l16]) + redb[q * 64 + 48 + l16];
            if (quad == 0 && l16 < cin) A.w0_bpart[(size_t)blockIdx.x * cin + l16] = t;
        }
    }
}

static size_t cnn_dgrad_plan(CnnDgradArgs& a, int rb) {
    a.rb = rb;
    a.MT = (rb * a.Pin + 15) / 16;
    a.NT = (a.cin + 15) / 16;
    const int cpg = a.cin / a.G;
    a.dc4o = make_fdiv(a.cout / 4);
    a.dc4i = make_fdiv(a.cin / 4);
    a.dPin = make_fdiv(a.Pin);
    a.dcpg = make_fdiv(cpg);
    a.dG = make_fdiv(a.G);
    size_t off = ((size_t)rb * a.P * (a.cout + 4) + 3) & ~(size_t)3;
    a.off_zin = (int)off;
    off += ((size_t)rb * a.Pin * (a.cin + 4) + 3) & ~(size_t)3;
    a.off_da = (int)off;
    off += ((size_t)rb * a.Pin * (a.cin + 4) + 3) & ~(size_t)3;
    a.off_perm = (int)off;
    off += (size_t)a.MT * 16;
    a.off_stat = (int)off;
    off += (size_t)rb * a.G * 2;
    a.off_gsum = (int)off;
    off += (size_t)8 * 2 * cpg;
    if (a.w0) {  // zero-bordered raw patches of the chunk [rb][f0 + 2][f0 + 2][cin0]
        off = (off + 3) & ~(size_t)3;
        a.cs0 = a.cin0;
        a.pix_per = (a.f0 + 2) * (a.f0 + 2) * a.cs0;
        a.off_pix = (int)off;
        off += (size_t)rb * a.pix_per;
        a.dpe0 = make_fdiv(a.cin0 * a.f0 * a.f0);
        a.dff0 = make_fdiv(a.f0 * a.f0);
        a.df0 = make_fdiv(a.f0);
        a.dhin0 = make_fdiv(a.hin);
        const size_t red = (size_t)8 * 2 * 256 + 8 * 64;  // the final tile reduction lives at the start of LDS
        if (off < red) off = red;
    }
    // Row tiles -> wave slots: contiguous ranges (one parity class -> one set of taps, so one
    // weight fragment feeds every tile of a range) of about equal cost; a tile costs one unit
    // plus one per tap of the classes it touches (1, 2, 2, 4 taps), <= kDgradTiles per slot.
    {
        const int ne = (a.hin + 1) / 2, no = a.hin / 2;
        int cstart[5] = {0, 0, 0, 0, 0};
        for (int c = 0; c < 4; ++c) cstart[c + 1] = cstart[c] + rb * ((c >> 1) ? no : ne) * ((c & 1) ? no : ne);
        const int wpn = 8 / a.NT > 0 ? 8 / a.NT : 1;
        int cost[64], total = 0;
        for (int t = 0; t < a.MT && t < 64; ++t) {
            int cmax = 0;
            for (int e = 0; e < 2; ++e) {
                const int m = t * 16 + e * 15;
                if (m >= cstart[4]) continue;
                int c = 0;
                while (c < 3 && m >= cstart[c + 1]) ++c;
                const int k = ((c >> 1) + 1) * ((c & 1) + 1);
                cmax = k > cmax ? k : cmax;
            }
            cost[t] = 1 + cmax;
            total += cost[t];
        }
        int t = 0, done = 0;
        for (int sl = 0; sl < wpn && sl < 16; ++sl) {
            a.tbeg[sl] = t;
            const int target = (total * (sl + 1) + wpn - 1) / wpn;
            int cnt = 0;
            while (t < a.MT && cnt < kDgradTiles &&
                   (done + cost[t] / 2 < target || a.MT - t > (wpn - 1 - sl) * kDgradTiles)) {
                done += cost[t];
                ++t;
                ++cnt;
            }
        }
        for (int sl = wpn; sl <= 16; ++sl) a.tbeg[sl] = a.MT;
        a.tbeg[wpn] = a.MT;
    }
    return off;
}

static int cnn_dgrad_rb(CnnDgradArgs& a) {
    // largest rb <= 8 whose panels fit 72 KiB (two workgroups per CU) with at most
    // kDgradTiles row tiles per wave, keeping >= 512 workgroups when the problem allows it
    const int nt = (a.cin + 15) / 16;
    const int wpn = 8 / nt;
    if (wpn < 1) return 0;
    const int rb_max = 8;
    const size_t lds_cap = (size_t)72 * 1024;
    for (int rb = rb_max < 8 ? (rb_max < 1 ? 1 : rb_max) : 8; rb >= 1; --rb) {
        if (cnn_dgrad_plan(a, rb) * sizeof(float) > lds_cap) continue;
        if (a.MT > kDgradTiles * wpn || a.MT > 64) continue;
        if (rb > 1 && cdiv(a.rows, rb) < tune_get("dgrad_min_chunks", 512)) continue;
        return rb;
    }
    return 0;
}

int cnn_dgrad_supported(const CnnDgradArgs& a0) {
    if (getenv("MARL_CNN_FUSED") && getenv("MARL_CNN_FUSED")[0] == '0') return 0;
    CnnDgradArgs a = a0;
    if ((a.cin & 3) || (a.cout & 3) || a.cin % a.G != 0) return 0;
    const int cpg = a.cin / a.G;
    if (cpg > 64 || (cpg & (cpg - 1)) || 8 % a.G != 0) return 0;  // lane <-> channel ownership
    if (a.hin > 255 || a.rows <= 0) return 0;
    return cnn_dgrad_rb(a) > 0;
}

// can the launch that produces dZ_0 also form layer 0's weight gradient? (cin = layer 0's output channels: one
// 16-wide tile; K0 = 9 * cin0 <= 32: two k tiles)
// EXPERIMENT BUILD ONLY (EXTRA=-DMARL_DGRAD_W0; round 5's knob dgrad_w0 is gone): measured SLOWER than the separate first-layer launch on every BASELINE shape (C3 7.68
// vs 7.59 ms, C4 3.71 vs 3.63, C5 18.45 vs 18.25: DESIGN 4.0c) - the extra phase adds two barriers, a scattered
// pixel gather and ~4 us of dependent latency to each chunk of a kernel that is a latency chain already, which costs
// more than the 100 us launch and the dZ_0 round trip it removes.
int cnn_dgrad_w0_ok(const CnnDgradArgs& a, int cin0, int f0) {
#ifdef MARL_DGRAD_W0  // experiment build only (make EXTRA=-DMARL_DGRAD_W0): the form lost every A/B, the product does not carry it
    return a.cin <= 16 && 9 * cin0 <= 32 && cin0 >= 1 && (f0 - 1) / 2 + 1 == a.hin;
#else
    (void)a, (void)cin0, (void)f0;
    return 0;
#endif
}

// Persistent grid: as many workgroups as are resident at once (occupancy x CUs), never more than
// there are chunks.  This is also the number of affine partial rows the kernel writes.
static int cnn_dgrad_grid(const CnnDgradArgs& a, int rb, size_t lds) {
    static int num_cu = 0;
    static size_t occ_lds[8];
    static int occ_val[8], occ_n = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        num_cu = prop.multiProcessorCount;
    }
    int occ = 0;
    const size_t key = lds * 2 + (a.w0 ? 1 : 0);  // (the two instantiations differ in registers)
    for (int i = 0; i < occ_n; ++i)
        if (occ_lds[i] == key) occ = occ_val[i];
    if (!occ) {
#ifdef MARL_DGRAD_W0
        const void* kern = a.w0 ? reinterpret_cast<const void*>(cnn_dgrad_kernel<true>)
                                     : reinterpret_cast<const void*>(cnn_dgrad_kernel<false>);
#else
        const void* kern = reinterpret_cast<const void*>(cnn_dgrad_kernel<false>);
#endif
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 512, lds) != hipSuccess || occ < 1) occ = 1;
        if (occ_n < 8) {
            occ_lds[occ_n] = key;
            occ_val[occ_n++] = occ;
        }
    }
    int64_t cap = (int64_t)tune_get("dgrad_wgs", 0);  // <= 0: every workgroup resident at once
    if (cap <= 0) cap = (int64_t)occ * num_cu;
    const int64_t chunks = cdiv(a.rows, rb);
    return (int)(chunks < cap ? chunks : (cap < 1 ? 1 : cap));
}

static void cnn_dgrad_raise_lds() {
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_dgrad_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
#ifdef MARL_DGRAD_W0
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cnn_dgrad_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
#endif
        raised = true;
    }
}

// device-independent upper bound of cnn_dgrad_blocks (the workspace layout is computed without a GPU)
int cnn_dgrad_blocks_max(const CnnDgradArgs& a0) {
    CnnDgradArgs a = a0;
    const int rb = cnn_dgrad_rb(a);
    return rb > 0 ? (int)cdiv(a.rows, rb) : 0;
}

int cnn_dgrad_blocks(const CnnDgradArgs& a0) {
    CnnDgradArgs a = a0;
    const int rb = cnn_dgrad_rb(a);
    if (rb <= 0) return 0;
    cnn_dgrad_raise_lds();
    return cnn_dgrad_grid(a, rb, cnn_dgrad_plan(a, rb) * sizeof(float));
}

int launch_cnn_dgrad(CnnDgradArgs& a, hipStream_t st) {
    const int rb = cnn_dgrad_rb(a);
    if (rb <= 0) {
        set_error("fused CNN layer backward: shape outside its range");
        return MARL_ELIMIT;
    }
    const size_t lds = cnn_dgrad_plan(a, rb) * sizeof(float);
    cnn_dgrad_raise_lds();
    const int grid = cnn_dgrad_grid(a, rb, lds);
#ifdef MARL_KERNEL_TS
    static long long* d_ts = nullptr;
    static int calls = 0;
    const int rec = ts_begin(&d_ts, calls) || ts_begin(&d_ts, calls - 1);
    ++calls;
    a.ts = rec ? d_ts : nullptr;
#endif
    prof_before(5, st);
    if (a.w0 && (!a.w0_part || !a.w0_bpart || !a.img || !a.pos)) {
        set_error("fused CNN layer backward: first-layer weight gradient without its buffers");
        return MARL_EINVAL;
    }
#ifdef MARL_DGRAD_W0
    if (a.w0)
        hipLaunchKernelGGL(cnn_dgrad_kernel<true>, dim3((unsigned)grid), dim3(512), lds, st, a);
    else
#else
    if (a.w0) {
        set_error("cnn_dgrad: the fused first-layer form is not in this build");
        return MARL_EINVAL;
    }
#endif
        hipLaunchKernelGGL(cnn_dgrad_kernel<false>, dim3((unsigned)grid), dim3(512), lds, st, a);
    prof_after(5, st);
    MARL_LAUNCH_CHECK();
#ifdef MARL_KERNEL_TS
    if (rec) {
        fprintf(stderr, "[ts] dgrad rb %d lds %zu cin %d cout %d blocks %d\n", rb, lds, a.cin, a.cout, grid);
        ts_report("cnn_dgrad", d_ts, 8);
    }
#endif
    return MARL_OK;
}

// prefetch depths the kernel is instantiated with: {dZ float4, input items} per thread
static const int kWgPref[2][2][2] = {{{2, 2}, {4, 4}},    // deeper layers (float4 of Z_{l-1})
                                     {{2, 4}, {4, 14}}};  // first layer (raw pixels)

// ---------------------------------------------------------------------------
// Convolution weight gradient from the activations (backward of networks/vision.py:33-35,
// training/trainer.py:115).  PERSISTENT workgroups (a few per CU) walk chunks of `rb` patches:
//   dZ_l chunk -> LDS; the layer's input -> LDS with a zero border (raw image patch gathered at
//   the saved positions for the first layer, SiLU(GroupNorm(Z_{l-1})) recomputed from the saved
//   pre-norm output otherwise); then dW[co][k] += sum_m dZ[m][co] * im2col(in)[m][k] on 16x16x4
//   f32 MFMA tiles whose B fragments are gathered straight from the LDS image (implicit im2col,
//   the zero border makes every tap address valid).  The accumulators live in registers across
//   ALL chunks; every workgroup writes one partial slab at the end and a fixed-order reduction
//   sums the slabs (bit-reproducible, no float atomics).  The next chunk's global loads are
//   issued before the current chunk's matrix phase (register prefetch).
// Wave roles: (cout-tile group, k-tile group, row-step phase); waves that share tiles but walk
// different row steps are summed through LDS once at the end.
// ---------------------------------------------------------------------------
// PD = float4 of dZ prefetched per thread; PI = prefetched input items per thread (float4 of
// Z_{l-1}, or raw pixels for the first layer).  Small values keep the kernel under 128 VGPRs so
// that two workgroups share a CU and hide each other's staging phases.
template <int NCT, int NKT, bool FIRST, int PD, int PI>
__global__ __launch_bounds__(512) void cnn_wgrad_kernel(const CnnWgradArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* In = lds;                                       // [rb][hp][hp][cs]
    float* Dz = lds + A.off_dz;                            // [Mpad][zs]
    int* tab = reinterpret_cast<int*>(lds + A.off_tab);    // [Mpad] float offset of row m's window
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = lane >> 4, l16 = lane & 15;
    const int cin = A.cin, cout = A.cout, P = A.P, K = A.K, hp = A.hp, cs = A.cs, zs = A.zs;
    const int rb = A.rb, in_per = A.in_per;
    const int Mfull = rb * P, Mpad = (Mfull + 3) & ~3;
    const int ms = A.ms;
    const int ms_id = wave % ms, tg = wave / ms;
    const int ktg = tg % A.tgk, ctg = tg / A.tgk;
    const int kt0 = blockIdx.y * A.nkt_slab + ktg * NKT;
    const int kt_end = min((int)(blockIdx.y + 1) * A.nkt_slab, A.nkt);
    const int ct0 = ctg * NCT;

    // per-lane constants: where inside a row's 3x3 window column kcol = kt * 16 + l16 lives
    int toff[NKT];
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
        int kc = (kt0 + j) * 16 + l16;
        kc = kc < K ? kc : K - 1;
        const int tap = fdiv(kc, A.dcin), ci = kc - tap * cin;
        const int kh = tap / 3, kw = tap - 3 * kh;
        toff[j] = (kh * hp + kw) * cs + ci;
    }
    int aoff[NCT];
#pragma unroll
    for (int i = 0; i < NCT; ++i) {
        const int ct = ct0 + i < A.nct ? ct0 + i : A.nct - 1;
        aoff[i] = ct * 16 + l16;
    }
    cf32x4 acc[NCT][NKT];
#pragma unroll
    for (int i = 0; i < NCT; ++i)
#pragma unroll
        for (int j = 0; j < NKT; ++j) acc[i][j] = cf32x4{0.f, 0.f, 0.f, 0.f};

    // ---- one-time LDS set-up: zeros (borders, padding rows), row table
    for (int i = tid; i < A.lds_floats; i += 512) lds[i] = 0.f;
    __syncthreads();
    for (int m = tid; m < Mpad; m += 512) {
        const int mm = m < Mfull ? m : 0;
        const int lr = fdiv(mm, A.dP), opos = mm - lr * P;
        const int oy = fdiv(opos, A.dhout), ox = opos - oy * A.hout;
        tab[m] = lr * in_per + (2 * oy * hp + 2 * ox) * cs;
    }

    // ---- staging roles: everything that does not depend on the chunk is worked out once
    // (512 is a multiple of cout / 4 and of cin / 4, so a thread keeps its channels)
    const int c4o = cout >> 2;
    const int dz_c = (tid % c4o) * 4, dz_m0 = tid / c4o, dz_mstep = 512 / c4o;
    const int Pin = A.hin * A.hin;
    // input items of this thread: patch-in-chunk, LDS float offset, global offset inside the
    // patch's source (image plane offset for the first layer, Z_{l-1} element otherwise)
    // (LEAN: the deep-prefetch first-layer form recomputes an item's indices where it needs them - three
    // tables of 14 registers kept the kernel at 146 VGPRs = one workgroup per CU, and with one 16x16-output
    // patch per chunk nothing then hides a chunk's two barriers and its staging)
    constexpr bool LEAN = FIRST && PI > 4;
    constexpr int NIT = LEAN ? 1 : PI;
    int it_lr[NIT], it_lo[NIT], it_go[NIT];
    float4 gm4 = make_float4(0.f, 0.f, 0.f, 0.f), bt4 = gm4;
    int zi_g = 0;
    // (t: the thread index - an opaque per-call copy in the LEAN form, or hipcc hoists the chunk-invariant
    // index math out of the chunk loop and the registers are back)
    auto first_item = [&](int t, int i, int& lr_, int& lo_, int& go_) __attribute__((always_inline)) {
        const int ff = Pin, pe = cin * ff;
        const int idx = t + i * 512;
        lr_ = -1;
        lo_ = go_ = 0;
        if (idx < rb * pe) {
            const int lr = fdiv(idx, A.dpe), e = idx - lr * pe;
            const int ci = fdiv(e, A.dff), e2 = e - ci * ff;
            const int iy = fdiv(e2, A.df), ix = e2 - iy * A.hin;
            lr_ = lr;
            lo_ = lr * in_per + ((iy + 1) * hp + ix + 1) * cs + ci;
            go_ = (ci * A.H + iy) * A.W + ix;
        }
    };
    if constexpr (FIRST) {
        if (!LEAN) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) first_item(tid, i, it_lr[i], it_lo[i], it_go[i]);
        }
    } else {
        const int c4i = cin >> 2;
        const int zi_c = (tid % c4i) * 4, zi_p0 = tid / c4i, zi_pstep = 512 / c4i;
        gm4 = *reinterpret_cast<const float4*>(A.gamma + zi_c);
        bt4 = *reinterpret_cast<const float4*>(A.beta + zi_c);
        zi_g = zi_c / (cin / A.G);
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int pl = zi_p0 + i * zi_pstep;  // patch-major input position
            it_lr[i] = -1;
            it_lo[i] = it_go[i] = 0;
            if (pl < rb * Pin) {
                const int lr = fdiv(pl, A.dPin), ipos = pl - lr * Pin;
                const int iy = fdiv(ipos, A.dhin), ix = ipos - iy * A.hin;
                it_lr[i] = lr;
                it_lo[i] = lr * in_per + ((iy + 1) * hp + ix + 1) * cs + zi_c;
                it_go[i] = pl * cin + zi_c;
            }
        }
    }
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 pd[PD];
    float4 pz[FIRST ? 1 : PI];
    float2 ps[FIRST ? 1 : PI];
    float pr[FIRST ? PI : 1];
    const float* imgf = static_cast<const float*>(A.img);
    const unsigned char* imgb = static_cast<const unsigned char*>(A.img);
    const int64_t plane = (int64_t)A.c_img * A.H * A.W;

    auto prefetch = [&](int chunk) {
        const int64_t row0 = (int64_t)chunk * rb;
        const int nrow = (int)(A.rows - row0 < rb ? A.rows - row0 : rb);
        const int M = nrow * P;
        const float* dsrc = A.dz + row0 * P * (int64_t)cout + dz_c;
#pragma unroll
        for (int i = 0; i < PD; ++i) {
            const int m = dz_m0 + i * dz_mstep;
            pd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < M) pd[i] = *reinterpret_cast<const float4*>(dsrc + (int64_t)m * cout);
        }
        if constexpr (FIRST) {
            int tv = tid;
            if (LEAN) asm volatile("" : "+v"(tv));
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                pr[i] = 0.f;
                int lr_, lo_, go_;
                if (LEAN) {
                    first_item(tv, i, lr_, lo_, go_);
                } else {
                    lr_ = it_lr[LEAN ? 0 : i];
                    go_ = it_go[LEAN ? 0 : i];
                }
                if ((unsigned)lr_ < (unsigned)nrow) {
                    const int64_t r = row0 + lr_;
                    const int p0 = A.pos[r * 2], p1 = A.pos[r * 2 + 1];
                    const int64_t off = (r % A.nb) * plane + (int64_t)p0 * A.W + p1 + go_;
                    pr[i] = A.img_u8 ? (float)imgb[off] / 255.0f : imgf[off];
                }
            }
        } else {
            const float* zsrc = A.zin + row0 * Pin * (int64_t)cin;
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                pz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                ps[i] = make_float2(0.f, 0.f);
                if ((unsigned)it_lr[i] < (unsigned)nrow) {
                    pz[i] = *reinterpret_cast<const float4*>(zsrc + it_go[i]);
                    ps[i] = *reinterpret_cast<const float2*>(A.gst + ((row0 + it_lr[i]) * A.G + zi_g) * 2);
                }
            }
        }
    };
    auto stage = [&](int chunk) {
        const int64_t row0 = (int64_t)chunk * rb;
        const int nrow = (int)(A.rows - row0 < rb ? A.rows - row0 : rb);
#pragma unroll
        for (int i = 0; i < PD; ++i) {
            const int m = dz_m0 + i * dz_mstep;
            if (m < Mpad) {  // rows past this chunk's M hold zeros (they multiply stale inputs)
                *reinterpret_cast<float4*>(Dz + m * zs + dz_c) = pd[i];
                bsum.x += pd[i].x;
                bsum.y += pd[i].y;
                bsum.z += pd[i].z;
                bsum.w += pd[i].w;
            }
        }
        if constexpr (FIRST) {
            int tv = tid;
            if (LEAN) asm volatile("" : "+v"(tv));
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                int lr_, lo_, go_;
                if (LEAN) {
                    first_item(tv, i, lr_, lo_, go_);
                } else {
                    lr_ = it_lr[LEAN ? 0 : i];
                    lo_ = it_lo[LEAN ? 0 : i];
                }
                if ((unsigned)lr_ < (unsigned)nrow) In[lo_] = pr[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < PI; ++i) {
                if ((unsigned)it_lr[i] < (unsigned)nrow) {
                    const float mean = ps[i].x, rstd = ps[i].y;
                    float4 v;
                    v.x = cnn_silu((pz[i].x - mean) * rstd * gm4.x + bt4.x);
                    v.y = cnn_silu((pz[i].y - mean) * rstd * gm4.y + bt4.y);
                    v.z = cnn_silu((pz[i].z - mean) * rstd * gm4.z + bt4.z);
                    v.w = cnn_silu((pz[i].w - mean) * rstd * gm4.w + bt4.w);
                    *reinterpret_cast<float4*>(In + it_lo[i]) = v;
                }
            }
        }
    };

    int chunk = blockIdx.x;
    if (chunk < A.nchunks) prefetch(chunk);
    for (; chunk < A.nchunks; chunk += gridDim.x) {
        __syncthreads();  // every wave is done with the previous chunk's LDS image
        stage(chunk);
        __syncthreads();
        if (chunk + (int)gridDim.x < A.nchunks) prefetch(chunk + gridDim.x);
        const int64_t row0 = (int64_t)chunk * rb;
        const int nrow = (int)(A.rows - row0 < rb ? A.rows - row0 : rb);
        const int msteps = (nrow * P + 3) >> 2, last = msteps - 1;
        // software-pipelined row steps: the fragments of step s + ms (and the row-table entry of
        // step s + 2 ms) are read while the matrix instructions of step s issue
        int s = ms_id;
        if (s < msteps) {
            float a[NCT], b[NKT];
            int rb2;
            {
                const int m = s * 4 + quad;
                const int rb1 = tab[m];
                rb2 = tab[min(s + ms, last) * 4 + quad];
#pragma unroll
                for (int i = 0; i < NCT; ++i) a[i] = Dz[m * zs + aoff[i]];
#pragma unroll
                for (int j = 0; j < NKT; ++j) b[j] = In[rb1 + toff[j]];
            }
            for (; s < msteps; s += ms) {
                float an[NCT], bn[NKT];
                const int mn = min(s + ms, last) * 4 + quad;
#pragma unroll
                for (int i = 0; i < NCT; ++i) an[i] = Dz[mn * zs + aoff[i]];
#pragma unroll
                for (int j = 0; j < NKT; ++j) bn[j] = In[rb2 + toff[j]];
                rb2 = tab[min(s + 2 * ms, last) * 4 + quad];
#pragma unroll
                for (int j = 0; j < NKT; ++j)
#pragma unroll
                    for (int i = 0; i < NCT; ++i)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NCT; ++i) a[i] = an[i];
#pragma unroll
                for (int j = 0; j < NKT; ++j) b[j] = bn[j];
            }
        }
    }

    // ---- waves that shared tiles (different row-step phases) are summed through LDS in a fixed
    // order, then the workgroup's partial slab goes to global
    __syncthreads();
    float* red = lds;  // [8 waves][NCT * NKT][64 lanes][4]
    if (ms > 1) {
#pragma unroll
        for (int i = 0; i < NCT; ++i)
#pragma unroll
            for (int j = 0; j < NKT; ++j)
                *reinterpret_cast<cf32x4*>(red + ((wave * (NCT * NKT) + i * NKT + j) * 64 + lane) * 4) = acc[i][j];
        __syncthreads();
    }
    if (ms_id == 0) {
        float* pw = A.part_w + (size_t)blockIdx.x * cout * K;  // grid.y slabs are disjoint in k
#pragma unroll
        for (int i = 0; i < NCT; ++i) {
            if (ct0 + i >= A.nct) continue;
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                const int kt = kt0 + j;
                if (kt >= kt_end) continue;
                cf32x4 v = acc[i][j];
                for (int q = 1; q < ms; ++q) {
                    const cf32x4 u = *reinterpret_cast<const cf32x4*>(
                        red + (((wave + q) * (NCT * NKT) + i * NKT + j) * 64 + lane) * 4);
                    v += u;
                }
                const int kcol = kt * 16 + l16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = (ct0 + i) * 16 + 4 * quad + r;
                    if (co < cout && kcol < K) pw[(size_t)co * K + kcol] = v[r];
                }
            }
        }
    }
    // bias gradient: column sums of dZ, taken from the staged values (threads with equal
    // tid % (cout / 4) staged the same four channels)
    if (A.part_b && blockIdx.y == 0) {
        __syncthreads();
        float4* sh4 = reinterpret_cast<float4*>(lds);
        sh4[tid] = bsum;
        __syncthreads();
        if (tid < c4o) {
            float4 t = sh4[tid];
            for (int q = 1; q < dz_mstep; ++q) {
                const float4 u = sh4[q * c4o + tid];
                t.x += u.x;
                t.y += u.y;
                t.z += u.z;
                t.w += u.w;
            }
            *reinterpret_cast<float4*>(A.part_b + (size_t)blockIdx.x * cout + tid * 4) = t;
        }
    }
}

// ---------------------------------------------------------------------------
// The same weight gradient on the bf16 matrix pipe ("bf16x6") for the layers whose input has
// >= 16 channels (the 3x3 weight gradients are matrix-pipe-bound on the exact-fp32 MFMA).
//   * the staged dZ chunk and the recomputed layer input are split into three bf16 terms WHILE they go to
//     LDS (split_pair: x0 + x1 + x2 == x), three planes each: Dz3[plane][m][co], In3[plane][patch][y][x][ci]
//     (channels last, zero border);
//   * dW[co][k] += sum_m dZ[m][co] * im2col[m][k] has the contraction index m as the ROW index of both
//     operands, so the fragments of v_mfma_f32_32x32x16_bf16 (8 consecutive m per lane) come out of LDS
//     with the transposing read ds_read_b64_tr_b16 exactly as in gemm_tn3_kernel - here with one address per
//     lane: the implicit im2col is the per-lane gather (window origin of row m from the row table + the
//     tap / channel offset of the lane's 4 columns; 4 consecutive ci never straddle a tap: cin % 16 == 0);
//   * six products per fp32 product, smallest terms first, fp32 accumulation - the arithmetic of every other
//     bf16x6 kernel; tiles are 32 (co) x 32 (k) x 16 (m), a wave owns one co tile and NKT k tiles;
//   * row strides (cs2, zs2) are chosen so that the four rows of a 16-lane group and the two column halves
//     of a 32-lane group fall on disjoint banks.
// Same persistent-workgroup / partial-slab / fixed-order-reduction structure as cnn_wgrad_kernel.
// ---------------------------------------------------------------------------
typedef short wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef float wg_f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) wg_s16x4* wg_tr_ptr;

template <int NKT, int PD, int PI>
__global__ __launch_bounds__(512) void cnn_wgrad3_kernel(const CnnWgradArgs A) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = A.cin, cout = A.cout, P = A.P, K = A.K, hp = A.hp, cs = A.cs, zs = A.zs;
    const int rb = A.rb, in_per = A.in_per;  // (elements)
    const int Mfull = rb * P, Mpad = (Mfull + 15) & ~15;
    char* In3 = smem;                         // [3][rb][hp][hp][cs] bf16
    char* Dz3 = smem + A.off_dz;              // [3][Mpad][zs] bf16   (off_dz, off_tab: bytes)
    int* tab = reinterpret_cast<int*>(smem + A.off_tab);  // [Mpad] BYTE offset of row m's window in a plane
    const int in_plane = A.in_plane, dz_plane = Mpad * zs * 2;  // bytes
    const int ms = A.ms;
    const int ms_id = wave % ms, tg = wave / ms;
    const int ktg = tg % A.tgk, ct = tg / A.tgk;  // (tgc == nct: one 32-wide co tile per group)
    const int kt0 = blockIdx.y * A.nkt_slab + ktg * NKT;
    const int kt_end = min((int)(blockIdx.y + 1) * A.nkt_slab, A.nkt);

    // fragment geometry (gemm_tn3_kernel): 16-lane group g16 -> column half sub, row half kg; lanes 4q .. 4q+3
    // supply row q of a [4 rows][16 columns] block, 4 columns each
    const int g16 = lane >> 4, sub = g16 & 1, kg = g16 >> 1, qrow = (lane >> 2) & 3, piece = lane & 3;
    const int mrow = 8 * kg + qrow;            // row of read h = 0 inside a 16-row step (h = 1: + 4)
    const int ccol = 16 * sub + 4 * piece;     // first of the lane's 4 columns inside a 32-wide tile
    int toff[NKT];                             // BYTE offset of the lane's columns inside a row's window
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
        int kc = (kt0 + j) * 32 + ccol;
        kc = kc < K ? kc : K - 4;              // (columns past K: computed on valid data, never stored)
        const int tap = fdiv(kc, A.dcin), ci = kc - tap * cin;
        const int kh = tap / 3, kw = tap - 3 * kh;
        toff[j] = ((kh * hp + kw) * cs + ci) * 2;
    }
    const int aoff = (ct * 32 + ccol) * 2;
    wg_f32x16 acc[NKT];
#pragma unroll
    for (int j = 0; j < NKT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // ---- one-time LDS set-up: zeros (borders, padding rows), row table
    for (int i = tid; i < A.lds_floats; i += 512) reinterpret_cast<float*>(smem)[i] = 0.f;
    __syncthreads();
    for (int m = tid; m < Mpad; m += 512) {
        const int mm = m < Mfull ? m : 0;
        const int lr = fdiv(mm, A.dP), opos = mm - lr * P;
        const int oy = fdiv(opos, A.dhout), ox = opos - oy * A.hout;
        tab[m] = (lr * in_per + (2 * oy * hp + 2 * ox) * cs) * 2;
    }

    // ---- staging roles (as cnn_wgrad_kernel: a thread keeps its four channels)
    const int c4o = cout >> 2;
    const int dz_c = (tid % c4o) * 4, dz_m0 = tid / c4o, dz_mstep = 512 / c4o;
    const int Pin = A.hin * A.hin;
    const int c4i = cin >> 2;
    const int zi_c = (tid % c4i) * 4, zi_p0 = tid / c4i, zi_pstep = 512 / c4i;
    const float4 gm4 = *reinterpret_cast<const float4*>(A.gamma + zi_c);
    const float4 bt4 = *reinterpret_cast<const float4*>(A.beta + zi_c);
    const int zi_g = zi_c / (cin / A.G);
    int it_lr[PI], it_lo[PI], it_go[PI];
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int pl = zi_p0 + i * zi_pstep;  // patch-major input position
        it_lr[i] = -1;
        it_lo[i] = it_go[i] = 0;
        if (pl < rb * Pin) {
            const int lr = fdiv(pl, A.dPin), ipos = pl - lr * Pin;
            const int iy = fdiv(ipos, A.dhin), ix = ipos - iy * A.hin;
            it_lr[i] = lr;
            it_lo[i] = (lr * in_per + ((iy + 1) * hp + ix + 1) * cs + zi_c) * 2;
            it_go[i] = pl * cin + zi_c;
        }
    }
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pd[PD], pz[PI];
    float2 ps[PI];

    auto prefetch = [&](int chunk) {
        const int64_t row0 = (int64_t)chunk * rb;
        const int nrow = (int)(A.rows - row0 < rb ? A.rows - row0 : rb);
        const int M = nrow * P;
        const float* dsrc = A.dz + row0 * P * (int64_t)cout + dz_c;
#pragma unroll
        for (int i = 0; i < PD; ++i) {
            const int m = dz_m0 + i * dz_mstep;
            pd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < M) pd[i] = *reinterpret_cast<const float4*>(dsrc + (int64_t)m * cout);
        }
        const float* zsrc = A.zin + row0 * Pin * (int64_t)cin;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            pz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            ps[i] = make_float2(0.f, 0.f);
            if ((unsigned)it_lr[i] < (unsigned)nrow) {
                pz[i] = *reinterpret_cast<const float4*>(zsrc + it_go[i]);
                ps[i] = *reinterpret_cast<const float2*>(A.gst + ((row0 + it_lr[i]) * A.G + zi_g) * 2);
            }
        }
    };
    // four consecutive values -> 8 bytes in each of the three planes
    auto put3 = [&](char* base, int plane_bytes, float a, float b, float c, float d) {
        uint32_t a0, a1, a2, b0, b1, b2;
        split_pair(a, b, a0, a1, a2);
        split_pair(c, d, b0, b1, b2);
        *reinterpret_cast<uint2*>(base) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(base + plane_bytes) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(base + 2 * plane_bytes) = make_uint2(a2, b2);
    };
    auto stage = [&](int chunk) {
        const int64_t row0 = (int64_t)chunk * rb;
        const int nrow = (int)(A.rows - row0 < rb ? A.rows - row0 : rb);
#pragma unroll
        for (int i = 0; i < PD; ++i) {
            const int m = dz_m0 + i * dz_mstep;
            if (m < Mpad) {  // rows past this chunk's M hold zeros (they multiply stale inputs)
                put3(Dz3 + (m * zs + dz_c) * 2, dz_plane, pd[i].x, pd[i].y, pd[i].z, pd[i].w);
                bsum.x += pd[i].x;
                bsum.y += pd[i].y;
                bsum.z += pd[i].z;
                bsum.w += pd[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            if ((unsigned)it_lr[i] < (unsigned)nrow) {
                const float mean = ps[i].x, rstd = ps[i].y;
                put3(In3 + it_lo[i], in_plane, cnn_silu((pz[i].x - mean) * rstd * gm4.x + bt4.x),
                     cnn_silu((pz[i].y - mean) * rstd * gm4.y + bt4.y), cnn_silu((pz[i].z - mean) * rstd * gm4.z + bt4.z),
                     cnn_silu((pz[i].w - mean) * rstd * gm4.w + bt4.w));
            }
        }
    };
#define WG3_FRAG(p0_, p1_)                                                                         \
    __builtin_shufflevector(__builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_tr_ptr)(p0_)),             \
                            __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_tr_ptr)(p1_)), 0, 1, 2, 3, 4, 5, 6, 7)

    int chunk = blockIdx.x;
    if (chunk < A.nchunks) prefetch(chunk);
    for (; chunk < A.nchunks; chunk += gridDim.x) {
        __syncthreads();  // every wave is done with the previous chunk's LDS image
        stage(chunk);
        __syncthreads();
        if (chunk + (int)gridDim.x < A.nchunks) prefetch(chunk + gridDim.x);
        const int64_t row0 = (int64_t)chunk * rb;
        const int nrow = (int)(A.rows - row0 < rb ? A.rows - row0 : rb);
        const int msteps = (nrow * P + 15) >> 4;
        for (int s = ms_id; s < msteps; s += ms) {
            const int m0 = s * 16 + mrow;
            const int w0 = tab[m0], w1 = tab[m0 + 4];
            const char* da = Dz3 + (m0 * zs) * 2 + aoff;
            wg_bf16x8 fa[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) fa[p] = WG3_FRAG(da + p * dz_plane, da + p * dz_plane + 4 * zs * 2);
#pragma unroll
            for (int j = 0; j < NKT; ++j) {
                wg_bf16x8 fb[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    fb[p] = WG3_FRAG(In3 + p * in_plane + w0 + toff[j], In3 + p * in_plane + w1 + toff[j]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[j], 0, 0, 0);
            }
        }
    }
#undef WG3_FRAG

    // ---- waves that shared tiles (different row-step phases) are summed through LDS in a fixed order, one k
    // tile at a time (a 32 x 32 tile of every wave = 32 KB), then the partial slab goes to global:
    // acc[j][r] = dW[co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)][k = kt * 32 + (lane & 31)]
    float* red = reinterpret_cast<float*>(smem);  // [8 waves][64 lanes][16]
    float* pw = A.part_w + (size_t)blockIdx.x * cout * K;  // grid.y slabs are disjoint in k
#pragma unroll
    for (int j = 0; j < NKT; ++j) {
        wg_f32x16 v = acc[j];
        if (ms > 1) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<cf32x4*>(red + (wave * 64 + lane) * 16 + 4 * q) =
                    cf32x4{acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
            __syncthreads();
            if (ms_id == 0)
                for (int q = 1; q < ms; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += red[((wave + q) * 64 + lane) * 16 + r];
        }
        const int kt = kt0 + j, kcol = kt * 32 + (lane & 31);
        if (ms_id == 0 && kt < kt_end && kcol < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < cout) pw[(size_t)co * K + kcol] = v[r];
            }
        }
    }
    // bias gradient: column sums of dZ, taken from the staged values
    if (A.part_b && blockIdx.y == 0) {
        __syncthreads();
        float4* sh4 = reinterpret_cast<float4*>(smem);
        sh4[tid] = bsum;
        __syncthreads();
        if (tid < c4o) {
            float4 t = sh4[tid];
            for (int q = 1; q < dz_mstep; ++q) {
                const float4 u = sh4[q * c4o + tid];
                t.x += u.x;
                t.y += u.y;
                t.z += u.z;
                t.w += u.w;
            }
            *reinterpret_cast<float4*>(A.part_b + (size_t)blockIdx.x * cout + tid * 4) = t;
        }
    }
#endif
}

// wave roles / chunk size of the bf16x6 form; returns the dynamic LDS BYTES (0 = this form does not apply)
static size_t cnn_wgrad3_plan(CnnWgradArgs& a) {
    if (a.first || !split_mode() || tune_get("wgrad3", 1) == 0) return 0;
    if ((a.cin & 15) || (a.cout & 31) || a.cout > 256 || 512 % (a.cout / 4) != 0 || 512 % (a.cin / 4) != 0 ||
        a.cin % a.G != 0 || ((a.cin / a.G) & 3) || a.hin > 64 || a.rows <= 0)
        return 0;
    const int nct = a.cout / 32;
    if (nct != 1 && nct != 2 && nct != 4 && nct != 8) return 0;
    a.nct = nct;
    a.nkt = (a.K + 31) / 32;
    // waves: tgc = nct co tiles x tgk k-tile groups x ms row-step phases; NKT k tiles per wave (5, or 9 when a
    // workgroup would otherwise need a second slab = a second staging pass over every patch)
    int best_nkt = 0, best_tgk = 0, best_slabs = 1 << 30;
    // (3 k tiles per wave at <= 128 registers - two workgroups per CU - spills 31 registers into the matrix loop:
    // not instantiated)
    for (int nktw : {5, 9}) {
        const int tgk_max = 8 / nct;
        int tgk = 1;
        while (tgk < tgk_max && tgk * nktw < a.nkt) tgk <<= 1;
        const int slabs = (int)cdiv(a.nkt, tgk * nktw);
        if (slabs < best_slabs) best_slabs = slabs, best_nkt = nktw, best_tgk = tgk;
    }
    if (best_slabs > 2) return 0;
    a.slabs = best_slabs;
    a.nkt_slab = (int)cdiv(a.nkt, a.slabs);
    a.slabs = (int)cdiv(a.nkt, a.nkt_slab);
    a.tgc = nct;
    a.tgk = best_tgk;
    a.ms = 8 / (nct * best_tgk);
    a.sct = 1;
    a.skt = best_nkt;
    a.hp = a.hin + 2;
    // row strides in bf16 elements: consecutive m (two pixels / one dZ row apart) 64 bytes apart modulo the 256-byte
    // bank row, so the four rows of a 16-lane group and the two column halves (+32 bytes) never share a bank
    a.cs = a.cin;
    while ((4 * a.cs) % 256 != 64 && (4 * a.cs) % 256 != 192) a.cs += 8;
    a.zs = a.cout;
    while ((2 * a.zs) % 256 != 64 && (2 * a.zs) % 256 != 192) a.zs += 8;
    a.in_per = a.hp * a.hp * a.cs;
    a.dP = make_fdiv(a.P);
    a.dhout = make_fdiv(a.hout);
    a.dPin = make_fdiv(a.hin * a.hin);
    a.dhin = make_fdiv(a.hin);
    a.dcin = make_fdiv(a.cin);
    if (best_nkt == 0) return 0;
    const int lds_cap_kb = best_nkt == 9 ? 150 : 76;
    int rb_cap = 16;
    if (rb_cap < 1) rb_cap = 1;
    for (int v = 0; v < 2; ++v) {
        const int PD = kWgPref[0][v][0], PI = kWgPref[0][v][1];
        for (int rb = rb_cap; rb >= 1; --rb) {
            const int M = rb * a.P, Mpad = (M + 15) & ~15;
            if ((int64_t)Mpad * a.cout > (int64_t)PD * 512 * 4) continue;
            if ((int64_t)rb * a.hin * a.hin * a.cin > (int64_t)PI * 512 * 4) continue;
            const size_t in_plane = (((size_t)rb * a.in_per * 2) + 15) & ~(size_t)15;
            const size_t off_dz = 3 * in_plane;
            const size_t off_tab = off_dz + 3 * (size_t)Mpad * a.zs * 2;
            size_t tot = off_tab + (size_t)Mpad * 4;
            if (tot < 32768) tot = 32768;  // the final tile reduction (8 waves x 64 lanes x 16 floats) lives at the start of LDS
            tot = (tot + 15) & ~(size_t)15;
            if (tot > (size_t)lds_cap_kb * 1024 && rb > 1) continue;
            if (tot > 150 * 1024) break;
            a.rb = rb;
            a.pd = PD;
            a.pi = PI;
            a.in_plane = (int)in_plane;
            a.off_dz = (int)off_dz;
            a.off_tab = (int)off_tab;
            a.lds_floats = (int)(tot / 4);
            a.nchunks = (int)cdiv(a.rows, rb);
            const int per_cu = best_nkt == 9 ? 1 : 2;
            int blocks = 256 * per_cu / a.slabs;
            if (blocks < 64) blocks = 64;
            a.blocks = a.nchunks < blocks ? a.nchunks : blocks;
            return tot;
        }
    }
    return 0;
}

// picks the wave roles and the chunk size; returns the dynamic LDS floats (0 = unsupported)
static size_t cnn_wgrad_plan(CnnWgradArgs& a) {
    if ((a.cout & 3) || 512 % (a.cout / 4) != 0 || a.cout > 2048) return 0;
    if (!a.first && ((a.cin & 3) || 512 % (a.cin / 4) != 0 || a.cin % a.G != 0 || ((a.cin / a.G) & 3)))
        return 0;
    if (a.hin > 200 || a.rows <= 0) return 0;
    if (a.first && (int64_t)a.c_img * a.H * a.W >= (1ll << 31)) return 0;
    a.nct = (a.cout + 15) / 16;
    a.nkt = (a.K + 15) / 16;
    // tile slabs over grid.y when one workgroup (8 waves x <= 18 tile slots) cannot hold them all
    const int max_tiles = 8 * 18;
    a.slabs = (int)cdiv((int64_t)a.nct * a.nkt, max_tiles);
    a.nkt_slab = (int)cdiv(a.nkt, a.slabs);
    a.slabs = (int)cdiv(a.nkt, a.nkt_slab);
    // wave roles: best slot utilisation, then the fewest row-step phases
    static const int kShapes[][2] = {{1, 1}, {1, 2}, {1, 3}, {1, 5}, {1, 9}, {2, 5}, {2, 9}};
    double best = -1.0;
    int bct = 0, bkt = 0, btgc = 0, btgk = 0, bms = 0;
    for (const auto& s : kShapes) {
        if (a.first && (s[0] != 1 || s[1] > 3)) continue;  // instantiated shapes
        for (int tgc = 1; tgc <= 8; tgc *= 2)
            for (int tgk = 1; tgc * tgk <= 8; tgk *= 2) {
                const int ms = 8 / (tgc * tgk);
                if (tgc * s[0] < a.nct || tgk * s[1] < a.nkt_slab) continue;
                const double util = (double)a.nct * a.nkt_slab * ms / (8.0 * s[0] * s[1]);
                const double score = util - 1e-3 * ms - 1e-4 * s[0] * s[1];
                if (score > best) {
                    best = score;
                    bct = s[0];
                    bkt = s[1];
                    btgc = tgc;
                    btgk = tgk;
                    bms = ms;
                }
            }
    }
    if (best < 0) return 0;
    a.tgc = btgc;
    a.tgk = btgk;
    a.ms = bms;
    a.sct = bct;
    a.skt = bkt;
    a.hp = a.hin + 2;
    a.cs = (a.cin & 3) ? a.cin : a.cin + 8;
    a.zs = ((a.cout + 15) / 32) * 32 + 16;
    a.in_per = a.hp * a.hp * a.cs;
    a.dP = make_fdiv(a.P);
    a.dhout = make_fdiv(a.hout);
    a.dPin = make_fdiv(a.hin * a.hin);
    a.dhin = make_fdiv(a.hin);
    a.dcin = make_fdiv(a.cin);
    a.dc4o = make_fdiv(a.cout / 4);
    a.dc4i = make_fdiv(a.cin >= 4 ? a.cin / 4 : 1);
    a.dpe = make_fdiv(a.cin * a.hin * a.hin);
    a.dff = make_fdiv(a.hin * a.hin);
    a.df = make_fdiv(a.hin);
    const int lds_cap_kb = 76;
    int rb_cap = 16;
    if (rb_cap < 1) rb_cap = 1;
    const int wg_per_cu = 0;  // 0 = by layer size
    const size_t red = a.ms > 1 ? (size_t)8 * bct * bkt * 256 : 0;
    // the shallow prefetch variant first (fewer registers -> two workgroups per CU)
    for (int v = 0; v < 2; ++v) {
        const int PD = kWgPref[a.first ? 1 : 0][v][0], PI = kWgPref[a.first ? 1 : 0][v][1];
        for (int rb = rb_cap; rb >= 1; --rb) {
            const int M = rb * a.P, Mpad = (M + 3) & ~3;
            if ((int64_t)M * a.cout > (int64_t)PD * 512 * 4) continue;
            if (a.first ? (int64_t)rb * a.cin * a.hin * a.hin > (int64_t)PI * 512
                        : (int64_t)rb * a.hin * a.hin * a.cin > (int64_t)PI * 512 * 4)
                continue;
            size_t off = ((size_t)rb * a.in_per + 3) & ~(size_t)3;
            const size_t off_dz = off;
            off += (size_t)Mpad * a.zs;
            const size_t off_tab = off;
            off += Mpad;
            size_t tot = off > red ? off : red;
            if (tot < 2048) tot = 2048;  // bias reduction scratch
            if (tot * sizeof(float) > (size_t)lds_cap_kb * 1024 && rb > 1) continue;
            if (tot * sizeof(float) > 150 * 1024) break;
            a.rb = rb;
            a.pd = PD;
            a.pi = PI;
            a.off_dz = (int)off_dz;
            a.off_tab = (int)off_tab;
            a.lds_floats = (int)tot;
            a.nchunks = (int)cdiv(a.rows, rb);
            // persistent workgroups: an upper bound here (it sizes the partial-slab scratch); the
            // launcher trims it to what is actually resident.  Small layers are staging-bound and
            // want more workgroups per CU in flight, their slabs are small.
            const int per_cu = wg_per_cu > 0 ? wg_per_cu : ((int64_t)a.cout * a.K <= 8192 ? 4 : 2);
            int blocks = 256 * per_cu / a.slabs;
            if (blocks < 64) blocks = 64;
            a.blocks = a.nchunks < blocks ? a.nchunks : blocks;
            return tot;
        }
    }
    return 0;
}

int cnn_wgrad_supported(const CnnWgradArgs& a0) {
    if (getenv("MARL_CNN_FUSED") && getenv("MARL_CNN_FUSED")[0] == '0') return 0;
    CnnWgradArgs a = a0;
    return cnn_wgrad_plan(a) > 0;  // (the bf16x6 form covers a subset of these shapes)
}

int cnn_wgrad_blocks(const CnnWgradArgs& a0) {
    // (an upper bound for the scratch: the larger of the two forms' workgroup counts)
    CnnWgradArgs a = a0, b = a0;
    const int n1 = cnn_wgrad_plan(a) > 0 ? a.blocks : 0;
    const int n3 = cnn_wgrad3_plan(b) > 0 ? b.blocks : 0;
    return n1 > n3 ? n1 : n3;
}

template <int NCT, int NKT, bool FIRST, int PD, int PI>
static int wgrad_launch(const CnnWgradArgs& a, size_t lds, hipStream_t st) {
    auto kern = cnn_wgrad_kernel<NCT, NKT, FIRST, PD, PI>;
    if (lds > 64 * 1024) {
        static bool raised = false;  // per process; one process drives one GPU (see marl_hip.h)
        if (!raised) {
            MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            raised = true;
        }
    }
    prof_before(5, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)a.blocks, (unsigned)a.slabs), dim3(512), lds, st, a);
    prof_after(5, st);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// resident workgroups per CU of one instantiation at this LDS size (registers, LDS, wave slots)
template <int NCT, int NKT, bool FIRST, int PD, int PI>
static int wgrad_occupancy(size_t lds) {
    static size_t seen_lds[8];  // (the same instantiation serves layers with different LDS sizes)
    static int seen_occ[8], nseen = 0;
    for (int i = 0; i < nseen; ++i)
        if (seen_lds[i] == lds) return seen_occ[i];
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &occ, reinterpret_cast<const void*>(cnn_wgrad_kernel<NCT, NKT, FIRST, PD, PI>), 512, lds) !=
            hipSuccess || occ < 1)
        occ = 1;
    if (nseen < 8) {
        seen_lds[nseen] = lds;
        seen_occ[nseen++] = occ;
    }
    return occ;
}

template <int NKT, int PD, int PI>
static int wgrad3_launch(CnnWgradArgs& a, float* part_w, size_t lds, hipStream_t st) {
    auto kern = cnn_wgrad3_kernel<NKT, PD, PI>;
    static bool raised = false;  // per process; one process drives one GPU (see marl_hip.h)
    if (!raised) {
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        raised = true;
    }
    static size_t seen_lds[8];
    static int seen_occ[8], nseen = 0;
    int occ = 0;
    for (int i = 0; i < nseen; ++i)
        if (seen_lds[i] == lds) occ = seen_occ[i];
    if (!occ) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(kern), 512, lds) != hipSuccess ||
            occ < 1)
            occ = 1;
        if (nseen < 8) seen_lds[nseen] = lds, seen_occ[nseen++] = occ;
    }
    int res = 256 * occ / a.slabs;
    if (res < 64) res = 64;
    if (res < a.blocks) a.blocks = res;
    a.part_b = part_w + (size_t)a.blocks * a.cout * a.K;
    prof_before(5, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)a.blocks, (unsigned)a.slabs), dim3(512), lds, st, a);
    prof_after(5, st);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

int launch_cnn_wgrad(CnnWgradArgs& a, hipStream_t st) {
    float* part_w = a.part_w;
    {   // layers with >= 16 input channels: the bf16x6 form
        CnnWgradArgs b = a;
        const size_t lds3 = cnn_wgrad3_plan(b);
        if (lds3) {
            a = b;
            if (a.skt == 9) return a.pd == 4 ? wgrad3_launch<9, 4, 4>(a, part_w, lds3, st) : wgrad3_launch<9, 2, 2>(a, part_w, lds3, st);
            return a.pd == 4 ? wgrad3_launch<5, 4, 4>(a, part_w, lds3, st) : wgrad3_launch<5, 2, 2>(a, part_w, lds3, st);
        }
    }
    const size_t fl = cnn_wgrad_plan(a);
    if (!fl) {
        set_error("conv weight gradient: shape outside the fused kernel's range");
        return MARL_ELIMIT;
    }
    const size_t lds = fl * sizeof(float);
    const int sct = a.sct, skt = a.skt;
    const bool deep = a.pd == 4;
    // grid = what is resident at once (a persistent workgroup that has to wait for a slot only
    // lengthens the tail); never more than the bound the scratch was sized for
#define MARL_WG_GO(...)                                                                          \
    {                                                                                            \
        const int occ_ = wgrad_occupancy<__VA_ARGS__>(lds);                                      \
        int res_ = 256 * occ_ / a.slabs;                                                         \
        if (res_ < 64) res_ = 64;                                                                \
        if (res_ < a.blocks) a.blocks = res_;                                                    \
        a.part_b = part_w + (size_t)a.blocks * a.cout * a.K;                                     \
        return wgrad_launch<__VA_ARGS__>(a, lds, st);                                            \
    }
#define MARL_WG(CT, KT)                                                                          \
    if (sct == CT && skt == KT) {                                                                \
        if (a.first) {                                                                           \
            if (deep) MARL_WG_GO(1, (KT <= 3 ? KT : 1), true, 4, 14)                             \
            MARL_WG_GO(1, (KT <= 3 ? KT : 1), true, 2, 4)                                        \
        }                                                                                        \
        if (deep) MARL_WG_GO(CT, KT, false, 4, 4)                                                \
        MARL_WG_GO(CT, KT, false, 2, 2)                                                          \
    }
    MARL_WG(1, 1) MARL_WG(1, 2) MARL_WG(1, 3) MARL_WG(1, 5) MARL_WG(1, 9) MARL_WG(2, 5) MARL_WG(2, 9)
#undef MARL_WG
#undef MARL_WG_GO
    set_error("conv weight gradient: no kernel for tile shape %d x %d", sct, skt);
    return MARL_ELIMIT;
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
