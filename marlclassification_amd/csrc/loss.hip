// A2C loss of Trainer.train_epoch (training/trainer.py:76-111) and its gradient with
// respect to the episode outputs, in a handful of small HBM-bound kernels:
//   rewards (functions.py:7-32) -> vote error (trainer.py:78-87) -> discounted returns
//   (functions.py:35-51, flip-cumsum order) -> global mean / unbiased std
//   (functions.py:54-55) -> path / critic terms and gradients (trainer.py:96-111).
// All cross-thread sums go through fixed-order partials (fp64) -> bit-reproducible.
#include "common.h"

namespace marl {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

struct LossLayout {
    double* part_adv;   // [blocksC][2]
    double* part_loss;  // [blocksE][2]
    double* part_err;   // [1]
    float* rew;         // [Ns*R]
    float* ret;         // [Ns*R]
    float* adv;         // [Ns*R]
    float* err;         // [Ns*Nb]
    int blocksC, blocksE;
};

static LossLayout loss_layout(float* scratch, int ns, int na, int nb) {
    LossLayout L;
    const int64_t R = (int64_t)na * nb, NR = R * ns;
    L.blocksC = (int)cdiv(R, 256);
    L.blocksE = (int)cdiv(NR, 256);
    double* d = reinterpret_cast<double*>(scratch);
    L.part_adv = d;
    L.part_loss = d + 2 * L.blocksC;
    L.part_err = L.part_loss + 2 * L.blocksE;
    float* f = reinterpret_cast<float*>(L.part_err + 2);
    L.rew = f;
    L.ret = f + NR;
    L.adv = f + 2 * NR;
    L.err = f + 3 * NR;
    return L;
}

size_t loss_scratch_floats(int ns, int na, int nb) {
    const int64_t R = (int64_t)na * nb, NR = R * ns;
    const int64_t dbl = 2 * cdiv(R, 256) + 2 * cdiv(NR, 256) + 2;
    return (size_t)(2 * dbl + 3 * NR + (int64_t)ns * nb + 16);
}

// rew[t, r] = (log nC - CE(preds[t, r, :], y[b])) / log nC ; one wave per (t, r)
__global__ __launch_bounds__(256) void loss_rewards_kernel(const float* __restrict__ preds,
                                                           const int64_t* __restrict__ y,
                                                           float* __restrict__ rew, int64_t NR,
                                                           int nb, int nc, float rnd) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= NR) return;
    const float* p = preds + row * nc;
    float mx = -INFINITY;
    for (int c = lane; c < nc; c += 64) mx = fmaxf(mx, p[c]);
    mx = wmax(mx);
    float s = 0.f;
    for (int c = lane; c < nc; c += 64) s += expf(p[c] - mx);
    s = wsum(s);
    if (lane == 0) {
        const int b = (int)(row % nb);
        const float ce = -(p[y[b]] - mx - logf(s));
        rew[row] = (rnd - ce) / rnd;
    }
}

// err[t, b] = CE(mean_a preds[t, a, b, :], y[b]);  g_preds[t, a, b, c] = (softmax - onehot) / R
__global__ __launch_bounds__(256) void loss_error_kernel(const float* __restrict__ preds,
                                                         const int64_t* __restrict__ y,
                                                         float* __restrict__ err,
                                                         float* __restrict__ g_preds, int ld_gp,
                                                         int ns, int na, int nb, int nc) {
    __shared__ float pbar[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tb = (int64_t)blockIdx.x * 4 + wave;
    if (tb >= (int64_t)ns * nb) return;
    const int t = (int)(tb / nb), b = (int)(tb % nb);
    const int64_t R = (int64_t)na * nb;
    float mx = -INFINITY;
    for (int c = lane; c < nc; c += 64) {
        float s = 0.f;
        for (int a = 0; a < na; ++a) s += preds[((int64_t)t * R + (int64_t)a * nb + b) * nc + c];
        s = s / (float)na;
        pbar[wave][c] = s;
        mx = fmaxf(mx, s);
    }
    mx = wmax(mx);
    float s = 0.f;
    for (int c = lane; c < nc; c += 64) s += expf(pbar[wave][c] - mx);
    s = wsum(s);
    const int yb = (int)y[b];
    if (lane == 0) err[tb] = -(pbar[wave][yb] - mx - logf(s));
    if (g_preds) {
        const float inv = 1.0f / (float)R;
        for (int c = lane; c < nc; c += 64) {
            const float g = (expf(pbar[wave][c] - mx) / s - (c == yb ? 1.0f : 0.0f)) * inv;
            for (int a = 0; a < na; ++a)
                g_preds[((int64_t)t * R + (int64_t)a * nb + b) * ld_gp + c] = g;
        }
    }
}

// thread per row r: returns by the reference's flip-cumsum-flip of rew * gamma^t, / gamma^t
__global__ __launch_bounds__(256) void loss_returns_kernel(const float* __restrict__ rew,
                                                           const float* __restrict__ values,
                                                           float* __restrict__ ret,
                                                           float* __restrict__ adv,
                                                           double* __restrict__ part, int ns,
                                                           int64_t R, float gamma) {
    __shared__ double sh[2][256];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    if (r < R) {
        float S = 0.f;
        for (int t = ns - 1; t >= 0; --t) {
            const float gt = powf(gamma, (float)t);
            S += rew[(int64_t)t * R + r] * gt;
            const float rt = S / gt;
            const float ad = rt - values[(int64_t)t * R + r];
            ret[(int64_t)t * R + r] = rt;
            adv[(int64_t)t * R + r] = ad;
            s1 += (double)ad;
            s2 += (double)ad * (double)ad;
        }
    }
    sh[0][threadIdx.x] = s1;
    sh[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = sh[0][0];
        part[2 * blockIdx.x + 1] = sh[1][0];
    }
}

// one 256-thread block; strided partial sums + fixed-order tree -> deterministic
__global__ __launch_bounds__(256) void loss_stats_kernel(
    const double* __restrict__ part, int nblocks, const float* __restrict__ err, int64_t nerr,
    double* __restrict__ part_err, double* __restrict__ adv_stats, double n) {
    __shared__ double sh[3][256];
    double s1 = 0.0, s2 = 0.0, e = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        s1 += part[2 * i];
        s2 += part[2 * i + 1];
    }
    for (int64_t i = threadIdx.x; i < nerr; i += 256) e += (double)err[i];
    sh[0][threadIdx.x] = s1;
    sh[1][threadIdx.x] = s2;
    sh[2][threadIdx.x] = e;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        adv_stats[0] = n;
        adv_stats[1] = sh[0][0];
        adv_stats[2] = sh[1][0];
        part_err[0] = sh[2][0];
    }
}

__global__ __launch_bounds__(256) void loss_grads_kernel(
    const float* __restrict__ logp, const float* __restrict__ values,
    const float* __restrict__ ret, const float* __restrict__ adv,
    const double* __restrict__ adv_stats, float* __restrict__ g_logp,
    float* __restrict__ g_values, double* __restrict__ part, int64_t NR, int64_t R) {
    __shared__ double sh[2][256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    if (i < NR) {
        const double n = adv_stats[0];
        const double mean_d = adv_stats[1] / n;
        const double var_d = (adv_stats[2] - adv_stats[1] * mean_d) / (n - 1.0);
        const float mean = (float)mean_d;
        const float sd = (float)sqrt(var_d > 0.0 ? var_d : 0.0);
        const float advn = (adv[i] - mean) / (sd + 1e-8f);
        const float invR = 1.0f / (float)R;
        s1 = (double)(-logp[i] * advn);
        const float d = values[i] - ret[i];
        const float ad = fabsf(d);
        s2 = (double)(ad < 1.0f ? 0.5f * d * d : ad - 0.5f);
        if (g_logp) g_logp[i] = -advn * invR;
        if (g_values) g_values[i] = (ad < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f)) * invR;
    }
    sh[0][threadIdx.x] = s1;
    sh[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = sh[0][0];
        part[2 * blockIdx.x + 1] = sh[1][0];
    }
}

// scalars = {loss, path.sum(0).mean(), error.mean(), critic.sum(0).mean()} (trainer.py:111,119-122)
__global__ __launch_bounds__(256) void loss_final_kernel(
    const double* __restrict__ part, int nblocks, const double* __restrict__ part_err,
    float* __restrict__ scalars, int ns, int nb, int64_t R) {
    __shared__ double sh[2][256];
    double p = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        p += part[2 * i];
        c += part[2 * i + 1];
    }
    sh[0][threadIdx.x] = p;
    sh[1][threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        p = sh[0][0];
        c = sh[1][0];
        const double path = p / (double)R, critic = c / (double)R;
        const double esum = part_err[0];
        scalars[0] = (float)(path + esum / (double)nb + critic);
        scalars[1] = (float)path;
        scalars[2] = (float)(esum / ((double)ns * nb));
        scalars[3] = (float)critic;
    }
}

int launch_loss(const LossArgs& a, hipStream_t st) {
    if (a.nc > 1024) {
        set_error("nb_class %d > 1024 unsupported by the loss kernel", a.nc);
        return MARL_ELIMIT;
    }
    const int64_t R = (int64_t)a.na * a.nb, NR = R * a.ns;
    LossLayout L = loss_layout(a.scratch, a.ns, a.na, a.nb);
    double* stats = reinterpret_cast<double*>(a.adv_stats);
    if (a.phase == 0 || a.phase == 1) {
        const float rnd = (float)log((double)a.nc);
        hipLaunchKernelGGL(loss_rewards_kernel, dim3((unsigned)cdiv(NR, 4)), dim3(256), 0, st,
                           a.preds, a.y, L.rew, NR, a.nb, a.nc, rnd);
        MARL_LAUNCH_CHECK();
        hipLaunchKernelGGL(loss_error_kernel, dim3((unsigned)cdiv((int64_t)a.ns * a.nb, 4)),
                           dim3(256), 0, st, a.preds, a.y, L.err, a.g_preds, a.ld_gp, a.ns, a.na,
                           a.nb, a.nc);
        MARL_LAUNCH_CHECK();
        hipLaunchKernelGGL(loss_returns_kernel, dim3((unsigned)L.blocksC), dim3(256), 0, st, L.rew,
                           a.values, L.ret, L.adv, L.part_adv, a.ns, R, a.gamma);
        MARL_LAUNCH_CHECK();
        hipLaunchKernelGGL(loss_stats_kernel, dim3(1), dim3(256), 0, st, L.part_adv, L.blocksC,
                           L.err, (int64_t)a.ns * a.nb, L.part_err, stats, (double)NR);
        MARL_LAUNCH_CHECK();
    }
    if (a.phase == 0 || a.phase == 2) {
        hipLaunchKernelGGL(loss_grads_kernel, dim3((unsigned)L.blocksE), dim3(256), 0, st, a.logp,
                           a.values, L.ret, L.adv, stats, a.g_logp, a.g_values, L.part_loss, NR, R);
        MARL_LAUNCH_CHECK();
        hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, L.part_loss, L.blocksE,
                           L.part_err, a.scalars, a.ns, a.nb, R);
        MARL_LAUNCH_CHECK();
    }
    return MARL_OK;
}

}  // namespace marl
