// Device code shared by sample_kernel (rowops.hip) and the sampling workgroups that ride along
// with the decoder panel launch (panel.hip): policy output layer + softmax + multinomial
// (argmax p/q) + log-prob + bounded move (networks/policy.py:15-16, core/agent.py:53-61,
// core/environment.py:56-66,128-150) and the position embedding of the next step
// (networks/state.py:14-16).
#pragma once

#include "common.h"

namespace marl {

__device__ __forceinline__ float sample_silu(float y) { return silu_fast(y); }

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): a
// counter-based generator - four 32-bit words per (key, counter), no state to carry between
// kernels, so every row draws its own stream wherever and whenever it runs.
struct Philox4 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ Philox4 philox4x32_10(uint64_t key, uint64_t ctr_hi, uint32_t c0, uint32_t c1) {
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    Philox4 c{c0, c1, (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = Philox4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
// uniform in (0, 1): never 0 (log) and never 1
__device__ __forceinline__ float philox_u01(uint32_t v) { return ((float)(v >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// p[j] += sum over this lane's columns base + lane + 64u of a[u] * w1[j][col]; the output
// layer's rows are read four actions at a time so that all loads of a group are in flight
// (MAXA: compile-time bound of the action loops - 4 for the usual four moves, MARL_MAX_ACTIONS otherwise.
// The loops are fully unrolled; at 16 the kernel was 24 KB of code of which a four-action model executes a
// quarter, branching over the rest line by line)
template <int MAXA>
__device__ __forceinline__ void sample_logits_chunk(const SampleArgs& A, const float (&a)[8], int base,
                                                    float (&p)[MAXA], int lane) {
#pragma unroll
    for (int j0 = 0; j0 < MAXA; j0 += 4) {
        if (j0 < A.nA) {
            float wv[4][8];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = j0 + jj < A.nA ? j0 + jj : A.nA - 1;
                const float* wj = A.w1 + (size_t)j * A.ldw;
#pragma unroll
                for (int u = 0; u < 8; ++u) {  // (clamped address, no branch between two loads)
                    const int k = base + lane + 64 * u;
                    const float v = wj[k < A.nla ? k : 0];
                    wv[jj][u] = k < A.nla ? v : 0.f;
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float s = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) s += a[u] * wv[jj][u];
                p[j0 + jj] += s;
            }
        }
    }
}

// One wave = one row r: the activation row is read into registers 512 columns at a time
template <int MAXA>
__device__ __forceinline__ void sample_row_logits(const SampleArgs& A, int r, float (&p)[MAXA], int lane) {
    const float* ar = A.a_pol + (size_t)r * A.ld_a;
#pragma unroll
    for (int j = 0; j < MAXA; ++j) p[j] = 0.f;
    for (int base = 0; base < A.nla; base += 512) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = base + lane + 64 * u;
            const float v = ar[k < A.nla ? k : 0];
            a[u] = k < A.nla ? v : 0.f;
        }
        sample_logits_chunk<MAXA>(A, a, base, p, lane);
    }
}

// Everything of a row whose address does not depend on the draw, requested BEFORE the logits are waited
// for: the kernel is one dependent chain per row (loads -> logits -> draw -> move -> embedding) and every
// load left inside the chain costs it a round trip (the ISA of the round-3 form had ~35 dependent waits).
template <int MAXA>
struct SamplePre {
    float b1[MAXA];
    float nz[MAXA];
    uint64_t ctr;
    int pi0, pi1, forced;
    // position embedding: this lane's four consecutive columns 4 * lane .. + 3 (pe_nd <= 256, aligned)
    bool pe_fast;
    float4 qb, qw[4], qg, qbt;
};
__device__ __forceinline__ bool sample_pe_fast(const SampleArgs& A) {
    const uintptr_t al = (uintptr_t)A.pe_W | (uintptr_t)A.pe_b | (uintptr_t)A.pe_gamma | (uintptr_t)A.pe_beta |
                         (uintptr_t)A.pe_z | (uintptr_t)A.pe_out;
    return A.pe_W && A.step_logp && (al & 15) == 0 && A.pe_nd <= 256 && ((A.pe_nd | A.pe_ldz | A.pe_ldo) & 3) == 0 &&
           (A.pe_col0 & 3) == 0;
}
// (PE = false: without the embedding's parameters - for the ride-along form inside the 80-register panel kernel)
template <int MAXA, bool PE = true>
__device__ __forceinline__ void sample_prefetch(const SampleArgs& A, int r, int lane, SamplePre<MAXA>& S) {
#pragma unroll
    for (int j = 0; j < MAXA; ++j) S.b1[j] = A.b1[j < A.nA ? j : A.nA - 1];
    if (A.noise) {
        const float* nr = A.noise + (size_t)r * A.nA;
#pragma unroll
        for (int j = 0; j < MAXA; ++j) S.nz[j] = nr[j < A.nA ? j : A.nA - 1];
    } else {
#pragma unroll
        for (int j = 0; j < MAXA; ++j) S.nz[j] = 1.0f;
    }
    S.ctr = A.rng_ctr + (A.rng_off_dev ? (*A.rng_off_dev << 16) : 0ull);
    S.pi0 = S.pi1 = S.forced = 0;
    if (A.step_logp) {
        S.pi0 = A.pos_in[r * 2];
        S.pi1 = A.pos_in[r * 2 + 1];
        if (A.forced) S.forced = (int)A.forced[r];
    }
    S.pe_fast = PE && sample_pe_fast(A);
    if (PE && S.pe_fast) {
        const int j4 = 4 * lane < A.pe_nd ? 4 * lane : 0;
        S.qb = *reinterpret_cast<const float4*>(A.pe_b + j4);
        S.qg = *reinterpret_cast<const float4*>(A.pe_gamma + j4);
        S.qbt = *reinterpret_cast<const float4*>(A.pe_beta + j4);
#pragma unroll
        for (int q = 0; q < 4; ++q) S.qw[q] = *reinterpret_cast<const float4*>(A.pe_W + 4 * (j4 + q));
    }
}

// p[] holds this lane's partial logits of row r; every lane of the wave must call this
template <int MAXA>
__device__ __forceinline__ void sample_finish(const SampleArgs& A, int r, float (&p)[MAXA], int lane,
                                              const SamplePre<MAXA>& S) {
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXA; ++j) {
        if (j < A.nA) {
            p[j] = wave_sum(p[j]) + S.b1[j];
            mx = fmaxf(mx, p[j]);
        } else {
            p[j] = 0.f;
        }
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < MAXA; ++j)
        if (j < A.nA) {
            p[j] = expf(p[j] - mx);
            den += p[j];
        }
    int act = 0;
    float best = -INFINITY, pa = 0.f;
#pragma unroll
    for (int j = 0; j < MAXA; ++j)
        if (j < A.nA) {
            p[j] = p[j] / den;
            if (A.noise) {
                const float sc = p[j] / S.nz[j];
                if (sc > best) {
                    best = sc;
                    act = j;
                }
            }
        }
    if (!A.noise && A.rng_on) {
        // th.multinomial(p, 1) == argmax_j p_j / q_j with q ~ Exp(1) (core/agent.py:53-55); here
        // q_j = -log(u_j) from this row's own Philox stream (wave-uniform, every lane agrees)
#pragma unroll
        for (int j0 = 0; j0 < MAXA; j0 += 4) {
            if (j0 < A.nA) {
                const Philox4 u = philox4x32_10(A.rng_seed, S.ctr, (uint32_t)r, (uint32_t)(j0 >> 2));
                const uint32_t uv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if (j0 + jj < A.nA) {
                        const float sc = p[j0 + jj] / -__logf(philox_u01(uv[jj]));
                        if (sc > best) {
                            best = sc;
                            act = j0 + jj;
                        }
                    }
                }
            }
        }
    }
    if (!A.step_logp) {  // standalone step API: probabilities only
        if (lane == 0)
#pragma unroll
            for (int j = 0; j < MAXA; ++j)
                if (j < A.nA) A.probs[(size_t)r * A.nA + j] = p[j];
        return;
    }
    if (A.forced) {  // teacher forcing; out-of-range indices are clamped like transition_kernel does
        const int fa = S.forced;
        act = fa < 0 ? 0 : (fa >= A.nA ? A.nA - 1 : fa);
    }
    int mv0 = 0, mv1 = 0;  // (a select chain: the table sits in the kernel arguments)
#pragma unroll
    for (int j = 0; j < MAXA; ++j)
        if (j == act) {
            pa = p[j];
            mv0 = A.table[j][0];
            mv1 = A.table[j][1];
        }
    // every lane computes the move (wave-uniform), lane 0 stores it
    const int pi0 = S.pi0, pi1 = S.pi1;
    const int q0 = pi0 + mv0, q1 = pi1 + mv1;
    const bool ok = q0 >= 0 && q0 + A.f < A.H && q1 >= 0 && q1 + A.f < A.W;
    const int n0 = ok ? q0 : pi0, n1 = ok ? q1 : pi1;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < MAXA; ++j)
            if (j < A.nA) A.probs[(size_t)r * A.nA + j] = p[j];
        A.actions_i32[r] = act;
        A.step_logp[r] = logf(pa);
        A.pos_out[r * 2] = n0;
        A.pos_out[r * 2 + 1] = n1;
        if (A.step_pos) {
            A.step_pos[(size_t)r * 2] = n0;
            A.step_pos[(size_t)r * 2 + 1] = n1;
        }
        if (A.step_actions) A.step_actions[r] = act;
    }
    // position embedding of the NEXT step (networks/state.py:14-16 on pos / size), so that no
    // separate launch sits between the move and the next LSTM
    if (A.pe_W) {
        const float p0 = (float)n0 / (float)A.H, p1 = (float)n1 / (float)A.W;
        const int nd = A.pe_nd;
        if (lane == 0 && A.pe_npos) {
            A.pe_npos[(size_t)r * 4] = p0;
            A.pe_npos[(size_t)r * 4 + 1] = p1;
        }
        if (S.pe_fast) {
            // four consecutive columns per lane, every parameter already in registers
            const bool on = 4 * lane < nd;
            const float b4[4] = {S.qb.x, S.qb.y, S.qb.z, S.qb.w}, g4[4] = {S.qg.x, S.qg.y, S.qg.z, S.qg.w};
            const float t4[4] = {S.qbt.x, S.qbt.y, S.qbt.z, S.qbt.w};
            float v[4], sm = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = on ? b4[q] + p0 * S.qw[q].x + p1 * S.qw[q].y : 0.f;
                sm += v[q];
            }
            if (on) *reinterpret_cast<float4*>(A.pe_z + (size_t)r * A.pe_ldz + 4 * lane) = make_float4(v[0], v[1], v[2], v[3]);
            const float mean = wave_sum(sm) / (float)nd;
            float qq = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dd = on ? v[q] - mean : 0.f;
                qq += dd * dd;
            }
            const float rstd = 1.0f / sqrtf(wave_sum(qq) / (float)nd + 1e-5f);
            if (lane == 0 && A.pe_stats) {
                A.pe_stats[(size_t)r * 2] = mean;
                A.pe_stats[(size_t)r * 2 + 1] = rstd;
            }
            if (on) {
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = sample_silu((v[q] - mean) * rstd * g4[q] + t4[q]);
                *reinterpret_cast<float4*>(A.pe_out + (size_t)r * A.pe_ldo + 4 * lane) = make_float4(o[0], o[1], o[2], o[3]);
                if (A.pe_img) {  // the same values into the image of U[t+1]
                    const int col = A.pe_col0 + 4 * lane;
                    img_store4(A.pe_img + img_off((int64_t)A.pe_row0 + r, col >> 4, A.pe_steps), col, o[0], o[1], o[2], o[3]);
                }
            }
            return;
        }
        float s = 0.f;
        for (int j = lane; j < nd; j += 64) {
            const float v = A.pe_b[j] + p0 * A.pe_W[4 * j] + p1 * A.pe_W[4 * j + 1];
            A.pe_z[(size_t)r * A.pe_ldz + j] = v;
            s += v;
        }
        const float mean = wave_sum(s) / (float)nd;
        float q = 0.f;
        for (int j = lane; j < nd; j += 64) {
            const float dd = (A.pe_b[j] + p0 * A.pe_W[4 * j] + p1 * A.pe_W[4 * j + 1]) - mean;
            q += dd * dd;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)nd + 1e-5f);
        if (lane == 0 && A.pe_stats) {
            A.pe_stats[(size_t)r * 2] = mean;
            A.pe_stats[(size_t)r * 2 + 1] = rstd;
        }
        for (int j = lane; j < nd; j += 64) {
            const float v = A.pe_b[j] + p0 * A.pe_W[4 * j] + p1 * A.pe_W[4 * j + 1];
            A.pe_out[(size_t)r * A.pe_ldo + j] = sample_silu((v - mean) * rstd * A.pe_gamma[j] + A.pe_beta[j]);
        }
        if (A.pe_img) {  // the same values, four consecutive columns per lane, into the image of U[t+1]
            for (int j4 = 4 * lane; j4 < nd; j4 += 256) {
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = j4 + q;
                    const float v = A.pe_b[j] + p0 * A.pe_W[4 * j] + p1 * A.pe_W[4 * j + 1];
                    o[q] = sample_silu((v - mean) * rstd * A.pe_gamma[j] + A.pe_beta[j]);
                }
                const int col = A.pe_col0 + j4;
                img_store4(A.pe_img + img_off((int64_t)A.pe_row0 + r, col >> 4, A.pe_steps), col, o[0], o[1], o[2], o[3]);
            }
        }
    }
}

}  // namespace marl
