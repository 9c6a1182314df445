// fp32 matrix kernels on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32,
// bitwise an fmaf chain, 157 TF peak - MI355X_MICROARCH.md).
//
//  gemm_nt_kernel : C[M,N] (+)= sum_s A_s[M,K_s] * B_s[N,K_s]^T + bias   (activations x weights)
//                   optional fused LSTM-cell epilogue (networks/recurrent.py:19-35)
//  gemm_tn_kernel : C[NI,NJ] = sum_r A[r,i] * B[r,j]                      (weight gradients)
//
// Tiling is for 64-lane waves: a 256-thread workgroup = 4 waves, each wave owns
// TM x TN accumulator tiles of 32x32 (16 VGPRs each).  Operand tiles are staged
// global -> registers -> LDS (double buffered, one barrier per K tile); LDS rows are
// padded to 20 floats so that the ds_read_b128 fragment reads are bank-conflict free.
#include <stdlib.h>

#include "common.h"

namespace marl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;          // row depth of one staged tile of the TN kernel


// (gate non-linearities of the fused LSTM epilogue: sigmoid_acc / tanh_fast of common.h; the accurate
// libm forms cost ~10 us per launch in the epilogue of a 120 us kernel)

// LDS-only workgroup barrier: the prefetch loads of the next K tile stay in flight across it
// (a plain __syncthreads() also drains vmcnt).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// GROUPS = 2 ("ping-pong"): 8 waves, two per SIMD.  Group g owns rows [g * BM, (g + 1) * BM) of
// a (2 * BM) x BN tile and half of the B rows of every K tile.  The groups run the same loop
// half a step apart - while one is in its matrix phase (64 back-to-back MFMAs per wave) the
// other waits for its prefetch, writes the next tile to LDS and issues the following loads -
// so the matrix pipe of every SIMD always has exactly one wave feeding it.  Two co-resident
// 4-wave workgroups running the plain loop (GROUPS = 1) start together and tend to stay in
// lockstep, staging at the same time with the pipe idle.
template <int BM, int BN, int WM, int WN, bool LSTM, int GROUPS>
__global__ __launch_bounds__(256 * GROUPS) void gemm_nt_kernel(const GemmBatch batch) {
    constexpr int BK = 32;
    constexpr int LDS_K = BK + 4;  // padded LDS row stride (floats): conflict-free b128 reads
    constexpr int KC = BK / 4;     // float4 chunks per tile row
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int A_CH = BM * KC / 256;            // A chunks per thread (own row half)
    constexpr int B_CH = BN * KC / 256 / GROUPS;   // B chunks per thread (own share of the rows)
    constexpr int BUF = (GROUPS * BM + BN) * LDS_K;
    static_assert(WM * WN == 4, "4 waves per group");
    static_assert(A_CH >= 1 && A_CH <= 4 && B_CH >= 1 && B_CH <= 4, "1..4 chunks per thread");
    static_assert(!LSTM || (WN == 1 && BN == 128), "LSTM tile = 4 gates x 32 units");

    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][BUF]

    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (batch.xcd_map) xcd_tile(batch.gx, batch.gy, batch.count, bx, by, bz);
    // by value: the fields of a dynamically indexed kernel argument are otherwise re-loaded (scalar
    // load + s_waitcnt lgkmcnt(0)) at every use, e.g. once per stored row in the epilogue
    const GemmProb P = batch.p[bz];
    const int M = P.m;
    const int N = P.n;  // LSTM: number of hidden units (B has 4*N rows)
    const int n0 = by * (LSTM ? 32 : BN);
    if ((int)(bx * (BM * GROUPS)) >= M || n0 >= N) return;

    const int tid = threadIdx.x & 255;  // thread index inside its group
    const int grp = GROUPS == 2 ? (int)(threadIdx.x >> 8) : 0;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave % WN;
    const int m0 = bx * (BM * GROUPS) + grp * BM;  // first row of this group

    const int t0 = (P.seg[0].k + BK - 1) / BK;
    const int t1 = P.nseg > 1 ? (P.seg[1].k + BK - 1) / BK : 0;
    const int T = t0 + t1;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Staging: thread t of a group moves A chunks c = t + 256 * i (tile row c / KC of the group's
    // rows) and B chunks of the group's share of the B rows, floats [(t % KC) * 4, +4) of the K
    // tile.  Addresses are a UNIFORM base (segment pointer + k0, advanced on the scalar unit)
    // plus a fixed per-thread 32-bit byte offset, so the steady state issues no vector integer
    // work beside the matrix instructions.  Rows beyond M / N are clamped to the last valid row
    // (their results are never stored).
    const int koff = (tid % KC) * 4;
    uint32_t aofX0 = 0, aofX1 = 0, aofX2 = 0, aofX3 = 0, bofX0 = 0, bofX1 = 0, bofX2 = 0, bofX3 = 0;
    const char* abase = nullptr;
    const char* bbase = nullptr;
    int K4cur = 0;
#define MARL_SETA(idx_, sg_)                                                               \
    if (A_CH > (idx_)) {                                                                   \
        int gm_ = m0 + (tid + 256 * (idx_)) / KC;                                          \
        gm_ = gm_ < M ? gm_ : M - 1;                                                       \
        aofX##idx_ = ((uint32_t)gm_ * (uint32_t)P.seg[sg_].lda + (uint32_t)koff) * 4u;     \
    }
#define MARL_SETB(idx_, sg_)                                                               \
    if (B_CH > (idx_)) {                                                                   \
        const int row_ = (tid + 256 * (idx_)) / KC + grp * (BN / GROUPS);                  \
        int gn_;                                                                           \
        if (LSTM) {                                                                        \
            int unit_ = n0 + (row_ & 31);                                                  \
            unit_ = unit_ < N ? unit_ : N - 1;                                             \
            gn_ = (row_ >> 5) * N + unit_;                                                 \
        } else {                                                                           \
            gn_ = n0 + row_;                                                               \
            gn_ = gn_ < N ? gn_ : N - 1;                                                   \
        }                                                                                  \
        bofX##idx_ = ((uint32_t)gn_ * (uint32_t)P.seg[sg_].ldb + (uint32_t)koff) * 4u;     \
    }
#define MARL_SET_SEG(sg_)                                                                  \
    {                                                                                      \
        MARL_SETA(0, sg_) MARL_SETA(1, sg_) MARL_SETA(2, sg_) MARL_SETA(3, sg_)            \
        MARL_SETB(0, sg_) MARL_SETB(1, sg_) MARL_SETB(2, sg_) MARL_SETB(3, sg_)            \
        abase = reinterpret_cast<const char*>(P.seg[sg_].a);                               \
        bbase = reinterpret_cast<const char*>(P.seg[sg_].b);                               \
        K4cur = (P.seg[sg_].k + 3) & ~3;                                                   \
    }

    // Loads are UNCONDITIONAL (a predicated load makes hipcc branch around it and drain vmcnt
    // per load).  Only the LAST K tile of a segment can reach past round4(K): there the chunk
    // address is clamped and the B side is zeroed when the tile goes to LDS (weights are
    // finite, so finite-garbage * 0 is exact); all other tiles take the mask-free path.
    // two sets of staging registers (X, Y): the small-tile plan keeps TWO tiles in flight
    float4 raX0, raX1, raX2, raX3, rbX0, rbX1, rbX2, rbX3;
    float4 raY0, raY1, raY2, raY3, rbY0, rbY1, rbY2, rbY3;
    float mkX = 1.f, mkY = 1.f;
    bool maskedX = false, maskedY = false;  // uniform: the tile held in the set is a tail tile
#define MARL_LOADA(S_, idx_, d_) \
    if (A_CH > (idx_)) ra##S_##idx_ = *reinterpret_cast<const float4*>(abase + (aofX##idx_ + (d_)));
#define MARL_LOADB(S_, idx_, d_) \
    if (B_CH > (idx_)) rb##S_##idx_ = *reinterpret_cast<const float4*>(bbase + (bofX##idx_ + (d_)));
#define MARL_LOAD_ALL(S_, d_)                                                              \
    MARL_LOADA(S_, 0, d_) MARL_LOADA(S_, 1, d_) MARL_LOADA(S_, 2, d_) MARL_LOADA(S_, 3, d_) \
    MARL_LOADB(S_, 0, d_) MARL_LOADB(S_, 1, d_) MARL_LOADB(S_, 2, d_) MARL_LOADB(S_, 3, d_)
#define MARL_LOAD_TILE_S(S_, tile_)                                                        \
    {                                                                                      \
        if ((tile_) == t0) MARL_SET_SEG(1)                                                 \
        const int k0_ = ((tile_) >= t0 ? (tile_) - t0 : (tile_)) * BK;                     \
        masked##S_ = k0_ + BK > K4cur;                                                     \
        const int k_ = k0_ + koff;                                                         \
        const uint32_t d_ = (!masked##S_ || k_ < K4cur) ? 0u : (uint32_t)((K4cur - 4 - k_) * 4); \
        mk##S_ = (!masked##S_ || k_ < K4cur) ? 1.f : 0.f;                                  \
        MARL_LOAD_ALL(S_, d_)                                                              \
        abase += BK * 4;                                                                   \
        bbase += BK * 4;                                                                   \
    }
#define MARL_LOAD_TILE(tile_) MARL_LOAD_TILE_S(X, tile_)
// UNCONDITIONAL form for the two-tiles-ahead loop: a load behind a branch makes the compiler wait
// for ALL outstanding loads (vmcnt(0)) wherever it needs the older set, which would serialise
// the two sets again.  Tiles past the end re-read the last tile (valid addresses, never stored).
#define MARL_LOAD_TILE_U(S_, tile_)                                                        \
    {                                                                                      \
        const bool live_ = (tile_) < T;                                                    \
        if (live_ && (tile_) == t0) MARL_SET_SEG(1)                                        \
        const int tc_ = live_ ? (tile_) : T - 1;                                           \
        const int k0_ = (tc_ >= t0 ? tc_ - t0 : tc_) * BK;                                 \
        masked##S_ = k0_ + BK > K4cur;                                                     \
        const int k_ = k0_ + koff;                                                         \
        const uint32_t d_ = (!masked##S_ || k_ < K4cur) ? 0u : (uint32_t)((K4cur - 4 - k_) * 4); \
        mk##S_ = (!masked##S_ || k_ < K4cur) ? 1.f : 0.f;                                  \
        abase -= live_ ? 0 : BK * 4; /* (64-bit base: the 32-bit lane offsets must not wrap) */ \
        bbase -= live_ ? 0 : BK * 4;                                                       \
        MARL_LOAD_ALL(S_, d_)                                                              \
        abase += BK * 4;                                                                   \
        bbase += BK * 4;                                                                   \
    }
#define MARL_STOREA(S_, idx_)                                                              \
    if (A_CH > (idx_))                                                                     \
        *reinterpret_cast<float4*>(As_ + ((tid + 256 * (idx_)) / KC) * LDS_K + koff) = ra##S_##idx_;
#define MARL_STOREB(S_, idx_)                                                              \
    if (B_CH > (idx_)) {                                                                   \
        if (masked##S_) {                                                                  \
            rb##S_##idx_.x *= mk##S_;                                                      \
            rb##S_##idx_.y *= mk##S_;                                                      \
            rb##S_##idx_.z *= mk##S_;                                                      \
            rb##S_##idx_.w *= mk##S_;                                                      \
        }                                                                                  \
        *reinterpret_cast<float4*>(Bs_ + ((tid + 256 * (idx_)) / KC) * LDS_K + koff) = rb##S_##idx_; \
    }
#define MARL_STORE_TILE_S(S_, buf_)                                                        \
    {                                                                                      \
        float* As_ = smem + (buf_) * BUF + grp * BM * LDS_K;                               \
        float* Bs_ = smem + (buf_) * BUF + (GROUPS * BM + grp * (BN / GROUPS)) * LDS_K;    \
        MARL_STOREA(S_, 0) MARL_STOREA(S_, 1) MARL_STOREA(S_, 2) MARL_STOREA(S_, 3)        \
        MARL_STOREB(S_, 0) MARL_STOREB(S_, 1) MARL_STOREB(S_, 2) MARL_STOREB(S_, 3)        \
    }
#define MARL_STORE_TILE(buf_) MARL_STORE_TILE_S(X, buf_)
    // Matrix phase: all fragments of an 8-deep K group are read first, then the MFMAs walk the
    // accumulators round-robin (k-major), so that consecutive matrix instructions never hit
    // the same accumulator and nothing else is issued between them.
#define MARL_MFMA_Q(q_)                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                         \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                     \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].q_, b4[j].q_, acc[i][j], 0, 0, 0);
#define MARL_COMPUTE_TILE(buf_)                                                            \
    {                                                                                      \
        const float* As = smem + (buf_) * BUF +                                            \
                          (grp * BM + wm * (BM / WM) + frag_row) * LDS_K + frag_k;         \
        const float* Bs = smem + (buf_) * BUF + GROUPS * BM * LDS_K +                      \
                          (wn * (BN / WN) + frag_row) * LDS_K + frag_k;                    \
        if (LSTM) __builtin_amdgcn_s_setprio(1); /* compile-time: see launch_nt_variant */ \
        _Pragma("unroll") for (int kk = 0; kk < BK / 8; ++kk) {                            \
            float4 a4[TM], b4[TN];                                                         \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                 \
                a4[i] = *reinterpret_cast<const float4*>(As + i * 32 * LDS_K + kk * 8);    \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                 \
                b4[j] = *reinterpret_cast<const float4*>(Bs + j * 32 * LDS_K + kk * 8);    \
            MARL_MFMA_Q(x) MARL_MFMA_Q(y) MARL_MFMA_Q(z) MARL_MFMA_Q(w)                    \
        }                                                                                  \
        if (LSTM) __builtin_amdgcn_s_setprio(0);                                           \
    }

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;

    // LSTM: the previous cell state and the biases are only needed by the epilogue; issuing their
    // loads here keeps them in flight behind the whole K loop instead of starting a dependent
    // HBM round trip when every workgroup of the launch reaches its epilogue at the same time
    float cprev[LSTM ? 16 : 1];
    float lbias[LSTM ? 4 : 1];
    if (LSTM) {
        const int unit_ = n0 + (lane & 31);
        const int uc_ = unit_ < N ? unit_ : N - 1;
#pragma unroll
        for (int g_ = 0; g_ < 4; ++g_) lbias[LSTM ? g_ : 0] = P.bias[g_ * N + uc_];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row_ = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            row_ = row_ < M ? row_ : M - 1;
            cprev[LSTM ? r : 0] = P.c_prev[(size_t)row_ * P.ld_state + uc_];
        }
    }

    MARL_SET_SEG(0)
    MARL_LOAD_TILE(0)
    constexpr bool PF2 = GROUPS == 1 && !LSTM && A_CH + B_CH <= 4;
    // 128-wide tiles: the same two-tiles-ahead loads with ONE LDS stage (a second barrier per tile)
    // - 168 registers allow three workgroups per CU, two 37 KB stages only two (-1.5 % on the class)
    constexpr bool PF2S = GROUPS == 1 && !LSTM && A_CH + B_CH > 4;
    if (PF2S && batch.single_buf) {
        MARL_LOAD_TILE_U(Y, 1)
        int tile = 0;
        for (; tile + 1 < T; tile += 2) {
            if (tile > 0) lds_barrier();
            MARL_STORE_TILE_S(X, 0)
            lds_barrier();
            MARL_LOAD_TILE_U(X, tile + 2)
            MARL_COMPUTE_TILE(0)
            lds_barrier();
            MARL_STORE_TILE_S(Y, 0)
            lds_barrier();
            MARL_LOAD_TILE_U(Y, tile + 3)
            MARL_COMPUTE_TILE(0)
        }
        if (tile < T) {
            if (tile > 0) lds_barrier();
            MARL_STORE_TILE_S(X, 0)
            lds_barrier();
            MARL_COMPUTE_TILE(0)
        }
    } else if (PF2) {
        MARL_LOAD_TILE_U(Y, 1)
        int tile = 0;
        for (; tile + 1 < T; tile += 2) {  // pairs of tiles: no branch around any load
            MARL_STORE_TILE_S(X, 0)
            lds_barrier();
            MARL_LOAD_TILE_U(X, tile + 2)
            MARL_COMPUTE_TILE(0)
            MARL_STORE_TILE_S(Y, 1)
            lds_barrier();
            MARL_LOAD_TILE_U(Y, tile + 3)
            MARL_COMPUTE_TILE(1)
        }
        if (tile < T) {  // odd tile count: the last tile sits in X
            MARL_STORE_TILE_S(X, 0)
            lds_barrier();
            MARL_COMPUTE_TILE(0)
        }
    } else if (GROUPS == 1) {
        // One barrier per tile: LDS buffer b was last read two tiles ago and every wave has
        // passed the previous tile's barrier only after finishing those reads.
        for (int tile = 0; tile < T; ++tile) {
            const int buf = tile & 1;
            MARL_STORE_TILE(buf)
            lds_barrier();
            if (tile + 1 < T) MARL_LOAD_TILE(tile + 1)
            MARL_COMPUTE_TILE(buf)
        }
    } else {
        // Half-steps h = 0, 1, 2, ...: group 0 computes tile t at h = 2t and stages tile t + 1
        // at h = 2t + 1; group 1 stages tile t + 1 at h = 2t (one step ahead, so that its half
        // of the B rows is in LDS before group 0 needs it) and computes tile t at h = 2t + 1.
        // A barrier closes every half-step; both groups execute the same number of them.
        MARL_STORE_TILE(0)
        if (grp == 1 && T > 1) MARL_LOAD_TILE(1)
        lds_barrier();
        if (grp == 0) {
            for (int tile = 0; tile < T; ++tile) {
                if (tile + 1 < T) MARL_LOAD_TILE(tile + 1)
                MARL_COMPUTE_TILE(tile & 1)
                lds_barrier();
                if (tile + 1 < T) MARL_STORE_TILE((tile + 1) & 1)
                lds_barrier();
            }
        } else {
            for (int tile = 0; tile < T; ++tile) {
                if (tile + 1 < T) MARL_STORE_TILE((tile + 1) & 1)
                if (tile + 2 < T) MARL_LOAD_TILE(tile + 2)
                lds_barrier();
                MARL_COMPUTE_TILE(tile & 1)
                lds_barrier();
            }
        }
    }
#undef MARL_SETA
#undef MARL_SETB
#undef MARL_SET_SEG
#undef MARL_LOADA
#undef MARL_LOADB
#undef MARL_LOAD_ALL
#undef MARL_LOAD_TILE
#undef MARL_LOAD_TILE_S
#undef MARL_LOAD_TILE_U
#undef MARL_STORE_TILE_S
#undef MARL_STOREA
#undef MARL_STOREB
#undef MARL_STORE_TILE
#undef MARL_COMPUTE_TILE
#undef MARL_MFMA_Q

    // ---- epilogue: acc[i][j][r] is C[row(r), col], col = lane & 31,
    //      row(r) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
    if (!LSTM) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * (BN / WN) + j * 32 + col_l;
                if (col >= N) continue;
                const float bv = P.bias ? P.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                    if (row < M) {
                        float* cp = P.c + (size_t)row * P.ldc + col;
                        float v = acc[i][j][r] + bv;
                        if (P.accumulate) v += *cp;
                        *cp = v;
                    }
                }
            }
    } else {
        const int unit = n0 + col_l;
        if (unit < N) {
            const float bi = lbias[0], bf = lbias[LSTM ? 1 : 0], bg = lbias[LSTM ? 2 : 0], bo = lbias[LSTM ? 3 : 0];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                if (row < M) {
                    const float gi = sigmoid_acc(acc[0][0][r] + bi);
                    const float gf = sigmoid_acc(acc[0][1][r] + bf);
                    const float gg = tanh_fast(acc[0][2][r] + bg);
                    const float go = sigmoid_acc(acc[0][3][r] + bo);
                    const size_t so = (size_t)row * P.ld_state + unit;
                    const float cn = gf * cprev[LSTM ? r : 0] + gi * gg;
                    P.c_next[so] = cn;
                    P.h_next[so] = go * tanh_fast(cn);
                    if (P.gates) {
                        float* gp = P.gates + (size_t)row * P.ld_gates + unit;
                        gp[0] = gi;
                        gp[N] = gf;
                        gp[2 * N] = gg;
                        gp[3 * N] = go;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// TN: contraction over rows.  LDS tiles are [BK rows][BM or BN columns]; the MFMA
// fragments are ds_read_b32 with consecutive lanes on consecutive columns.
// ---------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int BK, int NBUF = 2>
__device__ __forceinline__ void gemm_tn_body(const float* __restrict__ A, int lda,
                                             const float* __restrict__ B, int ldb,
                                             float* __restrict__ out, int ldo,
                                             int64_t out_split_stride, int NI, int NJ,
                                             int64_t rows, int64_t rows_per_split,
                                             float* __restrict__ csum, int prio, unsigned bx, unsigned by,
                                             unsigned bz) {
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int A_CH = BK * (BM / 4) / 256;
    constexpr int B_CH = BK * (BN / 4) / 256;
    static_assert(A_CH >= 1 && B_CH >= 1, "tile too small");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][BK * (BM + BN)]

    const int i0 = bx * BM;
    const int j0 = by * BN;
    const int64_t r_begin = (int64_t)bz * rows_per_split;
    int64_t r_end = r_begin + rows_per_split;
    if (r_end > rows) r_end = rows;
    const int NI4 = (NI + 3) & ~3, NJ4 = (NJ + 3) & ~3;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int T = r_end > r_begin ? (int)((r_end - r_begin + BK - 1) / BK) : 0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Staging: thread t moves chunks c = t + 256 * i: tile row c / (BM / 4), 4 columns at
    // (c % (BM / 4)) * 4 (clamped into the padded width; columns past NI / NJ are never stored).
    // Addresses are a UNIFORM base (row r_begin + tile * BK, advanced on the scalar unit) plus a
    // fixed per-thread 32-bit byte offset; only the LAST tile of a split can reach past r_end:
    // there the row is clamped and the A side zeroed at LDS-store time, all other tiles take
    // the mask-free path.
    static_assert(A_CH == B_CH && A_CH <= 4, "square tiles, at most 4 chunks per thread");
    uint32_t aof0 = 0, aof1 = 0, aof2 = 0, aof3 = 0, bof0 = 0, bof1 = 0, bof2 = 0, bof3 = 0;
#define MARL_TN_SET(idx_)                                                              \
    if (A_CH > (idx_)) {                                                               \
        const int c_ = tid + 256 * (idx_);                                             \
        const int kr_ = c_ / (BM / 4);                                                 \
        const int ic_ = i0 + (c_ % (BM / 4)) * 4, jc_ = j0 + (c_ % (BN / 4)) * 4;      \
        aof##idx_ = ((uint32_t)kr_ * (uint32_t)lda + (uint32_t)(ic_ < NI4 ? ic_ : NI4 - 4)) * 4u; \
        bof##idx_ = ((uint32_t)kr_ * (uint32_t)ldb + (uint32_t)(jc_ < NJ4 ? jc_ : NJ4 - 4)) * 4u; \
    }
    MARL_TN_SET(0) MARL_TN_SET(1) MARL_TN_SET(2) MARL_TN_SET(3)
#undef MARL_TN_SET
    const char* abase = reinterpret_cast<const char*>(A + (size_t)r_begin * lda);
    const char* bbase = reinterpret_cast<const char*>(B + (size_t)r_begin * ldb);
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    float mk0 = 1.f, mk1 = 1.f, mk2 = 1.f, mk3 = 1.f;
    bool masked = false;  // uniform: the staged tile is the split's (partial) last tile
    // column sums of A (the bias gradient that goes with this weight gradient): taken for
    // free from the staged A tiles by the workgroups of the first column-tile
    const bool do_csum = csum != nullptr && by == 0;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#define MARL_TN_LOAD1(idx_, clamp_)                                                    \
    if (A_CH > (idx_)) {                                                               \
        uint32_t da_ = 0u, db_ = 0u;                                                   \
        if (clamp_) {                                                                  \
            const int64_t gr_ = base_ + (tid + 256 * (idx_)) / (BM / 4);               \
            const int64_t back_ = gr_ < r_end ? 0 : gr_ - (r_end - 1);                 \
            da_ = (uint32_t)(-(back_ * lda * 4));                                      \
            db_ = (uint32_t)(-(back_ * ldb * 4));                                      \
            mk##idx_ = gr_ < r_end ? 1.f : 0.f;                                        \
        }                                                                              \
        ra##idx_ = *reinterpret_cast<const float4*>(abase + (aof##idx_ + da_));        \
        rb##idx_ = *reinterpret_cast<const float4*>(bbase + (bof##idx_ + db_));        \
    }
#define MARL_TN_LOAD(tile_)                                                            \
    {                                                                                  \
        const int64_t base_ = r_begin + (int64_t)(tile_) * BK;                         \
        masked = base_ + BK > r_end;                                                   \
        if (!masked) {                                                                 \
            MARL_TN_LOAD1(0, false) MARL_TN_LOAD1(1, false) MARL_TN_LOAD1(2, false) MARL_TN_LOAD1(3, false) \
        } else {                                                                       \
            MARL_TN_LOAD1(0, true) MARL_TN_LOAD1(1, true) MARL_TN_LOAD1(2, true) MARL_TN_LOAD1(3, true) \
        }                                                                              \
        abase += (size_t)BK * lda * 4;                                                 \
        bbase += (size_t)BK * ldb * 4;                                                 \
    }
#define MARL_TN_STORE(idx_)                                                            \
    if (A_CH > (idx_)) {                                                               \
        const int c_ = tid + 256 * (idx_);                                             \
        float4 v_ = ra##idx_;                                                          \
        if (masked) {                                                                  \
            v_.x *= mk##idx_;                                                          \
            v_.y *= mk##idx_;                                                          \
            v_.z *= mk##idx_;                                                          \
            v_.w *= mk##idx_;                                                          \
        }                                                                              \
        if (do_csum) {                                                                 \
            cs.x += v_.x;                                                              \
            cs.y += v_.y;                                                              \
            cs.z += v_.z;                                                              \
            cs.w += v_.w;                                                              \
        }                                                                              \
        *reinterpret_cast<float4*>(As_ + (c_ / (BM / 4)) * BM + (c_ % (BM / 4)) * 4) = v_; \
        *reinterpret_cast<float4*>(Bs_ + (c_ / (BN / 4)) * BN + (c_ % (BN / 4)) * 4) = rb##idx_; \
    }

    const int fcol = lane & 31;
    const int fk = lane >> 5;
    if (T > 0) MARL_TN_LOAD(0)
    for (int tile = 0; tile < T; ++tile) {
        // NBUF == 1: ONE LDS stage (32 KB for 128-wide tiles) and a second barrier per tile.  The
        // 168 registers of the 128-wide plan allow three workgroups per CU, two stages of LDS only
        // two: with one stage all 768 workgroups of a launch are resident at once (measured
        // 2.07 -> 2.00 ms for the weight gradients of an iteration).
        const int buf = NBUF == 2 ? (tile & 1) : 0;
        if (NBUF == 1 && tile > 0) lds_barrier();  // every wave is done reading the previous tile
        {
            float* As_ = smem + buf * BK * (BM + BN);
            float* Bs_ = As_ + BK * BM;
            MARL_TN_STORE(0) MARL_TN_STORE(1) MARL_TN_STORE(2) MARL_TN_STORE(3)
        }
        lds_barrier();
        if (tile + 1 < T) MARL_TN_LOAD(tile + 1)
        const float* As = smem + buf * BK * (BM + BN) + wm * (BM / WM) + fcol;
        const float* Bs = smem + buf * BK * (BM + BN) + BK * BM + wn * (BN / WN) + fcol;
        if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(ks * 2 + fk) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[(ks * 2 + fk) * BN + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
    }
#undef MARL_TN_LOAD1
#undef MARL_TN_LOAD
#undef MARL_TN_STORE

    if (do_csum) {  // threads with equal (tid % (BM/4)) staged the same 4 columns
        constexpr int CPR = BM / 4;     // column chunks per tile row
        constexpr int SH = 256 / CPR;   // threads sharing a chunk
        __syncthreads();
        float4* sh4 = reinterpret_cast<float4*>(smem);
        sh4[(tid / CPR) * CPR + (tid % CPR)] = cs;
        __syncthreads();
        if (tid < CPR) {
            float4 t = sh4[tid];
#pragma unroll
            for (int q = 1; q < SH; ++q) {
                const float4 u = sh4[q * CPR + tid];
                t.x += u.x;
                t.y += u.y;
                t.z += u.z;
                t.w += u.w;
            }
            float* co = csum + (size_t)bz * NI;
            const int ic = i0 + tid * 4;
            if (ic < NI) co[ic] = t.x;
            if (ic + 1 < NI) co[ic + 1] = t.y;
            if (ic + 2 < NI) co[ic + 2] = t.z;
            if (ic + 3 < NI) co[ic + 3] = t.w;
        }
    }

    float* o = out + (size_t)bz * out_split_stride;
    const int row_h = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = j0 + wn * (BN / WN) + j * 32 + fcol;
            if (col >= NJ) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                if (row < NI) o[(size_t)row * ldo + col] = acc[i][j][r];
            }
        }
}

template <int BM, int BN, int WM, int WN, int BK, int NBUF = 2>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, int lda,
                                                      const float* __restrict__ B, int ldb,
                                                      float* __restrict__ out, int ldo,
                                                      int64_t out_split_stride, int NI, int NJ,
                                                      int64_t rows, int64_t rows_per_split,
                                                      float* __restrict__ csum, int gx, int gy,
                                                      int gz, int prio) {
    // gx > 0: 1-D launch, XCD-aware order (the tiles of one row slab run on one XCD, so its A
    // and B row panels come from that XCD's L2 for all but the first tile)
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (gx > 0) xcd_tile(gx, gy, gz, bx, by, bz);
    gemm_tn_body<BM, BN, WM, WN, BK, NBUF>(A, lda, B, ldb, out, ldo, out_split_stride, NI, NJ, rows, rows_per_split,
                                           csum, prio, bx, by, bz);
}

// Several SMALL row contractions in one launch (the ~8 weight gradients of an iteration whose output
// is at most a few 64 x 64 tiles: each alone fills a fraction of the chip for 10-50 us).  Workgroup
// b belongs to the problem whose [first, first + tiles * splits) range holds it.
__global__ __launch_bounds__(256) void gemm_tn_batch_kernel(const TnBatch T) {
    int pi = 0;
    while (pi + 1 < T.count && (int)blockIdx.x >= T.p[pi + 1].first_block) ++pi;
    const TnBatchProb& P = T.p[pi];
    const unsigned local = blockIdx.x - (unsigned)P.first_block;
    const unsigned per = (unsigned)(P.gx * P.gy);
    const unsigned bz = local / per, r = local - bz * per;
    gemm_tn_body<64, 64, 2, 2, 32, 2>(P.a, P.lda, P.b, P.ldb, P.out, P.ldo, P.stride, P.ni, P.nj, P.rows, P.rows_per_split,
                                      P.csum, 0, r / (unsigned)P.gy, r % (unsigned)P.gy, bz);
}

// Fixed-order reduction of partial slabs: out[e] = sum_z part[z * stride + e].  A 1024-thread
// workgroup owns 64 consecutive elements: wave g sums the slabs z = g, g + 16, ... (four loads in
// flight), the 16 wave sums meet in LDS and are added in wave order - the same order whatever
// the grid, so the result is bit-reproducible.  Workgroups past `main_blocks` do the same for
// the bias partials bpart[z * ni + i].
__global__ __launch_bounds__(1024) void slab_reduce_kernel(
    const float* __restrict__ part, int64_t stride, int splits, float* __restrict__ c, int ldc,
    int NI, int NJ, const float* __restrict__ bpart, float* __restrict__ bias, int main_blocks) {
    __shared__ float sh[16][64];
    const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
    const bool is_bias = (int)blockIdx.x >= main_blocks;
    const int64_t n = is_bias ? NI : (int64_t)NI * NJ;
    const int64_t e = (int64_t)(is_bias ? blockIdx.x - main_blocks : blockIdx.x) * 64 + el;
    const float* src = is_bias ? bpart : part;
    const int64_t st = is_bias ? NI : stride;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int z = g;
        for (; z + 48 < splits; z += 64) {
            s0 += src[(size_t)z * st + e];
            s1 += src[(size_t)(z + 16) * st + e];
            s2 += src[(size_t)(z + 32) * st + e];
            s3 += src[(size_t)(z + 48) * st + e];
        }
        for (; z < splits; z += 16) s0 += src[(size_t)z * st + e];
    }
    sh[g][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && e < n) {
        float t = sh[0][el];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += sh[q][el];
        if (is_bias) {
            bias[e] = t;
        } else {
            const int i = (int)(e / NJ), j = (int)(e - (int64_t)i * NJ);
            c[(size_t)i * ldc + j] = t;
        }
    }
}

// Four consecutive elements per lane (16-byte loads, 256 elements per workgroup) when rows of the
// slab and of the output keep float4 alignment; same per-element summation order as above.
__global__ __launch_bounds__(1024) void slab_reduce4_kernel(
    const float* __restrict__ part, int64_t stride, int splits, float* __restrict__ c, int ldc,
    int NI, int NJ, const float* __restrict__ bpart, float* __restrict__ bias, int main_blocks) {
    __shared__ float4 sh[16][64];
    const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
    if ((int)blockIdx.x >= main_blocks) {  // bias partials: scalar, as in slab_reduce_kernel
        float* shf = reinterpret_cast<float*>(&sh[0][0]);
        const int64_t e = (int64_t)(blockIdx.x - main_blocks) * 64 + el;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (e < NI) {
            int z = g;
            for (; z + 48 < splits; z += 64) {
                s0 += bpart[(size_t)z * NI + e];
                s1 += bpart[(size_t)(z + 16) * NI + e];
                s2 += bpart[(size_t)(z + 32) * NI + e];
                s3 += bpart[(size_t)(z + 48) * NI + e];
            }
            for (; z < splits; z += 16) s0 += bpart[(size_t)z * NI + e];
        }
        shf[g * 64 + el] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (g == 0 && e < NI) {
            float t = shf[el];
#pragma unroll
            for (int q = 1; q < 16; ++q) t += shf[q * 64 + el];
            bias[e] = t;
        }
        return;
    }
    const int64_t n = (int64_t)NI * NJ;
    const int64_t e = ((int64_t)blockIdx.x * 64 + el) * 4;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
#define MARL_ADD4(d_, v_) \
    {                     \
        const float4 t_ = (v_); \
        d_.x += t_.x;     \
        d_.y += t_.y;     \
        d_.z += t_.z;     \
        d_.w += t_.w;     \
    }
    if (e < n) {
        int z = g;
        for (; z + 48 < splits; z += 64) {
            const float4 v0 = *reinterpret_cast<const float4*>(part + (size_t)z * stride + e);
            const float4 v1 = *reinterpret_cast<const float4*>(part + (size_t)(z + 16) * stride + e);
            const float4 v2 = *reinterpret_cast<const float4*>(part + (size_t)(z + 32) * stride + e);
            const float4 v3 = *reinterpret_cast<const float4*>(part + (size_t)(z + 48) * stride + e);
            MARL_ADD4(s0, v0) MARL_ADD4(s1, v1) MARL_ADD4(s2, v2) MARL_ADD4(s3, v3)
        }
        for (; z < splits; z += 16) MARL_ADD4(s0, *reinterpret_cast<const float4*>(part + (size_t)z * stride + e))
    }
#undef MARL_ADD4
    sh[g][el] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y),
                            (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (g == 0 && e < n) {
        float4 t = sh[0][el];
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const float4 u = sh[q][el];
            t.x += u.x;
            t.y += u.y;
            t.z += u.z;
            t.w += u.w;
        }
        const int i = (int)(e / NJ), j = (int)(e - (int64_t)i * NJ);  // NJ % 4 == 0: one row
        *reinterpret_cast<float4*>(c + (size_t)i * ldc + j) = t;
    }
}

int launch_slab_reduce(const float* part, int64_t stride, int splits, float* c, int ldc, int ni,
                       int nj, const float* bpart, float* bias, hipStream_t st) {
    if ((nj & 3) == 0 && (stride & 3) == 0 && (ldc & 3) == 0 && ((reinterpret_cast<uintptr_t>(part) |
                                                                  reinterpret_cast<uintptr_t>(c)) & 15) == 0) {
        const int main_blocks = (int)cdiv((int64_t)ni * nj, 256);
        const int extra = (bpart && bias) ? (int)cdiv(ni, 64) : 0;
        hipLaunchKernelGGL(slab_reduce4_kernel, dim3((unsigned)(main_blocks + extra)), dim3(1024), 0, st, part,
                           stride, splits, c, ldc, ni, nj, bpart, bias, main_blocks);
        MARL_LAUNCH_CHECK();
        return MARL_OK;
    }
    const int main_blocks = (int)cdiv((int64_t)ni * nj, 64);
    const int extra = (bpart && bias) ? (int)cdiv(ni, 64) : 0;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)(main_blocks + extra)), dim3(1024), 0, st,
                       part, stride, splits, c, ldc, ni, nj, bpart, bias, main_blocks);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// Batched form of the same reduction: every 1024-thread workgroup owns 64 elements of ONE queued
// descriptor (see RedQueue in common.h); 16 waves take the parts z = g, g + 16, ... with eight
// loads in flight, the wave sums are added in wave order.
__global__ __launch_bounds__(1024) void red_batch_kernel(const RedBatch B) {
    __shared__ float sh[16][64];
    __shared__ int sdesc;
    const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        int i = 0;
        while (i + 1 < B.count && (int)blockIdx.x >= B.first_block[i + 1]) ++i;
        sdesc = i;
    }
    __syncthreads();
    const RedDesc D = B.d[sdesc];
    const int nb = (D.n + 63) >> 6, local = (int)blockIdx.x - B.first_block[sdesc];
    const int f = local / nb, e = (local - f * nb) * 64 + el;
    const int z0 = f * D.per, zn = D.count - z0 < D.per ? D.count - z0 : D.per;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < D.n) {
        const float* p = D.part + (size_t)z0 * D.stride + e;
        int z = g;
        for (; z + 112 < zn; z += 128) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(z + 16 * u) * D.stride];
            s0 += v[0] + v[4];
            s1 += v[1] + v[5];
            s2 += v[2] + v[6];
            s3 += v[3] + v[7];
        }
        for (; z < zn; z += 16) s0 += p[(size_t)z * D.stride];
    }
    sh[g][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && e < D.n) {
        float t = sh[0][el];
#pragma unroll
        for (int q = 1; q < 16; ++q) t += sh[q][el];
        float* o;
        if (D.folds > 1) {
            o = D.out0 + (size_t)f * D.n + e;
        } else if (e < D.split) {
            const int i = e / D.nj, j = e - i * D.nj;
            o = D.out0 + (size_t)i * D.ldc + j;
        } else {
            o = D.out1 + (e - D.split);
        }
        if (D.accumulate) t += *o;
        *o = t;
    }
}

int launch_red_batch(const RedBatch& b, hipStream_t st) {
    if (b.count <= 0) return MARL_OK;
    hipLaunchKernelGGL(red_batch_kernel, dim3((unsigned)b.first_block[b.count]), dim3(1024), 0, st, b);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

void RedQueue::reset(float* s, size_t floats, hipStream_t stream) {
    scratch = s;
    cap = floats;
    off = 0;
    st = stream;
    rc = MARL_OK;
    stage1.count = stage2.count = 0;
    stage1.first_block[0] = stage2.first_block[0] = 0;
}

float* RedQueue::take(size_t floats, bool may_flush) {
    floats = (floats + 63) & ~(size_t)63;
    if (off + floats > cap) {
        if (!may_flush) return nullptr;
        // everything handed out so far has been pushed: after the queued reductions (stream
        // order) the space is free again
        (void)flush();
        if (floats > cap) {
            rc = MARL_ESIZE;
            set_error("reduction scratch too small (%zu > %zu floats)", floats, cap);
            return scratch;
        }
    }
    float* p = scratch + off;
    off += floats;
    return p;
}

static void red_add(RedBatch& b, const RedDesc& d) {
    b.d[b.count] = d;
    b.first_block[b.count + 1] = b.first_block[b.count] + d.folds * (int)cdiv(d.n, 64);
    ++b.count;
}

void RedQueue::push(const float* part, int64_t stride, int count, int n, float* out0, int split, int nj,
                    int ldc, float* out1, int accumulate) {
    if (rc != MARL_OK || n <= 0 || count <= 0) return;
    // long reductions: fold `count` parts into <= 64 temporaries first (stage 1), so that no
    // thread walks more than a few dozen dependent loads
    int folds = count > 1024 ? 64 : (count > 256 ? 16 : 1);
    // (the scratch stays handed out here: `part` of this very call may live in it)
    if (stage2.count + 1 > kMaxRed || stage1.count + (folds > 1) > kMaxRed) (void)launch_pending();
    RedDesc d{part, stride, count, n, out0, split, nj, ldc, out1, accumulate, 1, count};
    if (folds > 1) {
        const int per = (int)cdiv(count, folds);
        folds = (int)cdiv(count, per);
        float* tmp = take((size_t)folds * n, false);
        if (tmp) {  // (else: no room for the temporaries - one slow stage, still correct)
            red_add(stage1, RedDesc{part, stride, count, n, tmp, n, n, n, nullptr, 0, folds, per});
            d.part = tmp;
            d.stride = n;
            d.count = d.per = folds;
        }
    }
    red_add(stage2, d);
}

int RedQueue::launch_pending() {
    if (rc == MARL_OK && stage1.count) rc = launch_red_batch(stage1, st);
    if (rc == MARL_OK && stage2.count) rc = launch_red_batch(stage2, st);
    stage1.count = stage2.count = 0;
    return rc;
}

int RedQueue::flush() {
    (void)launch_pending();
    off = 0;
    return rc;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
GemmProb gemm_prob(const float* a, int lda, const float* b, int ldb, int k, float* c, int ldc,
                   int m, int n, const float* bias, int accumulate) {
    GemmProb p{};
    p.seg[0] = GemmSeg{a, b, lda, ldb, k};
    p.nseg = 1;
    p.m = m;
    p.n = n;
    p.c = c;
    p.ldc = ldc;
    p.bias = bias;
    p.accumulate = accumulate;
    return p;
}

void gemm_add_seg(GemmProb& p, const float* a, int lda, const float* b, int ldb, int k) {
    p.seg[1] = GemmSeg{a, b, lda, ldb, k};
    p.nseg = 2;
}


template <int BM, int BN, int WM, int WN, bool LSTM, int GROUPS>
static int launch_nt_variant(dim3 grid, const GemmBatch& batch_in, hipStream_t st) {
    GemmBatch batch = batch_in;
    batch.gx = (int)grid.x;
    batch.gy = (int)grid.y;
    // XCD-ordered tiles (column block fastest inside an XCD's chunk) when the launch is ONE
    // product: its row panels are then read through one L2 once.  Batched launches of unequal
    // problems would load the XCDs unevenly (measured slower), they keep the plain order.
    batch.xcd_map = xcd_map_enabled() || (batch.count == 1 && !LSTM && tune_get("nt_xcd", 1));
    if (batch.xcd_map) grid = dim3(grid.x * grid.y * grid.z);
    constexpr size_t lds2 = (size_t)2 * (GROUPS * BM + BN) * (32 + 4) * sizeof(float);
    batch.single_buf = GROUPS == 1 && !LSTM && BM == 128 && true;
    // (LSTM instantiation: s_setprio(1) around the matrix phase - the scheduler prefers the wave
    // that is feeding the matrix pipe over a co-resident one that is staging: -3 % on the launch.
    // Compile-time only: even a never-taken run-time branch around the cluster cost the other
    // plans 3-4 %, it splits the block in which the compiler interleaves LDS reads and MFMAs.)
    const size_t lds = batch.single_buf ? lds2 / 2 : lds2;
    auto kern = gemm_nt_kernel<BM, BN, WM, WN, LSTM, GROUPS>;
    if (lds2 > 64 * 1024) {
        static bool raised = false;  // opt in to > 64 KiB of dynamic LDS once per kernel
        if (!raised) {
            MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            raised = true;
        }
    }
    hipLaunchKernelGGL(kern, grid, dim3(256 * GROUPS), lds, st, batch);
    return MARL_OK;
}

int xcd_map_enabled() { return 0; }  // (the fp32-operand NT kernels: measured neutral-to-negative; the image kernels map by knob nt_xcd / tn_xcd)

// 1 = plain 4-wave loop (default), 2 = ping-pong wave groups (MARL_GEMM_GROUPS=2).  Measured on
// MI355X: the two-group schedule gains 2-7 % on single large products (M = 65536) but loses
// on this workload's small / batched products (bigger tiles -> worse tail), so it is opt-in.
static int gemm_groups() { return 1; }

static int check_prob(const GemmProb& p) {
    for (int s = 0; s < p.nseg; ++s) {
        const GemmSeg& g = p.seg[s];
        if (!g.a || !g.b || g.k <= 0 || (g.lda & 3) || (g.ldb & 3) ||
            (reinterpret_cast<uintptr_t>(g.a) & 15) || (reinterpret_cast<uintptr_t>(g.b) & 15) ||
            g.lda < p4(g.k) || g.ldb < p4(g.k)) {
            set_error("gemm: bad operand (seg %d: a=%p lda=%d b=%p ldb=%d k=%d)", s, (const void*)g.a,
                      g.lda, (const void*)g.b, g.ldb, g.k);
            return MARL_EINVAL;
        }
        // the staging offsets are 32-bit byte offsets from the segment base
        const int64_t brows = p.h_next ? 4 * (int64_t)p.n : (int64_t)p.n;
        if ((int64_t)p.m * g.lda >= (1ll << 30) || brows * g.ldb >= (1ll << 30)) {
            set_error("gemm: operand of seg %d spans more than 4 GiB", s);
            return MARL_ELIMIT;
        }
    }
    if (p.m <= 0 || p.n <= 0) {
        set_error("gemm: empty problem m=%d n=%d", p.m, p.n);
        return MARL_EINVAL;
    }
    return MARL_OK;
}

int launch_gemm_nt(const GemmBatch& batch, hipStream_t st) {
    if (batch.count < 1 || batch.count > kMaxGemmBatch) return MARL_EINVAL;
    int max_m = 0, max_n = 0;
    int64_t blocks128 = 0;
    for (int i = 0; i < batch.count; ++i) {
        MARL_TRY(check_prob(batch.p[i]));
        if (!batch.p[i].c) return MARL_EINVAL;
        max_m = batch.p[i].m > max_m ? batch.p[i].m : max_m;
        max_n = batch.p[i].n > max_n ? batch.p[i].n : max_n;
        blocks128 += cdiv(batch.p[i].m, 128) * cdiv(batch.p[i].n, 128);
    }
    static const bool log_shapes = getenv("MARL_GEMM_LOG") != nullptr;  // perf debugging aid
    if (log_shapes)
        for (int i = 0; i < batch.count; ++i)
            fprintf(stderr, "[gemm_nt] %d/%d m=%d n=%d k=%d+%d acc=%d bias=%d lda=%d ldc=%d\n", i,
                    batch.count, batch.p[i].m, batch.p[i].n, batch.p[i].seg[0].k,
                    batch.p[i].nseg > 1 ? batch.p[i].seg[1].k : 0, batch.p[i].accumulate,
                    batch.p[i].bias != nullptr, batch.p[i].seg[0].lda, batch.p[i].ldc);
    prof_before(1, st);
    if (split_mode()) {
        MARL_TRY(launch_gemm_nt_split(batch, max_m, max_n, blocks128, st));
    } else if (blocks128 >= 256 && max_n >= 96) {
        const int g = gemm_groups();
        dim3 grid((unsigned)cdiv(max_m, 128 * g), (unsigned)cdiv(max_n, 128), (unsigned)batch.count);
        if (g == 2)
            MARL_TRY((launch_nt_variant<128, 128, 2, 2, false, 2>(grid, batch, st)));
        else
            MARL_TRY((launch_nt_variant<128, 128, 2, 2, false, 1>(grid, batch, st)));
    } else {
        const int g = gemm_groups();
        dim3 grid((unsigned)cdiv(max_m, 64 * g), (unsigned)cdiv(max_n, 64), (unsigned)batch.count);
        if (g == 2)
            MARL_TRY((launch_nt_variant<64, 64, 2, 2, false, 2>(grid, batch, st)));
        else
            MARL_TRY((launch_nt_variant<64, 64, 2, 2, false, 1>(grid, batch, st)));
    }
    prof_after(1, st);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// optional per-launch timing of one kernel class with HIP events recorded on the launch
// stream (bench.py's roofline figure).  class 0 = fused LSTM GEMM, 1 = plain NT GEMM,
// 2 = TN GEMM, 3 = fused CNN forward, 4 = row-panel MLP kernels, 5 = CNN backward (layer
// backward + weight gradients).  Off by default: no events are created or recorded.
// ---------------------------------------------------------------------------
static int g_prof_class = -1;
static int g_prof_cap = 0, g_prof_n = 0;
static hipEvent_t* g_prof_ev = nullptr;  // [2 * cap]

void prof_before(int cls, hipStream_t st) {
    if (cls == g_prof_class && g_prof_n < g_prof_cap) (void)hipEventRecord(g_prof_ev[2 * g_prof_n], st);
}
void prof_after(int cls, hipStream_t st) {
    if (cls == g_prof_class && g_prof_n < g_prof_cap) {
        (void)hipEventRecord(g_prof_ev[2 * g_prof_n + 1], st);
        ++g_prof_n;
    }
}

int profile_begin(int cls, int max_launches) {
    if (g_prof_ev) return MARL_EINVAL;
    g_prof_ev = new hipEvent_t[2 * (size_t)max_launches];
    for (int i = 0; i < 2 * max_launches; ++i) MARL_HIP_CHECK(hipEventCreate(&g_prof_ev[i]));
    g_prof_cap = max_launches;
    g_prof_n = 0;
    g_prof_class = cls;
    return MARL_OK;
}

int profile_end(double* total_ms, int* launches) {
    if (!g_prof_ev) return MARL_EINVAL;
    double tot = 0.0;
    for (int i = 0; i < g_prof_n; ++i) {
        MARL_HIP_CHECK(hipEventSynchronize(g_prof_ev[2 * i + 1]));
        float ms = 0.f;
        MARL_HIP_CHECK(hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]));
        tot += ms;
    }
    for (int i = 0; i < 2 * g_prof_cap; ++i) (void)hipEventDestroy(g_prof_ev[i]);
    delete[] g_prof_ev;
    g_prof_ev = nullptr;
    if (total_ms) *total_ms = tot;
    if (launches) *launches = g_prof_n;
    g_prof_class = -1;
    g_prof_cap = g_prof_n = 0;
    return MARL_OK;
}

int launch_gemm_lstm(const GemmBatch& batch, hipStream_t st) {
    if (batch.count < 1 || batch.count > kMaxGemmBatch) return MARL_EINVAL;
    int max_m = 0, max_n = 0;
    for (int i = 0; i < batch.count; ++i) {
        const GemmProb& p = batch.p[i];
        MARL_TRY(check_prob(p));
        if (!p.bias || !p.c_prev || !p.h_next || !p.c_next) return MARL_EINVAL;
        max_m = p.m > max_m ? p.m : max_m;
        max_n = p.n > max_n ? p.n : max_n;
    }
    const int g = gemm_groups();
    dim3 grid((unsigned)cdiv(max_m, 128 * g), (unsigned)cdiv(max_n, 32), (unsigned)batch.count);
    prof_before(0, st);
    if (split_mode())
        MARL_TRY(launch_gemm_lstm_split(batch, max_m, max_n, st));
    else if (g == 2)
        MARL_TRY((launch_nt_variant<128, 128, 4, 1, true, 2>(grid, batch, st)));
    else
        MARL_TRY((launch_nt_variant<128, 128, 4, 1, true, 1>(grid, batch, st)));
    prof_after(0, st);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

struct TnPlan {
    int bm;  // 128 or 64 (square tiles)
    int splits;
    int64_t rows_per_split;
};

static TnPlan tn_plan(int ni, int nj, int64_t rows) {
    TnPlan p;
    // 128-wide tiles only when there are enough of them: with < 4 tiles the 768-workgroup
    // target turns into hundreds of row slabs and the slab reduction costs more than the product
    p.bm = (ni >= 96 && nj >= 96 && cdiv(ni, 128) * cdiv(nj, 128) >= 4) ? 128 : 64;
    const int64_t tiles = cdiv(ni, p.bm) * cdiv(nj, p.bm);
    // (the bf16x6 kernel keeps two workgroups per CU resident, the fp32 one three)
    const int target = (p.bm == 128 && split_mode()) ? 512 : 768;
    int64_t s = cdiv(target, tiles);
    const int64_t max_s = cdiv(rows, 4 * BK);  // at least 4 K tiles per split
    if (s > max_s) s = max_s;
    if (s > 512) s = 512;
    if (s < 1) s = 1;
    int64_t rps = cdiv(cdiv(rows, s), BK) * BK;
    p.splits = (int)cdiv(rows, rps);
    p.rows_per_split = rps;
    return p;
}

size_t gemm_tn_scratch_bytes(int ni, int nj, int64_t rows) {
    TnPlan p = tn_plan(ni, nj, rows);
    // split partials of the product + of the optional column sums
    return p.splits > 1 ? ((size_t)p.splits * ni * nj + (size_t)p.splits * ni) * sizeof(float) : 0;
}

bool gemm_tn_is_small(int ni, int nj, int64_t rows) { return tn_plan(ni, nj, rows).bm == 64; }

int launch_tn_queue(TnQueue& tq, RedQueue* q, hipStream_t st) {
    if (tq.n == 0) return MARL_OK;
    if (!q) return MARL_EINVAL;
    // all slabs must come out of the scratch in one piece of time: no flush between the takes
    size_t need = 0;
    for (int i = 0; i < tq.n; ++i)
        need += (gemm_tn_scratch_bytes(tq.it[i].ni, tq.it[i].nj, tq.it[i].rows) / sizeof(float) + 63) & ~(size_t)63;
    if (q->off + need > q->cap) MARL_TRY(q->flush());
    TnBatch tb{};
    float* scr[kMaxTnBatch];
    TnPlan plan[kMaxTnBatch];
    int blocks = 0;
    for (int i = 0; i < tq.n; ++i) {
        const TnQueue::Item& it = tq.it[i];
        plan[i] = tn_plan(it.ni, it.nj, it.rows);
        scr[i] = nullptr;
        TnBatchProb& P = tb.p[i];
        P.a = it.a;
        P.b = it.b;
        P.lda = it.lda;
        P.ldb = it.ldb;
        P.ni = it.ni;
        P.nj = it.nj;
        P.rows = it.rows;
        P.rows_per_split = plan[i].rows_per_split;
        P.gx = (int)cdiv(it.ni, 64);
        P.gy = (int)cdiv(it.nj, 64);
        P.first_block = blocks;
        blocks += P.gx * P.gy * plan[i].splits;
        if (plan[i].splits > 1) {
            scr[i] = q->take(gemm_tn_scratch_bytes(it.ni, it.nj, it.rows) / sizeof(float), false);
            if (!scr[i]) {
                set_error("tn queue: reduction scratch too small");
                return MARL_ESIZE;
            }
            P.out = scr[i];
            P.ldo = it.nj;
            P.stride = (int64_t)it.ni * it.nj;
            P.csum = it.colsum ? scr[i] + (size_t)plan[i].splits * it.ni * it.nj : nullptr;
        } else {
            P.out = it.c;
            P.ldo = it.ldc;
            P.stride = 0;
            P.csum = it.colsum;
        }
    }
    tb.count = tq.n;
    prof_before(2, st);
    hipLaunchKernelGGL(gemm_tn_batch_kernel, dim3((unsigned)blocks), dim3(256), (size_t)2 * 32 * 2 * 64 * sizeof(float), st, tb);
    prof_after(2, st);
    MARL_LAUNCH_CHECK();
    for (int i = 0; i < tq.n; ++i) {
        const TnQueue::Item& it = tq.it[i];
        if (plan[i].splits <= 1) continue;
        q->push(scr[i], (int64_t)it.ni * it.nj, plan[i].splits, it.ni * it.nj, it.c, it.ni * it.nj, it.nj, it.ldc, nullptr, 0);
        if (it.colsum) q->push(tb.p[i].csum, it.ni, plan[i].splits, it.ni, it.colsum, it.ni, it.ni, it.ni, nullptr, 0);
    }
    tq.n = 0;
    return q->rc;
}

int launch_gemm_tn(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int ni,
                   int nj, int64_t rows, float* scratch, size_t scratch_bytes, hipStream_t st,
                   float* colsum_out, RedQueue* q, TnQueue* tq) {
    if (!a || !b || !c || ni <= 0 || nj <= 0 || rows <= 0 || (lda & 3) || (ldb & 3) ||
        lda < p4(ni) || ldb < p4(nj) || (reinterpret_cast<uintptr_t>(a) & 15) ||
        (reinterpret_cast<uintptr_t>(b) & 15)) {
        set_error("gemm_tn: bad operand ni=%d nj=%d rows=%lld lda=%d ldb=%d", ni, nj,
                  (long long)rows, lda, ldb);
        return MARL_EINVAL;
    }
    TnPlan p = tn_plan(ni, nj, rows);
    if (tq && q && p.bm == 64) {  // small product: runs with the others, later
        if (tq->n == kMaxTnBatch) MARL_TRY(launch_tn_queue(*tq, q, st));
        tq->it[tq->n++] = TnQueue::Item{a, b, c, colsum_out, lda, ldb, ldc, ni, nj, rows};
        return MARL_OK;
    }
    float* out = c;
    int ldo = ldc;
    int64_t stride = 0;
    if (p.splits > 1 && q) {  // partial slabs live in the queue's scratch until its flush
        scratch = q->take(gemm_tn_scratch_bytes(ni, nj, rows) / sizeof(float));
        scratch_bytes = gemm_tn_scratch_bytes(ni, nj, rows);
        if (q->rc != MARL_OK) return q->rc;
    }
    if (p.splits > 1) {
        if (!scratch || scratch_bytes < gemm_tn_scratch_bytes(ni, nj, rows)) {
            set_error("gemm_tn: scratch too small");
            return MARL_ESIZE;
        }
        out = scratch;
        ldo = nj;
        stride = (int64_t)ni * nj;
    }
    float* csum = colsum_out;  // split partials live behind the product partials
    if (colsum_out && p.splits > 1) csum = scratch + (size_t)p.splits * ni * nj;
    static const bool log_shapes = getenv("MARL_GEMM_LOG") != nullptr;
    if (log_shapes)
        fprintf(stderr, "[gemm_tn] ni=%d nj=%d rows=%lld bm=%d splits=%d colsum=%d\n", ni, nj,
                (long long)rows, p.bm, p.splits, colsum_out != nullptr);
    dim3 grid((unsigned)cdiv(ni, p.bm), (unsigned)cdiv(nj, p.bm), (unsigned)p.splits);
    // XCD-ordered tiles by default: all tiles of a row slab run on one XCD and read its A / B row
    // panels through that XCD's L2 once (measured: FETCH_SIZE 706 -> 292 MB per launch = 1.07x the
    // algorithmic bytes, same duration - the kernel is matrix-pipe bound)
    int gx = 0, gy = 0, gz = 0;
    if (xcd_map_enabled() || tune_get("tn_xcd", 1)) {
        gx = (int)grid.x;
        gy = (int)grid.y;
        gz = (int)grid.z;
        grid = dim3(grid.x * grid.y * grid.z);
    }
    prof_before(2, st);
    // s_setprio(1) around the matrix phase (see launch_nt_variant): -3 % on the weight gradients.
    // (Kept behind this run-time flag on purpose: the unconditional form compiles to a 20 % SLOWER
    // loop - measured 307 vs 254 us - the branch changes how hipcc schedules the cluster.)
    const int tn_prio = 1;
    constexpr int tbk = 32;
#define MARL_TN_LAUNCH(BM_, BK_)                                                               \
    hipLaunchKernelGGL((gemm_tn_kernel<BM_, BM_, 2, 2, BK_>), grid, dim3(256),                 \
                       (size_t)2 * BK_ * 2 * BM_ * sizeof(float), st, a, lda, b, ldb, out, ldo, \
                       stride, ni, nj, rows, p.rows_per_split, csum, gx, gy, gz, tn_prio)
    if (p.bm == 128 && split_mode())
        MARL_TRY(launch_gemm_tn_split(a, lda, b, ldb, out, ldo, stride, ni, nj, rows, p.rows_per_split, csum,
                                      grid, gx, gy, gz, st));
    else if (p.bm == 128 && tbk == 32)
        hipLaunchKernelGGL((gemm_tn_kernel<128, 128, 2, 2, 32, 1>), grid, dim3(256),
                           (size_t)32 * 2 * 128 * sizeof(float), st, a, lda, b, ldb, out, ldo, stride, ni, nj,
                           rows, p.rows_per_split, csum, gx, gy, gz, tn_prio);
    else if (p.bm == 128 && tbk == 32)
        MARL_TN_LAUNCH(128, 32);
    else if (p.bm == 128)
        MARL_TN_LAUNCH(128, 16);
    else if (tbk == 32)
        MARL_TN_LAUNCH(64, 32);
    else
        MARL_TN_LAUNCH(64, 16);
#undef MARL_TN_LAUNCH
    prof_after(2, st);
    MARL_LAUNCH_CHECK();
    if (p.splits > 1 && q) {
        q->push(scratch, stride, p.splits, ni * nj, c, ni * nj, nj, ldc, nullptr, 0);
        if (colsum_out) q->push(csum, ni, p.splits, ni, colsum_out, ni, ni, ni, nullptr, 0);
        return q->rc;
    }
    if (p.splits > 1)
        MARL_TRY(launch_slab_reduce(scratch, stride, p.splits, c, ldc, ni, nj,
                                    colsum_out ? csum : nullptr, colsum_out, st));
    return MARL_OK;
}

}  // namespace marl
