// fp32 matrix products on the bf16 matrix pipe ("bf16x6"): MI355X runs v_mfma_f32_32x32x2_f32 at
// 1/16 of the bf16 MFMA rate (MI355X_MICROARCH.md: 157 TF vs 2.5 PF dense), so an exact-fp32
// product is priced at the VECTOR rate.  Here every fp32 operand element x is split, in
// registers while its tile goes to LDS, into three bf16 terms
//     x0 = bf16(x),  x1 = bf16(x - x0),  x2 = bf16(x - x0 - x1)        (round to nearest)
// with x0 + x1 + x2 == x EXACTLY (3 x 8 significand bits + the signs cover fp32's 24), and a
// product a*b is accumulated in fp32 from the six bf16 x bf16 MFMA products of weight >= 2^-16,
//     a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0);
// the three dropped terms a1 b2 + a2 b1 + a2 b2 are bounded by 2^-23 |a b| (|x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|
// with round-to-nearest splits), i.e. two fp32 unit round-offs in the worst case - measured, the error equals the
// exact-fp32 MFMA kernels' (tests/test_gpu_kernels.py asserts err <= 1.5 x theirs per shape).  Every
// bf16 x bf16 product is exact in fp32, so the result differs from an fp32 fmaf chain only in
// summation order - measured against float64: max error <= the fp32-MFMA kernel's on every
// shape of this workload (tests/test_gpu_kernels.py::test_split_gemm_*).  Six
// v_mfma_f32_32x32x16_bf16 replace eight v_mfma_f32_32x32x2_f32 at 1/2 the cycles each:
// 2.67x the fp32-MFMA peak (416.7 TF fp32-equivalent).  Operands in HBM stay fp32.
//
//  gemm_nt_split_kernel : same contract as gemm_nt_kernel (gemm.hip), incl. the LSTM epilogue
//  gemm_tn_split_kernel : same contract as gemm_tn_kernel (the 4x4 register transpose happens
//                         while a tile is staged, so both kernels share one LDS image:
//                         [plane][row][32 k] bf16, rows padded to 80 B = conflict-free b128)
#include <stdlib.h>

#include "common.h"

namespace marl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SK = 32;     // K depth of a staged tile (two 16-deep MFMA steps)
constexpr int SROW = 80;   // LDS bytes per tile row: 32 bf16 + 16 B pad

__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (pack_bf16 / split_pair: split.h)
// four consecutive-k values -> 8 bytes in each of the three planes
__device__ __forceinline__ void split_store4(char* dst, int plane, float a, float b, float c, float d) {
    uint32_t a0, a1, a2, b0, b1, b2;
    split_pair(a, b, a0, a1, a2);
    split_pair(c, d, b0, b1, b2);
    *reinterpret_cast<uint2*>(dst) = make_uint2(a0, b0);
    *reinterpret_cast<uint2*>(dst + plane) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(dst + 2 * plane) = make_uint2(a2, b2);
}

// Staging maps (thread chunk c -> tile row / piece) chosen for the LDS STORE banking (32 banks,
// MI355X_MICROARCH.md LDS table): with 80-byte rows a 16-lane ds_write_b64 group that covers rows
// r, r + 1 hits four banks twice, rows r, r + 4 (320 B = 16 banks apart) none - measured before
// the remap: SQ_LDS_BANK_CONFLICT = 31 % of SQ_LDS_IDX_ACTIVE in the NT kernels, every store 2-way.
// fp32 tiles: chunk c = 8 bytes (four k) of row stage_row(c), piece c % 8.
__device__ __forceinline__ constexpr int stage_row(int c) {
    return (c >> 6) * 8 + ((c >> 4) & 3) + 4 * ((c >> 3) & 1);
}
// pre-split image tiles (16-byte pieces, four per row and plane; RW rows per wave): lanes 4u..4u+3
// hold one (row, plane) unit; units 2v, 2v + 1 of an 8-lane ds_write_b128 group are rows r, r + 4
template <int RW>
__device__ __forceinline__ void image_unit(int c, int& row, int& plane) {
    const int wave = (c & 255) >> 6, u = ((c & 63) >> 2) + 16 * (c >> 8);
    const int v = u >> 1, rb = v % (RW / 2);
    plane = v / (RW / 2);
    row = wave * RW + (rb >> 2) * 8 + (rb & 3) + 4 * (u & 1);
}

// Matrix phase of one staged tile.  al / bl: this lane's fragment address in plane 0 of the A / B
// image (row = tile row of the wave + lane % 32, k = (lane / 32) * 8); planes APL / BPL bytes apart.
// All fragments of a 16-deep step are read first, then the MFMAs walk the accumulators round
// robin, smallest terms first.
template <int TM, int TN, int APL, int BPL>
__device__ __forceinline__ void split_compute(const char* al, const char* bl, f32x16 (&acc)[TM][TN]) {
    constexpr int JC = TM >= 2 ? 1 : (TN > 2 ? 2 : TN);  // column tiles per pass (bounds the fragment registers)
#pragma unroll
    for (int kk = 0; kk < SK / 16; ++kk) {
        bf16x8 a[3][TM];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[p][i] = *reinterpret_cast<const bf16x8*>(al + p * APL + i * 32 * SROW + kk * 32);
#pragma unroll
        for (int j0 = 0; j0 < TN; j0 += JC) {
            bf16x8 b[3][JC];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < JC; ++j)
                    b[p][j] = *reinterpret_cast<const bf16x8*>(bl + p * BPL + (j0 + j) * 32 * SROW + kk * 32);
#define MARL_SPLIT_P(pa_, pb_)                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                         \
        _Pragma("unroll") for (int j = 0; j < JC; ++j)                                     \
            acc[i][j0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa_][i], b[pb_][j], acc[i][j0 + j], 0, 0, 0);
            MARL_SPLIT_P(1, 1) MARL_SPLIT_P(0, 2) MARL_SPLIT_P(2, 0)
            MARL_SPLIT_P(0, 1) MARL_SPLIT_P(1, 0) MARL_SPLIT_P(0, 0)
#undef MARL_SPLIT_P
        }
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// NT: C[M,N] (+)= sum_s A_s[M,K_s] * B_s[N,K_s]^T + bias, optional LSTM-cell epilogue.
// 128 x BN tiles, 4 waves; two sets of staging registers keep two K tiles in flight over ONE
// LDS stage (two workgroups per CU: one stages while the other feeds the matrix pipe).
// ---------------------------------------------------------------------------
// BPRE: the B operands are weights whose bf16x3 image already exists in the weights workspace
// (split_weights_kernel: [row][k / 32][plane][32] bf16, zero-padded to whole K tiles) - their tiles
// are copied, not split: the staging arithmetic of the kernel halves.
template <int BN, bool LSTM, bool BPRE>
__global__ __launch_bounds__(256, 2) void gemm_nt_split_kernel(const GemmBatch batch) {
    constexpr int BM = 128;
    constexpr int B3_CH = BN * 12 / 256;       // 16-byte chunks of a pre-split B tile per thread
    constexpr int WM = LSTM ? 4 : 2, WN = LSTM ? 1 : 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int KC = SK / 4;                 // float4 chunks per tile row
    constexpr int A_CH = BM * KC / 256;        // 4
    constexpr int B_CH = BN * KC / 256;        // 4 or 2
    constexpr int APL = BM * SROW, BPL = BN * SROW;
    static_assert(!LSTM || BN == 128, "LSTM tile = 4 gates x 32 units");

    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* const As = smem_c;
    char* const Bs = smem_c + 3 * APL;

    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (batch.xcd_map) xcd_tile(batch.gx, batch.gy, batch.count, bx, by, bz);
    const GemmProb P = batch.p[bz];
    const int M = P.m;
    const int N = P.n;  // LSTM: hidden units (B has 4*N rows)
    const int n0 = by * (LSTM ? 32 : BN);
    const int m0 = bx * BM;
    if (m0 >= M || n0 >= N) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int t0 = (P.seg[0].k + SK - 1) / SK;
    const int t1 = P.nseg > 1 ? (P.seg[1].k + SK - 1) / SK : 0;
    const int T = t0 + t1;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: thread t moves A chunks c = t + 256 i (tile row c / KC, floats [(c % KC) * 4, +4)),
    // B likewise; address = uniform base (advanced per tile on the scalar unit) + fixed 32-bit
    // per-thread byte offset.  Rows beyond M / N are clamped (never stored).
    const int koff = (tid % KC) * 4;
    uint32_t aof[A_CH], bof[BPRE ? B3_CH : B_CH];
    const char* abase = nullptr;
    const char* bbase = nullptr;
    int K4cur = 0;
    auto set_seg_a = [&](int sg) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            int gm = m0 + stage_row(tid + 256 * i);
            gm = gm < M ? gm : M - 1;
            aof[i] = ((uint32_t)gm * (uint32_t)P.seg[sg].lda + (uint32_t)koff) * 4u;
        }
        abase = reinterpret_cast<const char*>(P.seg[sg].a);
        K4cur = (P.seg[sg].k + 3) & ~3;
    };
    auto set_seg_b = [&](int sg) {
#pragma unroll
        for (int i = 0; i < (BPRE ? B3_CH : B_CH); ++i) {
            const int c = tid + 256 * i;
            int row = stage_row(c), plane = 0;
            if (BPRE) image_unit<BN / 4>(c, row, plane);
            int gn;
            if (LSTM) {
                int unit = n0 + (row & 31);
                unit = unit < N ? unit : N - 1;
                gn = (row >> 5) * N + unit;
            } else {
                gn = n0 + row;
                gn = gn < N ? gn : N - 1;
            }
            if (BPRE)  // row gn of the image: kt tiles of 192 bytes; chunk c % 12 of the tile
                bof[i] = (uint32_t)gn * (uint32_t)P.seg[sg].kt3 * 192u + (uint32_t)(plane * 64 + (c & 3) * 16);
            else
                bof[i] = ((uint32_t)gn * (uint32_t)P.seg[sg].ldb + (uint32_t)koff) * 4u;
        }
        bbase = BPRE ? reinterpret_cast<const char*>(P.seg[sg].b3) : reinterpret_cast<const char*>(P.seg[sg].b);
    };

    float4 raX[A_CH], raY[A_CH];
    float4 rbX[BPRE ? 1 : B_CH], rbY[BPRE ? 1 : B_CH];  // fp32 B: two sets like A
    u32x4 rb3[BPRE ? B3_CH : 1];                         // pre-split B: ONE set, a tile ahead (L2-resident)
    float mkX = 1.f, mkY = 1.f;
    bool maskedX = false, maskedY = false;
    // UNCONDITIONAL loads (a load behind a branch makes hipcc drain vmcnt): tiles past the end
    // re-read the last tile; only the last tile of a segment can reach past round4(K) - there
    // the chunk address is clamped into the row (finite values) and the B side is zero: the
    // fp32 form is zeroed when it goes to LDS, the pre-split image is zero-padded.
    // Issue order: the B tile (needed first) before the A tile that stays in flight longer.
#define MARL_SP_LOADB3(tile_)                                                              \
    if (BPRE) {                                                                            \
        const bool live_ = (tile_) < T;                                                    \
        if (live_ && (tile_) == t0 && t1 > 0) set_seg_b(1);                                \
        bbase -= live_ ? 0 : 192;                                                          \
        _Pragma("unroll") for (int i = 0; i < B3_CH; ++i)                                  \
            rb3[BPRE ? i : 0] = *reinterpret_cast<const u32x4*>(bbase + bof[i]);           \
        bbase += 192;                                                                      \
    }
#define MARL_SP_LOAD(ra_, rb_, mk_, masked_, tile_)                                        \
    {                                                                                      \
        const bool live_ = (tile_) < T;                                                    \
        if (live_ && (tile_) == t0 && t1 > 0) {                                            \
            set_seg_a(1);                                                                  \
            if (!BPRE) set_seg_b(1);                                                       \
        }                                                                                  \
        const int tc_ = live_ ? (tile_) : T - 1;                                           \
        const int k0_ = (tc_ >= t0 ? tc_ - t0 : tc_) * SK;                                 \
        masked_ = k0_ + SK > K4cur;                                                        \
        const int k_ = k0_ + koff;                                                         \
        const uint32_t d_ = (!masked_ || k_ < K4cur) ? 0u : (uint32_t)((K4cur - 4 - k_) * 4); \
        mk_ = (!masked_ || k_ < K4cur) ? 1.f : 0.f;                                        \
        abase -= live_ ? 0 : SK * 4;                                                       \
        if (!BPRE) bbase -= live_ ? 0 : SK * 4;                                            \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i)                                   \
            ra_[i] = *reinterpret_cast<const float4*>(abase + (aof[i] + d_));              \
        if (!BPRE) {                                                                       \
            _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                               \
                rb_[BPRE ? 0 : i] = *reinterpret_cast<const float4*>(bbase + (bof[i] + d_)); \
            bbase += SK * 4;                                                               \
        }                                                                                  \
        abase += SK * 4;                                                                   \
    }
#ifdef MARL_KERNEL_TS
#define MARL_SP_DBGWAIT() if (!BPRE) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); MARL_TS(); }
#define MARL_SP_DBGTS() MARL_TS();
#else
#define MARL_SP_DBGWAIT()
#define MARL_SP_DBGTS()
#endif
#define MARL_SP_STORE(ra_, rb_, mk_, masked_)                                              \
    {                                                                                      \
        MARL_SP_DBGWAIT()                                                                  \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                 \
            const int c_ = tid + 256 * i;                                                  \
            split_store4(As + stage_row(c_) * SROW + (c_ % KC) * 8, APL, ra_[i].x, ra_[i].y, \
                         ra_[i].z, ra_[i].w);                                              \
        }                                                                                  \
        MARL_SP_DBGTS()                                                                    \
        if (BPRE) {                                                                        \
            _Pragma("unroll") for (int i = 0; i < B3_CH; ++i) {                            \
                int row_, pl_;                                                             \
                image_unit<BN / 4>(tid + 256 * i, row_, pl_);                              \
                *reinterpret_cast<u32x4*>(Bs + pl_ * BPL + row_ * SROW + (tid & 3) * 16) = \
                    rb3[BPRE ? i : 0];                                                     \
            }                                                                              \
        } else {                                                                           \
            _Pragma("unroll") for (int i = 0; i < B_CH; ++i) {                             \
                const int c_ = tid + 256 * i;                                              \
                float4 v_ = rb_[BPRE ? 0 : i];                                             \
                if (masked_) {                                                             \
                    v_.x *= mk_;                                                           \
                    v_.y *= mk_;                                                           \
                    v_.z *= mk_;                                                           \
                    v_.w *= mk_;                                                           \
                }                                                                          \
                split_store4(Bs + stage_row(c_) * SROW + (c_ % KC) * 8, BPL, v_.x, v_.y, v_.z, v_.w); \
            }                                                                              \
        }                                                                                  \
    }

    const char* al = As + (wm * (BM / WM) + (lane & 31)) * SROW + (lane >> 5) * 16;
    const char* bl = Bs + (wn * (BN / WN) + (lane & 31)) * SROW + (lane >> 5) * 16;

    MARL_TS_DECL(batch.ts);
    MARL_TS();
    set_seg_a(0);
    set_seg_b(0);
    MARL_SP_LOADB3(0)
    MARL_SP_LOAD(raX, rbX, mkX, maskedX, 0)
    MARL_SP_LOAD(raY, rbY, mkY, maskedY, 1)
    int tile = 0;
    MARL_TS();
    for (; tile + 1 < T; tile += 2) {
        if (tile > 0) lds_barrier();
        MARL_TS();
        MARL_SP_STORE(raX, rbX, mkX, maskedX)
        MARL_TS();
        lds_barrier();
        MARL_TS();
        MARL_SP_LOADB3(tile + 1)
        MARL_SP_LOAD(raX, rbX, mkX, maskedX, tile + 2)
        MARL_TS();
        split_compute<TM, TN, APL, BPL>(al, bl, acc);
        MARL_TS();
        lds_barrier();
        MARL_TS();
        MARL_SP_STORE(raY, rbY, mkY, maskedY)
        MARL_TS();
        lds_barrier();
        MARL_SP_LOADB3(tile + 2)
        MARL_SP_LOAD(raY, rbY, mkY, maskedY, tile + 3)
        MARL_TS();
        split_compute<TM, TN, APL, BPL>(al, bl, acc);
        MARL_TS();
    }
    if (tile < T) {
        if (tile > 0) lds_barrier();
        MARL_SP_STORE(raX, rbX, mkX, maskedX)
        lds_barrier();
        split_compute<TM, TN, APL, BPL>(al, bl, acc);
    }
#undef MARL_SP_LOADB3
#undef MARL_SP_LOAD
#undef MARL_SP_STORE

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
            // (previous cell state and biases are fetched here: holding them across the K loop
            // costs 20 registers this kernel does not have at two workgroups per CU)
            const float bi = P.bias[unit], bf = P.bias[N + unit], bg = P.bias[2 * N + unit], bo = P.bias[3 * N + unit];
            float cprev[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row_ = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                row_ = row_ < M ? row_ : M - 1;
                cprev[r] = P.c_prev[(size_t)row_ * P.ld_state + unit];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                if (row < M) {
                    const float gi = sigmoid_acc(acc[0][0][r] + bi);
                    const float gf = sigmoid_acc(acc[0][LSTM ? 1 : 0][r] + bf);
                    const float gg = tanh_fast(acc[0][LSTM ? 2 : 0][r] + bg);
                    const float go = sigmoid_acc(acc[0][LSTM ? 3 : 0][r] + bo);
                    const size_t so = (size_t)row * P.ld_state + unit;
                    const float cn = gf * cprev[r] + gi * gg;
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
// TN: C[NI,NJ] = sum_r A[r,i] * B[r,j] over the row slab of this workgroup (weight gradients).
// A thread stages one 4 x 4 block (4 rows x 4 columns) of each operand per tile and transposes
// it in registers: LDS row = matrix column, 4 consecutive rows r = 8 bytes of a plane.
// ---------------------------------------------------------------------------
// NW = 4: a thread stages one block of EACH operand, a wave owns 64 x 64 of the tile.  NW = 8 (512
// threads): threads 0..255 stage A, 256..511 stage B (half the staging registers), a wave owns
// 32 x 64 - twice the waves per SIMD to cover barrier and operand latency.
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void gemm_tn_split_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
    float* __restrict__ out, int ldo, int64_t out_split_stride, int NI, int NJ, int64_t rows,
    int64_t rows_per_split, float* __restrict__ csum, int gx, int gy, int gz) {
    constexpr int BM = 128, BN = 128, WMn = NW / 2, TM = BM / WMn / 32, TN = 2;
    constexpr bool HALF = NW == 8;  // one operand per thread
    constexpr int APL = BM * SROW, BPL = BN * SROW;
    static_assert(APL == BPL, "shared plane stride");
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* const As = smem_c;
    char* const Bs = smem_c + 3 * APL;

    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (gx > 0) xcd_tile(gx, gy, gz, bx, by, bz);
    const int i0 = bx * BM, j0 = by * BN;
    const int64_t r_begin = (int64_t)bz * rows_per_split;
    int64_t r_end = r_begin + rows_per_split;
    if (r_end > rows) r_end = rows;
    const int NI4 = (NI + 3) & ~3, NJ4 = (NJ + 3) & ~3;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int T = r_end > r_begin ? (int)((r_end - r_begin + SK - 1) / SK) : 0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // block of this thread: tile rows [4 rb, +4), columns [4 cb, +4) (clamped into the padded
    // width; columns past NI / NJ are never stored)
    const int st = tid & 255;
    const bool isB = HALF && tid >= 256;
    const int rb = st & 7, cb = st >> 3;
    const int ic = i0 + 4 * cb, jc = j0 + 4 * cb;
    const uint32_t acol = (uint32_t)(ic < NI4 ? ic : NI4 - 4) * 4u;
    const uint32_t bcol = (uint32_t)(jc < NJ4 ? jc : NJ4 - 4) * 4u;
    const char* abase = reinterpret_cast<const char*>(A + (size_t)r_begin * lda);
    const char* bbase = reinterpret_cast<const char*>(B + (size_t)r_begin * ldb);
    if (HALF && isB) {  // this thread's ONE operand goes through the "a" names below
        abase = bbase;
        lda = ldb;
    }
    const uint32_t col1 = isB ? bcol : acol;
    const bool do_csum = csum != nullptr && by == 0 && !isB;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);

    float4 raX[4], rbX[HALF ? 1 : 4], raY[4], rbY[HALF ? 1 : 4];
    float mX[4] = {1.f, 1.f, 1.f, 1.f}, mY[4] = {1.f, 1.f, 1.f, 1.f};
    bool maskedX = false, maskedY = false;
    // unconditional loads; tiles past the end re-read the last tile; rows past r_end are clamped
    // to the last valid row and the A side zeroed when the tile goes to LDS
#define MARL_TS_LOAD(ra_, rb_, m_, masked_, tile_)                                         \
    {                                                                                      \
        const bool live_ = (tile_) < T;                                                    \
        abase -= live_ ? (size_t)0 : (size_t)SK * lda * 4;                                 \
        if (!HALF) bbase -= live_ ? (size_t)0 : (size_t)SK * ldb * 4;                      \
        const int64_t base_ = r_begin + (int64_t)(live_ ? (tile_) : T - 1) * SK;           \
        masked_ = base_ + SK > r_end;                                                      \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                    \
            int rr_ = 4 * rb + q;                                                          \
            m_[q] = 1.f;                                                                   \
            if (masked_ && base_ + rr_ >= r_end) {                                         \
                rr_ = (int)(r_end - 1 - base_);                                            \
                m_[q] = 0.f;                                                               \
            }                                                                              \
            ra_[q] = *reinterpret_cast<const float4*>(abase + ((uint32_t)rr_ * (uint32_t)lda * 4u + col1)); \
            if (!HALF)                                                                     \
                rb_[HALF ? 0 : q] = *reinterpret_cast<const float4*>(bbase + ((uint32_t)rr_ * (uint32_t)ldb * 4u + bcol)); \
        }                                                                                  \
        abase += (size_t)SK * lda * 4;                                                     \
        if (!HALF) bbase += (size_t)SK * ldb * 4;                                          \
    }
#define MARL_TS_STORE(ra_, rb_, m_, masked_)                                               \
    {                                                                                      \
        if (masked_) { /* (both operands zeroed: with one operand per thread either would do) */ \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                \
                ra_[q].x *= m_[q];                                                         \
                ra_[q].y *= m_[q];                                                         \
                ra_[q].z *= m_[q];                                                         \
                ra_[q].w *= m_[q];                                                         \
            }                                                                              \
        }                                                                                  \
        if (do_csum) {                                                                     \
            cs.x += (ra_[0].x + ra_[1].x) + (ra_[2].x + ra_[3].x);                         \
            cs.y += (ra_[0].y + ra_[1].y) + (ra_[2].y + ra_[3].y);                         \
            cs.z += (ra_[0].z + ra_[1].z) + (ra_[2].z + ra_[3].z);                         \
            cs.w += (ra_[0].w + ra_[1].w) + (ra_[2].w + ra_[3].w);                         \
        }                                                                                  \
        char* da_ = (isB ? Bs : As) + (4 * cb) * SROW + rb * 8;                            \
        split_store4(da_, APL, ra_[0].x, ra_[1].x, ra_[2].x, ra_[3].x);                    \
        split_store4(da_ + SROW, APL, ra_[0].y, ra_[1].y, ra_[2].y, ra_[3].y);             \
        split_store4(da_ + 2 * SROW, APL, ra_[0].z, ra_[1].z, ra_[2].z, ra_[3].z);         \
        split_store4(da_ + 3 * SROW, APL, ra_[0].w, ra_[1].w, ra_[2].w, ra_[3].w);         \
        if (!HALF) {                                                                       \
            char* db_ = Bs + (4 * cb) * SROW + rb * 8;                                     \
            split_store4(db_, BPL, rb_[0].x, rb_[HALF ? 0 : 1].x, rb_[HALF ? 0 : 2].x, rb_[HALF ? 0 : 3].x); \
            split_store4(db_ + SROW, BPL, rb_[0].y, rb_[HALF ? 0 : 1].y, rb_[HALF ? 0 : 2].y, rb_[HALF ? 0 : 3].y); \
            split_store4(db_ + 2 * SROW, BPL, rb_[0].z, rb_[HALF ? 0 : 1].z, rb_[HALF ? 0 : 2].z, rb_[HALF ? 0 : 3].z); \
            split_store4(db_ + 3 * SROW, BPL, rb_[0].w, rb_[HALF ? 0 : 1].w, rb_[HALF ? 0 : 2].w, rb_[HALF ? 0 : 3].w); \
        }                                                                                  \
    }

    const char* al = As + (wm * (BM / WMn) + (lane & 31)) * SROW + (lane >> 5) * 16;
    const char* bl = Bs + (wn * 64 + (lane & 31)) * SROW + (lane >> 5) * 16;

    if (T > 0) {
        MARL_TS_LOAD(raX, rbX, mX, maskedX, 0)
        MARL_TS_LOAD(raY, rbY, mY, maskedY, 1)
    }
    int tile = 0;
    for (; tile + 1 < T; tile += 2) {
        if (tile > 0) lds_barrier();
        MARL_TS_STORE(raX, rbX, mX, maskedX)
        lds_barrier();
        MARL_TS_LOAD(raX, rbX, mX, maskedX, tile + 2)
        split_compute<TM, TN, APL, BPL>(al, bl, acc);
        lds_barrier();
        MARL_TS_STORE(raY, rbY, mY, maskedY)
        lds_barrier();
        MARL_TS_LOAD(raY, rbY, mY, maskedY, tile + 3)
        split_compute<TM, TN, APL, BPL>(al, bl, acc);
    }
    if (tile < T) {
        if (tile > 0) lds_barrier();
        MARL_TS_STORE(raX, rbX, mX, maskedX)
        lds_barrier();
        split_compute<TM, TN, APL, BPL>(al, bl, acc);
    }
#undef MARL_TS_LOAD
#undef MARL_TS_STORE

    if (csum != nullptr && by == 0) {  // the 8 threads rb = 0..7 of a column block staged the same 4 columns
        __syncthreads();
        float4* sh4 = reinterpret_cast<float4*>(smem_c);
        if (!isB) sh4[rb * 32 + cb] = cs;
        __syncthreads();
        if (tid < 32) {
            float4 t = sh4[tid];
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                const float4 u = sh4[q * 32 + tid];
                t.x += u.x;
                t.y += u.y;
                t.z += u.z;
                t.w += u.w;
            }
            float* co = csum + (size_t)bz * NI;
            const int c0 = i0 + tid * 4;
            if (c0 < NI) co[c0] = t.x;
            if (c0 + 1 < NI) co[c0 + 1] = t.y;
            if (c0 + 2 < NI) co[c0 + 2] = t.z;
            if (c0 + 3 < NI) co[c0 + 3] = t.w;
        }
    }

    float* o = out + (size_t)bz * out_split_stride;
    const int fcol = lane & 31;
    const int row_h = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = j0 + wn * 64 + j * 32 + fcol;
            if (col >= NJ) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * (BM / WMn) + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                if (row < NI) o[(size_t)row * ldo + col] = acc[i][j][r];
            }
        }
}

// ---------------------------------------------------------------------------
// pre-split images of the weights
// ---------------------------------------------------------------------------
__global__ void split_weights_kernel(const SplitBatch B) {
    const SplitDesc& d = B.d[blockIdx.y];
    const int64_t tot = (int64_t)d.rows * d.kt * 16;  // one thread per pair of consecutive k
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int pr = (int)(idx & 15);
        const int64_t rt = idx >> 4;  // row * kt + tile
        const int t = (int)(rt % d.kt);
        const int64_t r = rt / d.kt;
        const int k = t * 32 + pr * 2;
        const float x = k < d.k ? d.src[r * d.ld + k] : 0.f;
        const float y = k + 1 < d.k ? d.src[r * d.ld + k + 1] : 0.f;
        uint32_t p0, p1, p2;
        split_pair(x, y, p0, p1, p2);
        uint32_t* o = static_cast<uint32_t*>(d.dst) + rt * 48 + pr;
        o[0] = p0;
        o[16] = p1;
        o[32] = p2;
    }
}

size_t split_image_floats(int rows, int k) { return (size_t)rows * ((k + 31) / 32) * 48; }

int launch_split_weights(const SplitBatch& b, hipStream_t st) {
    if (b.count <= 0) return MARL_OK;
    if (b.count > kMaxSplitDesc) return MARL_EINVAL;
    int64_t mx = 0;
    for (int i = 0; i < b.count; ++i) {
        const int64_t t = (int64_t)b.d[i].rows * b.d[i].kt * 16;
        mx = t > mx ? t : mx;
    }
    int64_t gx = cdiv(mx, 256);
    gx = gx > 512 ? 512 : (gx < 1 ? 1 : gx);
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)gx, (unsigned)b.count), dim3(256), 0, st, b);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

namespace {
struct SplitReg {
    const float* base;
    size_t floats;
    int ld, k;
    const char* image;
};
// (per THREAD and per CALL: ctypes releases the GIL, two engines may be inside the library at once; every
// entry point that registers clears the table again before it returns - SplitRegistryScope - so a later
// kernel-level call can never meet the image of a workspace that was freed or rewritten since)
thread_local SplitReg g_reg[kMaxSplitDesc];
thread_local int g_nreg = 0;
// b = base + row0 * ld of a registered matrix with the same row stride and depth -> its image rows
bool split_lookup(const GemmSeg& g, const void*& b3, int& kt) {
    for (int i = 0; i < g_nreg; ++i) {
        const SplitReg& r = g_reg[i];
        if (g.b < r.base || g.b >= r.base + r.floats || g.ldb != r.ld || g.k != r.k) continue;
        const size_t off = (size_t)(g.b - r.base);
        if (off % (size_t)r.ld) return false;
        kt = (r.k + 31) / 32;
        b3 = r.image + (off / (size_t)r.ld) * (size_t)kt * 192;
        return true;
    }
    return false;
}
}  // namespace

void split_registry_reset() { g_nreg = 0; }
void split_registry_add(const float* base, int rows, int ld, int k, const void* image) {
    if (g_nreg < kMaxSplitDesc)
        g_reg[g_nreg++] = SplitReg{base, (size_t)rows * ld, ld, k, static_cast<const char*>(image)};
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int split_mode() { return tune_get("mfma_split", 1); }

template <int BN, bool LSTM>
static int launch_nt_split_variant(dim3 grid, const GemmBatch& batch_in, hipStream_t st) {
    GemmBatch batch = batch_in;
    batch.gx = (int)grid.x;
    batch.gy = (int)grid.y;
    batch.xcd_map = batch.count == 1 && !LSTM && tune_get("nt_xcd", 1);
    if (batch.xcd_map) grid = dim3(grid.x * grid.y * grid.z);
    // every B operand a registered weight matrix: the kernel copies their pre-split tiles
    bool pre = true;
    for (int i = 0; i < batch.count && pre; ++i)
        for (int sg = 0; sg < batch.p[i].nseg && pre; ++sg) {
            GemmSeg& g = batch.p[i].seg[sg];
            pre = split_lookup(g, g.b3, g.kt3);
            // 32-bit byte offsets into the image
            if (pre && (int64_t)(LSTM ? 4 : 1) * batch.p[i].n * g.kt3 * 192 >= (1ll << 32)) pre = false;
        }
    constexpr size_t lds = (size_t)3 * (128 + BN) * SROW;
#ifdef MARL_KERNEL_TS
    static long long* d_ts = nullptr;
    static int calls = 0;
    const int rec = ts_begin(&d_ts, calls++);
    batch.ts = rec ? d_ts : nullptr;
#endif
    if (pre)
        hipLaunchKernelGGL((gemm_nt_split_kernel<BN, LSTM, true>), grid, dim3(256), lds, st, batch);
    else
        hipLaunchKernelGGL((gemm_nt_split_kernel<BN, LSTM, false>), grid, dim3(256), lds, st, batch);
#ifdef MARL_KERNEL_TS
    if (rec) ts_report(pre ? "nt_split_pre" : "nt_split", d_ts, 4);
#endif
    return MARL_OK;
}

int launch_gemm_nt_split(const GemmBatch& batch, int max_m, int max_n, int64_t blocks128, hipStream_t st) {
    // 128-wide column tiles when they fill the chip (two workgroups per CU), else 128 x 64
    if (blocks128 >= 384 && max_n >= 96) {
        dim3 grid((unsigned)cdiv(max_m, 128), (unsigned)cdiv(max_n, 128), (unsigned)batch.count);
        return launch_nt_split_variant<128, false>(grid, batch, st);
    }
    dim3 grid((unsigned)cdiv(max_m, 128), (unsigned)cdiv(max_n, 64), (unsigned)batch.count);
    return launch_nt_split_variant<64, false>(grid, batch, st);
}

int launch_gemm_lstm_split(const GemmBatch& batch, int max_m, int max_n, hipStream_t st) {
    dim3 grid((unsigned)cdiv(max_m, 128), (unsigned)cdiv(max_n, 32), (unsigned)batch.count);
    return launch_nt_split_variant<128, true>(grid, batch, st);
}

int launch_gemm_tn_split(const float* a, int lda, const float* b, int ldb, float* out, int ldo,
                         int64_t stride, int ni, int nj, int64_t rows, int64_t rows_per_split,
                         float* csum, dim3 grid, int gx, int gy, int gz, hipStream_t st) {
    if (tune_get("tn_split_waves", 8) == 8)
        hipLaunchKernelGGL(gemm_tn_split_kernel<8>, grid, dim3(512), (size_t)3 * 256 * SROW, st, a, lda, b, ldb,
                           out, ldo, stride, ni, nj, rows, rows_per_split, csum, gx, gy, gz);
    else
        hipLaunchKernelGGL(gemm_tn_split_kernel<4>, grid, dim3(256), (size_t)3 * 256 * SROW, st, a, lda, b, ldb,
                           out, ldo, stride, ni, nj, rows, rows_per_split, csum, gx, gy, gz);
    return MARL_OK;
}

}  // namespace marl
