// fp32 matrix products on the bf16 matrix pipe ("bf16x6"): MI355X runs v_mfma_f32_32x32x2_f32 at
// 1/16 of the bf16 MFMA rate (MI355X_MICROARCH.md: 157 TF vs 2.5 PF dense), so an exact-fp32
// product is priced at the VECTOR rate.  Here every fp32 operand element x is split, in
// registers while its tile goes to LDS, into three bf16 terms
//     x0 = bf16(x),  x1 = bf16(x - x0),  x2 = bf16(x - x0 - x1)        (round to nearest)
// with x0 + x1 + x2 == x EXACTLY (3 x 8 significand bits + the signs cover fp32's 24), and a
// product a*b is accumulated in fp32 from the six bf16 x bf16 MFMA products of weight >= 2^-16,
//     a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0);
// the three dropped terms are below 2^-25 |a b|, i.e. under fp32's own rounding step.  Every
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
constexpr int plane_bytes(int rows) { return rows * SROW; }

__device__ __forceinline__ float sigmoid_acc(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
    return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x));
}
// (sched_barrier: nothing - in particular no matrix instruction, which touches registers only -
// may be scheduled across the barrier: each tile body is one scheduling region)
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// two fp32 values -> one dword (x low half, y high half) per bf16 term
__device__ __forceinline__ uint32_t pack_bf16(float x, float y) {
    const f32x2_t v = {x, y};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));  // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ void split_pair(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pack_bf16(x, y);
    const float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = pack_bf16(rx, ry);
    p2 = pack_bf16(rx - __uint_as_float(p1 << 16), ry - __uint_as_float(p1 & 0xffff0000u));
}
// four consecutive-k values -> 8 bytes in each of the three planes
__device__ __forceinline__ void split_store4(char* dst, int plane, float a, float b, float c, float d) {
    uint32_t a0, a1, a2, b0, b1, b2;
    split_pair(a, b, a0, a1, a2);
    split_pair(c, d, b0, b1, b2);
    *reinterpret_cast<uint2*>(dst) = make_uint2(a0, b0);
    *reinterpret_cast<uint2*>(dst + plane) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(dst + 2 * plane) = make_uint2(a2, b2);
}

// Matrix phase of one staged tile.  al / bl: this lane's fragment address in plane 0 of the A / B
// image (row = tile row of the wave + lane % 32, k = (lane / 32) * 8); planes APL / BPL bytes apart.
// All 12 + 12 fragment reads of the tile are issued up front (the wave has the registers), the
// MFMAs walk the accumulators round robin, smallest terms first.
template <int TM, int TN>
struct SplitFrags {
    bf16x8 a[3][TM], b[3][TN];
};
template <int TM, int TN, int APL, int BPL>
__device__ __forceinline__ void split_read(const char* al, const char* bl, int kk, SplitFrags<TM, TN>& f) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            f.a[p][i] = *reinterpret_cast<const bf16x8*>(al + p * APL + i * 32 * SROW + kk * 32);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            f.b[p][j] = *reinterpret_cast<const bf16x8*>(bl + p * BPL + j * 32 * SROW + kk * 32);
    }
}
template <int TM, int TN>
__device__ __forceinline__ void split_mfma(const SplitFrags<TM, TN>& f, f32x16 (&acc)[TM][TN]) {
#define MARL_SPLIT_P(pa_, pb_)                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                         \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                     \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[pa_][i], f.b[pb_][j], acc[i][j], 0, 0, 0);
    MARL_SPLIT_P(1, 1) MARL_SPLIT_P(0, 2) MARL_SPLIT_P(2, 0)
    MARL_SPLIT_P(0, 1) MARL_SPLIT_P(1, 0) MARL_SPLIT_P(0, 0)
#undef MARL_SPLIT_P
}
template <int TM, int TN, int APL, int BPL>
__device__ __forceinline__ void split_compute(const char* al, const char* bl, f32x16 (&acc)[TM][TN]) {
    SplitFrags<TM, TN> f0, f1;
    split_read<TM, TN, APL, BPL>(al, bl, 0, f0);
    split_read<TM, TN, APL, BPL>(al, bl, 1, f1);
    split_mfma<TM, TN>(f0, acc);
    split_mfma<TM, TN>(f1, acc);
}

// Issue order of one pipelined tile body (a single scheduling region): the fragment reads and
// the memory requests first, then every MFMA followed by NV single-issue instructions of the
// staging arithmetic (they run in the 32-cycle shadow of the matrix instruction) and, every
// other MFMA, one LDS store.  hipcc by itself emits the staging arithmetic as ONE block in
// front of 48 back-to-back MFMAs (no overlap at all: 2.1 us per tile instead of 0.8).
template <int NMFMA, int NV, int NREAD, int NVMEM>
__device__ __forceinline__ void split_pipeline() {
    __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);   // DS reads
    __builtin_amdgcn_sched_group_barrier(0x020, NVMEM, 0);   // VMEM reads
#pragma unroll
    for (int m = 0; m < NMFMA; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);  // VALU
        if (m & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// Both kernels are software-pipelined inside ONE wave per SIMD (256 threads, one workgroup per
// CU, two LDS stages of 60 KB, up to 512 registers): while the 48 MFMAs of tile t run from LDS
// stage t % 2, the same wave splits tile t + 1 (already in registers) into the other stage and
// issues the loads of tile t + 3 - the VALU / LDS / memory instructions sit in the 32-cycle
// shadows of the matrix instructions (tools/ubench_split.hip; MI355X_MICROARCH.md: up to 5
// single-issue instructions per v_mfma_f32_32x32x16_bf16), one barrier per tile.  Two
// independent workgroups per CU measured 2x slower per phase (they drift into lockstep), a
// ping-pong pair of wave groups likewise.
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// NT: C[M,N] (+)= sum_s A_s[M,K_s] * B_s[N,K_s]^T + bias, optional LSTM-cell epilogue; 128 x BN tiles.
// BPRE: the B operands are weights whose bf16x3 image already exists in the weights workspace
// (split_weights_kernel: [row][k / 32][plane][32] bf16, zero-padded to whole K tiles) - their tiles
// are copied, not split.
// ---------------------------------------------------------------------------
template <int BN, bool LSTM, bool BPRE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_nt_split_kernel(const GemmBatch batch) {
    constexpr int BM = 128;
    constexpr int WM = LSTM ? 4 : 2, WN = LSTM ? 1 : 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int KC = SK / 4;                 // float4 chunks per tile row
    constexpr int A_CH = BM * KC / 256;        // 4
    constexpr int B_CH = BN * KC / 256;        // 4 or 2
    constexpr int B3_CH = BN * 12 / 256;       // 16-byte chunks of a pre-split B tile per thread
    constexpr int APL = plane_bytes(BM), BPL = plane_bytes(BN);
    constexpr int GSZ = 3 * (APL + BPL);       // LDS bytes of one stage
    // A pre-split B tile row is 3 planes x 64 bytes: the 256 / BN threads of a row take the
    // 16-byte columns [h * BH, +BH) of every plane - one global offset and one LDS address per
    // thread plus immediates cover all of a thread's chunks.
    constexpr int BR = 256 / BN;   // threads per B row: 2 or 4
    constexpr int BH = 4 / BR;     // 16-byte columns per thread and plane: 2 or 1
    static_assert(3 * BH == B3_CH, "chunk count");
    static_assert(!LSTM || BN == 128, "LSTM tile = 4 gates x 32 units");

    extern __shared__ __attribute__((aligned(16))) char smem_c[];

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
    // the two K segments (LSTM: [u_t | h]); wave-uniform scalar state
    const int ks0 = P.seg[0].k, ks1 = P.nseg > 1 ? P.seg[1].k : 0;
    const int t0 = (ks0 + SK - 1) / SK;
    const int T = t0 + (ks1 + SK - 1) / SK;
    const int K40 = (ks0 + 3) & ~3, K41 = (ks1 + 3) & ~3;
    const char* const a0p = reinterpret_cast<const char*>(P.seg[0].a);
    const char* const a1p = reinterpret_cast<const char*>(P.seg[1].a);
    const char* const b0p = reinterpret_cast<const char*>(BPRE ? P.seg[0].b3 : (const void*)P.seg[0].b);
    const char* const b1p = reinterpret_cast<const char*>(BPRE ? P.seg[1].b3 : (const void*)P.seg[1].b);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: thread t moves A chunks c = t + 256 i (tile row c / KC, floats [(c % KC) * 4, +4)),
    // fp32 B likewise.  Rows beyond M / N are clamped (never stored).  Byte offsets of both
    // segments are kept in registers (this kernel has 512 of them).
    const int koff = (tid % KC) * 4;
    uint32_t aof0[A_CH], aof1[A_CH], bof0[BPRE ? 1 : B_CH], bof1[BPRE ? 1 : B_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        int r = m0 + (tid + 256 * i) / KC;
        r = r < M ? r : M - 1;
        aof0[i] = (uint32_t)r * (uint32_t)P.seg[0].lda * 4u;
        aof1[i] = (uint32_t)r * (uint32_t)P.seg[1].lda * 4u;
    }
#pragma unroll
    for (int i = 0; i < (BPRE ? 1 : B_CH); ++i) {
        const int row = BPRE ? tid / BR : (tid + 256 * i) / KC;
        int gn;
        if (LSTM) {
            int unit = n0 + (row & 31);
            unit = unit < N ? unit : N - 1;
            gn = (row >> 5) * N + unit;
        } else {
            gn = n0 + row;
            gn = gn < N ? gn : N - 1;
        }
        if (BPRE) {  // row gn of the image: kt tiles of 192 bytes; this thread's byte column
            bof0[i] = (uint32_t)gn * (uint32_t)P.seg[0].kt3 * 192u + (uint32_t)(tid % BR) * (BH * 16);
            bof1[i] = (uint32_t)gn * (uint32_t)P.seg[1].kt3 * 192u + (uint32_t)(tid % BR) * (BH * 16);
        } else {
            bof0[i] = (uint32_t)gn * (uint32_t)P.seg[0].ldb * 4u;
            bof1[i] = (uint32_t)gn * (uint32_t)P.seg[1].ldb * 4u;
        }
    }

    // four sets of staging registers: tile u + 4 is requested while tile u is multiplied (three
    // tile times of latency cover: with two sets 42 % of the wave cycles were s_waitcnt vmcnt)
    float4 ra0[A_CH], ra1[A_CH], ra2[A_CH], ra3[A_CH];
    float4 rb0[BPRE ? 1 : B_CH], rb1[BPRE ? 1 : B_CH], rb2[BPRE ? 1 : B_CH], rb3[BPRE ? 1 : B_CH];
    u32x4 r30[BPRE ? B3_CH : 1], r31[BPRE ? B3_CH : 1], r32[BPRE ? B3_CH : 1], r33[BPRE ? B3_CH : 1];
    float mk0 = 1.f, mk1 = 1.f, mk2 = 1.f, mk3 = 1.f;
    // Loads are UNCONDITIONAL; tiles past the end re-read the last tile.  Only the last tile of a
    // segment can reach past round4(K): there the chunk address is clamped into the row (finite
    // values) and the B side is zero - the fp32 form is multiplied by a 0 / 1 mask when it goes
    // to LDS, the pre-split image is zero-padded.
#define MARL_SP_LOAD(S_, q_)                                                               \
    {                                                                                      \
        const int qd_ = (q_) < T ? (q_) : T - 1;                                           \
        const bool s1_ = qd_ >= t0;                                                        \
        const int tq_ = qd_ - (s1_ ? t0 : 0);                                              \
        const int kq_ = tq_ * SK;                                                          \
        const int K4_ = s1_ ? K41 : K40;                                                   \
        const bool in_ = kq_ + koff < K4_;                                                 \
        const uint32_t d_ = (uint32_t)((in_ ? koff : K4_ - 4 - kq_) * 4);                  \
        mk##S_ = in_ ? 1.f : 0.f;                                                          \
        if (BPRE) {                                                                        \
            const char* bp_ = (s1_ ? b1p : b0p) + (size_t)tq_ * 192 + (s1_ ? bof1[0] : bof0[0]); \
            _Pragma("unroll") for (int i = 0; i < B3_CH; ++i)                              \
                r3##S_[BPRE ? i : 0] = *reinterpret_cast<const u32x4*>(bp_ + (i / BH) * 64 + (i % BH) * 16); \
        }                                                                                  \
        const char* ap_ = (s1_ ? a1p : a0p) + (size_t)kq_ * 4;                             \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i)                                   \
            ra##S_[i] = *reinterpret_cast<const float4*>(ap_ + ((s1_ ? aof1[i] : aof0[i]) + d_)); \
        if (!BPRE) {                                                                       \
            const char* bp_ = (s1_ ? b1p : b0p) + (size_t)kq_ * 4;                         \
            _Pragma("unroll") for (int i = 0; i < B_CH; ++i)                               \
                rb##S_[BPRE ? 0 : i] = *reinterpret_cast<const float4*>(                   \
                    bp_ + ((s1_ ? bof1[BPRE ? 0 : i] : bof0[BPRE ? 0 : i]) + d_));         \
        }                                                                                  \
    }
#define MARL_SP_STORE(S_, buf_)                                                            \
    {                                                                                      \
        char* As_ = smem_c + (buf_) * GSZ;                                                 \
        char* Bs_ = As_ + 3 * APL;                                                         \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i) {                                 \
            const int c_ = tid + 256 * i;                                                  \
            split_store4(As_ + (c_ / KC) * SROW + (c_ % KC) * 8, APL, ra##S_[i].x, ra##S_[i].y, \
                         ra##S_[i].z, ra##S_[i].w);                                        \
        }                                                                                  \
        if (BPRE) {                                                                        \
            char* d3_ = Bs_ + (tid / BR) * SROW + (tid % BR) * (BH * 16);                  \
            _Pragma("unroll") for (int i = 0; i < B3_CH; ++i)                              \
                *reinterpret_cast<u32x4*>(d3_ + (i / BH) * BPL + (i % BH) * 16) = r3##S_[BPRE ? i : 0]; \
        } else {                                                                           \
            _Pragma("unroll") for (int i = 0; i < B_CH; ++i) {                             \
                const int c_ = tid + 256 * i;                                              \
                const float4 v_ = rb##S_[BPRE ? 0 : i];                                    \
                split_store4(Bs_ + (c_ / KC) * SROW + (c_ % KC) * 8, BPL, v_.x * mk##S_, v_.y * mk##S_, \
                             v_.z * mk##S_, v_.w * mk##S_);                                \
            }                                                                              \
        }                                                                                  \
    }
#define MARL_SP_COMPUTE(buf_)                                                              \
    split_compute<TM, TN, APL, BPL>(smem_c + (buf_) * GSZ + (wm * (BM / WM) + (lane & 31)) * SROW + (lane >> 5) * 16, \
                                    smem_c + (buf_) * GSZ + 3 * APL + (wn * (BN / WN) + (lane & 31)) * SROW + (lane >> 5) * 16, acc);
    // body of tile u = t + j (set j, stage j & 1): request tile u + 4 into the set tile u just
    // left, split tile u + 1 into the other stage, multiply tile u; one barrier
#define MARL_SP_BODY(j_, jn_)                                                              \
    MARL_SP_LOAD(j_, t + j_ + 4)                                                           \
    MARL_SP_STORE(jn_, (j_ + 1) & 1)                                                       \
    MARL_SP_COMPUTE(j_ & 1)                                                                \
    split_pipeline<TM * TN * 12, BPRE ? 3 : 5, (TM + TN) * 6, A_CH + (BPRE ? B3_CH : B_CH)>(); \
    lds_barrier();

    MARL_SP_LOAD(0, 0)
    MARL_SP_LOAD(1, 1)
    MARL_SP_LOAD(2, 2)
    MARL_SP_LOAD(3, 3)
    MARL_SP_STORE(0, 0)
    lds_barrier();
    for (int t = 0;; t += 4) {
        MARL_SP_BODY(0, 1)
        if (t + 1 >= T) break;
        MARL_SP_BODY(1, 2)
        if (t + 2 >= T) break;
        MARL_SP_BODY(2, 3)
        if (t + 3 >= T) break;
        MARL_SP_BODY(3, 0)
        if (t + 4 >= T) break;
    }
#undef MARL_SP_LOAD
#undef MARL_SP_STORE
#undef MARL_SP_COMPUTE
#undef MARL_SP_BODY

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
template <bool CSUM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_tn_split_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
    float* __restrict__ out, int ldo, int64_t out_split_stride, int NI, int NJ, int64_t rows,
    int64_t rows_per_split, float* __restrict__ csum, int gx, int gy, int gz) {
    constexpr int BM = 128, BN = 128, TM = 2, TN = 2;
    constexpr int APL = plane_bytes(BM), BPL = plane_bytes(BN);
    constexpr int GSZ = 3 * (APL + BPL);
    extern __shared__ __attribute__((aligned(16))) char smem_c[];

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
    // T full 32-row tiles go through the pipelined loop, a last partial tile (only the last
    // slab can have one) through the masked tail below
    const int64_t nrows = r_end > r_begin ? r_end - r_begin : 0;
    const int T = (int)(nrows / SK);
    const int tail = (int)(nrows - (int64_t)T * SK);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // block of this thread: tile rows [4 rb, +4), columns [4 cb, +4) (clamped into the padded
    // width; columns past NI / NJ are never stored)
    const int rb = tid & 7, cb = tid >> 3;
    const int ic = i0 + 4 * cb, jc = j0 + 4 * cb;
    uint32_t aof[4], bof[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        aof[q] = ((uint32_t)(4 * rb + q) * (uint32_t)lda + (uint32_t)(ic < NI4 ? ic : NI4 - 4)) * 4u;
        bof[q] = ((uint32_t)(4 * rb + q) * (uint32_t)ldb + (uint32_t)(jc < NJ4 ? jc : NJ4 - 4)) * 4u;
    }
    const char* const abase = reinterpret_cast<const char*>(A + (size_t)r_begin * lda);
    const char* const bbase = reinterpret_cast<const char*>(B + (size_t)r_begin * ldb);
    const bool do_csum = CSUM && by == 0;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);

    // four sets of staging registers (see gemm_nt_split_kernel); unconditional loads, tiles past
    // the end re-read the last full tile (staged into a stage that is never multiplied)
    float4 ra0[4], rb0[4], ra1[4], rb1[4], ra2[4], rb2[4], ra3[4], rb3[4];
#define MARL_TS_LOAD(S_, q_)                                                               \
    {                                                                                      \
        const int qd_ = (q_) < T ? (q_) : (T > 0 ? T - 1 : 0);                             \
        const char* ap_ = abase + (size_t)qd_ * SK * lda * 4;                              \
        const char* bp_ = bbase + (size_t)qd_ * SK * ldb * 4;                              \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                    \
            ra##S_[q] = *reinterpret_cast<const float4*>(ap_ + aof[q]);                    \
            rb##S_[q] = *reinterpret_cast<const float4*>(bp_ + bof[q]);                    \
        }                                                                                  \
    }
#define MARL_TS_STORE(S_, buf_, live_)                                                     \
    {                                                                                      \
        if (CSUM) { /* (tiles past the end are staged too - no branch in the body - with weight 0) */ \
            const float w_ = (live_) ? 1.f : 0.f;                                          \
            cs.x += w_ * ((ra##S_[0].x + ra##S_[1].x) + (ra##S_[2].x + ra##S_[3].x));      \
            cs.y += w_ * ((ra##S_[0].y + ra##S_[1].y) + (ra##S_[2].y + ra##S_[3].y));      \
            cs.z += w_ * ((ra##S_[0].z + ra##S_[1].z) + (ra##S_[2].z + ra##S_[3].z));      \
            cs.w += w_ * ((ra##S_[0].w + ra##S_[1].w) + (ra##S_[2].w + ra##S_[3].w));      \
        }                                                                                  \
        char* da_ = smem_c + (buf_) * GSZ + (4 * cb) * SROW + rb * 8;                      \
        char* db_ = da_ + 3 * APL;                                                         \
        split_store4(da_, APL, ra##S_[0].x, ra##S_[1].x, ra##S_[2].x, ra##S_[3].x);        \
        split_store4(da_ + SROW, APL, ra##S_[0].y, ra##S_[1].y, ra##S_[2].y, ra##S_[3].y); \
        split_store4(da_ + 2 * SROW, APL, ra##S_[0].z, ra##S_[1].z, ra##S_[2].z, ra##S_[3].z); \
        split_store4(da_ + 3 * SROW, APL, ra##S_[0].w, ra##S_[1].w, ra##S_[2].w, ra##S_[3].w); \
        split_store4(db_, BPL, rb##S_[0].x, rb##S_[1].x, rb##S_[2].x, rb##S_[3].x);        \
        split_store4(db_ + SROW, BPL, rb##S_[0].y, rb##S_[1].y, rb##S_[2].y, rb##S_[3].y); \
        split_store4(db_ + 2 * SROW, BPL, rb##S_[0].z, rb##S_[1].z, rb##S_[2].z, rb##S_[3].z); \
        split_store4(db_ + 3 * SROW, BPL, rb##S_[0].w, rb##S_[1].w, rb##S_[2].w, rb##S_[3].w); \
    }
#define MARL_TS_COMPUTE(buf_)                                                              \
    split_compute<TM, TN, APL, BPL>(smem_c + (buf_) * GSZ + (wm * 64 + (lane & 31)) * SROW + (lane >> 5) * 16, \
                                    smem_c + (buf_) * GSZ + 3 * APL + (wn * 64 + (lane & 31)) * SROW + (lane >> 5) * 16, acc);
    // body of tile u = t + j: request tile u + 4, split tile u + 1 (if it exists: the column sums
    // must see every tile exactly once), multiply tile u; one barrier
#define MARL_TS_BODY(j_, jn_)                                                              \
    MARL_TS_LOAD(j_, t + j_ + 4)                                                           \
    MARL_TS_STORE(jn_, (j_ + 1) & 1, t + j_ + 1 < T)                                       \
    MARL_TS_COMPUTE(j_ & 1)                                                                \
    split_pipeline<48, 5, 24, 8>();                                                        \
    lds_barrier();

    if (T > 0) {
        MARL_TS_LOAD(0, 0)
        MARL_TS_LOAD(1, 1)
        MARL_TS_LOAD(2, 2)
        MARL_TS_LOAD(3, 3)
        MARL_TS_STORE(0, 0, true)
        lds_barrier();
        for (int t = 0;; t += 4) {
            MARL_TS_BODY(0, 1)
            if (t + 1 >= T) break;
            MARL_TS_BODY(1, 2)
            if (t + 2 >= T) break;
            MARL_TS_BODY(2, 3)
            if (t + 3 >= T) break;
            MARL_TS_BODY(3, 0)
            if (t + 4 >= T) break;
        }
    }
    if (tail > 0) {  // the slab's last rows: clamped row addresses, A rows past the end zeroed
        const char* ap_ = abase + (size_t)T * SK * lda * 4;
        const char* bp_ = bbase + (size_t)T * SK * ldb * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = 4 * rb + q;
            const bool in = rr < tail;
            const uint32_t back = in ? 0u : (uint32_t)(rr - (tail - 1));
            const float m = in ? 1.f : 0.f;
            ra0[q] = *reinterpret_cast<const float4*>(ap_ + (aof[q] - back * (uint32_t)lda * 4u));
            rb0[q] = *reinterpret_cast<const float4*>(bp_ + (bof[q] - back * (uint32_t)ldb * 4u));
            ra0[q].x *= m;
            ra0[q].y *= m;
            ra0[q].z *= m;
            ra0[q].w *= m;
        }
        MARL_TS_STORE(0, 0, true)
        lds_barrier();
        MARL_TS_COMPUTE(0)
        lds_barrier();
    }
#undef MARL_TS_LOAD
#undef MARL_TS_STORE
#undef MARL_TS_COMPUTE
#undef MARL_TS_BODY

    if (do_csum) {  // the 8 threads rb = 0..7 of a column block staged the same 4 columns
        __syncthreads();
        float4* sh4 = reinterpret_cast<float4*>(smem_c);
        sh4[rb * 32 + cb] = cs;
        __syncthreads();
        if (tid < 32) {
            float4 t4 = sh4[tid];
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                const float4 u = sh4[q * 32 + tid];
                t4.x += u.x;
                t4.y += u.y;
                t4.z += u.z;
                t4.w += u.w;
            }
            float* co = csum + (size_t)bz * NI;
            const int c0 = i0 + tid * 4;
            if (c0 < NI) co[c0] = t4.x;
            if (c0 + 1 < NI) co[c0 + 1] = t4.y;
            if (c0 + 2 < NI) co[c0 + 2] = t4.z;
            if (c0 + 3 < NI) co[c0 + 3] = t4.w;
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
                const int row = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
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
SplitReg g_reg[kMaxSplitDesc];
int g_nreg = 0;
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
    bool pre = tune_get("split_pre", 1) != 0;
    for (int i = 0; i < batch.count && pre; ++i)
        for (int sg = 0; sg < batch.p[i].nseg && pre; ++sg) {
            GemmSeg& g = batch.p[i].seg[sg];
            pre = split_lookup(g, g.b3, g.kt3);
            // 32-bit byte offsets into the image
            if (pre && (int64_t)(LSTM ? 4 : 1) * batch.p[i].n * g.kt3 * 192 >= (1ll << 32)) pre = false;
        }
    constexpr int lds = 2 * 3 * plane_bytes(128) + 2 * 3 * plane_bytes(BN);  // two stages
    static bool raised = false;  // > 64 KiB of dynamic LDS: opt in once per instantiation
    if (!raised) {
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_split_kernel<BN, LSTM, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_split_kernel<BN, LSTM, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        raised = true;
    }
    if (pre)
        hipLaunchKernelGGL((gemm_nt_split_kernel<BN, LSTM, true>), grid, dim3(256), lds, st, batch);
    else
        hipLaunchKernelGGL((gemm_nt_split_kernel<BN, LSTM, false>), grid, dim3(256), lds, st, batch);
    return MARL_OK;
}

int launch_gemm_nt_split(const GemmBatch& batch, int max_m, int max_n, int64_t blocks128, hipStream_t st) {
    // 128-wide column tiles when they fill the chip (two workgroups per CU), else 128 x 64
    if (blocks128 >= tune_get("nts_min_blocks128", 384) && max_n >= 96) {
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
    constexpr int lds = 2 * 3 * 2 * plane_bytes(128);
    static bool raised = false;
    if (!raised) {
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_split_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_split_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        raised = true;
    }
    if (csum)
        hipLaunchKernelGGL(gemm_tn_split_kernel<true>, grid, dim3(256), lds, st, a, lda, b, ldb,
                           out, ldo, stride, ni, nj, rows, rows_per_split, csum, gx, gy, gz);
    else
        hipLaunchKernelGGL(gemm_tn_split_kernel<false>, grid, dim3(256), lds, st, a, lda, b, ldb,
                           out, ldo, stride, ni, nj, rows, rows_per_split, csum, gx, gy, gz);
    return MARL_OK;
}

}  // namespace marl
