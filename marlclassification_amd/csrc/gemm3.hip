// Image GEMMs ("g3"): fp32-accurate products whose operands are BOTH pre-split in HBM.
//
// gemm_split.hip splits fp32 operands into three bf16 terms while it stages them - 5-6 VALU
// instructions per MFMA, 48 staging registers, one LDS stage, two barriers per K tile: the 128 x 128
// two-barrier structure tops out near 1/3 of the bf16 matrix pipe (MI355X guide, "step-3 ceiling").
// Here the PRODUCER of every operand writes its k16 image (split.h: [row][k/16][plane][16] bf16,
// 6 bytes per element) and the product kernel is a pure bf16 kernel:
//   * tiles go HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane), no staging registers,
//     no VALU; the LDS stage is the image itself (96-byte rows; a (32-row block, K step) chunk is 3 KB
//     contiguous in memory = three whole instructions), the 16-byte piece index XOR-ed with bit 3 of
//     the row on the SOURCE address and on the fragment read (conflict-free ds_read_b128,
//     tools/lds_conflicts.py model);
//   * a ring of NST stages of one 16-deep K step each, ONE barrier per step, counted vmcnt: NST - 2
//     steps stay in flight across every barrier;
//   * the fragments of step s + 1 are read (other register set) while the 6 x TM x TN MFMAs of step
//     s issue - the matrix pipe only drains at the barrier itself;
//   * six products per fp32 product, smallest terms first, fp32 accumulation (gemm_split.hip).
// The LSTM form fuses the cell update in the epilogue like gemm_nt_split_kernel<.., LSTM> and ALSO
// writes the image of h' (through an LDS transposition, 96 contiguous bytes per lane), which is
// the A operand of the next step's launch and of the batched heads.
#include <stdlib.h>

#include "common.h"
#include "split.h"

namespace marl {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr;

namespace {

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// LDS-DMA by inline asm (round 6).  hipcc's waitcnt pass treats the transposing LDS read intrinsic
// (ds_read_b64_tr_b16) as an LDS access that may collide with a pending `buffer_load ... lds` and puts
// `s_waitcnt vmcnt(0)` in front of the first such read that follows an LDS-DMA builtin - in the row-contraction
// kernels that drained the whole ring once per K step, right behind the issue of the newest stage (every step paid
// a full memory round trip: 192 us against 132 us of matrix work, profiles/r06_clock_tn_ablate.csv; plain
// ds_read_b128 - the NT kernels - are not treated that way).  An asm-issued DMA is invisible to that pass; these
// kernels count vmcnt by hand anyway (wait_vm).  m0 = LDS byte address of the 1 KB this wave-instruction fills.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 dma_rsrc(const void* base) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return i32x4{(int)__builtin_amdgcn_readfirstlane((unsigned)a), (int)(__builtin_amdgcn_readfirstlane((unsigned)(a >> 32)) & 0xffff),
                 0x7fffffff, 0x00020000};
}
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds_addr, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop

}  // namespace

// BM x BN tile, WM x WN waves (wave tile (BM / WM) x (BN / WN) in 32 x 32 blocks), NST ring stages.
// LSTM: BN = 4 gates x 32 units, WN = 1 (a lane holds the four gates of its (row, unit) elements).
template <int BM, int BN, int WM, int WN, int NST, bool LSTM, int ABL>
__global__ __launch_bounds__(WM* WN * 64, 2) void gemm_nt3_kernel(const G3Batch batch) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass only needs the stub: the buffer-resource type is device-only)
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_BYTES = BM * kImgRowBytes, B_BYTES = BN * kImgRowBytes, ST_BYTES = A_BYTES + B_BYTES;
    constexpr int A_INS = A_BYTES / 1024, B_INS = B_BYTES / 1024;  // wave-instructions per step
    static_assert(A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0, "whole LDS-DMA instructions");
    // the A_INS + B_INS instructions of a step are dealt round robin: wave w issues t = w + NW i
    constexpr int T_INS = A_INS + B_INS, NI_LO = T_INS / NW, NI_HI = (T_INS + NW - 1) / NW;
    constexpr int I_EXTRA = T_INS % NW;  // waves [0, I_EXTRA) issue NI_HI instructions
    // LSTM tile = 4 gates x 32 units: all four gates in one wave (WN == 1: a lane holds the gates of its
    // elements) or one gate per wave (WN == 4, "gate split": the small-batch plans - a wave's MFMA chain is a
    // quarter as long and a 32-row tile spreads over the four SIMDs of its CU; the gates meet in LDS)
    static_assert(!LSTM || (BN == 128 && (WN == 1 || WN == 4)), "LSTM tile = 4 gates x 32 units");

    // ABL (perf diagnosis only): 0 product build, 1 every step fully waited for (SAFE), 2 no LDS-DMA in
    // the loop, 3 no MFMA, 4 block-major source addresses (1 KB contiguous per instruction; wrong data),
    // 5 no fragment reads in the loop
    constexpr bool SAFE = ABL == 1;
    extern __shared__ __attribute__((aligned(16))) char sm[];

    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (batch.xcd_map) xcd_tile(batch.gx, batch.gy, batch.count, bx, by, bz);
    const G3Prob& P = batch.p[bz];
    const int M = P.m;
    const int N = P.n;  // LSTM: hidden units (B has 4 N rows)
    const int n0 = by * (LSTM ? 32 : BN);
    const int m0 = bx * BM;
    if (m0 >= M || n0 >= N) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MARL_G3_ABLATE
    const bool probe = batch.clk != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0;
    if (probe) {
        batch.clk[0] = __builtin_readcyclecounter();
        batch.clk[1] = wall_clock64();
    }
#endif
    const int wm = wave / WN, wn = wave % WN;
    const bool extra_i = wave < I_EXTRA;
    const int S0 = P.seg[0].steps, S = S0 + (P.nseg > 1 ? P.seg[1].steps : 0);

    // ---- LDS-DMA source offsets of this lane: instruction j covers pieces q = 64 j + lane of the
    // stage image, piece q = (row q / 6, plane (q % 6) / 2, 16-byte half (q % 6) % 2)
    int voff[NI_HI];
    __amdgpu_buffer_rsrc_t rA, rB;
    auto set_seg = [&](int sg) {
        const G3Seg& g = P.seg[sg];
        const int blk = g.steps * kImgChunkBytes;  // bytes from one row block to the next
        // A: offsets relative to the tile's first row block (64-bit base), B: to the image start
        const int64_t a_first = (int64_t)g.a_row0 + m0;
        const int a_blk0 = (int)(a_first >> 5);
#pragma unroll
        for (int i = 0; i < NI_HI; ++i) {
            const int t = wave + NW * i;
            const bool isA = t < A_INS;
            const int q = (isA ? t : t - A_INS) * 64 + lane;
            int row = q / 6;
            const int w = q - row * 6;
            const int sw = (row >> 3) & 1;
            int grow;
            if (isA) {
                row = m0 + row < M ? row : M - 1 - m0;
                grow = (int)(a_first + row - ((int64_t)a_blk0 << 5));
            } else if (LSTM) {
                int unit = n0 + (row & 31);
                unit = unit < N ? unit : N - 1;
                grow = g.b_row0 + (row >> 5) * N + unit;
            } else {
                grow = n0 + row;
                grow = g.b_row0 + (grow < N ? grow : N - 1);
            }
            voff[i] = (grow >> 5) * blk + (grow & 31) * kImgRowBytes + (w >> 1) * 32 + (((w & 1) ^ sw) << 4);
        }
        rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g.a3) + (size_t)a_blk0 * blk, 0, 0x7fffffff, 0x00020000);
        rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g.b3), 0, 0x7fffffff, 0x00020000);
    };
    int gi = 0, islot = 0;  // next step to issue and its ring slot
    auto issue = [&]() {
        if (gi == S0) set_seg(1);
        const int soff = (gi >= S0 ? gi - S0 : gi) * kImgChunkBytes;
        char* base = sm + islot * ST_BYTES;
#pragma unroll
        for (int i = 0; i < NI_HI; ++i) {
            const int t = wave + NW * i;
            if (i == NI_LO && !extra_i) break;  // (NI_LO < NI_HI: only the first I_EXTRA waves)
            if (t < A_INS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)(base + t * 1024), 16, voff[i], soff, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr)(base + t * 1024), 16, voff[i], soff, 0, 0);
        }
        ++gi;
        islot = islot + 1 == NST ? 0 : islot + 1;
    };

    // fragment addresses: row = wave tile row + lane % 32, 16-byte half (lane / 32) ^ bit 3 of the row
    const int fsw = ((lane >> 5) ^ ((lane >> 3) & 1)) << 4;
    const char* lA = sm + (wm * (BM / WM) + (lane & 31)) * kImgRowBytes + fsw;
    const char* lB = sm + A_BYTES + (wn * (BN / WN) + (lane & 31)) * kImgRowBytes + fsw;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int NF = 3 * (TM + TN), NMMA = 6 * TM * TN;  // fragment reads / MFMAs of one step
    bf16x8 fa0[3][TM], fb0[3][TN], fa1[3][TM], fb1[3][TN];
#define G3_LOADF(fa_, fb_, slot_)                                                           \
    {                                                                                       \
        const char* pa_ = lA + (slot_) * ST_BYTES;                                          \
        const char* pb_ = lB + (slot_) * ST_BYTES;                                          \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                     \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                  \
                fa_[p][i] = *reinterpret_cast<const bf16x8*>(pa_ + i * 32 * kImgRowBytes + p * 32); \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                  \
                fb_[p][j] = *reinterpret_cast<const bf16x8*>(pb_ + j * 32 * kImgRowBytes + p * 32); \
        }                                                                                   \
    }
#define G3_P(fa_, fb_, pa_, pb_)                                                            \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                          \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[pa_][i], fb_[pb_][j], acc[i][j], 0, 0, 0);
#define G3_MMA(fa_, fb_)                                                                    \
    G3_P(fa_, fb_, 1, 1) G3_P(fa_, fb_, 0, 2) G3_P(fa_, fb_, 2, 0)                          \
    G3_P(fa_, fb_, 0, 1) G3_P(fa_, fb_, 1, 0) G3_P(fa_, fb_, 0, 0)
    // one K step; `cur` holds (is receiving) the fragments of step s.  MAIN: a step is left to issue and
    // NST - 2 steps stay in flight across the barrier; else (the last NST steps) everything is waited for.
#define G3_WAIT(k_)                                                                         \
    if (I_EXTRA > 0 && extra_i) wait_vm<(k_) * NI_HI>(); else wait_vm<(k_) * NI_LO>();
#define G3_BODY(fa_, fb_, fan_, fbn_, MAIN_)                                                \
    {                                                                                       \
        if (MAIN_ && !SAFE) { G3_WAIT(NST - 2) } else wait_vm<0>();                         \
        __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): my reads of step s are done */   \
        __builtin_amdgcn_s_barrier();                                                       \
        if (ABL == 2) { if (gi < S) ++gi; } else if (MAIN_ || gi < S) issue();              \
        rslot = rslot + 1 == NST ? 0 : rslot + 1;                                           \
        if (ABL != 5 && (MAIN_ || s + 1 < S)) G3_LOADF(fan_, fbn_, rslot)                   \
        if (ABL != 3 && ABL != 6) { G3_MMA(fa_, fb_) } else { G3_KEEP(fa_, fb_) }                       \
        if (MAIN_ && ABL != 3 && ABL != 5 && ABL != 6) { /* next step's fragment reads between the FIRST MFMAs */ \
            _Pragma("unroll") for (int q = 0; q < (NF + 1) / 2; ++q) {                      \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          \
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                          \
            }                                                                               \
            __builtin_amdgcn_sched_group_barrier(0x008, NMMA - (NF + 1) / 2, 0);            \
        }                                                                                   \
        ++s;                                                                                \
    }
#define G3_KEEP(fa_, fb_)                                                                   \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                         \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa_[p][i]));   \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb_[p][j]));   \
    }

    set_seg(0);
#pragma unroll
    for (int g = 0; g < NST; ++g)
        if (g < S) issue();
    if (!SAFE && S >= NST) { G3_WAIT(NST - 1) } else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    int s = 0, rslot = 0;
    G3_LOADF(fa0, fb0, 0)
    while (gi + 1 < S) {  // two steps per trip: both have a step to issue
        G3_BODY(fa0, fb0, fa1, fb1, true)
        G3_BODY(fa1, fb1, fa0, fb0, true)
    }
    while (s + 1 < S) {
        G3_BODY(fa0, fb0, fa1, fb1, false)
        G3_BODY(fa1, fb1, fa0, fb0, false)
    }
    if (s < S) G3_BODY(fa0, fb0, fa1, fb1, false)
#undef G3_BODY
#undef G3_KEEP
#undef G3_WAIT
#undef G3_MMA
#undef G3_P
#undef G3_LOADF

    // ---- epilogue.  acc[i][j][r] is C[row(r), col], col = lane & 31, row(r) = (r & 3) + 8 * (r >> 2) +
    // 4 * (lane >> 5): one column and 16 rows per lane, i.e. 4-byte stores - and a store costs its ISSUE
    // slot (~70 cycles per wave-instruction and CU), not its bytes (MI355X guide, T21).  Every 32 x 32
    // block therefore goes through a wave-private LDS panel and leaves as FOUR 16-byte stores per lane
    // (8 rows x 128 contiguous bytes per instruction) instead of sixteen 4-byte ones.
    constexpr int HLD = 36;  // floats per panel row (16-byte aligned rows, conflict-free enough)
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
    float* hp = reinterpret_cast<float*>(sm) + wave * 32 * HLD;
    const int tr = lane >> 3, tc = (lane & 7) * 4;  // read-back: rows tr + 8 q, columns tc .. tc + 3
    __syncthreads();  // every wave is done with the ring before the panels overwrite it
#define G3_PANEL_PUT(val_)                                                                  \
    _Pragma("unroll") for (int r = 0; r < 16; ++r)                                          \
        hp[((r & 3) + 8 * (r >> 2) + row_h) * HLD + col_l] = (val_);
#define G3_PANEL_GET(q_) (*reinterpret_cast<const float4*>(hp + (tr + 8 * (q_)) * HLD + tc))
    if constexpr (!LSTM) {
        const bool vec = (P.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(P.c) & 15) == 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cb = n0 + wn * (BN / WN) + j * 32;  // first column of the block
                if (cb >= N) continue;
                const int rb = m0 + wm * (BM / WM) + i * 32;
                G3_PANEL_PUT(acc[i][j][r])
                wait_lgkm0();
                const int col = cb + tc;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (P.bias) {
                    bv.x = col < N ? P.bias[col] : 0.f;
                    bv.y = col + 1 < N ? P.bias[col + 1] : 0.f;
                    bv.z = col + 2 < N ? P.bias[col + 2] : 0.f;
                    bv.w = col + 3 < N ? P.bias[col + 3] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = rb + tr + 8 * q;
                    float4 v = G3_PANEL_GET(q);
                    v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                    if (row >= M || col >= N) continue;
                    float* cp = P.c + (size_t)row * P.ldc + col;
                    if (vec && col + 3 < N) {
                        if (P.accumulate) {
                            const float4 o = *reinterpret_cast<const float4*>(cp);
                            v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
                        }
                        *reinterpret_cast<float4*>(cp) = v;
                    } else {
                        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (col + u < N) cp[u] = P.accumulate ? cp[u] + e[u] : e[u];
                    }
                }
                wait_lgkm0();  // (the panel is rewritten by the next block)
            }
    } else if constexpr (WN == 4) {
        // ---- gate split: acc[0][0] of wave (wm, g = wn) = pre-activations of gate g, rows m0 + 32 wm ..,
        // units n0 ..  Same arithmetic, in the same order, as the one-wave form below (bit-identical results).
        const int g = wn;
        const int unit = n0 + col_l;
        const int uc = unit < N ? unit : N - 1;
        const float bg_ = P.bias[g * N + uc];
        if (g == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] = tanh_fast(acc[0][0][r] + bg_);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] = sigmoid_acc(acc[0][0][r] + bg_);
        }
        G3_PANEL_PUT(acc[0][0][r])  // panel `wave` = (row group wm, gate g)
        __syncthreads();
        // the 256 threads of a row group share its 32 x 32 (row, unit) elements: four consecutive units each
        const int t = wn * 64 + lane, er = t >> 3, ec = (t & 7) * 4;
        const float* gpn = reinterpret_cast<const float*>(sm) + (wm * 4) * 32 * HLD + er * HLD + ec;
        const float4 vi = *reinterpret_cast<const float4*>(gpn), vf = *reinterpret_cast<const float4*>(gpn + 32 * HLD),
                     vg = *reinterpret_cast<const float4*>(gpn + 2 * 32 * HLD),
                     vo = *reinterpret_cast<const float4*>(gpn + 3 * 32 * HLD);
        const int row = m0 + wm * 32 + er, col = n0 + ec;
        const bool rok = row < M;
        const int rowc = rok ? row : M - 1;
        const bool svec = (P.ld_state & 3) == 0;
        const bool col_ok = col < ((N + 3) & ~3);
        float cp[4];
        if (svec && col_ok) {
            const float4 c4 = *reinterpret_cast<const float4*>(P.c_prev + (size_t)rowc * P.ld_state + col);
            cp[0] = c4.x, cp[1] = c4.y, cp[2] = c4.z, cp[3] = c4.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) cp[e] = P.c_prev[(size_t)rowc * P.ld_state + (col + e < N ? col + e : N - 1)];
        }
        const float gi4[4] = {vi.x, vi.y, vi.z, vi.w}, gf4[4] = {vf.x, vf.y, vf.z, vf.w},
                    gg4[4] = {vg.x, vg.y, vg.z, vg.w}, go4[4] = {vo.x, vo.y, vo.z, vo.w};
        float cn[4], hn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool uok = col + e < N;
            const float c_ = gf4[e] * cp[e] + gi4[e] * gg4[e];
            cn[e] = uok ? c_ : 0.f;
            hn[e] = uok ? go4[e] * tanh_fast(c_) : 0.f;
        }
        // h' of the row group -> its own LDS panel (behind the gate panels) for the image stores below
        float* hpn = reinterpret_cast<float*>(sm) + (WM * 4 + wm) * 32 * HLD;
        *reinterpret_cast<float4*>(hpn + er * HLD + ec) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        if (rok) {
            float* cdst = P.c_next + (size_t)row * P.ld_state + col;
            float* hdst = P.h_next + (size_t)row * P.ld_state + col;
            if (svec) {
                if (col_ok) {
                    *reinterpret_cast<float4*>(cdst) = make_float4(cn[0], cn[1], cn[2], cn[3]);
                    *reinterpret_cast<float4*>(hdst) = make_float4(hn[0], hn[1], hn[2], hn[3]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < N) cdst[e] = cn[e], hdst[e] = hn[e];
            }
            if (P.gates) {
                const bool gvec = (N & 3) == 0 && (P.ld_gates & 3) == 0;
                float* gd = P.gates + (size_t)row * P.ld_gates + col;
                if (gvec) {
                    if (col_ok) {
                        *reinterpret_cast<float4*>(gd) = vi;
                        *reinterpret_cast<float4*>(gd + N) = vf;
                        *reinterpret_cast<float4*>(gd + 2 * N) = vg;
                        *reinterpret_cast<float4*>(gd + 3 * N) = vo;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < N) gd[e] = gi4[e], gd[N + e] = gf4[e], gd[2 * N + e] = gg4[e], gd[3 * N + e] = go4[e];
                }
            }
        }
        if (P.h3) {
            __syncthreads();
            if (wn == 0) {  // one image step of one row per lane: 96 contiguous bytes
                const int lr = lane >> 1, u0 = (lane & 1) * 16;
                const int irow = m0 + wm * 32 + lr;
                float v[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t4 = *reinterpret_cast<const float4*>(hpn + lr * HLD + u0 + 4 * q);
                    v[4 * q] = t4.x, v[4 * q + 1] = t4.y, v[4 * q + 2] = t4.z, v[4 * q + 3] = t4.w;
                }
                if (irow < M && n0 + u0 < ((N + 15) & ~15))
                    img_store16(P.h3 + img_off((int64_t)P.h3_row0 + irow, (n0 + u0) >> 4, P.h3_steps), v);
            }
        }
    } else {
        // (TM == 1, TN == 4: acc[0][g] = gate g of unit n0 + col_l, 16 rows of the wave's 32)
        const int unit = n0 + col_l;
        const bool uok = unit < N;
        const int uc = uok ? unit : N - 1;
        const float bi = P.bias[uc], bf = P.bias[N + uc], bg = P.bias[2 * N + uc], bo = P.bias[3 * N + uc];
        const int rb = m0 + wm * 32;
        float cprev[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row_ = rb + (r & 3) + 8 * (r >> 2) + row_h;
            row_ = row_ < M ? row_ : M - 1;
            cprev[r] = P.c_prev[(size_t)row_ * P.ld_state + uc];
        }
        // activated gates in place of the pre-activations, then c' and h' (zeros past the last unit: the
        // padding columns of the state buffers and of the image stay zero)
        float cn[16], hn[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float gi_ = sigmoid_acc(acc[0][0][r] + bi);
            const float gf = sigmoid_acc(acc[0][LSTM ? 1 : 0][r] + bf);
            const float gg = tanh_fast(acc[0][LSTM ? 2 : 0][r] + bg);
            const float go = sigmoid_acc(acc[0][LSTM ? 3 : 0][r] + bo);
            acc[0][0][r] = gi_;
            acc[0][LSTM ? 1 : 0][r] = gf;
            acc[0][LSTM ? 2 : 0][r] = gg;
            acc[0][LSTM ? 3 : 0][r] = go;
            const float c_ = gf * cprev[r] + gi_ * gg;
            cn[r] = uok ? c_ : 0.f;
            hn[r] = uok ? go * tanh_fast(c_) : 0.f;
        }
        const int col = n0 + tc;
        const bool col_ok = col < ((N + 3) & ~3);  // state rows are pad4(N) wide (zeros behind N)
        // one [32 rows][32 units] block -> dst[row * ld + col0 + unit]; vec: 16-byte stores allowed
#define G3_PANEL_STORE(dst_, ld_, col0_, vec_)                                              \
        {                                                                                   \
            wait_lgkm0();                                                                   \
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                 \
                const int row = rb + tr + 8 * q;                                            \
                const float4 v = G3_PANEL_GET(q);                                           \
                if (row >= M) continue;                                                     \
                float* cp = (dst_) + (size_t)row * (ld_) + (col0_) + col;                   \
                if (vec_) {                                                                 \
                    if (col_ok) *reinterpret_cast<float4*>(cp) = v;                         \
                } else {                                                                    \
                    if (col < N) cp[0] = v.x;                                               \
                    if (col + 1 < N) cp[1] = v.y;                                           \
                    if (col + 2 < N) cp[2] = v.z;                                           \
                    if (col + 3 < N) cp[3] = v.w;                                           \
                }                                                                           \
            }                                                                               \
            wait_lgkm0();                                                                   \
        }
        const bool svec = (P.ld_state & 3) == 0;
        G3_PANEL_PUT(cn[r])
        G3_PANEL_STORE(P.c_next, P.ld_state, 0, svec)
        if (P.gates) {
            const bool gvec = (N & 3) == 0 && (P.ld_gates & 3) == 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                G3_PANEL_PUT(acc[0][LSTM ? g : 0][r])
                G3_PANEL_STORE(P.gates, P.ld_gates, g * N, gvec)
            }
        }
        G3_PANEL_PUT(hn[r])
        G3_PANEL_STORE(P.h_next, P.ld_state, 0, svec)
        if (P.h3) {
            // (the h' panel is still in LDS) lane = (row lane / 2, 16 units (lane & 1) * 16 ..) = one image
            // step of that row: 96 contiguous bytes
            const int lr = lane >> 1, u0 = (lane & 1) * 16;
            const int row = rb + lr;
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(hp + lr * HLD + u0 + 4 * q);
                v[4 * q] = t.x;
                v[4 * q + 1] = t.y;
                v[4 * q + 2] = t.z;
                v[4 * q + 3] = t.w;
            }
            if (row < M && n0 + u0 < ((N + 15) & ~15))
                img_store16(P.h3 + img_off((int64_t)P.h3_row0 + row, (n0 + u0) >> 4, P.h3_steps), v);
        }
#undef G3_PANEL_STORE
    }
#undef G3_PANEL_PUT
#undef G3_PANEL_GET
#ifdef MARL_G3_ABLATE
    if (probe) {
        batch.clk[2] = __builtin_readcyclecounter();
        batch.clk[3] = wall_clock64();
    }
#endif
#endif
}

// ---------------------------------------------------------------------------
// Phase-pipelined NT / fused LSTM cell (round 6).  Same operands, same six products in the same order and the same
// epilogue arithmetic as gemm_nt3_kernel - bit-identical results - on ONE 256-row workgroup per CU whose K step is
// cut into phases (one per 32-row block of A, the wider side of the wave tile): the fragments of phase ph + 1 and of
// step s + 1 are read, and the LDS-DMA pieces of step s + NST - 1 are issued, BETWEEN the MFMAs of the phase before
// they are needed, so a wave's matrix chain only breaks at the two barriers of a step (B0 at the step boundary:
// slot s - 1 is free; B1 mid step: the DMA of step s + 1 has landed).  Measured on the row-contraction twin of this
// loop with memory out of the way (32 workgroups, tools/tn_pipe_probe.py): 100.5 us against 98 us for MFMAs and
// barriers alone (round 5's loop: 142 us).  LDS-DMA is issued by inline asm (dma16): the order inside the pinned
// phase is fixed with sched_barrier(0) - an asm statement has no scheduling class.
// LSTM: tile = 4 gates x 32 units, WN = 2: wave (wm, wn) holds gates 2 wn, 2 wn + 1 of its rows; the gates meet in
// LDS panels for the cell update (the arithmetic of the one-wave form, element by element).
// ---------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int NST, bool LSTM>
__global__ __launch_bounds__(WM* WN * 64, 1) void gemm_nt3p_kernel(const G3Batch batch) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NP = TM, NIN = TN;  // phases walk the A blocks; B's fragments stay for the step
    static_assert(TM >= TN && (NP == 2 || NP == 4) && NST >= 3, "phase pipeline: 2 or 4 A blocks per wave, >= 3 stages");
    static_assert(!LSTM || (BN == 128 && WN * TN == 4), "LSTM tile = 4 gates x 32 units");
    constexpr int A_BYTES = BM * kImgRowBytes, B_BYTES = BN * kImgRowBytes, ST_BYTES = A_BYTES + B_BYTES;
    constexpr int A_INS = A_BYTES / 1024, B_INS = B_BYTES / 1024;
    static_assert(A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0 && A_INS % NW == 0, "whole LDS-DMA instructions, A's first");
    constexpr int T_INS = A_INS + B_INS, NI_LO = T_INS / NW, NI_HI = (T_INS + NW - 1) / NW;
    constexpr int I_EXTRA = T_INS % NW;
    extern __shared__ __attribute__((aligned(16))) char sm[];

    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (batch.xcd_map) xcd_tile(batch.gx, batch.gy, batch.count, bx, by, bz);
    const G3Prob& P = batch.p[bz];
    const int M = P.m;
    const int N = P.n;  // LSTM: hidden units (B has 4 N rows)
    const int n0 = by * (LSTM ? 32 : BN);
    const int m0 = bx * BM;
    if (m0 >= M || n0 >= N) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool extra_i = wave < I_EXTRA;
    const int S0 = P.seg[0].steps, S = S0 + (P.nseg > 1 ? P.seg[1].steps : 0);
    int safe_i = batch.safe;
    asm volatile("" : "+s"(safe_i));
    const bool safe = safe_i == 1;

    // ---- LDS-DMA source offsets (as gemm_nt3_kernel): instruction t = wave + NW i covers pieces q = 64 t' + lane
    int voff[NI_HI];
    i32x4 rA, rB;
    auto set_seg = [&](int sg) {
        const G3Seg& g = P.seg[sg];
        const int blk = g.steps * kImgChunkBytes;
        const int64_t a_first = (int64_t)g.a_row0 + m0;
        const int a_blk0 = (int)(a_first >> 5);
#pragma unroll
        for (int i = 0; i < NI_HI; ++i) {
            const int t = wave + NW * i;
            const bool isA = i < A_INS / NW;
            const int q = (isA ? t : t - A_INS) * 64 + lane;
            int row = q / 6;
            const int w = q - row * 6;
            const int sw = (row >> 3) & 1;
            int grow;
            if (isA) {
                row = m0 + row < M ? row : M - 1 - m0;
                grow = (int)(a_first + row - ((int64_t)a_blk0 << 5));
            } else if (LSTM) {
                int unit = n0 + (row & 31);
                unit = unit < N ? unit : N - 1;
                grow = g.b_row0 + (row >> 5) * N + unit;
            } else {
                grow = n0 + row;
                grow = g.b_row0 + (grow < N ? grow : N - 1);
            }
            voff[i] = (grow >> 5) * blk + (grow & 31) * kImgRowBytes + (w >> 1) * 32 + (((w & 1) ^ sw) << 4);
        }
        rA = dma_rsrc(g.a3 + (size_t)a_blk0 * blk);
        rB = dma_rsrc(g.b3);
    };
    const unsigned sm_lds = (unsigned)reinterpret_cast<unsigned long long>((lds_ptr)sm);
    int gi = 0, islot = 0;
    auto issue_begin = [&]() {
        if (gi == S0) set_seg(1);
    };
#ifdef MARL_G3_ABLATE
    const bool nodma = batch.safe == 2, nomma = batch.safe == 3;  // perf diagnosis (wrong results)
#endif
    auto issue_one = [&](int i) {
        if (i == NI_LO && !extra_i) return;
#ifdef MARL_G3_ABLATE
        if (nodma) return;
#endif
        dma16(i < A_INS / NW ? rA : rB, sm_lds + islot * ST_BYTES + (wave + NW * i) * 1024, voff[i],
              (gi >= S0 ? gi - S0 : gi) * kImgChunkBytes);
    };
    auto issue_done = [&]() {
        ++gi;
        islot = islot + 1 == NST ? 0 : islot + 1;
    };
    auto issue = [&]() {
        issue_begin();
#pragma unroll
        for (int i = 0; i < NI_HI; ++i) issue_one(i);
        issue_done();
    };

    const int fsw = ((lane >> 5) ^ ((lane >> 3) & 1)) << 4;
    const char* lA = sm + (wm * (BM / WM) + (lane & 31)) * kImgRowBytes + fsw;
    const char* lB = sm + A_BYTES + (wn * (BN / WN) + (lane & 31)) * kImgRowBytes + fsw;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    bf16x8 fi0[3][NIN], fi1[3][NIN], fo0[3], fo1[3];  // B of step s / s + 1; A block of phase ph / ph + 1
#define NP_LOAD_IN(dst_, so_, p_lo_, p_hi_)                                                           \
    _Pragma("unroll") for (int p = (p_lo_); p < (p_hi_); ++p)                                         \
        _Pragma("unroll") for (int b = 0; b < NIN; ++b)                                                \
            dst_[p][b] = *reinterpret_cast<const bf16x8*>(lB + (so_) + b * 32 * kImgRowBytes + p * 32);
#define NP_LOAD_OUT1(dst_, so_, ob_, p_) dst_[p_] = *reinterpret_cast<const bf16x8*>(lA + (so_) + (ob_) * 32 * kImgRowBytes + (p_) * 32);
#define NP_LOAD_OUT(dst_, so_, ob_) _Pragma("unroll") for (int p = 0; p < 3; ++p) NP_LOAD_OUT1(dst_, so_, ob_, p)
#define NP_P(in_, out_, ob_, pa_, pb_)                                                                \
    _Pragma("unroll") for (int b = 0; b < NIN; ++b)                                                   \
        acc[ob_][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(out_[pa_], in_[pb_][b], acc[ob_][b], 0, 0, 0);
#define NP_MMA(in_, out_, ob_)                                                                        \
    NP_P(in_, out_, ob_, 1, 1) NP_P(in_, out_, ob_, 0, 2) NP_P(in_, out_, ob_, 2, 0)                  \
    NP_P(in_, out_, ob_, 0, 1) NP_P(in_, out_, ob_, 1, 0) NP_P(in_, out_, ob_, 0, 0)
#define NP_SCHED(nr_)                                                                                 \
    {                                                                                                 \
        _Pragma("unroll") for (int q = 0; q < (nr_); ++q) {                                           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
        }                                                                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * NIN - (nr_), 0);                              \
    }
#define NP_WAIT(k_)                                                                                   \
    if (I_EXTRA > 0 && extra_i) wait_vm<(k_) * NI_HI>(); else wait_vm<(k_) * NI_LO>();
#define NP_PIN() __builtin_amdgcn_sched_barrier(0);
    // pieces q, q + 6, ... of the step behind product q of the pinned phase
#define NP_PIECES(q_)                                                                                 \
    _Pragma("unroll") for (int i = (q_); i < NI_HI; i += 6) issue_one(i);
#define NP_PHASE0_PINNED(ina_)                                                                        \
    {                                                                                                 \
        issue_begin();                                                                                \
        NP_P(ina_, fo0, 0, 1, 1) NP_PIN() NP_PIECES(0) NP_LOAD_OUT1(fo1, so, 1, 0) NP_PIN()           \
        NP_P(ina_, fo0, 0, 0, 2) NP_PIN() NP_PIECES(1) NP_PIN()                                       \
        NP_P(ina_, fo0, 0, 2, 0) NP_PIN() NP_PIECES(2) NP_LOAD_OUT1(fo1, so, 1, 1) NP_PIN()           \
        NP_P(ina_, fo0, 0, 0, 1) NP_PIN() NP_PIECES(3) NP_PIN()                                       \
        NP_P(ina_, fo0, 0, 1, 0) NP_PIN() NP_PIECES(4) NP_LOAD_OUT1(fo1, so, 1, 2) NP_PIN()           \
        NP_P(ina_, fo0, 0, 0, 0) NP_PIN() NP_PIECES(5) NP_PIN()                                       \
        issue_done();                                                                                 \
    }
#define NP_PHASE(ina_, inb_, ph_, MAIN_)                                                              \
    {                                                                                                 \
        constexpr int ph = (ph_);                                                                     \
        constexpr bool second = ph >= NP / 2, last = ph == NP - 1;                                    \
        constexpr int h2 = NP / 2, pl = (NP == 2 || ph == h2) ? 0 : 2, phi = NP == 2 ? 3 : (ph == h2 ? 2 : 3); \
        const bool nxt = MAIN_ || s + 1 < S;                                                          \
        if constexpr (!last) {                                                                        \
            if constexpr (ph & 1) { NP_LOAD_OUT(fo0, so, ph + 1) } else { NP_LOAD_OUT(fo1, so, ph + 1) } \
        } else if (nxt) { NP_LOAD_OUT(fo0, sn, 0) }                                                   \
        if constexpr (second) { if (nxt) { NP_LOAD_IN(inb_, sn, pl, phi) } }                          \
        if constexpr (ph & 1) { NP_MMA(ina_, fo1, ph) } else { NP_MMA(ina_, fo0, ph) }                \
        if (MAIN_) NP_SCHED(3 + (second ? NIN * (phi - pl) : 0))                                      \
    }
#define NP_BODY(ina_, inb_, MAIN_)                                                                    \
    {                                                                                                 \
        __builtin_amdgcn_s_waitcnt(0xc07f);                                                           \
        __builtin_amdgcn_s_barrier(); /* B0 */                                                        \
        const int rn = rs + 1 == NST ? 0 : rs + 1;                                                    \
        const int so = rs * ST_BYTES, sn = rn * ST_BYTES;                                             \
        if (MAIN_) { NP_PHASE0_PINNED(ina_) }                                                         \
        else {                                                                                        \
            if (gi < S) issue();                                                                      \
            NP_PHASE(ina_, inb_, 0, MAIN_)                                                            \
        }                                                                                             \
        if constexpr (NP == 4) NP_PHASE(ina_, inb_, 1, MAIN_)                                         \
        if (MAIN_ && !safe) { NP_WAIT(NST - 2) } else wait_vm<0>();                                   \
        __builtin_amdgcn_s_barrier(); /* B1 */                                                        \
        if constexpr (NP == 4) { NP_PHASE(ina_, inb_, 2, MAIN_) NP_PHASE(ina_, inb_, 3, MAIN_) }      \
        else NP_PHASE(ina_, inb_, 1, MAIN_)                                                           \
        rs = rn;                                                                                      \
        ++s;                                                                                          \
    }
    set_seg(0);
    {
#pragma unroll
        for (int g = 0; g < NST - 1; ++g)
            if (g < S) issue();
        if (!safe && S >= NST - 1) { NP_WAIT(NST - 2) } else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        int s = 0, rs = 0;
        NP_LOAD_IN(fi0, 0, 0, 3)
        NP_LOAD_OUT(fo0, 0, 0)
        while (gi + 1 < S) {
            NP_BODY(fi0, fi1, true)
            NP_BODY(fi1, fi0, true)
        }
        while (s + 1 < S) {
            NP_BODY(fi0, fi1, false)
            NP_BODY(fi1, fi0, false)
        }
        if (s < S) NP_BODY(fi0, fi1, false)
    }
#undef NP_BODY
#undef NP_PHASE
#undef NP_PHASE0_PINNED
#undef NP_PIECES
#undef NP_PIN
#undef NP_WAIT
#undef NP_SCHED
#undef NP_MMA
#undef NP_P
#undef NP_LOAD_OUT
#undef NP_LOAD_OUT1
#undef NP_LOAD_IN

    // ---- epilogue (wave-private LDS panels, 16-byte stores: see gemm_nt3_kernel)
    constexpr int HLD = 36;
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
    float* hp = reinterpret_cast<float*>(sm) + wave * 32 * HLD;
    const int tr = lane >> 3, tc = (lane & 7) * 4;
    __syncthreads();
    if constexpr (!LSTM) {
        const bool vec = (P.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(P.c) & 15) == 0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cb = n0 + wn * (BN / WN) + j * 32;
                if (cb >= N) continue;
                const int rb = m0 + wm * (BM / WM) + i * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) hp[((r & 3) + 8 * (r >> 2) + row_h) * HLD + col_l] = acc[i][j][r];
                wait_lgkm0();
                const int col = cb + tc;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (P.bias) {
                    bv.x = col < N ? P.bias[col] : 0.f;
                    bv.y = col + 1 < N ? P.bias[col + 1] : 0.f;
                    bv.z = col + 2 < N ? P.bias[col + 2] : 0.f;
                    bv.w = col + 3 < N ? P.bias[col + 3] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = rb + tr + 8 * q;
                    float4 v = *reinterpret_cast<const float4*>(hp + (tr + 8 * q) * HLD + tc);
                    v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                    if (row >= M || col >= N) continue;
                    float* cp = P.c + (size_t)row * P.ldc + col;
                    if (vec && col + 3 < N) {
                        if (P.accumulate) {
                            const float4 o = *reinterpret_cast<const float4*>(cp);
                            v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
                        }
                        *reinterpret_cast<float4*>(cp) = v;
                    } else {
                        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (col + u < N) cp[u] = P.accumulate ? cp[u] + e[u] : e[u];
                    }
                }
                wait_lgkm0();
            }
    } else {
        // ---- gates meet in LDS: per 32-row block i of the wave tiles, panel (wm, gate) <- activated gate; then the
        // WN x 64 threads of row group wm share the block's 32 x 32 (row, unit) elements, eight consecutive units each
        static_assert(WN == 2 && TN == 2, "LSTM epilogue: two gates per wave");
        constexpr int PANEL = 32 * HLD;                                // floats
        float* gpan = reinterpret_cast<float*>(sm) + wm * 5 * PANEL;   // 4 gate panels + the h' panel of this row group
        static_assert((size_t)WM * 5 * PANEL * 4 <= (size_t)NST * ST_BYTES, "gate panels fit the ring");
        const int unit = n0 + col_l;
        const int uc = unit < N ? unit : N - 1;
        const float bg0 = P.bias[(2 * wn) * N + uc], bg1 = P.bias[(2 * wn + 1) * N + uc];
        const int t = wn * 64 + lane, er = t >> 2, ec = (t & 3) * 8;  // this thread's element group in the block
        const bool svec = (P.ld_state & 3) == 0;
        const bool gvec = (N & 3) == 0 && (P.ld_gates & 3) == 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i > 0) __syncthreads();  // the panels of block i - 1 have been read
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int g = 2 * wn + j;
                const float bg_ = j ? bg1 : bg0;
                float* pp = gpan + g * PANEL;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float x = acc[i][j][r] + bg_;
                    pp[((r & 3) + 8 * (r >> 2) + row_h) * HLD + col_l] = g == 2 ? tanh_fast(x) : sigmoid_acc(x);
                }
            }
            __syncthreads();
            const int row = m0 + wm * (BM / WM) + i * 32 + er;
            const bool rok = row < M;
            const int rowc = rok ? row : M - 1;
            float* hpn = gpan + 4 * PANEL;
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {  // two groups of four consecutive units
                const int ecq = ec + 4 * hq, col = n0 + ecq;
                const float* gpn = gpan + er * HLD + ecq;
                const float4 vi = *reinterpret_cast<const float4*>(gpn), vf = *reinterpret_cast<const float4*>(gpn + PANEL),
                             vg = *reinterpret_cast<const float4*>(gpn + 2 * PANEL), vo = *reinterpret_cast<const float4*>(gpn + 3 * PANEL);
                const bool col_ok = col < ((N + 3) & ~3);
                float cp[4];
                if (svec && col_ok) {
                    const float4 c4 = *reinterpret_cast<const float4*>(P.c_prev + (size_t)rowc * P.ld_state + col);
                    cp[0] = c4.x, cp[1] = c4.y, cp[2] = c4.z, cp[3] = c4.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) cp[e] = P.c_prev[(size_t)rowc * P.ld_state + (col + e < N ? col + e : N - 1)];
                }
                const float gi4[4] = {vi.x, vi.y, vi.z, vi.w}, gf4[4] = {vf.x, vf.y, vf.z, vf.w},
                            gg4[4] = {vg.x, vg.y, vg.z, vg.w}, go4[4] = {vo.x, vo.y, vo.z, vo.w};
                float cn[4], hn[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool uok = col + e < N;
                    const float c_ = gf4[e] * cp[e] + gi4[e] * gg4[e];
                    cn[e] = uok ? c_ : 0.f;
                    hn[e] = uok ? go4[e] * tanh_fast(c_) : 0.f;
                }
                *reinterpret_cast<float4*>(hpn + er * HLD + ecq) = make_float4(hn[0], hn[1], hn[2], hn[3]);
                if (rok) {
                    float* cdst = P.c_next + (size_t)row * P.ld_state + col;
                    float* hdst = P.h_next + (size_t)row * P.ld_state + col;
                    if (svec) {
                        if (col_ok) {
                            *reinterpret_cast<float4*>(cdst) = make_float4(cn[0], cn[1], cn[2], cn[3]);
                            *reinterpret_cast<float4*>(hdst) = make_float4(hn[0], hn[1], hn[2], hn[3]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (col + e < N) cdst[e] = cn[e], hdst[e] = hn[e];
                    }
                    if (P.gates) {
                        float* gd = P.gates + (size_t)row * P.ld_gates + col;
                        if (gvec) {
                            if (col_ok) {
                                *reinterpret_cast<float4*>(gd) = vi;
                                *reinterpret_cast<float4*>(gd + N) = vf;
                                *reinterpret_cast<float4*>(gd + 2 * N) = vg;
                                *reinterpret_cast<float4*>(gd + 3 * N) = vo;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (col + e < N) gd[e] = gi4[e], gd[N + e] = gf4[e], gd[2 * N + e] = gg4[e], gd[3 * N + e] = go4[e];
                        }
                    }
                }
            }
            if (P.h3) {
                __syncthreads();
                if (wn == 0) {  // one image step of one row per lane: 96 contiguous bytes
                    const int lr = lane >> 1, u0 = (lane & 1) * 16;
                    const int irow = m0 + wm * (BM / WM) + i * 32 + lr;
                    float v[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 t4 = *reinterpret_cast<const float4*>(hpn + lr * HLD + u0 + 4 * q);
                        v[4 * q] = t4.x, v[4 * q + 1] = t4.y, v[4 * q + 2] = t4.z, v[4 * q + 3] = t4.w;
                    }
                    if (irow < M && n0 + u0 < ((N + 15) & ~15))
                        img_store16(P.h3 + img_off((int64_t)P.h3_row0 + irow, (n0 + u0) >> 4, P.h3_steps), v);
                }
            }
        }
    }
#endif
}

// ---------------------------------------------------------------------------
// TN: C[NI,NJ] = sum_r A[r,i] * B[r,j] over the row slab of this workgroup (weight gradients), both
// operands k16 images whose ROWS are the contraction index.  A stage = 16 rows x (BI + BJ) columns:
// per 16-column step of an operand 1536 contiguous bytes (half a row block), so the LDS stage is
// [column step][16 row slots][3 planes][16 columns] and the MFMA fragments (8 consecutive ROWS of one
// column per lane) come out of it with the transposing read ds_read_b64_tr_b16: lanes 4q..4q+3 of a
// 16-lane group supply row q of a [4 rows][16 columns] block, lane j receives column j.  Row slot s of
// an ODD column step holds row s ^ 4 (on the DMA source address and on the read): the two 16-lane
// groups of a read then hit disjoint banks.  Same ring / barrier / MFMA structure as the NT kernel.
// csum (nullable, column tile 0 only): column sums of A from three more MFMAs per row block against a
// fragment of ones.
// ---------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr;
#ifndef MARL_TN_PIN
#define MARL_TN_PIN 1
#endif
[[maybe_unused]] constexpr bool kTnPin = MARL_TN_PIN != 0;  // pipelined loop: LDS-DMA pieces pinned between the MFMAs of phase 0

// DB: fragments double-buffered in registers (step s + 1 read while step s multiplies, NST - 2 steps in
// flight); !DB (the 256 x 256 tile: 128 accumulator registers leave room for one fragment set): the
// fragments of step s are read right behind the barrier, NST - 1 steps in flight.
// PIPE (round 6): the phase-pipelined step - see the block comment at its loop below.
template <int BI, int BJ, int WI, int WJ, int NST, bool DB, int ABL = 0, bool CS = true, bool PIPE = false>
__device__ __forceinline__ void gemm_tn3_body(const G3TnArgs& P, char* sm, unsigned bx, int j0, unsigned bz, bool first_col_tile) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = WI * WJ;
    constexpr int TM = BI / WI / 32, TN = BJ / WJ / 32;
    constexpr int CS_BYTES = 16 * kImgRowBytes;  // one column step of a stage: 16 rows x 96 bytes
    constexpr int A_BYTES = (BI / 16) * CS_BYTES, B_BYTES = (BJ / 16) * CS_BYTES, ST_BYTES = A_BYTES + B_BYTES;
    constexpr int A_INS = A_BYTES / 1024, B_INS = B_BYTES / 1024;
    static_assert(A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0, "whole LDS-DMA instructions");
    constexpr int T_INS = A_INS + B_INS, NI_LO = T_INS / NW, NI_HI = (T_INS + NW - 1) / NW;
    constexpr int I_EXTRA = T_INS % NW;
    constexpr int NF = 6 * (TM + TN), NMMA = 6 * TM * TN;

    const int i0 = bx * BI;
    const int64_t r_begin = (int64_t)bz * P.rows_per_split;
    int64_t r_end = r_begin + P.rows_per_split;
    if (r_end > P.rows) r_end = P.rows;
    const int S = r_end > r_begin ? (int)((r_end - r_begin) >> 4) : 0;  // stages of 16 rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WJ, wn = wave % WJ;
    const bool extra_i = wave < I_EXTRA;

    // LDS-DMA source offsets: instruction t covers pieces q = 64 t' + lane of its operand's stage,
    // piece q = (column step q / 96, row slot (q % 96) / 6, 16-byte piece (q % 96) % 6)
    int voff[NI_HI];
#pragma unroll
    for (int i = 0; i < NI_HI; ++i) {
        const int t = wave + NW * i;
        const bool isA = t < A_INS;
        const int q = (isA ? t : t - A_INS) * 64 + lane;
        const int csl = q / 96, w = q - csl * 96, slot = w / 6, pw = w - slot * 6;
        const int steps = isA ? P.a_steps : P.b_steps;
        int cs = (isA ? i0 : j0) / 16 + csl;
        cs = cs < steps ? cs : steps - 1;  // (columns past the operand's width: never stored)
        voff[i] = cs * kImgChunkBytes + (slot ^ ((csl & 1) << 2)) * kImgRowBytes + pw * 16;
    }
    const int a_blk = P.a_steps * kImgChunkBytes, b_blk = P.b_steps * kImgChunkBytes;
    const i32x4 rA = dma_rsrc(P.a3 + (size_t)((P.a_row0 + r_begin) >> 5) * a_blk);
    const i32x4 rB = dma_rsrc(P.b3 + (size_t)((P.b_row0 + r_begin) >> 5) * b_blk);
    const unsigned sm_lds = (unsigned)reinterpret_cast<unsigned long long>((lds_ptr)sm);
    // wave w issues instructions t = w + NW i: with A_INS a multiple of NW, instruction i of EVERY wave belongs to the
    // same operand (no per-instruction branch on the resource)
    constexpr bool A_STATIC = A_INS % NW == 0;
    int gi = 0, islot = 0;
    // instruction i of step gi (the pinned phase of the pipelined loop places them one by one between MFMAs)
    auto issue_one = [&](int i) {
        if (i == NI_LO && !extra_i) return;
        const int half = (gi & 1) * CS_BYTES;  // rows 0..15 / 16..31 of the row block gi / 2
        const bool isA = A_STATIC ? i < A_INS / NW : wave + NW * i < A_INS;
        dma16(isA ? rA : rB, sm_lds + islot * ST_BYTES + (wave + NW * i) * 1024, voff[i],
              (gi >> 1) * (isA ? a_blk : b_blk) + half);
    };
    auto issue_done = [&]() {
        ++gi;
        islot = islot + 1 == NST ? 0 : islot + 1;
    };
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < NI_HI; ++i) issue_one(i);
        issue_done();
    };
    int safe_i = P.safe;  // (pinned in a scalar register: hipcc re-loaded the kernel argument inside the K loop,
    asm volatile("" : "+s"(safe_i));  //  and its s_waitcnt lgkmcnt(0) also waited for the LDS reads in flight)
    const bool safe = safe_i != 0;

    // fragment addresses (see the header): 16-lane group g = lane / 16 -> column sub-step g & 1, row
    // group g / 2; lane j = 4 q + piece; the two reads of a fragment take rows 4 h + q, h = 0, 1
    const int g16 = lane >> 4, sub = g16 & 1, kg = g16 >> 1, qrow = (lane >> 2) & 3, piece = lane & 3;
    const int f0 = sub * CS_BYTES + (8 * kg + 4 * sub + qrow) * kImgRowBytes + piece * 8;        // h = 0
    const int f1 = sub * CS_BYTES + (8 * kg + 4 * (1 - sub) + qrow) * kImgRowBytes + piece * 8;  // h = 1
    const char* lA0 = sm + wm * (BI / WI / 16) * CS_BYTES + f0;
    const char* lA1 = sm + wm * (BI / WI / 16) * CS_BYTES + f1;
    const char* lB0 = sm + A_BYTES + wn * (BJ / WJ / 16) * CS_BYTES + f0;
    const char* lB1 = sm + A_BYTES + wn * (BJ / WJ / 16) * CS_BYTES + f1;

    f32x16 acc[TM][TN], accs[CS ? TM : 1];  // (CS = false: no column sums of A - their accumulators are not even allocated)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[CS ? i : 0][r] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    const bool do_csum = CS && P.csum != nullptr && first_col_tile;
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (short)0x3f80;  // bf16 1.0

    bf16x8 fa0[PIPE ? 1 : 3][PIPE ? 1 : TM], fb0[PIPE ? 1 : 3][PIPE ? 1 : TN], fa1[DB && !PIPE ? 3 : 1][DB && !PIPE ? TM : 1],
        fb1[DB && !PIPE ? 3 : 1][DB && !PIPE ? TN : 1];
#define G3T_FRAG(p0_, p1_, off_)                                                                     \
    __builtin_shufflevector(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)((p0_) + (off_))),    \
                            __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)((p1_) + (off_))), 0, 1, 2, 3, 4, 5, 6, 7)
#define G3T_LOADF(fa_, fb_, slot_)                                                                   \
    {                                                                                                \
        const int so_ = (slot_) * ST_BYTES;                                                          \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                              \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
                fa_[p][i] = G3T_FRAG(lA0, lA1, so_ + i * 2 * CS_BYTES + p * 32);                     \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                           \
                fb_[p][j] = G3T_FRAG(lB0, lB1, so_ + j * 2 * CS_BYTES + p * 32);                     \
        }                                                                                            \
    }
#define G3T_P(fa_, fb_, pa_, pb_)                                                                    \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                   \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                               \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[pa_][i], fb_[pb_][j], acc[i][j], 0, 0, 0);
#define G3T_MMA(fa_, fb_)                                                                            \
    G3T_P(fa_, fb_, 1, 1) G3T_P(fa_, fb_, 0, 2) G3T_P(fa_, fb_, 2, 0)                                \
    G3T_P(fa_, fb_, 0, 1) G3T_P(fa_, fb_, 1, 0) G3T_P(fa_, fb_, 0, 0)                                \
    if (do_csum) {                                                                                   \
        _Pragma("unroll") for (int p = 2; p >= 0; --p)                                               \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                           \
                accs[CS ? i : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa_[p][i], ones, accs[CS ? i : 0], 0, 0, 0); \
    }
#define G3T_WAIT(k_)                                                                                 \
    if (I_EXTRA > 0 && extra_i) wait_vm<(k_) * NI_HI>(); else wait_vm<(k_) * NI_LO>();
#define G3T_BODY(fa_, fb_, fan_, fbn_, MAIN_)                                                        \
    {                                                                                                \
        if (MAIN_ && !safe) { G3T_WAIT(NST - 2) } else wait_vm<0>();                               \
        __builtin_amdgcn_s_waitcnt(0xc07f);                                                          \
        __builtin_amdgcn_s_barrier();                                                                \
        if (MAIN_ || gi < S) issue();                                                                \
        rslot = rslot + 1 == NST ? 0 : rslot + 1;                                                    \
        if (MAIN_ || s + 1 < S) G3T_LOADF(fan_, fbn_, rslot)                                         \
        G3T_MMA(fa_, fb_)                                                                            \
        if (MAIN_) {                                                                                 \
            _Pragma("unroll") for (int q = 0; q < NF / 4; ++q) {                                     \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                   \
            }                                                                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, NMMA - NF / 4, 0);                           \
        }                                                                                            \
        ++s;                                                                                         \
    }

    if constexpr (PIPE) {
        // ---- phase-pipelined step (round 6).  The one-fragment-set loop below runs
        // [barrier] [issue DMA] [read ALL fragments of the step] [48 MFMAs]: the eight waves of the workgroup pass
        // the barrier together, so every SIMD's matrix pipe idles while both of its waves issue LDS-DMA and wait
        // for 36 transposing reads (measured: 192 us against 132 us for the same launch with neither, 256
        // workgroups - profiles/r06_clock_tn_ablate.csv).  Here a step is cut into NP phases, one per 32-column
        // block of the WIDER side of the wave tile ("outer" operand, one block = 3 fragments live at a time, two
        // buffers); the narrower side's fragments ("inner", 3 x NIN) stay for the whole step and have a second set
        // that receives step s + 1.  Every read and every LDS-DMA instruction sits between MFMAs of the phase
        // BEFORE its data is needed (sched_group_barrier pins the interleave), so the pipe only drains at the two
        // barriers of a step:
        //   B0 (step boundary): all reads of slot s - 1 are done          -> DMA of step s + NST - 1 goes there
        //   B1 (mid step):      the DMA of step s + 1 has landed (vmcnt)  -> its fragments are read in the 2nd half
        // NST - 2 steps of DMA stay in flight across B1.  Per accumulator the six products of a step keep the
        // order of the loop below: results are bit-identical (tested against the fully-waited build).
        constexpr bool OJ = TN >= TM;
        constexpr int NP = OJ ? TN : TM, NIN = OJ ? TM : TN;
        static_assert(NP % 2 == 0 && NST >= 3 && (!CS || OJ), "phase pipeline: even phase count, >= 3 stages");
        const char* lI0 = OJ ? lA0 : lB0;
        const char* lI1 = OJ ? lA1 : lB1;
        const char* lO0 = OJ ? lB0 : lA0;
        const char* lO1 = OJ ? lB1 : lA1;
        bf16x8 fi0[3][NIN], fi1[3][NIN], fo0[3], fo1[3];
#define TP_LOAD_IN(dst_, so_, p_lo_, p_hi_)                                                           \
        _Pragma("unroll") for (int p = (p_lo_); p < (p_hi_); ++p)                                     \
            _Pragma("unroll") for (int b = 0; b < NIN; ++b)                                            \
                dst_[p][b] = G3T_FRAG(lI0, lI1, (so_) + b * 2 * CS_BYTES + p * 32);
#define TP_LOAD_OUT(dst_, so_, ob_)                                                                   \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) dst_[p] = G3T_FRAG(lO0, lO1, (so_) + (ob_) * 2 * CS_BYTES + p * 32);
#define TP_P(in_, out_, ob_, pa_, pb_)                                                                \
        _Pragma("unroll") for (int b = 0; b < NIN; ++b) {                                             \
            if constexpr (OJ)                                                                         \
                acc[b][ob_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(in_[pa_][b], out_[pb_], acc[b][ob_], 0, 0, 0); \
            else                                                                                      \
                acc[ob_][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(out_[pa_], in_[pb_][b], acc[ob_][b], 0, 0, 0); \
        }
#define TP_MMA(in_, out_, ob_)                                                                        \
        TP_P(in_, out_, ob_, 1, 1) TP_P(in_, out_, ob_, 0, 2) TP_P(in_, out_, ob_, 2, 0)              \
        TP_P(in_, out_, ob_, 0, 1) TP_P(in_, out_, ob_, 1, 0) TP_P(in_, out_, ob_, 0, 0)
        // interleave of one phase: `nr_` transposing reads two per MFMA, `nv_` LDS-DMA pieces one per MFMA behind
        // them, the remaining MFMAs of the phase (6 NIN in all, + `xm_` column-sum MFMAs) back to back
#define TP_SCHED(nr_, nv_, xm_)                                                                       \
        {                                                                                             \
            _Pragma("unroll") for (int q = 0; q < ((nr_) + 1) / 2; ++q) {                             \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
            }                                                                                         \
            _Pragma("unroll") for (int q = 0; q < (nv_); ++q) {                                       \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                                    \
            }                                                                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * NIN + (xm_) - ((nr_) + 1) / 2 - (nv_), 0); \
        }
        // phase ph of step s: its outer block sits in buffer ph & 1; the other buffer receives block ph + 1 (the
        // last phase: block 0 of step s + 1); the 2nd-half phases also fill the other inner set
#define TP_PHASE(ina_, inb_, ph_, MAIN_)                                                              \
        {                                                                                             \
            constexpr int ph = (ph_);                                                                 \
            constexpr bool second = ph >= NP / 2, last = ph == NP - 1;                                \
            constexpr int h2 = NP / 2, pl = (NP == 2 || ph == h2) ? 0 : 2, phi = NP == 2 ? 3 : (ph == h2 ? 2 : 3); /* inner planes read in this phase */ \
            const bool nxt = MAIN_ || s + 1 < S;                                                      \
            if constexpr (!last) {                                                                    \
                if constexpr (ph & 1) { TP_LOAD_OUT(fo0, so, ph + 1) } else { TP_LOAD_OUT(fo1, so, ph + 1) } \
            } else if (nxt) { TP_LOAD_OUT(fo0, sn, 0) }                                               \
            if constexpr (second) { if (nxt) { TP_LOAD_IN(inb_, sn, pl, phi) } }                      \
            if constexpr (ph & 1) { TP_MMA(ina_, fo1, ph) } else { TP_MMA(ina_, fo0, ph) }            \
            if constexpr (CS && ph == 0) { /* (CS bodies always form the column sums: the caller picks CS by do_csum) */ \
                _Pragma("unroll") for (int p = 2; p >= 0; --p)                                        \
                    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                    \
                        accs[CS ? i : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ina_[p][CS ? i : 0], ones, accs[CS ? i : 0], 0, 0, 0); \
            }                                                                                         \
            if (MAIN_) TP_SCHED(6 + (second ? 2 * NIN * (phi - pl) : 0), ph == 0 ? NI_HI : 0, CS && ph == 0 ? 3 * TM : 0) \
        }
        // phase 0 of a main-loop step, hand-ordered: the six products of the phase (NIN MFMAs each) with ONE LDS-DMA
        // piece of step s + NST - 1 behind each of the first NI_HI and one fragment of outer block 1 behind every
        // second - an asm-issued DMA has no scheduling class, so sched_barrier(0) pins the order (the pieces cost
        // 60-185 cycles of issue each, MI355X_MICROARCH.md: behind a barrier, all waves at once, they idled the pipe)
#define TP_PIN() __builtin_amdgcn_sched_barrier(0);
#define TP_PHASE0_PINNED(ina_)                                                                        \
        {                                                                                             \
            TP_P(ina_, fo0, 0, 1, 1) TP_PIN() issue_one(0); fo1[0] = G3T_FRAG(lO0, lO1, so + 2 * CS_BYTES); TP_PIN() \
            TP_P(ina_, fo0, 0, 0, 2) TP_PIN() issue_one(1); TP_PIN()                                  \
            TP_P(ina_, fo0, 0, 2, 0) TP_PIN() issue_one(2); fo1[1] = G3T_FRAG(lO0, lO1, so + 2 * CS_BYTES + 32); TP_PIN() \
            TP_P(ina_, fo0, 0, 0, 1) TP_PIN() issue_one(3); TP_PIN()                                  \
            TP_P(ina_, fo0, 0, 1, 0) TP_PIN() if constexpr (NI_HI > 4) issue_one(4); fo1[2] = G3T_FRAG(lO0, lO1, so + 2 * CS_BYTES + 64); TP_PIN() \
            TP_P(ina_, fo0, 0, 0, 0) TP_PIN() if constexpr (NI_HI > 5) issue_one(5); TP_PIN()        \
            _Pragma("unroll") for (int i = 6; i < NI_HI; ++i) issue_one(i);                           \
            issue_done();                                                                             \
            if constexpr (CS) {                                                                       \
                _Pragma("unroll") for (int p = 2; p >= 0; --p)                                        \
                    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                    \
                        accs[CS ? i : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ina_[p][CS ? i : 0], ones, accs[CS ? i : 0], 0, 0, 0); \
            }                                                                                         \
            TP_PIN()                                                                                  \
        }
#define TP_BODY(ina_, inb_, MAIN_)                                                                    \
        {                                                                                             \
            __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0): my reads of slot s - 1 (and of step s's first fragments) */ \
            __builtin_amdgcn_s_barrier();       /* B0 */                                              \
            const int rn = rs + 1 == NST ? 0 : rs + 1;                                                \
            const int so = rs * ST_BYTES, sn = rn * ST_BYTES;                                         \
            if (MAIN_ && kTnPin) { TP_PHASE0_PINNED(ina_) }                                           \
            else {                                                                                    \
                if (MAIN_ || gi < S) issue();                                                         \
                TP_PHASE(ina_, inb_, 0, MAIN_)                                                        \
            }                                                                                         \
            if constexpr (NP == 4) TP_PHASE(ina_, inb_, 1, MAIN_)                                     \
            if (MAIN_ && !safe) { G3T_WAIT(NST - 2) } else wait_vm<0>();                            \
            __builtin_amdgcn_s_barrier();       /* B1 */                                              \
            if constexpr (NP == 4) { TP_PHASE(ina_, inb_, 2, MAIN_) TP_PHASE(ina_, inb_, 3, MAIN_) }  \
            else TP_PHASE(ina_, inb_, 1, MAIN_)                                                       \
            rs = rn;                                                                                  \
            ++s;                                                                                      \
        }
        static_assert(NP == 2 || NP == 4, "phase pipeline: 2 or 4 phases");
        if (S > 0) {
#pragma unroll
            for (int g = 0; g < NST - 1; ++g)
                if (g < S) issue();
            if (!safe && S >= NST - 1) { G3T_WAIT(NST - 2) } else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            int s = 0, rs = 0;
            TP_LOAD_IN(fi0, 0, 0, 3)
            TP_LOAD_OUT(fo0, 0, 0)
            while (gi + 1 < S) {  // two steps per trip (the inner sets alternate): both have a step to issue
                TP_BODY(fi0, fi1, true)
                TP_BODY(fi1, fi0, true)
            }
            while (s + 1 < S) {
                TP_BODY(fi0, fi1, false)
                TP_BODY(fi1, fi0, false)
            }
            if (s < S) TP_BODY(fi0, fi1, false)
        }
#undef TP_BODY
#undef TP_PHASE0_PINNED
#undef TP_PIN
#undef TP_PHASE
#undef TP_SCHED
#undef TP_MMA
#undef TP_P
#undef TP_LOAD_OUT
#undef TP_LOAD_IN
    } else
    if (S > 0 && DB) {
#pragma unroll
        for (int g = 0; g < NST; ++g)
            if (g < S) issue();
        if (!safe && S >= NST) { G3T_WAIT(NST - 1) } else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        int s = 0, rslot = 0;
        G3T_LOADF(fa0, fb0, 0)
        while (gi + 1 < S) {
            G3T_BODY(fa0, fb0, fa1, fb1, true)
            G3T_BODY(fa1, fb1, fa0, fb0, true)
        }
        while (s + 1 < S) {
            G3T_BODY(fa0, fb0, fa1, fb1, false)
            G3T_BODY(fa1, fb1, fa0, fb0, false)
        }
        if (s < S) G3T_BODY(fa0, fb0, fa1, fb1, false)
    } else if (S > 0) {
        // one fragment set: [wait step s] [barrier] [issue step s + NST - 1] [read step s] [multiply]
#pragma unroll
        for (int g = 0; g < NST - 1; ++g)
            if (g < S) issue();
        int rslot = 0;
        // ABL (perf diagnosis, tools/tn_ablate.py; wrong results): 1 no LDS-DMA in the loop, 2 no MFMA, 3 no
        // fragment reads, 4 neither DMA nor fragment reads
        if (ABL == 3 || ABL == 4) G3T_LOADF(fa0, fb0, 0)
        for (int s = 0; s < S; ++s) {
            if (!safe && gi < S) { G3T_WAIT(NST - 2) } else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (ABL == 1 || ABL == 4) { if (gi < S) ++gi; } else if (gi < S) issue();
            if (ABL != 3 && ABL != 4) G3T_LOADF(fa0, fb0, rslot)
            if (ABL != 2) { G3T_MMA(fa0, fb0) } else {
                _Pragma("unroll") for (int p = 0; p < 3; ++p) {
                    _Pragma("unroll") for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa0[p][i]));
                    _Pragma("unroll") for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb0[p][j]));
                }
            }
            rslot = rslot + 1 == NST ? 0 : rslot + 1;
        }
    }
#undef G3T_BODY
#undef G3T_WAIT
#undef G3T_MMA
#undef G3T_P
#undef G3T_LOADF
#undef G3T_FRAG

    // ---- epilogue: the slab of this split, through the wave-private LDS panels (16-byte stores)
    constexpr int HLD = 36;
    const int col_l = lane & 31, row_h = 4 * (lane >> 5);
    float* hp = reinterpret_cast<float*>(sm) + wave * 32 * HLD;
    const int tr = lane >> 3, tc = (lane & 7) * 4;
    __syncthreads();
    float* o = P.out + (size_t)bz * P.out_split_stride;
    const bool vec = (P.ldo & 3) == 0 && (P.out_split_stride & 3) == 0 && (reinterpret_cast<uintptr_t>(P.out) & 15) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rb = i0 + wm * (BI / WI) + i * 32;
        if (do_csum && wn == 0 && col_l == 0) {  // every column of accs[i] holds the same sums
            float* co = P.csum + (size_t)bz * P.ni;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb + (r & 3) + 8 * (r >> 2) + row_h;
                if (row < P.ni) co[row] = accs[CS ? i : 0][r];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cb = j0 + wn * (BJ / WJ) + j * 32;
            if (cb >= P.nj || rb >= P.ni) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) hp[((r & 3) + 8 * (r >> 2) + row_h) * HLD + col_l] = acc[i][j][r];
            wait_lgkm0();
            const int col = cb + tc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = rb + tr + 8 * q;
                const float4 v = *reinterpret_cast<const float4*>(hp + (tr + 8 * q) * HLD + tc);
                if (row >= P.ni || col >= P.nj) continue;
                float* cp = o + (size_t)row * P.ldo + col;
                if (vec && col + 3 < P.nj) {
                    *reinterpret_cast<float4*>(cp) = v;
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (col + u < P.nj) cp[u] = e[u];
                }
            }
            wait_lgkm0();
        }
    }
#endif
}

template <int BI, int BJ, int WI, int WJ, int NST, bool DB, int ABL = 0, bool PIPE = false>
__global__ __launch_bounds__(WI* WJ * 64, 2) void gemm_tn3_kernel(const G3TnArgs P) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (P.gx > 0) xcd_tile(P.gx, P.gy, P.gz, bx, by, bz);
    if constexpr (PIPE) {  // (the pipelined body forms the column sums unconditionally when it is compiled with them)
        if (P.csum != nullptr && by == 0)
            gemm_tn3_body<BI, BJ, WI, WJ, NST, DB, ABL, true, true>(P, sm, bx, P.j_first + (int)by * BJ, bz, true);
        else
            gemm_tn3_body<BI, BJ, WI, WJ, NST, DB, ABL, false, true>(P, sm, bx, P.j_first + (int)by * BJ, bz, false);
    } else
        gemm_tn3_body<BI, BJ, WI, WJ, NST, DB, ABL, true, false>(P, sm, bx, P.j_first + (int)by * BJ, bz, by == 0);
}

// "column passes": a workgroup owns a 256-column tile of A and a row slab and walks ALL columns of B in
// 256-wide passes with a 128-wide last pass when at most 128 columns remain - no half-empty 256 x 256
// tile (NJ = 368 -> 256 + 128, NJ = 624 -> 256 + 256 + 128), equal work per workgroup.  A's slab is
// re-read by every pass (L2 / MALL).
template <bool PIPE>
__global__ __launch_bounds__(512, 2) void gemm_tn3_passes_kernel(const G3TnArgs P) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    unsigned bx = blockIdx.x, by = 0, bz = blockIdx.z;
    if (P.gx > 0) xcd_tile(P.gx, 1, P.gz, bx, by, bz);
    for (int j0 = 0; j0 < P.nj; j0 += 256) {
        if (j0 > 0) __syncthreads();  // the epilogue panels of the previous pass live in the ring
        const bool cs = P.csum != nullptr && j0 == 0;
        if (P.nj - j0 > 128) {
            if (PIPE && !cs) gemm_tn3_body<256, 256, 4, 2, 3, false, 0, false, PIPE>(P, sm, bx, j0, bz, false);
            else gemm_tn3_body<256, 256, 4, 2, 3, false, 0, true, PIPE>(P, sm, bx, j0, bz, j0 == 0);
        } else {
            if (PIPE && !cs) gemm_tn3_body<256, 128, 4, 2, 3, false, 0, false, PIPE>(P, sm, bx, j0, bz, false);
            else gemm_tn3_body<256, 128, 4, 2, 3, false, 0, true, PIPE>(P, sm, bx, j0, bz, j0 == 0);
        }
    }
}

// One LSTM cell's two weight gradients in one launch (common.h, G3TnCell): workgroup = (row slab, 256-column
// tile of G, column tile k of [U | H]), k fastest in the XCD-contiguous order - the nt workgroups that read the
// same slab of G are dispatched next to each other on one XCD and walk it together.
template <bool PIPE>
__global__ __launch_bounds__(512, 2) void gemm_tn3_cell_kernel(const G3TnCell P) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    unsigned bx, by, bz;
    if (P.teams) {
        // equal-work teams: per (row slab, PAIR of 256-column tiles of G): the 256-wide column tiles of both G tiles
        // and ONE 512 (both G tiles) x 128 workgroup for the narrow last column tile - 2 n256 + 1 workgroups of the
        // same length that walk the slab in step (a 256 x 128 tile would run twice as fast, away from its team)
        const unsigned per = 2 * P.n256 + 1, T = (P.gx >> 1) * per * P.gz, L = blockIdx.x;
        const unsigned chunk = T >> 3, rem = T & 7, xcd = L & 7, slot = L >> 3;
        unsigned q = (xcd < rem ? xcd * (chunk + 1) : rem * (chunk + 1) + (xcd - rem) * chunk) + slot;
        const unsigned m = q % per;
        q /= per;
        const unsigned pair = q % (P.gx >> 1);
        bz = q / (P.gx >> 1);
        if (m == 2 * (unsigned)P.n256) {
            const G3TnArgs& Q = P.t[P.n256];
            gemm_tn3_body<512, 128, 4, 2, 2, false, 0, false>(Q, sm, pair, Q.j_first, bz, false);
            return;
        }
        bx = 2 * pair + m / P.n256;
        by = m % P.n256;
    } else {
        xcd_tile(P.gx, P.nt, P.gz, bx, by, bz);
    }
    const G3TnArgs& Q = P.t[by];
    if ((int)by < P.n256) {
        if (PIPE && Q.csum == nullptr) gemm_tn3_body<256, 256, 4, 2, 3, false, 0, false, PIPE>(Q, sm, bx, Q.j_first, bz, false);
        else gemm_tn3_body<256, 256, 4, 2, 3, false, 0, true, PIPE>(Q, sm, bx, Q.j_first, bz, Q.csum != nullptr);
    } else
        gemm_tn3_body<256, 128, 4, 2, 3, false, 0, false, PIPE>(Q, sm, bx, Q.j_first, bz, false);
}

// ---------------------------------------------------------------------------
// fp32 matrix -> k16 image (weights after every optimiser step; operands of the kernel-level API)
// ---------------------------------------------------------------------------
__global__ void image_kernel(const ImgBatch B) {
    const ImgDesc& d = B.d[blockIdx.y];
    const int steps = (d.k + 15) >> 4;
    const int64_t rows32 = (d.rows + 31) & ~(int64_t)31;
    const int64_t tot = rows32 * steps * 2;  // one thread per 8 consecutive k
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < tot;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / (steps * 2);
        const int k0 = (int)(idx - r * steps * 2) * 8;
        float v[8];
        if (r >= d.rows) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = 0.f;
        } else {
            const float* s = d.src + r * d.ld + k0;
            if (k0 + 8 <= d.k && (d.ld & 3) == 0) {
                const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
                v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = k0 + q < d.k ? s[q] : 0.f;
            }
        }
        img_store8(static_cast<char*>(d.dst) + img_off(r, k0 >> 4, steps), k0, v);
    }
}

int launch_images(const ImgBatch& b, hipStream_t st) {
    if (b.count <= 0) return MARL_OK;
    if (b.count > kMaxImgDesc) return MARL_EINVAL;
    int64_t mx = 0;
    for (int i = 0; i < b.count; ++i) {
        const int64_t t = ((b.d[i].rows + 31) & ~(int64_t)31) * img_steps(b.d[i].k) * 2;
        mx = t > mx ? t : mx;
    }
    int64_t gx = cdiv(mx, 256);
    gx = gx > 2048 ? 2048 : (gx < 1 ? 1 : gx);
    hipLaunchKernelGGL(image_kernel, dim3((unsigned)gx, (unsigned)b.count), dim3(256), 0, st, b);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
G3Prob g3_prob(const void* a3, int a_row0, const void* b3, int b_row0, int k, float* c, int ldc, int m, int n,
               const float* bias, int accumulate) {
    G3Prob p{};
    p.seg[0] = G3Seg{static_cast<const char*>(a3), static_cast<const char*>(b3), a_row0, b_row0, img_steps(k)};
    p.nseg = 1;
    p.m = m;
    p.n = n;
    p.c = c;
    p.ldc = ldc;
    p.bias = bias;
    p.accumulate = accumulate;
    return p;
}
void g3_add_seg(G3Prob& p, const void* a3, int a_row0, const void* b3, int b_row0, int k) {
    p.seg[1] = G3Seg{static_cast<const char*>(a3), static_cast<const char*>(b3), a_row0, b_row0, img_steps(k)};
    p.nseg = 2;
}

template <int BM, int BN, int WM, int WN, int NST, bool LSTM>
static int launch_g3_variant(G3Batch batch, int max_m, int max_n, hipStream_t st) {
    dim3 grid((unsigned)cdiv(max_m, BM), (unsigned)cdiv(max_n, LSTM ? 32 : BN), (unsigned)batch.count);
    batch.gx = (int)grid.x;
    batch.gy = (int)grid.y;
    // XCD-contiguous tile order: single products always; the two-cell LSTM launch by knob (each half of the XCDs then
    // streams ONE cell's 3.8 MB of weight images - they fit its L2 - at the price of fetching A's row tiles twice)
    // (round 6: also several products of ONE shape - the two batched heads: their three 128-column tiles per row tile
    // were spread over the XCDs and A left HBM three times; knob nt_xcd_multi)
    bool same = true;
    for (int i = 1; i < batch.count; ++i) same = same && batch.p[i].m == batch.p[0].m && batch.p[i].n == batch.p[0].n;
    batch.xcd_map = tune_get("nt_xcd", 1) &&
                    !LSTM && (batch.count == 1 || same);  // (the two-cell LSTM launch keeps the dispatch order: measured twice)
    if (batch.xcd_map) grid = dim3(grid.x * grid.y * grid.z);
    batch.safe = tune_get("g3_safe", 0);
    constexpr size_t lds = (size_t)NST * (BM + BN) * kImgRowBytes;
    static_assert(lds <= 160 * 1024, "LDS ring");
    static_assert((size_t)WM * WN * 32 * 36 * 4 <= lds, "epilogue panels fit the ring");
    static_assert(!(LSTM && WN == 4) || (size_t)WM * 5 * 32 * 36 * 4 <= lds, "gate panels + h' panels fit the ring");
    const int abl = batch.safe;
#ifdef MARL_G3_ABLATE
    static long long* d_clk = nullptr;
    if (!d_clk) MARL_HIP_CHECK(hipMalloc(&d_clk, 4 * sizeof(long long)));
    batch.clk = tune_get("g3_clk", 0) ? d_clk : nullptr;
#endif
#define G3_LAUNCH(ABL_)                                                                                   \
    {                                                                                                     \
        auto kern = gemm_nt3_kernel<BM, BN, WM, WN, NST, LSTM, ABL_>;                                     \
        static bool raised = false;                                                                       \
        if (lds > 64 * 1024 && !raised) {                                                                 \
            MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                      \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    \
            raised = true;                                                                                \
        }                                                                                                 \
        hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, batch);                               \
    }
    if (abl == 0) G3_LAUNCH(0)
    else if (abl == 1) G3_LAUNCH(1)
#ifdef MARL_G3_ABLATE
    else if (abl == 2) G3_LAUNCH(2)
    else if (abl == 3) G3_LAUNCH(3)
    else if (abl == 4) G3_LAUNCH(4)
    else if (abl == 5) G3_LAUNCH(5)
    else if (abl == 6) G3_LAUNCH(6)
#endif
    else return MARL_EINVAL;
#undef G3_LAUNCH
#ifdef MARL_G3_ABLATE
    if (batch.clk) {
        long long h[4];
        MARL_HIP_CHECK(hipMemcpy(h, batch.clk, sizeof(h), hipMemcpyDeviceToHost));
        fprintf(stderr, "[g3 clk] block 0: %.1f us, %.0f MHz shader clock\n", (h[3] - h[1]) * 0.01,
                (double)(h[2] - h[0]) / ((h[3] - h[1]) * 0.01));
    }
#endif
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// the phase-pipelined kernels (gemm_nt3p_kernel): one 256-row workgroup per CU
template <int BM, int BN, int WM, int WN, int NST, bool LSTM>
static int launch_g3p_variant(G3Batch batch, int max_m, int max_n, hipStream_t st) {
    dim3 grid((unsigned)cdiv(max_m, BM), (unsigned)cdiv(max_n, LSTM ? 32 : BN), (unsigned)batch.count);
    batch.gx = (int)grid.x;
    batch.gy = (int)grid.y;
    bool same = true;
    for (int i = 1; i < batch.count; ++i) same = same && batch.p[i].m == batch.p[0].m && batch.p[i].n == batch.p[0].n;
    batch.xcd_map = tune_get("nt_xcd", 1) && !LSTM && (batch.count == 1 || same);
    if (batch.xcd_map) grid = dim3(grid.x * grid.y * grid.z);
    batch.safe = tune_get("g3_safe", 0);
#ifndef MARL_G3_ABLATE
    batch.safe = batch.safe != 0;
#endif
    constexpr size_t lds = (size_t)NST * (BM + BN) * kImgRowBytes;
    static_assert(lds <= 160 * 1024, "LDS ring");
    static_assert((size_t)WM * WN * 32 * 36 * 4 <= lds, "epilogue panels fit the ring");
    auto kern = gemm_nt3p_kernel<BM, BN, WM, WN, NST, LSTM>;
    static bool raised = false;
    if (!raised) {
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, batch);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

static int check_g3(const G3Prob& p, bool lstm) {
    if (p.m <= 0 || p.n <= 0 || p.nseg < 1 || p.nseg > 2) {
        set_error("g3: empty problem m=%d n=%d nseg=%d", p.m, p.n, p.nseg);
        return MARL_EINVAL;
    }
    for (int s = 0; s < p.nseg; ++s) {
        const G3Seg& g = p.seg[s];
        if (!g.a3 || !g.b3 || g.steps <= 0 || g.a_row0 < 0 || g.b_row0 < 0 ||
            (reinterpret_cast<uintptr_t>(g.a3) & 15) || (reinterpret_cast<uintptr_t>(g.b3) & 15)) {
            set_error("g3: bad operand image (seg %d)", s);
            return MARL_EINVAL;
        }
        // 32-bit byte offsets: within the row blocks of a tile for A, over all rows for B
        const int64_t brows = g.b_row0 + (lstm ? 4 * (int64_t)p.n : (int64_t)p.n) + 32;
        if ((brows >> 5) * g.steps * kImgChunkBytes >= (1ll << 31) || 10ll * g.steps * kImgChunkBytes >= (1ll << 31)) {
            set_error("g3: operand image of seg %d too large for 32-bit offsets", s);
            return MARL_ELIMIT;
        }
    }
    return MARL_OK;
}

// variant: 0 = automatic, else (perf experiments) 1: 256 x 128 / 8 waves, 2: 128 x 128 / 4 waves,
// 3: 128 x 64 / 4 waves, 7: 64 x 64 / 4 waves, 11: 128 x 64 / 8 waves, 12: 64 x 64 / 4 waves with a 3-stage ring
int launch_gemm_nt3(const G3Batch& batch, hipStream_t st, int variant) {
    if (batch.count < 1 || batch.count > kMaxG3) return MARL_EINVAL;
    int max_m = 0, max_n = 0;
    int64_t blocks128 = 0;
    for (int i = 0; i < batch.count; ++i) {
        MARL_TRY(check_g3(batch.p[i], false));
        if (!batch.p[i].c) return MARL_EINVAL;
        max_m = batch.p[i].m > max_m ? batch.p[i].m : max_m;
        max_n = batch.p[i].n > max_n ? batch.p[i].n : max_n;
        blocks128 += cdiv(batch.p[i].m, 128) * cdiv(batch.p[i].n, 128);
    }
    if (!variant) variant = tune_get("g3_nt_variant", 0);
    if (!variant) {
        // Tile plans by launch size (tools/small_r_lab.py, profiles/r05_small_r_lab.json):
        // big = one product with >= 1024 tiles of 128 x 128, mid = >= 256, small = everything else (the in-loop
        // backward batch at every batch size, the narrow heads, all launches of a 32-image batch).  Wave tiles
        // of 32 x 32 (plans 7, 11, 12: ~85 registers, 3-4 workgroups per CU) hide the ring's barriers better than
        // 64 x 64 ones wherever the launch is not long enough to amortise a 12 us prologue + epilogue.
        variant = max_n >= 96 && blocks128 >= 256 ? 2 : 7;
        // round 6: ONE long product whose width is a whole number of 256-column tiles (dU: [Ns R x 256] over both cells'
        // gate gradients) -> the phase-pipelined 256 x 256 plan: A is read once instead of once per 128-column tile
        // (lab 182 -> 165 us on [65536 x 256 x 1024]; iteration -0.02 ms; the heads, n = 384, keep 128 x 128: 88 vs 78 us)
        if (variant == 2 && blocks128 >= 1024 && batch.count == 1 && max_n % 256 == 0) variant = 21;
    }
    prof_before(1, st);
    int rc;
    if (variant == 1)
        rc = launch_g3_variant<256, 128, 4, 2, 4, false>(batch, max_m, max_n, st);
    else if (variant == 2)
        rc = launch_g3_variant<128, 128, 2, 2, 3, false>(batch, max_m, max_n, st);
#ifdef MARL_G3_ABLATE
    else if (variant == 4)  // one workgroup per CU, six stages
        rc = launch_g3_variant<128, 128, 2, 2, 6, false>(batch, max_m, max_n, st);
    else if (variant == 5)  // eight waves on a 128 x 128 tile
        rc = launch_g3_variant<128, 128, 4, 2, 3, false>(batch, max_m, max_n, st);
    else if (variant == 6)  // eight waves, 256 x 128, three stages (2 workgroups would need 216 KB: still 1 per CU)
        rc = launch_g3_variant<256, 128, 4, 2, 3, false>(batch, max_m, max_n, st);
#endif
    else if (variant == 7)
        rc = launch_g3_variant<64, 64, 2, 2, 4, false>(batch, max_m, max_n, st);
    else if (variant == 11)
        rc = launch_g3_variant<128, 64, 4, 2, 3, false>(batch, max_m, max_n, st);
    else if (variant == 12)
        rc = launch_g3_variant<64, 64, 2, 2, 3, false>(batch, max_m, max_n, st);
    else if (variant == 21)  // phase-pipelined, 256 x 256 / 8 waves (round 6)
        rc = launch_g3p_variant<256, 256, 2, 4, 3, false>(batch, max_m, max_n, st);
    else if (variant == 22)  // phase-pipelined, 256 x 128 / 8 waves
        rc = launch_g3p_variant<256, 128, 4, 2, 4, false>(batch, max_m, max_n, st);

    else
        rc = launch_g3_variant<128, 64, 2, 2, 4, false>(batch, max_m, max_n, st);
    prof_after(1, st);
    return rc;
}

// Tile plan of the fused LSTM launch.  128 x 128 tiles (variant 2) need >= 2 workgroups per CU to cover each
// other's prologue / epilogue; below that (BASELINE configs[3] / [4] at 32 images per GPU: 64 / 256 tiles on 256
// CUs) the gate-split plans run - 64- or 32-row tiles with one gate per wave, whichever still fills the chip.
int g3_lstm_plan(const G3Batch& batch) {
    int64_t t128 = 0, t64 = 0;
    for (int i = 0; i < batch.count; ++i) {
        t128 += cdiv(batch.p[i].m, 128) * cdiv(batch.p[i].n, 32);
        t64 += cdiv(batch.p[i].m, 64) * cdiv(batch.p[i].n, 32);
    }
    // measured (two cells, n = 256, K = 880): R = 512: 35.0 / 22.9 / 18.6 us (128-row / 64-row / 32-row plan),
    // R = 1024: 37.3 / 28.9 / 27.4, R = 2048: 45.3 / 47.0 / 56.1, R = 4096: 63-65 / 66-68 / 84
    if (t128 >= 192) return 2;
    return t64 >= 384 ? 4 : 3;
}

int launch_gemm_lstm3(const G3Batch& batch, hipStream_t st, int variant) {
    if (batch.count < 1 || batch.count > kMaxG3) return MARL_EINVAL;
    int max_m = 0, max_n = 0;
    for (int i = 0; i < batch.count; ++i) {
        const G3Prob& p = batch.p[i];
        MARL_TRY(check_g3(p, true));
        if (!p.bias || !p.c_prev || !p.h_next || !p.c_next) return MARL_EINVAL;
        max_m = p.m > max_m ? p.m : max_m;
        max_n = p.n > max_n ? p.n : max_n;
    }
    if (!variant) variant = tune_get("g3_lstm_variant", 0);
    if (!variant) variant = g3_lstm_plan(batch);
    prof_before(0, st);
    int rc;
    if (variant == 1)
        rc = launch_g3_variant<256, 128, 8, 1, 4, true>(batch, max_m, max_n, st);
    else if (variant == 3)
        rc = launch_g3_variant<32, 128, 1, 4, 4, true>(batch, max_m, max_n, st);
    else if (variant == 4)
        rc = launch_g3_variant<64, 128, 2, 4, 4, true>(batch, max_m, max_n, st);
    else if (variant == 5)  // phase-pipelined 256-row tiles (round 6): eight waves of 64 x 64 ...
        rc = launch_g3p_variant<256, 128, 4, 2, 4, true>(batch, max_m, max_n, st);
    else if (variant == 6)  // ... or four waves of 128 x 64 (one per SIMD)
        rc = launch_g3p_variant<256, 128, 2, 2, 4, true>(batch, max_m, max_n, st);
#ifdef MARL_G3_ABLATE
    else if (variant == 7)  // (ring depth probe, tools/lstm_p_depth.py: three stages instead of four)
        rc = launch_g3p_variant<256, 128, 2, 2, 3, true>(batch, max_m, max_n, st);
#endif
    else
        rc = launch_g3_variant<128, 128, 4, 1, 3, true>(batch, max_m, max_n, st);
    prof_after(0, st);
    return rc;
}

// knob g3_tn_pipe (default 1): the phase-pipelined step of the 256-column row-contraction bodies (round 6)
static bool tn_pipe() { return tune_get("g3_tn_pipe", 1) != 0; }

template <int BI, int BJ, int WI, int WJ, int NST, bool DB, int ABL = 0, bool PIPE = false>
static int launch_tn3_variant(const G3TnArgs& a, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)NST * ((BI + BJ) / 16) * 16 * kImgRowBytes;
    static_assert(lds <= 160 * 1024 && (size_t)WI * WJ * 32 * 36 * 4 <= lds, "LDS ring / epilogue panels");
    auto kern = gemm_tn3_kernel<BI, BJ, WI, WJ, NST, DB, ABL, PIPE>;
    static bool raised = false;
    if (lds > 64 * 1024 && !raised) {
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WI * WJ * 64), lds, st, a);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

// rows per split are whole row blocks; the slabs (and the column-sum slabs behind them) are reduced by
// the caller (launch_slab_reduce / the deferred queue) exactly like those of the fp32-operand kernels
G3TnPlan g3_tn_plan(int ni, int nj, int64_t rows) {
    G3TnPlan p;
    p.variant = tune_get("g3_tn_variant", 0);
    // 256-column tiles of A when both sides fill them (measured: wins from 32768 rows on): whole 256 x 256
    // tiles when B's width is (nearly) a multiple of 256, else column passes with a 128-wide last pass
    // (NJ = 368: 373 vs 386 us; NJ = 624: 604 vs 907 us, fp32-operand kernel 660)
    if (!p.variant) {
        const int rem = nj % 256;
        p.variant = !(ni >= 256 && nj >= 192 && rows >= 32768) ? 2 : (rem > 0 && rem <= 128 && nj > 256) ? 4 : 3;
    }
    const int bi = p.variant == 2 ? 128 : 256, bj = p.variant == 3 ? 256 : 128;
    const int64_t tiles = p.variant == 4 ? cdiv(ni, 256) : cdiv(ni, bi) * cdiv(nj, bj);
    int64_t s = cdiv(tune_get("g3_tn_wgs", p.variant == 2 ? 512 : 256), tiles);
    const int64_t max_s = cdiv(rows, 256);  // at least 16 stages per split
    if (s > max_s) s = max_s;
    if (s > 512) s = 512;
    if (s < 1) s = 1;
    const int64_t rps = cdiv(cdiv(rows, s), 32) * 32;
    p.splits = (int)cdiv(rows, rps);
    p.rows_per_split = rps;
    return p;
}
size_t g3_tn_scratch_bytes(int ni, int nj, int64_t rows) {
    const G3TnPlan p = g3_tn_plan(ni, nj, rows);
    return ((size_t)p.splits * ni * nj + (size_t)p.splits * ni) * sizeof(float);
}

int launch_gemm_tn3(G3TnArgs a, const G3TnPlan& plan, hipStream_t st) {
    if (!a.a3 || !a.b3 || !a.out || a.ni <= 0 || a.nj <= 0 || a.rows <= 0 || (a.rows & 31) || (a.a_row0 & 31) ||
        (a.b_row0 & 31) || a.a_steps < img_steps(a.ni) || a.b_steps < img_steps(a.nj)) {
        set_error("gemm_tn3: bad operand (ni=%d nj=%d rows=%lld)", a.ni, a.nj, (long long)a.rows);
        return MARL_EINVAL;
    }
    // 32-bit offsets: column steps of an operand x one row block, and the row blocks of one split
    if ((int64_t)(plan.rows_per_split >> 5) * a.a_steps * kImgChunkBytes >= (1ll << 31) ||
        (int64_t)(plan.rows_per_split >> 5) * a.b_steps * kImgChunkBytes >= (1ll << 31)) {
        set_error("gemm_tn3: split too long for 32-bit offsets");
        return MARL_ELIMIT;
    }
    a.rows_per_split = plan.rows_per_split;
    a.safe = tune_get("g3_safe", 0) != 0;
#ifdef MARL_G3_ABLATE
    a.abl = tune_get("g3_tn_abl", 0);
#endif
    const int bi = plan.variant == 2 ? 128 : 256, bj = plan.variant == 3 ? 256 : 128;
    dim3 grid((unsigned)cdiv(a.ni, bi), (unsigned)(plan.variant == 4 ? 1 : cdiv(a.nj, bj)), (unsigned)plan.splits);
    a.gx = a.gy = a.gz = 0;
    if (tune_get("tn_xcd", 1)) {
        a.gx = (int)grid.x;
        a.gy = (int)grid.y;
        a.gz = (int)grid.z;
        grid = dim3(grid.x * grid.y * grid.z);
    }
    prof_before(2, st);
    int rc;
    if (plan.variant == 4) {
        constexpr size_t lds = (size_t)3 * 32 * 16 * kImgRowBytes;  // the 256 x 256 ring covers the 256 x 128 one
        static bool raised = false;
        if (!raised) {
            MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn3_passes_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn3_passes_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            raised = true;
        }
        if (tn_pipe()) hipLaunchKernelGGL(gemm_tn3_passes_kernel<true>, grid, dim3(512), lds, st, a);
        else hipLaunchKernelGGL(gemm_tn3_passes_kernel<false>, grid, dim3(512), lds, st, a);
        rc = hipGetLastError() == hipSuccess ? MARL_OK : MARL_EHIP;
    } else if (plan.variant == 1)
        rc = launch_tn3_variant<256, 128, 4, 2, 4, true>(a, grid, st);
#ifdef MARL_G3_ABLATE
    else if (plan.variant == 3 && a.abl == 1) rc = launch_tn3_variant<256, 256, 4, 2, 3, false, 1>(a, grid, st);
    else if (plan.variant == 3 && a.abl == 2) rc = launch_tn3_variant<256, 256, 4, 2, 3, false, 2>(a, grid, st);
    else if (plan.variant == 3 && a.abl == 3) rc = launch_tn3_variant<256, 256, 4, 2, 3, false, 3>(a, grid, st);
    else if (plan.variant == 3 && a.abl == 4) rc = launch_tn3_variant<256, 256, 4, 2, 3, false, 4>(a, grid, st);
#endif
    else if (plan.variant == 3 && tn_pipe())
        rc = launch_tn3_variant<256, 256, 4, 2, 3, false, 0, true>(a, grid, st);
    else if (plan.variant == 3)
        rc = launch_tn3_variant<256, 256, 4, 2, 3, false>(a, grid, st);
    else
        rc = launch_tn3_variant<128, 128, 2, 2, 3, true>(a, grid, st);
    prof_after(2, st);
    return rc;
}

// ---- one cell's dW_ih and dW_hh in one launch ---------------------------------------------------------
// column tiles of one operand: whole 256-wide ones (a remainder of more than 128 columns takes one too) plus at most
// one narrow tile for a remainder of <= 128 columns
static void tn_cell_tiles(int nj, int& n256, int& n128) {
    n256 = nj / 256;
    const int rem = nj - n256 * 256;
    n128 = 0;
    if (rem > 128) ++n256;
    else if (rem > 0) n128 = 1;
}
bool g3_tn_cell_ok(int ni, int nj_ih, int nj_hh, int64_t rows) {
    if (tune_get("g3_tn_cell", 1) == 0 || ni < 256 || rows < 8192 || (rows & 31)) return false;
    int a256, a128, b256, b128;
    tn_cell_tiles(nj_ih, a256, a128);
    tn_cell_tiles(nj_hh, b256, b128);
    return a128 + b128 <= 1 && a256 + b256 + a128 + b128 <= kMaxTnCell && a256 + b256 >= 1;
}
// teams (see the kernel): a narrow last tile exists and G's 256-column tiles pair up
static bool g3_tn_cell_teams(int ni, int nj_ih, int nj_hh) {
    int a256, a128, b256, b128;
    tn_cell_tiles(nj_ih, a256, a128);
    tn_cell_tiles(nj_hh, b256, b128);
    return a128 + b128 == 1 && ni % 512 == 0;
}
G3TnPlan g3_tn_cell_plan(int ni, int nj_ih, int nj_hh, int64_t rows) {
    G3TnPlan p;
    p.variant = 5;
    // ONE round of equal workgroups: every workgroup of the launch is resident at once (<= 256: one per CU), walks a
    // long row slab and writes one partial tile - measured against 3 rounds of a third the length (760 workgroups):
    // C3 7.37 vs 7.47 ms, C5 17.60 vs 17.78 (a third of the partial-slab traffic, no dispatch stagger inside a team);
    // two rounds: no gain.
    int a256, a128, b256, b128;
    tn_cell_tiles(nj_ih, a256, a128);
    tn_cell_tiles(nj_hh, b256, b128);
    const int n256 = a256 + b256, n128 = a128 + b128;
    const int64_t per_slab = g3_tn_cell_teams(ni, nj_ih, nj_hh) ? (int64_t)(ni / 512) * (2 * n256 + 1)
                                                                : (int64_t)cdiv(ni, 256) * (n256 + n128);
    int64_t s = 256 / (per_slab > 0 ? per_slab : 1);
    const int64_t max_s = cdiv(rows, 256);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    const int64_t rps = cdiv(cdiv(rows, s), 32) * 32;
    p.splits = (int)cdiv(rows, rps);
    p.rows_per_split = rps;
    return p;
}
size_t g3_tn_cell_scratch_bytes(int ni, int nj_ih, int nj_hh, int64_t rows) {
    const G3TnPlan p = g3_tn_cell_plan(ni, nj_ih, nj_hh, rows);
    return (size_t)p.splits * ((size_t)ni * nj_ih + (size_t)ni * nj_hh + 2 * (size_t)ni) * sizeof(float);
}

int launch_gemm_tn3_cell(G3TnArgs ih, G3TnArgs hh, const G3TnPlan& plan, hipStream_t st) {
    for (const G3TnArgs* q : {&ih, &hh}) {
        const G3TnArgs& a = *q;
        if (!a.a3 || !a.b3 || !a.out || a.ni <= 0 || a.nj <= 0 || a.rows <= 0 || (a.rows & 31) || (a.a_row0 & 31) ||
            (a.b_row0 & 31) || a.a_steps < img_steps(a.ni) || a.b_steps < img_steps(a.nj) || a.ni != ih.ni ||
            a.rows != ih.rows || a.a3 != ih.a3) {
            set_error("gemm_tn3_cell: bad operand (ni=%d nj=%d rows=%lld)", a.ni, a.nj, (long long)a.rows);
            return MARL_EINVAL;
        }
        if ((int64_t)(plan.rows_per_split >> 5) * a.a_steps * kImgChunkBytes >= (1ll << 31) ||
            (int64_t)(plan.rows_per_split >> 5) * a.b_steps * kImgChunkBytes >= (1ll << 31)) {
            set_error("gemm_tn3_cell: split too long for 32-bit offsets");
            return MARL_ELIMIT;
        }
    }
    if (!g3_tn_cell_ok(ih.ni, ih.nj, hh.nj, ih.rows)) {
        set_error("gemm_tn3_cell: shape outside the plan (ni=%d nj=%d/%d)", ih.ni, ih.nj, hh.nj);
        return MARL_EINVAL;
    }
    G3TnCell c{};
    int n = 0;
    G3TnArgs last128{};
    bool have128 = false;
    for (G3TnArgs* q : {&ih, &hh}) {
        q->rows_per_split = plan.rows_per_split;
        q->safe = tune_get("g3_safe", 0) != 0;
        for (int j0 = 0; j0 < q->nj; j0 += 256) {
            G3TnArgs t = *q;
            t.j_first = j0;
            t.csum = nullptr;
            if (q->nj - j0 > 128) c.t[n++] = t;
            else last128 = t, have128 = true;
        }
    }
    // the column sums of G (the bias gradient) ride on the FIRST 256-wide tile, whichever product it belongs to (both
    // read the same G; the narrow tile's body carries no column-sum accumulators)
    c.t[0].csum = ih.csum ? ih.csum : hh.csum;
    c.n256 = n;
    if (have128) c.t[n++] = last128;
    c.nt = n;
    c.gx = (int)cdiv(ih.ni, 256);
    c.gz = plan.splits;
    c.teams = have128 && g3_tn_cell_teams(ih.ni, ih.nj, hh.nj);
    const unsigned total = c.teams ? (unsigned)((c.gx >> 1) * (2 * c.n256 + 1) * c.gz) : (unsigned)(c.gx * c.nt * c.gz);
    constexpr size_t lds = (size_t)3 * 32 * 16 * kImgRowBytes;
    static bool raised = false;
    if (!raised) {
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn3_cell_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn3_cell_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = true;
    }
    prof_before(2, st);
    if (tn_pipe()) hipLaunchKernelGGL(gemm_tn3_cell_kernel<true>, dim3(total), dim3(512), lds, st, c);
    else hipLaunchKernelGGL(gemm_tn3_cell_kernel<false>, dim3(total), dim3(512), lds, st, c);
    const int rc = hipGetLastError() == hipSuccess ? MARL_OK : MARL_EHIP;
    prof_after(2, st);
    return rc;
}

}  // namespace marl

// ---------------------------------------------------------------------------
// kernel-level C ABI (tests, tools; include/marl_hip.h)
// ---------------------------------------------------------------------------
extern "C" {

size_t marl_image_bytes(int64_t rows, int k) { return rows > 0 && k > 0 ? marl::img_bytes(rows, k) : 0; }

int marl_image_build(const float* src, int ld, int64_t rows, int k, void* image, void* stream) {
    using namespace marl;
    if (!src || !image || rows < 1 || k < 1 || ld < k || (reinterpret_cast<uintptr_t>(image) & 15)) {
        set_error("image_build: bad argument");
        return MARL_EINVAL;
    }
    ImgBatch b{};
    b.d[0] = ImgDesc{src, image, rows, k, ld};
    b.count = 1;
    return launch_images(b, static_cast<hipStream_t>(stream));
}

int marl_gemm_nt_images(const void* a3, const void* b3, const float* bias, float* c, int ldc, int m, int n,
                        int k, int accumulate, int variant, void* stream) {
    using namespace marl;
    if (k < 1 || m < 1 || n < 1 || !a3 || !b3 || !c || ldc < n) return MARL_EINVAL;
    G3Batch bt{};
    bt.p[0] = g3_prob(a3, 0, b3, 0, k, c, ldc, m, n, bias, accumulate);
    bt.count = 1;
    return launch_gemm_nt3(bt, static_cast<hipStream_t>(stream), variant);
}

// `count` products of equal M and K in one launch (the in-loop backward batch)
int marl_gemm_nt_images_batch(int count, const void* const* a3, const void* const* b3, float* const* c,
                              const int* n, const int* ldc, int m, int k, int accumulate, int variant,
                              void* stream) {
    using namespace marl;
    if (count < 1 || count > kMaxG3 || k < 1 || m < 1 || !a3 || !b3 || !c || !n || !ldc)
        return MARL_EINVAL;
    for (int i = 0; i < count; ++i)
        if (n[i] < 1 || ldc[i] < n[i] || !a3[i] || !b3[i] || !c[i]) return MARL_EINVAL;
    G3Batch bt{};
    for (int i = 0; i < count; ++i)
        bt.p[i] = g3_prob(a3[i], 0, b3[i], 0, k, c[i], ldc[i], m, n[i], nullptr, accumulate);
    bt.count = count;
    return launch_gemm_nt3(bt, static_cast<hipStream_t>(stream), variant);
}

// one LSTM cell: gates = u3 * wih3^T + h3 * whh3^T + bias, fused cell update; h3_next nullable
// (cells = 2: the same problem twice in one launch - the two-cell launch of the episode, for timing)
int marl_lstm_images(const void* u3, int nin, const void* h3, const void* wih3, const void* whh3,
                     const float* bias, const float* c_prev, float* h_next, float* c_next, float* gates,
                     void* h3_next, int m, int n, int ld_state, int ld_gates, int variant, int cells,
                     void* stream) {
    using namespace marl;
    if (nin < 1 || n < 1 || m < 1 || !u3 || !h3 || !wih3 || !whh3) return MARL_EINVAL;
    G3Batch bt{};
    G3Prob& p = bt.p[0];
    p = g3_prob(u3, 0, wih3, 0, nin, nullptr, 0, m, n, bias);
    g3_add_seg(p, h3, 0, whh3, 0, n);
    p.c_prev = c_prev;
    p.h_next = h_next;
    p.c_next = c_next;
    p.gates = gates;
    p.ld_state = ld_state;
    p.ld_gates = ld_gates;
    p.h3 = static_cast<char*>(h3_next);
    p.h3_row0 = 0;
    p.h3_steps = img_steps(n);
    bt.count = 1;
    if (cells == 2) {
        bt.p[1] = p;
        bt.count = 2;
    }
    return launch_gemm_lstm3(bt, static_cast<hipStream_t>(stream), variant);
}

// C[NI,NJ] = sum_r A[r,i] B[r,j] from images whose rows are the contraction index (rows % 32 == 0);
// colsum (nullable) [NI] = column sums of A; scratch >= marl_gemm_tn_images_scratch() bytes
size_t marl_gemm_tn_images_scratch(int ni, int nj, int64_t rows) {
    if (ni < 1 || nj < 1 || rows < 32 || rows % 32 != 0) return 0;  // (the plan divides by the tile count)
    return marl::g3_tn_scratch_bytes(ni, nj, rows);
}
int marl_gemm_tn_images(const void* a3, const void* b3, float* c, int ldc, int ni, int nj, int64_t rows,
                        float* colsum, float* scratch, size_t scratch_bytes, void* stream) {
    using namespace marl;
    if (ni < 1 || nj < 1 || rows < 32 || rows % 32 != 0 || !a3 || !b3 || !c) {
        set_error("gemm_tn_images: ni=%d nj=%d rows=%lld (rows must be a positive multiple of 32)", ni, nj,
                  (long long)rows);
        return MARL_EINVAL;
    }
    if (!scratch || scratch_bytes < g3_tn_scratch_bytes(ni, nj, rows)) {
        set_error("gemm_tn_images: scratch too small");
        return MARL_ESIZE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const G3TnPlan plan = g3_tn_plan(ni, nj, rows);
    G3TnArgs a{};
    a.a3 = static_cast<const char*>(a3);
    a.b3 = static_cast<const char*>(b3);
    a.a_steps = img_steps(ni);
    a.b_steps = img_steps(nj);
    a.out = scratch;
    a.ldo = nj;
    a.out_split_stride = (int64_t)ni * nj;
    a.ni = ni;
    a.nj = nj;
    a.rows = rows;
    a.csum = colsum ? scratch + (size_t)plan.splits * ni * nj : nullptr;
    MARL_TRY(launch_gemm_tn3(a, plan, st));
    return launch_slab_reduce(scratch, (int64_t)ni * nj, plan.splits, c, ldc, ni, nj, a.csum, colsum, st);
}

// both weight gradients of one LSTM cell (C_ih [NI, NIH] = G^T U, C_hh [NI, NHH] = G^T H, colsum = column sums of
// G) from one launch that reads G once (gemm_tn3_cell_kernel); 0 bytes = shape outside the plan
size_t marl_gemm_tn_images_cell_scratch(int ni, int nih, int nhh, int64_t rows) {
    if (ni < 1 || nih < 1 || nhh < 1 || rows < 32 || rows % 32 != 0 || !marl::g3_tn_cell_ok(ni, nih, nhh, rows)) return 0;
    return marl::g3_tn_cell_scratch_bytes(ni, nih, nhh, rows);
}
int marl_gemm_tn_images_cell(const void* g3, int ni, const void* u3, int nih, const void* h3, int nhh, int64_t rows,
                             float* c_ih, int ld_ih, float* c_hh, int ld_hh, float* colsum, float* scratch,
                             size_t scratch_bytes, void* stream) {
    using namespace marl;
    if (ni < 1 || nih < 1 || nhh < 1 || rows < 32 || rows % 32 != 0 || !g3 || !u3 || !h3 || !c_ih || !c_hh ||
        !g3_tn_cell_ok(ni, nih, nhh, rows)) {
        set_error("gemm_tn_images_cell: shape outside the plan (ni=%d nih=%d nhh=%d rows=%lld)", ni, nih, nhh, (long long)rows);
        return MARL_EINVAL;
    }
    if (!scratch || scratch_bytes < g3_tn_cell_scratch_bytes(ni, nih, nhh, rows)) {
        set_error("gemm_tn_images_cell: scratch too small");
        return MARL_ESIZE;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const G3TnPlan plan = g3_tn_cell_plan(ni, nih, nhh, rows);
    float* s_ih = scratch;
    float* s_hh = s_ih + (size_t)plan.splits * ni * nih;
    float* s_cs = s_hh + (size_t)plan.splits * ni * nhh;
    G3TnArgs ih{}, hh{};
    ih.a3 = hh.a3 = static_cast<const char*>(g3);
    ih.a_steps = hh.a_steps = img_steps(ni);
    ih.ni = hh.ni = ni;
    ih.rows = hh.rows = rows;
    ih.b3 = static_cast<const char*>(u3);
    ih.b_steps = img_steps(nih);
    ih.nj = ih.ldo = nih;
    ih.out = s_ih;
    ih.out_split_stride = (int64_t)ni * nih;
    ih.csum = colsum ? s_cs : nullptr;
    hh.b3 = static_cast<const char*>(h3);
    hh.b_steps = img_steps(nhh);
    hh.nj = hh.ldo = nhh;
    hh.out = s_hh;
    hh.out_split_stride = (int64_t)ni * nhh;
    MARL_TRY(launch_gemm_tn3_cell(ih, hh, plan, st));
    MARL_TRY(launch_slab_reduce(s_ih, (int64_t)ni * nih, plan.splits, c_ih, ld_ih, ni, nih, nullptr, nullptr, st));
    return launch_slab_reduce(s_hh, (int64_t)ni * nhh, plan.splits, c_hh, ld_hh, ni, nhh, ih.csum, colsum, st);
}

}  // extern "C"
