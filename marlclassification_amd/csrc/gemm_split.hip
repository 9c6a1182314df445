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

#include <utility>

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
// (+64: the planes of an image start 16 banks apart - the three 16-byte stores of a pre-split
// row that fall into one 8-lane group then never share a bank)
constexpr int plane_bytes(int rows) { return rows * SROW + 64; }

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

// Operand tiles are fetched with raw buffer loads written as inline assembly: descriptor + scalar
// tile offset in SGPRs, ONE persistent 32-bit VGPR offset per chunk, nothing else - and the
// request stays where it is written.  hipcc would (a) rebuild 64-bit addresses for every plain
// load, park them in the destination registers of loads still in flight and drain vmcnt at the
// top of every tile body, and (b) sink the requests of tile u + 4 three bodies down to their
// first use (even buffer-load builtins marked volatile): measured 37 % of the kernel time.
//
// The requested tiles wait in the accumulation registers a[128:255], addressed BY NUMBER from
// these asm statements only ("raw" storage: no C++ variable lives there).  Giving the compiler
// asm outputs for data that has not arrived does not work: it copies them (a -> v -> a, to
// satisfy its own register assignment) before the wait.  The kernels' own register demand
// stays below a128 - tests/test_host_logic.py::test_split_kernels_leave_the_staging_registers_alone
// scans the ISA for any compiler-generated access to a[128:255].
//   araw_load  : buffer_load_dwordx4 a[LO:LO+3]      (vmcnt is counted by hand: buf_wait<N>)
//   araw_read4 : four v_accvgpr_read into VGPR values (the split arithmetic needs VGPRs)
//   araw_lds16 : ds_write_b128 straight from a[LO:LO+3] (pre-split weight chunks: no VGPR at all)
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int kARawBase = 128;
__device__ __forceinline__ void araw_reserve() { asm volatile("" ::: "a255"); }  // kernel uses all 256 AGPRs
__device__ __forceinline__ i32x4 make_rsrc(const void* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)0xffffffffu);  // num_records: no reliance on the range check
    r.w = __builtin_amdgcn_readfirstlane(0x00020000);
    return r;
}
template <int LO>
__device__ __forceinline__ void araw_load(const i32x4& rsrc, uint32_t voff, uint32_t soff) {
    static_assert(LO >= kARawBase && LO + 3 <= 255 && (LO & 3) == 0, "staging register range");
    asm volatile("buffer_load_dwordx4 a[%3:%4], %0, %1, %2 offen"
                 :
                 : "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane((int)soff)), "n"(LO), "n"(LO + 3)
                 : "memory");
}
template <int LO>
__device__ __forceinline__ void araw_read4(float& x0, float& x1, float& x2, float& x3) {
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\t"
                 "v_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3)
                 : "n"(LO), "n"(LO + 1), "n"(LO + 2), "n"(LO + 3));
}
template <int LO>
__device__ __forceinline__ void araw_lds16(uint32_t lds_byte) {
    asm volatile("ds_write_b128 %0, a[%1:%2]" : : "v"(lds_byte), "n"(LO), "n"(LO + 3) : "memory");
}
// all but the N youngest requests have landed
template <int N>
__device__ __forceinline__ void buf_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
__device__ __forceinline__ void buf_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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

// ---------------------------------------------------------------------------
// Hand-placed software pipeline.  One wave per SIMD (256 threads, one workgroup per CU, two LDS
// stages, up to 512 registers) runs, for every 32-deep K tile u, a BODY of NM matrix
// instructions on LDS stage u % 2 while it splits tile u + 1 (already in registers) into the
// other stage and requests tile u + 4.  A wave issues in order, so the staging work only
// overlaps with the matrix pipe if it sits BETWEEN the MFMAs in program order: slot m of a body
// = MFMA m + a slice of the staging arithmetic (a few single-issue instructions, which run in
// the 32-cycle shadow of the MFMA: MI355X_MICROARCH.md) + at most one LDS access, pinned by
// sched_barrier.  Left to itself hipcc emits the whole split as ONE block in front of 48
// back-to-back MFMAs - no overlap, 2.1 us per tile instead of 0.8 (measured, as were two
// independent workgroups per CU and a ping-pong pair of wave groups: both ~1.5 us per tile).
//
// Staging arithmetic of a set of NP value pairs, in lock step (all pairs take step s before any
// takes step s + 1: seven dependent steps, 11 VALU per pair):
//   A: p0 = cvt(x)   B: t = unpack(p0)   C: x -= t   D: p1 = cvt(x)   E: t = unpack(p1)
//   F: x -= t        G: p2 = cvt(x)
// chunk c = pairs 2c, 2c + 1 = 8 bytes per plane; its plane-0 / 1 / 2 store follows the last
// op of step A / D / G it needs.
// ---------------------------------------------------------------------------
#ifdef MARL_KERNEL_TS
// cycle stamps kept in scalar registers (s_memtime; the lgkmcnt wait also completes the wave's
// LDS traffic - diagnosis only)
__device__ __forceinline__ uint64_t ts_stamp() {
    uint64_t t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    return t;
}
#define MARL_STAMP(i_) if (ts_on) ts_v[i_] = ts_stamp();
#else
#define MARL_STAMP(i_)
#endif

template <int TM, int TN>
struct Frags {
    bf16x8 a[2][3][TM], b[2][3][TN];
};

template <int NP>
struct SplitRegs {
    uint32_t p0[NP], p1[NP], p2[NP];
};

// MFMA m of a body: the first half of a body multiplies the SECOND 16-deep step of the previous
// tile (its fragments were read during that tile's body), the second half the first step of
// the current tile (read at the top of this body): every fragment read has half a body to
// land, nothing waits for LDS behind the barrier.  Products smallest terms first,
// accumulators round robin.
template <int TM, int TN, int m>
__device__ __forceinline__ void mfma_slot(const Frags<TM, TN>& f, f32x16 (&acc)[TM][TN]) {
    constexpr int per = 6 * TM * TN, kk = m < per ? 1 : 0, rem = m % per;
    constexpr int pi = rem / (TM * TN), q = rem % (TM * TN), i = q / TN, j = q % TN;
    constexpr int pa = pi == 0 ? 1 : (pi == 2 ? 2 : (pi == 4 ? 1 : 0));
    constexpr int pb = pi == 0 ? 1 : (pi == 1 ? 2 : (pi == 3 ? 1 : 0));
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[kk][pa][i], f.b[kk][pb][j], acc[i][j], 0, 0, 0);
}
// fragment read r (0 .. 3 (TM + TN) - 1) of the 16-deep step kk
template <int TM, int TN, int APL, int BPL, int kk, int r>
__device__ __forceinline__ void frag_read(const char* al, const char* bl, Frags<TM, TN>& f) {
    if constexpr (r < 3 * TM)
        f.a[kk][r / TM][r % TM] = *reinterpret_cast<const bf16x8*>(al + (r / TM) * APL + (r % TM) * 32 * SROW + kk * 32);
    else
        f.b[kk][(r - 3 * TM) / TN][(r - 3 * TM) % TN] = *reinterpret_cast<const bf16x8*>(
            bl + ((r - 3 * TM) / TN) * BPL + ((r - 3 * TM) % TN) * 32 * SROW + kk * 32);
}
template <int TM, int TN, int APL, int BPL, int kk, int... R>
__device__ __forceinline__ void frag_read_all(const char* al, const char* bl, Frags<TM, TN>& f,
                                              std::integer_sequence<int, R...>) {
    (frag_read<TM, TN, APL, BPL, kk, R>(al, bl, f), ...);
}

// Element layout of a staged set.  NT: x[4 c + e] = element e of chunk c (consecutive k).
// TN: x[16 o + 4 q + c] = operand o, tile row 4 rb + q, column 4 cb + c; chunk (o, c) = the four
// rows of a column (the 4 x 4 register transpose).
template <bool TRANS>
struct SetMap {
    static constexpr int e0(int pr) { return TRANS ? (pr / 8) * 16 + (2 * (pr % 2)) * 4 + (pr % 8) / 2 : 2 * pr; }
    static constexpr int e1(int pr) { return TRANS ? e0(pr) + 4 : 2 * pr + 1; }
};

// op n (0 .. 11 NP - 1) of the lock-step split of x; the chunk stores ride behind the ops that
// complete them.  dst(c) = wbase + WOFF(c); planes PL bytes apart.
template <int NP, bool TRANS, int PL, int n>
__device__ __forceinline__ void split_op(float (&x)[2 * NP], SplitRegs<NP>& r, char* wbase) {
    // steps: A (NP cvt) | BC (2 NP x {unpack, subtract}) | D (NP cvt) | EF (2 NP x 2) | G (NP cvt);
    // op n counts single instructions: 11 per pair
    constexpr int sA = NP, sC = sA + 4 * NP, sD = sC + NP, sF = sD + 4 * NP;
    using M = SetMap<TRANS>;
    // byte offset of chunk c from wbase.  NT: chunk c = tile row + 32 c.  TN: operand o = c / 4
    // (its image 3 planes further), column c % 4 = next LDS row
    auto woff = [](int c) constexpr { return TRANS ? (c / 4) * 3 * PL + (c % 4) * SROW : c * 32 * SROW; };
    if constexpr (n < sA) {
        r.p0[n] = pack_bf16(x[M::e0(n)], x[M::e1(n)]);
        if constexpr (n & 1) *reinterpret_cast<uint2*>(wbase + woff(n / 2)) = make_uint2(r.p0[n - 1], r.p0[n]);
    } else if constexpr (n < sC) {
        // element m = (n - sA) / 2: the op pair {unpack its bf16 term, subtract}; done on the odd op
        if constexpr ((n - sA) & 1) {
            constexpr int m = (n - sA) / 2;
            x[(m & 1) ? M::e1(m / 2) : M::e0(m / 2)] -=
                (m & 1) ? __uint_as_float(r.p0[m / 2] & 0xffff0000u) : __uint_as_float(r.p0[m / 2] << 16);
        }
    } else if constexpr (n < sD) {
        constexpr int m = n - sC;
        r.p1[m] = pack_bf16(x[M::e0(m)], x[M::e1(m)]);
        if constexpr (m & 1) *reinterpret_cast<uint2*>(wbase + woff(m / 2) + PL) = make_uint2(r.p1[m - 1], r.p1[m]);
    } else if constexpr (n < sF) {
        if constexpr ((n - sD) & 1) {
            constexpr int m = (n - sD) / 2;
            x[(m & 1) ? M::e1(m / 2) : M::e0(m / 2)] -=
                (m & 1) ? __uint_as_float(r.p1[m / 2] & 0xffff0000u) : __uint_as_float(r.p1[m / 2] << 16);
        }
    } else if constexpr (n < 11 * NP) {
        constexpr int m = n - sF;
        r.p2[m] = pack_bf16(x[M::e0(m)], x[M::e1(m)]);
        if constexpr (m & 1) *reinterpret_cast<uint2*>(wbase + woff(m / 2) + 2 * PL) = make_uint2(r.p2[m - 1], r.p2[m]);
    }
}
template <int NP, bool TRANS, int PL, int... N>
__device__ __forceinline__ void split_ops(float (&x)[2 * NP], SplitRegs<NP>& r, char* wbase,
                                          std::integer_sequence<int, N...>) {
    (split_op<NP, TRANS, PL, N>(x, r, wbase), ...);
}
template <int FIRST, int... N>
constexpr auto seq_from(std::integer_sequence<int, N...>) { return std::integer_sequence<int, (FIRST + N)...>{}; }
template <int FIRST, int COUNT>
constexpr auto seq_range() { return seq_from<FIRST>(std::make_integer_sequence<int, (COUNT > 0 ? COUNT : 0)>{}); }

// One pipelined body: NM MFMAs, every one followed by its slice of everything else the wave has
// to issue for this tile step (the "items" below), pinned by sched_barrier.  Nothing but the
// wait for the set about to be staged precedes the first MFMA.
//   slot m < NCH          : chunk m of the staged set -> VGPR values (Ops::take<m>)
//   slot m >= NCH         : Q ops of the lock-step split of those values (+ the LDS stores
//                           that ride behind them)
//   first half, in order  : this tile's first-step fragment reads, the pre-split B chunk
//                           copies, this body's memory requests (Ops::request<k>)
//   second half           : this tile's second-step fragment reads (used in the NEXT body; the
//                           first half still multiplies the previous tile's)
// Ops (kernel specific, all static-index templates): take<c>(x), request<k>(), copy_b3<i>().
template <int TM, int TN, int APL, int BPL, int NP, bool TRANS, int B3N, int NREQ, class Ops>
struct Body {
    static constexpr int NM = 12 * TM * TN;
    static constexpr int NR = 3 * (TM + TN);  // fragment reads per 16-deep step
    static constexpr int NCH = NP / 2;        // chunks of the staged set
    static constexpr int Q = (11 * NP + (NM - NCH) - 1) / (NM - NCH);
    static constexpr int NF = NR + B3N + NREQ;                  // first-half items
    static constexpr int PF = (NF + NM / 2 - 1) / (NM / 2);     // ... per slot
    static constexpr int PS = (NR + NM / 2 - 1) / (NM / 2);     // second-half items per slot
    static_assert(NCH < NM && PF >= 1 && PS >= 1, "slot plan");

    template <int i>
    static __device__ __forceinline__ void first_item(const char* al, const char* bl, Frags<TM, TN>& f, Ops& o) {
        if constexpr (i < NR) {
#ifndef MARL_EXP_NOREAD
            frag_read<TM, TN, APL, BPL, 0, i>(al, bl, f);
#endif
        } else if constexpr (i < NR + B3N) {
            o.template copy_b3<i - NR>();
        } else if constexpr (i < NF) {
            o.template request<i - NR - B3N>();
        }
    }
    template <int... I>
    static __device__ __forceinline__ void first_items(const char* al, const char* bl, Frags<TM, TN>& f, Ops& o,
                                                       std::integer_sequence<int, I...>) {
        (first_item<I>(al, bl, f, o), ...);
    }
    template <int... I>
    static __device__ __forceinline__ void second_items(const char* al, const char* bl, Frags<TM, TN>& f,
                                                        std::integer_sequence<int, I...>) {
#ifndef MARL_EXP_NOREAD
        ((I < NR ? frag_read<TM, TN, APL, BPL, 1, (I < NR ? I : 0)>(al, bl, f) : (void)0), ...);
#endif
    }
    template <int m>
    static __device__ __forceinline__ void slot(const char* al, const char* bl, Frags<TM, TN>& f,
                                                f32x16 (&acc)[TM][TN], float (&x)[2 * NP], SplitRegs<NP>& r,
                                                char* wbase, Ops& o) {
        mfma_slot<TM, TN, m>(f, acc);
#ifndef MARL_EXP_NOSTAGE
        if constexpr (m < NCH)
            o.template take<m>(x);
        else
            split_ops<NP, TRANS, APL>(x, r, wbase, seq_range<(m - NCH) * Q, Q>());
#endif
        if constexpr (m < NM / 2)
            first_items(al, bl, f, o, seq_range<m * PF, PF>());
        else
            second_items(al, bl, f, seq_range<(m - NM / 2) * PS, PS>());
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int... M>
    static __device__ __forceinline__ void slots(const char* al, const char* bl, Frags<TM, TN>& f,
                                                 f32x16 (&acc)[TM][TN], float (&x)[2 * NP], SplitRegs<NP>& r,
                                                 char* wbase, Ops& o, std::integer_sequence<int, M...>) {
        (slot<M>(al, bl, f, acc, x, r, wbase, o), ...);
    }
    // the second step of the last tile (after the loop)
    template <int... M>
    static __device__ __forceinline__ void flush_(const Frags<TM, TN>& f, f32x16 (&acc)[TM][TN],
                                                  std::integer_sequence<int, M...>) {
        (mfma_slot<TM, TN, M>(f, acc), ...);
    }
    static __device__ __forceinline__ void flush(const Frags<TM, TN>& f, f32x16 (&acc)[TM][TN]) {
        flush_(f, acc, std::make_integer_sequence<int, NM / 2>{});
    }
    // before the first body: no previous tile - its "second step" multiplies zeros
    static __device__ __forceinline__ void init(Frags<TM, TN>& f) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) f.a[1][p][i][e] = 0;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) f.b[1][p][j][e] = 0;
        }
    }
    static __device__ __forceinline__ void run(const char* al, const char* bl, Frags<TM, TN>& f,
                                               f32x16 (&acc)[TM][TN], float (&x)[2 * NP], SplitRegs<NP>& r,
                                               char* wbase, Ops& o) {
        slots(al, bl, f, acc, x, r, wbase, o, std::make_integer_sequence<int, NM>{});
    }
};

// ---- per-kernel item providers --------------------------------------------------------------
// NT: A set S at a[ALO + 16 S], pre-split B set P at a[BLO + 24 P]; this body stages A set SN and
// B set PN, and requests B tile -> set PR, then A tile -> set SR
template <int B3N, int ALO, int BLO, int SN, int PN, int SR, int PR>
struct NtOps {
    const i32x4& rsa;
    const i32x4& rsb;
    const uint32_t (&cao)[4];
    const uint32_t (&vb3)[B3N];
    const uint32_t (&db3)[B3N];
    uint32_t so_a, so_b, b3dst;
    template <int c>
    __device__ __forceinline__ void take(float (&x)[16]) {
        araw_read4<ALO + 16 * SN + 4 * c>(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);
    }
    template <int i>
    __device__ __forceinline__ void copy_b3() { araw_lds16<BLO + 24 * PN + 4 * i>(b3dst + db3[i]); }
    template <int k>
    __device__ __forceinline__ void request() {
        if constexpr (k < B3N)
            araw_load<BLO + 24 * PR + 4 * k>(rsb, vb3[k], so_b);
        else
            araw_load<ALO + 16 * SR + 4 * (k - B3N)>(rsa, cao[k - B3N], so_a);
    }
};
// TN: set S at a[128 + 32 S]: A rows q = 0..3, then B rows
template <int SN, int SR>
struct TnOps {
    const i32x4& rsa;
    const i32x4& rsb;
    const uint32_t (&aof)[4];
    const uint32_t (&bof)[4];
    uint32_t so_a, so_b;
    template <int c>
    __device__ __forceinline__ void take(float (&x)[32]) {
        // chunk c of the split = column c % 4 of operand c / 4; its four values are rows q = 0..3:
        // elements 16 o + 4 q + col.  Take the staged vector c (row c % 4 of operand c / 4) instead:
        // by chunk 8 every element is there, and the first split op only runs in slot 8
        araw_read4<kARawBase + 32 * SN + 4 * c>(x[4 * c], x[4 * c + 1], x[4 * c + 2], x[4 * c + 3]);
    }
    template <int i>
    __device__ __forceinline__ void copy_b3() {}
    template <int k>
    __device__ __forceinline__ void request() {
        if constexpr (k < 4)
            araw_load<kARawBase + 32 * SR + 4 * k>(rsa, aof[k], so_a);
        else
            araw_load<kARawBase + 32 * SR + 16 + 4 * (k - 4)>(rsb, bof[k - 4], so_b);
    }
};

// plain (not pipelined) matrix phase of one staged tile: tails
template <int TM, int TN, int APL, int BPL>
__device__ __forceinline__ void split_compute(const char* al, const char* bl, f32x16 (&acc)[TM][TN]) {
    Frags<TM, TN> f;
    frag_read_all<TM, TN, APL, BPL, 0>(al, bl, f, std::make_integer_sequence<int, 3 * (TM + TN)>{});
    frag_read_all<TM, TN, APL, BPL, 1>(al, bl, f, std::make_integer_sequence<int, 3 * (TM + TN)>{});
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#define MARL_SPLIT_P(pa_, pb_)                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                         \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                     \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[kk][pa_][i], f.b[kk][pb_][j], acc[i][j], 0, 0, 0);
        MARL_SPLIT_P(1, 1) MARL_SPLIT_P(0, 2) MARL_SPLIT_P(2, 0)
        MARL_SPLIT_P(0, 1) MARL_SPLIT_P(1, 0) MARL_SPLIT_P(0, 0)
#undef MARL_SPLIT_P
    }
}

}  // namespace

#ifdef MARL_EXP_NOBAR
#define MARL_EXP_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define MARL_EXP_BARRIER() lds_barrier()
#endif

// ---------------------------------------------------------------------------
// NT: C[M,N] (+)= sum_s A_s[M,K_s] * B_s[N,K_s]^T + bias, optional LSTM-cell epilogue; 128 x BN
// tiles.  The B operands are weights whose bf16x3 image exists in the weights workspace
// (split_weights_kernel: [row][k / 32][plane][32] bf16, zero-padded to whole K tiles): their
// tiles are copied, only A is split.  (Products with other B operands take the exact-fp32
// kernel of gemm.hip.)
// ---------------------------------------------------------------------------
template <int BN, bool LSTM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_nt_split_kernel(const GemmBatch batch) {
    constexpr int BM = 128;
    constexpr int WM = LSTM ? 4 : 2, WN = LSTM ? 1 : 2;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int KC = SK / 4;                 // float4 chunks per tile row
    constexpr int A_CH = BM * KC / 256;        // 4
    constexpr int B3_CH = BN * 12 / 256;       // 16-byte chunks of a pre-split B tile per thread
    constexpr int APL = plane_bytes(BM), BPL = plane_bytes(BN);
    constexpr int GSZ = 3 * (APL + BPL);       // LDS bytes of one stage
    constexpr int NP = 2 * A_CH;   // value pairs a thread splits per tile
    // staging registers (raw AGPRs): A set S, chunk i at a[128 + 16 S + 4 i]; B set P, chunk i at
    // a[192 + 24 P + 4 i]
    constexpr int ALO = kARawBase, BLO = kARawBase + 64;
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
    const char* const a0p = reinterpret_cast<const char*>(P.seg[0].a);
    const char* const a1p = reinterpret_cast<const char*>(P.seg[1].a);
    const char* const b0p = reinterpret_cast<const char*>(P.seg[0].b3);
    const char* const b1p = reinterpret_cast<const char*>(P.seg[1].b3);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: thread t moves A chunks c = t + 256 i (tile row c / KC, floats [(c % KC) * 4, +4)).
    // Rows beyond M / N are clamped (never stored).  Byte offsets of both segments stay in registers.
    const int koff = (tid % KC) * 4;
    uint32_t aof0[A_CH], aof1[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        int r = m0 + (tid + 256 * i) / KC;
        r = r < M ? r : M - 1;
        aof0[i] = (uint32_t)r * (uint32_t)P.seg[0].lda * 4u;
        aof1[i] = (uint32_t)r * (uint32_t)P.seg[1].lda * 4u;
    }
    // A pre-split B tile = BN rows x 192 bytes, and the image is tile-major: rows n0 .. n0 + BN - 1
    // of a K tile are contiguous (LSTM: four blocks of 32 rows).  Thread t copies the 16-byte
    // chunks c = t + 256 i (row c / 12, plane (c % 12) / 4, column (c % 12) % 4): a wave's 64
    // chunks are 1 KB of consecutive bytes - whole cache lines.  Same offsets in both segments.
    uint32_t vb3[B3_CH];   // byte offset of chunk i inside a K tile of the image
    uint32_t db3[B3_CH];   // its place in the LDS image (relative to the B image of a stage)
#pragma unroll
    for (int i = 0; i < B3_CH; ++i) {
        const int c = tid + 256 * i, row = c / 12, rem = c % 12;
        int gn;  // row of the image, relative to the row the segment's b3 points at
        if (LSTM) {
            int unit = n0 + (row & 31);
            unit = unit < N ? unit : N - 1;
            gn = (row >> 5) * N + unit;
        } else {
            gn = n0 + row;
            gn = gn < N ? gn : N - 1;
        }
        vb3[i] = (uint32_t)gn * 192u + (uint32_t)rem * 16u;
        db3[i] = (uint32_t)((rem / 4) * BPL + row * SROW + (rem % 4) * 16);
    }

    // staging: A (from HBM) four sets - tile u + 4 is requested while tile u is multiplied; the
    // pre-split B tiles (L2-resident weights) two sets, two tiles ahead.
    // Requests are UNCONDITIONAL; tiles past the end re-read the last tile.  The last tile of a
    // segment may reach past K: those A chunks read on into the next row (or up to 112 bytes past
    // the last row - A operands are slices of the episode workspace, whose every float is
    // finite) and meet the zero padding of the B image.  Chunk offsets / descriptor / tile
    // stride in use change ONCE per request stream, when it crosses into the second segment.
    araw_reserve();
    uint32_t cao[A_CH], cts = P.seg[0].ts3;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) cao[i] = aof0[i] + (uint32_t)koff * 4u;
    i32x4 rsa = make_rsrc(a0p), rsb = make_rsrc(b0p);
    static_assert(A_CH == 4, "four A chunks per thread and tile");
    // scalar part of the requests of A tile qa and B tile qb (each stream switches segment once)
#define MARL_SP_SCALARS(qa_, qb_)                                                          \
    if ((qa_) == t0 && (qa_) < T) {                                                        \
        _Pragma("unroll") for (int i = 0; i < A_CH; ++i) cao[i] = aof1[i] + (uint32_t)koff * 4u; \
        rsa = make_rsrc(a1p);                                                              \
    }                                                                                      \
    if ((qb_) == t0 && (qb_) < T) {                                                        \
        cts = P.seg[1].ts3;                                                                \
        rsb = make_rsrc(b1p);                                                              \
    }                                                                                      \
    const int qda_ = (qa_) < T ? (qa_) : T - 1, qdb_ = (qb_) < T ? (qb_) : T - 1;          \
    const uint32_t so_a_ = (uint32_t)(qda_ - (qda_ >= t0 ? t0 : 0)) * (SK * 4);            \
    const uint32_t so_b_ = (uint32_t)(qdb_ - (qdb_ >= t0 ? t0 : 0)) * cts;

    // LDS addresses of this thread: A chunk 0 (chunk i is 32 rows further), its fragments - in
    // stage 0; stage 1 is GSZ bytes further.  (The dynamic LDS starts at byte 0.)
    char* const wa = smem_c + (tid / KC) * SROW + (tid % KC) * 8;
    constexpr uint32_t wb3 = 3 * APL;  // B image of stage 0
    const char* const al = smem_c + (wm * (BM / WM) + (lane & 31)) * SROW + (lane >> 5) * 16;
    const char* const bl = smem_c + 3 * APL + (wn * (BN / WN) + (lane & 31)) * SROW + (lane >> 5) * 16;
    Frags<TM, TN> fr;
    float xs[2 * NP];
    SplitRegs<NP> sr;
    // body of tile u = t + j (A set j, stage j & 1): multiply tile u while tile u + 1 (A set jn,
    // B set 1 - par) goes to the other stage; request B tile u + 2 into B set par (tile u's:
    // copied during the previous body) and A tile u + 4 into A set j.  Landed before the body:
    // everything but the previous body's A request.
#define MARL_SP_BODY(j_, jn_, par_)                                                        \
    {                                                                                      \
        MARL_SP_SCALARS(t + j_ + 4, t + j_ + 2)                                            \
        using O_ = NtOps<B3_CH, ALO, BLO, jn_, 1 - par_, j_, par_>;                        \
        using B_ = Body<TM, TN, APL, BPL, NP, false, B3_CH, A_CH + B3_CH, O_>;             \
        O_ o_{rsa, rsb, cao, vb3, db3, so_a_, so_b_, wb3 + ((j_ + 1) & 1) * GSZ};          \
        buf_wait<A_CH>();                                                                  \
        B_::run(al + (j_ & 1) * GSZ, bl + (j_ & 1) * GSZ, fr, acc, xs, sr, wa + ((j_ + 1) & 1) * GSZ, o_); \
        MARL_EXP_BARRIER();                                                                \
    }
    using O0 = NtOps<B3_CH, ALO, BLO, 0, 0, 0, 0>;
    using B0 = Body<TM, TN, APL, BPL, NP, false, B3_CH, A_CH + B3_CH, O0>;
    B0::init(fr);
    {   // prologue: requests of tiles 0..3 (B: 0, 1), tile 0 -> stage 0 (plain)
        { MARL_SP_SCALARS(0, 0) O0 o{rsa, rsb, cao, vb3, db3, so_a_, so_b_, wb3};
          o.template request<0>(); o.template request<1 % B3_CH>(); o.template request<2 % B3_CH>();
          if (B3_CH > 3) { o.template request<3 % B3_CH>(); o.template request<4 % B3_CH>(); o.template request<5 % B3_CH>(); }
          o.template request<B3_CH>(); o.template request<B3_CH + 1>(); o.template request<B3_CH + 2>(); o.template request<B3_CH + 3>(); }
        { MARL_SP_SCALARS(1, 1) NtOps<B3_CH, ALO, BLO, 0, 0, 1, 1> o{rsa, rsb, cao, vb3, db3, so_a_, so_b_, wb3};
          o.template request<0>(); o.template request<1 % B3_CH>(); o.template request<2 % B3_CH>();
          if (B3_CH > 3) { o.template request<3 % B3_CH>(); o.template request<4 % B3_CH>(); o.template request<5 % B3_CH>(); }
          o.template request<B3_CH>(); o.template request<B3_CH + 1>(); o.template request<B3_CH + 2>(); o.template request<B3_CH + 3>(); }
        { MARL_SP_SCALARS(2, 1) NtOps<B3_CH, ALO, BLO, 0, 0, 2, 1> o{rsa, rsb, cao, vb3, db3, so_a_, so_b_, wb3};
          o.template request<B3_CH>(); o.template request<B3_CH + 1>(); o.template request<B3_CH + 2>(); o.template request<B3_CH + 3>(); }
        { MARL_SP_SCALARS(3, 1) NtOps<B3_CH, ALO, BLO, 0, 0, 3, 1> o{rsa, rsb, cao, vb3, db3, so_a_, so_b_, wb3};
          o.template request<B3_CH>(); o.template request<B3_CH + 1>(); o.template request<B3_CH + 2>(); o.template request<B3_CH + 3>(); }
        buf_wait<3 * A_CH + B3_CH>();  // tile 0 (A and B) has landed
        O0 o{rsa, rsb, cao, vb3, db3, 0u, 0u, wb3};
        o.template take<0>(xs); o.template take<1>(xs); o.template take<2>(xs); o.template take<3>(xs);
        split_ops<NP, false, APL>(xs, sr, wa, std::make_integer_sequence<int, 11 * NP>{});
        o.template copy_b3<0>(); o.template copy_b3<1 % B3_CH>(); o.template copy_b3<2 % B3_CH>();
        if (B3_CH > 3) { o.template copy_b3<3 % B3_CH>(); o.template copy_b3<4 % B3_CH>(); o.template copy_b3<5 % B3_CH>(); }
        lds_barrier();
    }
    // (the prologue already requested B tile 1 into set 1: body 0 requests B tile 2 into set 0, ...)
    for (int t = 0;; t += 4) {
        MARL_SP_BODY(0, 1, 0)
        if (t + 1 >= T) break;
        MARL_SP_BODY(1, 2, 1)
        if (t + 2 >= T) break;
        MARL_SP_BODY(2, 3, 0)
        if (t + 3 >= T) break;
        MARL_SP_BODY(3, 0, 1)
        if (t + 4 >= T) break;
    }
    buf_drain();  // (requests of tiles past the end are still in flight)
    B0::flush(fr, acc);
#undef MARL_SP_SCALARS
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

    // four sets of staging registers (see gemm_nt_split_kernel): set S at a[128 + 32 S], A rows
    // q = 0..3 then B rows; unconditional requests, tiles past the end re-read the last full tile
    // (staged into a stage that is never multiplied)
    constexpr int NP = 16;  // value pairs per tile: 4 x 4 block of A + 4 x 4 block of B
    araw_reserve();
    float xs[2 * NP];
    const i32x4 rsa = make_rsrc(abase), rsb = make_rsrc(bbase);
    const uint32_t tsa = (uint32_t)SK * (uint32_t)lda * 4u, tsb = (uint32_t)SK * (uint32_t)ldb * 4u;  // tile strides
    char* const wbase = smem_c + (4 * cb) * SROW + rb * 8;  // column 4 cb of the A image, stage 0
    const char* const al = smem_c + (wm * 64 + (lane & 31)) * SROW + (lane >> 5) * 16;
    const char* const bl = smem_c + 3 * APL + (wn * 64 + (lane & 31)) * SROW + (lane >> 5) * 16;
    Frags<TM, TN> fr;
    SplitRegs<NP> sr;
    using O0 = TnOps<0, 0>;
    using Bd0 = Body<TM, TN, APL, BPL, NP, true, 0, 8, O0>;
    // column sums of A (the bias gradient) from the raw values of the set that body u - 1 staged
    // (tile u): read once more from the staging registers, which the next request into that set
    // has not touched yet - it is issued later in the same body.  Tiles past the end: weight 0.
#define MARL_TS_CSUM(S_, live_)                                                            \
    if (CSUM) {                                                                            \
        float c_[16];                                                                      \
        araw_read4<kARawBase + 32 * S_>(c_[0], c_[1], c_[2], c_[3]);                       \
        araw_read4<kARawBase + 32 * S_ + 4>(c_[4], c_[5], c_[6], c_[7]);                   \
        araw_read4<kARawBase + 32 * S_ + 8>(c_[8], c_[9], c_[10], c_[11]);                 \
        araw_read4<kARawBase + 32 * S_ + 12>(c_[12], c_[13], c_[14], c_[15]);              \
        const float w_ = (live_) ? 1.f : 0.f;                                              \
        cs.x += w_ * ((c_[0] + c_[4]) + (c_[8] + c_[12]));                                 \
        cs.y += w_ * ((c_[1] + c_[5]) + (c_[9] + c_[13]));                                 \
        cs.z += w_ * ((c_[2] + c_[6]) + (c_[10] + c_[14]));                                \
        cs.w += w_ * ((c_[3] + c_[7]) + (c_[11] + c_[15]));                                \
    }
    // body of tile u = t + j: multiply tile u while tile u + 1 (set jn) is split into the other
    // stage; request tile u + 4 into set j.  Landed before the body: all but the two youngest sets.
#define MARL_TS_BODY(j_, jn_)                                                              \
    {                                                                                      \
        const uint32_t qd_ = (uint32_t)(t + j_ + 4 < T ? t + j_ + 4 : T - 1);              \
        using O_ = TnOps<jn_, j_>;                                                         \
        using B_ = Body<TM, TN, APL, BPL, NP, true, 0, 8, O_>;                             \
        O_ o_{rsa, rsb, aof, bof, qd_ * tsa, qd_ * tsb};                                   \
        buf_wait<16>();                                                                    \
        MARL_TS_CSUM(jn_, t + j_ + 1 < T)                                                  \
        B_::run(al + (j_ & 1) * GSZ, bl + (j_ & 1) * GSZ, fr, acc, xs, sr, wbase + ((j_ + 1) & 1) * GSZ, o_); \
        MARL_EXP_BARRIER();                                                                \
    }

    if (T > 0) {
        Bd0::init(fr);
        {   // prologue: requests of tiles 0..3, tile 0 -> stage 0 (plain)
#define MARL_TS_REQ(S_, q_)                                                                \
    {                                                                                      \
        const uint32_t qd_ = (uint32_t)((q_) < T ? (q_) : T - 1);                          \
        TnOps<0, S_> o{rsa, rsb, aof, bof, qd_ * tsa, qd_ * tsb};                          \
        o.template request<0>(); o.template request<1>(); o.template request<2>(); o.template request<3>(); \
        o.template request<4>(); o.template request<5>(); o.template request<6>(); o.template request<7>(); \
    }
            MARL_TS_REQ(0, 0)
            MARL_TS_REQ(1, 1)
            MARL_TS_REQ(2, 2)
            MARL_TS_REQ(3, 3)
#undef MARL_TS_REQ
            buf_wait<24>();
            MARL_TS_CSUM(0, true)
            O0 o{rsa, rsb, aof, bof, 0u, 0u};
            o.template take<0>(xs); o.template take<1>(xs); o.template take<2>(xs); o.template take<3>(xs);
            o.template take<4>(xs); o.template take<5>(xs); o.template take<6>(xs); o.template take<7>(xs);
            split_ops<NP, true, APL>(xs, sr, wbase, std::make_integer_sequence<int, 11 * NP>{});
            lds_barrier();
        }
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
        buf_drain();  // (requests of tiles past the end are still in flight)
        Bd0::flush(fr, acc);
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
            const float4 va = *reinterpret_cast<const float4*>(ap_ + (aof[q] - back * (uint32_t)lda * 4u));
            const float4 vb = *reinterpret_cast<const float4*>(bp_ + (bof[q] - back * (uint32_t)ldb * 4u));
            xs[4 * q] = va.x * m;
            xs[4 * q + 1] = va.y * m;
            xs[4 * q + 2] = va.z * m;
            xs[4 * q + 3] = va.w * m;
            xs[16 + 4 * q] = vb.x;
            xs[16 + 4 * q + 1] = vb.y;
            xs[16 + 4 * q + 2] = vb.z;
            xs[16 + 4 * q + 3] = vb.w;
        }
        if (CSUM) {
            cs.x += (xs[0] + xs[4]) + (xs[8] + xs[12]);
            cs.y += (xs[1] + xs[5]) + (xs[9] + xs[13]);
            cs.z += (xs[2] + xs[6]) + (xs[10] + xs[14]);
            cs.w += (xs[3] + xs[7]) + (xs[11] + xs[15]);
        }
        split_ops<NP, true, APL>(xs, sr, wbase, std::make_integer_sequence<int, 11 * NP>{});
        lds_barrier();
        split_compute<TM, TN, APL, BPL>(al, bl, acc);
        lds_barrier();
    }
#undef MARL_TS_CSUM
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
        const int64_t rt = idx >> 4;  // tile * rows + row
        const int t = (int)(rt / d.rows);
        const int64_t r = rt % d.rows;
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
bool split_lookup(const GemmSeg& g, const void*& b3, uint32_t& ts) {
    for (int i = 0; i < g_nreg; ++i) {
        const SplitReg& r = g_reg[i];
        if (g.b < r.base || g.b >= r.base + r.floats || g.ldb != r.ld || g.k != r.k) continue;
        const size_t off = (size_t)(g.b - r.base);
        if (off % (size_t)r.ld) return false;
        const size_t rows = r.floats / (size_t)r.ld;
        if (rows * 192 * (size_t)((r.k + 31) / 32) >= (1ull << 32)) return false;  // 32-bit tile offsets
        ts = (uint32_t)(rows * 192);
        b3 = r.image + (off / (size_t)r.ld) * 192;
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

// every B operand must be a registered weight matrix (its pre-split image is what the kernel
// copies); *ok = false -> the caller takes the exact-fp32 kernel instead
static bool split_prepare(GemmBatch& batch, bool lstm) {
    if (!tune_get("split_pre", 1)) return false;
    for (int i = 0; i < batch.count; ++i)
        for (int sg = 0; sg < batch.p[i].nseg; ++sg) {
            GemmSeg& g = batch.p[i].seg[sg];
            if (!split_lookup(g, g.b3, g.ts3)) return false;
        }
    return true;
}

template <int BN, bool LSTM>
static int launch_nt_split_variant(dim3 grid, GemmBatch& batch, hipStream_t st) {
    batch.gx = (int)grid.x;
    batch.gy = (int)grid.y;
    batch.xcd_map = batch.count == 1 && !LSTM && tune_get("nt_xcd", 1);
    if (batch.xcd_map) grid = dim3(grid.x * grid.y * grid.z);
    constexpr int lds = 2 * 3 * plane_bytes(128) + 2 * 3 * plane_bytes(BN);  // two stages
    static bool raised = false;  // > 64 KiB of dynamic LDS: opt in once per instantiation
    if (!raised) {
        MARL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_split_kernel<BN, LSTM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        raised = true;
    }
#ifdef MARL_KERNEL_TS
    static long long* d_ts = nullptr;
    static int calls = 0;
    const int rec = ts_begin(&d_ts, calls++);
    batch.ts = rec ? d_ts : nullptr;
#endif
    hipLaunchKernelGGL((gemm_nt_split_kernel<BN, LSTM>), grid, dim3(256), lds, st, batch);
#ifdef MARL_KERNEL_TS
    if (rec) {
        long long h[9];
        (void)hipMemcpy(h, d_ts, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[ts] nt_split body (cycles): top %lld | slots0-11 %lld | 12-23 %lld | 24-35 %lld | 36-47 %lld | barrier %lld | total %lld\n",
                h[1] - h[0], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[7] - h[5], h[8] - h[7], h[8] - h[0]);
    }
#endif
    return MARL_OK;
}

// returns MARL_OK with *done = 0 when the batch cannot take the bf16x6 kernel
int launch_gemm_nt_split(const GemmBatch& batch_in, int max_m, int max_n, int64_t blocks128, hipStream_t st,
                         int* done) {
    GemmBatch batch = batch_in;
    *done = 0;
    if (!split_prepare(batch, false)) return MARL_OK;
    *done = 1;
    // one workgroup per CU: 128-wide column tiles unless 128 x 64 tiles fit one round of 256
    // workgroups better
    int64_t blocks64 = 0;
    for (int i = 0; i < batch.count; ++i) blocks64 += cdiv(batch.p[i].m, 128) * cdiv(batch.p[i].n, 64);
    const bool wide = blocks128 >= tune_get("nts_min_blocks128", 192) || blocks64 > 256;
    if (wide && max_n >= 96) {
        dim3 grid((unsigned)cdiv(max_m, 128), (unsigned)cdiv(max_n, 128), (unsigned)batch.count);
        return launch_nt_split_variant<128, false>(grid, batch, st);
    }
    dim3 grid((unsigned)cdiv(max_m, 128), (unsigned)cdiv(max_n, 64), (unsigned)batch.count);
    return launch_nt_split_variant<64, false>(grid, batch, st);
}

int launch_gemm_lstm_split(const GemmBatch& batch_in, int max_m, int max_n, hipStream_t st, int* done) {
    GemmBatch batch = batch_in;
    *done = 0;
    if (!split_prepare(batch, true)) return MARL_OK;
    *done = 1;
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
