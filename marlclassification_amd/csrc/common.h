// Internal declarations shared by the HIP translation units of libmarl_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/marl_hip.h"
#include "split.h"

namespace marl {

constexpr int kWave = 64;  // gfx950 wavefront

inline int p4(int x) { return (x + 3) & ~3; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

void set_error(const char* fmt, ...);

// Tuning knobs (perf experiments, tests): value set by marl_tune(), else the environment variable
// MARL_<KEY IN UPPER CASE>, else `dflt`.  Read at every launch - cheap, host side only.
int tune_get(const char* key, int dflt);
int tune_set(const char* key, int value);

// Sum over the 64 lanes of a wave on the VALU (DPP row shifts + row broadcasts, then a
// readlane of lane 63): no LDS-pipe ds_bpermute round trips, result is wave-uniform.
// Call with all 64 lanes active.
// Activations on the transcendental unit: one v_exp_f32 and one v_rcp_f32 (each ~1 ulp; ~1e-7 absolute
// error on a sigmoid, inside the 1e-5 parity budget and checked by the golden-vector tests).  libm's
// expf is ~12 VALU instructions and an IEEE division 11 (so is __frcp_rn: it compiles to the full
// v_div_scale / v_div_fmas / v_div_fixup sequence, not to v_rcp_f32); on kernels whose row passes are
// VALU-issue-bound (cnn_dgrad, panel_*, ln_silu_*, sample) that was a third of their instructions.
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoid_acc(float x) { return rcp_fast(1.0f + __expf(-x)); }
// 1 - 2 / (1 + e^{2x}); e^{2x} -> inf gives 1, -> 0 gives -1
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * rcp_fast(1.0f + __expf(2.0f * x)); }
__device__ __forceinline__ float silu_fast(float y) { return y * sigmoid_acc(y); }
// d silu(y) / dy
__device__ __forceinline__ float silu_grad_fast(float y) {
    const float s = sigmoid_acc(y);
    return s * (1.0f + y * (1.0f - s));
}

__device__ __forceinline__ float wave_sum(float v) {
    int x = __float_as_int(v);
#define MARL_DPP_ADD(ctrl, rmask)                         \
    x = __float_as_int(__int_as_float(x) +                \
                       __int_as_float(__builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false)))
    MARL_DPP_ADD(0x111, 0xf);  // row_shr:1
    MARL_DPP_ADD(0x112, 0xf);  // row_shr:2
    MARL_DPP_ADD(0x114, 0xf);  // row_shr:4
    MARL_DPP_ADD(0x118, 0xf);  // row_shr:8   -> lane 15 of each row holds the row's sum
    MARL_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    MARL_DPP_ADD(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef MARL_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

#define MARL_HIP_CHECK(expr)                                                        \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            marl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                            __FILE__, __LINE__);                                    \
            return MARL_EHIP;                                                       \
        }                                                                           \
    } while (0)

#define MARL_LAUNCH_CHECK() MARL_HIP_CHECK(hipGetLastError())

#define MARL_TRY(expr)             \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != MARL_OK) return rc_; \
    } while (0)

// ---------------------------------------------------------------------------
// GEMM problem descriptors (gemm.hip)
// ---------------------------------------------------------------------------
// C[M,N] (+)= sum over segments s of A_s[M,K_s] * B_s[N,K_s]^T  (+ bias[N])
// Both operands are K-contiguous ("NT"): lda/ldb are multiples of 4 floats, bases
// 16-byte aligned, and columns [K_s, round4(K_s)) of both are zero.
struct GemmSeg {
    const float* a;
    const float* b;
    int lda, ldb, k;
    // filled by the bf16x6 launcher when b is a registered weight matrix: its pre-split image
    // [row][k / 32][plane][32] bf16 and the number of 32-deep K tiles per row
    const void* b3;
    int kt3;
};

struct GemmProb {
    GemmSeg seg[2];
    int nseg;
    int m, n;
    float* c;
    int ldc;
    const float* bias;
    int accumulate;  // C += result
    // fused LSTM-cell epilogue (gate rows of B are g * n + unit, g = i,f,g,o)
    const float* c_prev;
    float* h_next;
    float* c_next;
    float* gates;  // [M, ld_gates]: activated i | f | g | o
    int ld_state, ld_gates;
};

constexpr int kMaxGemmBatch = 4;
struct GemmBatch {
    GemmProb p[kMaxGemmBatch];
    int count;
    int gx, gy;   // logical tile grid (row blocks, column blocks); filled by the launcher
    int xcd_map;  // 1: 1-D launch with the XCD-aware tile order below
    int single_buf;  // 1: one LDS stage (filled by the launcher)
#ifdef MARL_KERNEL_TS
    long long* ts;  // phase timestamps of one workgroup (debug builds only)
#endif
};

// Tile order for tiled kernels whose neighbouring tiles share operand panels.  Workgroup b of
// a 1-D launch lands on XCD b % 8 (MI355X_MICROARCH.md, workgroup dispatch) and every XCD
// has a private L2, so XCD x gets a CONTIGUOUS chunk of the tile sequence (bijective for any
// tile count), ordered column block fastest: the tiles that run together on one XCD read the
// same A row panel (and the small B) from its L2 instead of HBM.
__device__ __forceinline__ void xcd_tile(unsigned gx, unsigned gy, unsigned gz, unsigned& bx,
                                         unsigned& by, unsigned& bz) {
    const unsigned T = gx * gy * gz, L = blockIdx.x;
    const unsigned per = T >> 3, rem = T & 7, xcd = L & 7, slot = L >> 3;
    const unsigned q = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    bz = q / (gx * gy);
    const unsigned r = q - bz * gx * gy;
    bx = r / gy;
    by = r - bx * gy;
}
int xcd_map_enabled();

GemmProb gemm_prob(const float* a, int lda, const float* b, int ldb, int k, float* c, int ldc,
                   int m, int n, const float* bias = nullptr, int accumulate = 0);
void gemm_add_seg(GemmProb& p, const float* a, int lda, const float* b, int ldb, int k);

int launch_gemm_nt(const GemmBatch& batch, hipStream_t st);
// LSTM cell: gates = seg products + bias over 4*n_units rows of B, then the cell update.
int launch_gemm_lstm(const GemmBatch& batch, hipStream_t st);

// bf16x6 forms of the three matrix kernels (gemm_split.hip): fp32 operands split into three bf16
// terms while they are staged, six bf16 MFMA products per fp32 product, fp32 accumulation.
// split_mode(): knob "mfma_split" (default 1); 0 = the exact-fp32 v_mfma_f32_32x32x2_f32 kernels.
int split_mode();
// Pre-split images of the weight matrices (built by marl_pack_weights next to their fp32 copies):
// a matrix [rows][ld] fp32 with k valid columns -> [rows][ceil(k / 32)][3][32] bf16, zero-padded.
size_t split_image_floats(int rows, int k);  // size of an image in units of 4 bytes
struct SplitDesc {
    const float* src;  // [rows][ld]
    void* dst;
    int rows, k, ld, kt;
};
constexpr int kMaxSplitDesc = 64;
struct SplitBatch {
    SplitDesc d[kMaxSplitDesc];
    int count;
};
int launch_split_weights(const SplitBatch& b, hipStream_t st);
// host-side table fp32 copy -> image, rebuilt by every entry point that lays out the weights
// workspace; the launchers look the B operands of a product up in it
void split_registry_reset();
void split_registry_add(const float* base, int rows, int ld, int k, const void* image);
struct SplitRegistryScope {  // clears the table when the registering entry point returns
    SplitRegistryScope() = default;
    SplitRegistryScope(const SplitRegistryScope&) = delete;
    ~SplitRegistryScope() { split_registry_reset(); }
};
int launch_gemm_nt_split(const GemmBatch& batch, int max_m, int max_n, int64_t blocks128, hipStream_t st);
int launch_gemm_lstm_split(const GemmBatch& batch, int max_m, int max_n, hipStream_t st);
int launch_gemm_tn_split(const float* a, int lda, const float* b, int ldb, float* out, int ldo,
                         int64_t stride, int ni, int nj, int64_t rows, int64_t rows_per_split,
                         float* csum, dim3 grid, int gx, int gy, int gz, hipStream_t st);

// ---------------------------------------------------------------------------
// Image GEMMs (gemm3.hip): both operands are k16 images (split.h) written by their producers
// ---------------------------------------------------------------------------
struct G3Seg {
    const char* a3;  // image of A (split.h: [row / 32][step][row % 32][96 bytes]); the product's row 0 is
    const char* b3;  // image row a_row0 / b_row0 (multiples of 32 keep the operand stream contiguous)
    int a_row0, b_row0;
    int steps;       // 16-deep K steps of both images (= K steps of this segment)
};
struct G3Prob {
    G3Seg seg[2];
    int nseg;
    int m, n;
    float* c;
    int ldc;
    const float* bias;
    int accumulate;
    // fused LSTM-cell epilogue (as GemmProb) + the image of h_next
    const float* c_prev;
    float* h_next;
    float* c_next;
    float* gates;
    int ld_state, ld_gates;
    char* h3;  // nullable; image of h_next with h3_steps steps, the product's row 0 = image row h3_row0
    int h3_row0, h3_steps;
};
constexpr int kMaxG3 = 4;
struct G3Batch {
    G3Prob p[kMaxG3];
    int count;
    int gx, gy, xcd_map;  // filled by the launcher
    int safe;             // debug: every step fully waited for (bit-identical results, no overlap)
#ifdef MARL_G3_ABLATE
    long long* clk;       // perf diagnosis: cycle / wall-clock counters of workgroup 0
#endif
};
G3Prob g3_prob(const void* a3, int a_row0, const void* b3, int b_row0, int k, float* c, int ldc, int m, int n,
               const float* bias = nullptr, int accumulate = 0);
void g3_add_seg(G3Prob& p, const void* a3, int a_row0, const void* b3, int b_row0, int k);
int launch_gemm_nt3(const G3Batch& batch, hipStream_t st, int variant = 0);
int launch_gemm_lstm3(const G3Batch& batch, hipStream_t st, int variant = 0);
int g3_lstm_plan(const G3Batch& batch);  // the tile plan launch_gemm_lstm3 takes for this batch (marl_plan_query reports it)
// weight-gradient form: contraction over the ROWS of two images (gemm_tn3_kernel)
struct G3TnArgs {
    const char* a3;
    const char* b3;
    int a_steps, b_steps;  // 16-column steps of each image (its row-block stride / 3072)
    int a_row0, b_row0;    // image row of contraction row 0 (multiples of 32)
    float* out;            // [splits][ni][ldo] partial slabs
    int ldo;
    int64_t out_split_stride;
    int ni, nj;
    int64_t rows, rows_per_split;
    float* csum;           // nullable: [splits][ni] column sums of A
    int gx, gy, gz, safe;  // filled by the launcher
    int j_first;           // first column of B (and of the output) this launch covers (column tile by starts there)
#ifdef MARL_G3_ABLATE
    int abl;
#endif
};
struct G3TnPlan {
    int variant, splits;
    int64_t rows_per_split;
};
G3TnPlan g3_tn_plan(int ni, int nj, int64_t rows);
// Both weight gradients of ONE LSTM cell - dW_ih = G^T U and dW_hh = G^T H, the same gate-gradient image G as
// A - as one launch (gemm_tn3_cell_kernel): the column tiles of U and of H that belong to the same (256-column
// tile of G, row slab) are adjacent workgroups on one XCD, run in step and share G's slab through that XCD's L2,
// so G (403 MB at C3) leaves HBM once instead of once per column tile.  `ih` / `hh` are complete descriptions of
// the two products (out = their slabs, csum on `hh` or `ih`); the plan fixes one split count for both.
constexpr int kMaxTnCell = 4;
struct G3TnCell {
    G3TnArgs t[kMaxTnCell];  // one entry per column tile: 256-wide ones first, then (at most one) 128-wide
    int nt, n256;            // tiles in all, 256-wide ones
    int gx, gz;
    int teams;               // equal-work team dispatch (see the kernel)
};
bool g3_tn_cell_ok(int ni, int nj_ih, int nj_hh, int64_t rows);
G3TnPlan g3_tn_cell_plan(int ni, int nj_ih, int nj_hh, int64_t rows);
size_t g3_tn_cell_scratch_bytes(int ni, int nj_ih, int nj_hh, int64_t rows);
int launch_gemm_tn3_cell(G3TnArgs ih, G3TnArgs hh, const G3TnPlan& plan, hipStream_t st);
size_t g3_tn_scratch_bytes(int ni, int nj, int64_t rows);
int launch_gemm_tn3(G3TnArgs a, const G3TnPlan& plan, hipStream_t st);

// fp32 [rows][ld] (k valid columns) -> its image (rows padded to whole blocks: zeros)
struct ImgDesc {
    const float* src;
    void* dst;
    int64_t rows;
    int k, ld;
};
constexpr int kMaxImgDesc = 64;
struct ImgBatch {
    ImgDesc d[kMaxImgDesc];
    int count;
};
int launch_images(const ImgBatch& b, hipStream_t st);

// per-launch HIP-event timing of one kernel class (see marl_profile_begin); no-ops when off
constexpr int kProfClasses = 6;
void prof_before(int cls, hipStream_t st);
void prof_after(int cls, hipStream_t st);
int profile_begin(int cls, int max_launches);
int profile_end(double* total_ms, int* launches);

struct RedQueue;
// C[NI,NJ] = sum_r A[r,i] * B[r,j]  (weight gradients; contraction over rows).
size_t gemm_tn_scratch_bytes(int ni, int nj, int64_t rows);
// colsum_out (nullable): also writes out[i] = sum_r A[r,i] (the matching bias gradient)
// q != null: the slab reduction is queued there instead of launched (scratch comes from q too)
struct TnQueue;
// tq != null (needs q): a product on the 64 x 64 plan is queued there instead of launched
int launch_gemm_tn(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int ni,
                   int nj, int64_t rows, float* scratch, size_t scratch_bytes, hipStream_t st,
                   float* colsum_out = nullptr, RedQueue* q = nullptr, TnQueue* tq = nullptr);

// Small row contractions (output of at most a few 64 x 64 tiles) of a backward pass collected and run as
// ONE launch at its end: alone each fills a fraction of the chip for 10-50 us (gemm_tn_batch_kernel).
struct TnBatchProb {
    const float* a;
    const float* b;
    float* out;
    float* csum;
    int64_t stride, rows, rows_per_split;
    int lda, ldb, ldo, ni, nj, gx, gy, first_block;
};
constexpr int kMaxTnBatch = 12;
struct TnBatch {
    TnBatchProb p[kMaxTnBatch];
    int count;
};
struct TnQueue {
    struct Item {
        const float* a;
        const float* b;
        float* c;
        float* colsum;
        int lda, ldb, ldc, ni, nj;
        int64_t rows;
    };
    Item it[kMaxTnBatch];
    int n = 0;
};
bool gemm_tn_is_small(int ni, int nj, int64_t rows);  // takes the 64 x 64 plan
// launches what is queued; the split slabs live in q's scratch and their reductions are queued there
int launch_tn_queue(TnQueue& tq, RedQueue* q, hipStream_t st);

// ---------------------------------------------------------------------------
// Deferred fixed-order reductions (gemm.hip).  The backward pass produces ~40 sets of partial
// sums (split-K slabs of the weight-gradient GEMMs, per-workgroup LayerNorm / GroupNorm affine
// partials, conv weight-gradient slabs) whose results nobody reads before the end of the pass.
// Instead of one or two tiny launches each (5-9 us + a kernel boundary apiece), they are queued
// and reduced by ONE launch (two when a long reduction is folded in two stages).
//   out[e] (+)= sum over z < count (fixed order) of part[z * stride + e],  e < n
//   element e < split goes to out0[(e / nj) * ldc + e % nj], the others to out1[e - split]
// ---------------------------------------------------------------------------
struct RedDesc {
    const float* part;
    int64_t stride;
    int count, n;
    float* out0;
    int split, nj, ldc;
    float* out1;
    int accumulate;
    // folds > 1 (stage 1 of a long reduction): fold f sums the parts [f * per, (f + 1) * per)
    // into out0[f * n + e]
    int folds, per;
};
constexpr int kMaxRed = 40;
struct RedBatch {
    RedDesc d[kMaxRed];
    int first_block[kMaxRed + 1];  // workgroups (fold, 64 elements) of descriptor i: [first_block[i], first_block[i + 1])
    int count;
};
struct RedQueue {
    RedBatch stage1, stage2;   // stage1: long reductions folded into temporaries first
    float* scratch = nullptr;  // temporaries of stage 1 + bump space handed out by take()
    size_t cap = 0, off = 0;
    hipStream_t st = nullptr;
    int rc = MARL_OK;
    void reset(float* s, size_t floats, hipStream_t stream);
    // scratch that stays valid until the next flush; when full: flushes (callers push what they
    // took before they take again), or returns null if !may_flush
    float* take(size_t floats, bool may_flush = true);
    void push(const float* part, int64_t stride, int count, int n, float* out0, int split, int nj,
              int ldc, float* out1, int accumulate);
    int launch_pending();  // launch what is queued, keep the scratch handed out
    int flush();           // + hand the scratch out anew
};
int launch_red_batch(const RedBatch& b, hipStream_t st);

// ---------------------------------------------------------------------------
// row-wise kernels (rowops.hip)
// ---------------------------------------------------------------------------
// optional wdot / bdot / dot_out (n <= 384): dot_out[r] = out[r] . wdot + bdot[0] (a one-output
// layer on top of the activation)
int launch_ln_silu_fwd(const float* z, int ldz, const float* gamma, const float* beta, float* out,
                       int ldo, float* stats, int64_t m, int n, hipStream_t st,
                       const float* wdot = nullptr, const float* bdot = nullptr,
                       float* dot_out = nullptr);
// dz = d(loss)/d(z) from da = d(loss)/d(silu out); dgamma/dbeta partial sums are written
// to part[nblk][2][n]; returns nblk through *nblk_out.
int ln_bwd_blocks(int64_t m, int n);
// dz = LayerNorm+SiLU backward of da[r][c] = sum_{j<kin} g[r][j] * bt[c][j], kin <= 4, n <= 384
int launch_ln_silu_bwd_rank(const float* g, int ldg, int kin, const float* bt, int ldbt,
                            const float* z, int ldz, const float* stats, const float* gamma,
                            const float* beta, float* dz, int lddz, float* part, int64_t m, int n,
                            hipStream_t st);
int launch_ln_silu_bwd(const float* da, int ldda, const float* z, int ldz, const float* stats,
                       const float* gamma, const float* beta, float* dz, int lddz, float* part,
                       int64_t m, int n, hipStream_t st);
// LayerNorm/GroupNorm partials [nparts][2][n] -> dgamma[n], dbeta[n]
int launch_reduce_affine(float* part, int64_t nparts, int n, float* dgamma, float* dbeta,
                         int accumulate, hipStream_t st, RedQueue* q = nullptr);

// GroupNorm + SiLU over NHWC rows: z [rows, P, C] -> a.  out_chw != 0 writes element
// (pos, c) at out[row * ldo + c * P + pos] (reference Flatten order), else NHWC with ldo = P*C.
// cols != null: additionally scatters the activations into the NEXT layer's im2col matrix
// (3x3 stride 2 pad 1; hin = this layer's output side); out may then be null.
int gn_fwd_im2col_supported(int P, int C);
int launch_gn_silu_fwd(const float* z, const float* gamma, const float* beta, float* out,
                       int64_t ldo, int out_chw, float* stats, int64_t rows, int P, int C, int G,
                       hipStream_t st, float* cols = nullptr, int ldk = 0, int hin = 0);
int gn_bwd_blocks(int64_t rows, int C);
int launch_gn_silu_bwd(const float* da, int64_t ldda, int da_chw, const float* z,
                       const float* stats, const float* gamma, const float* beta, float* dz,
                       float* part, int64_t rows, int P, int C, int G, hipStream_t st);

// message mean over the other agents (networks/message.py:5-17); self-adjoint, so the
// same kernel is its own backward.
int launch_agg_msg(const float* m, float* out, int ld, int na, int nb, int n, hipStream_t st);

// map_pos (networks/state.py): lambda = SiLU(LN(W * (pos / size) + b)) -> out (ld ldo)
// (npos_in != null: use these normalised positions [rows,2] instead of pos / size)
int launch_pos_embed_fwd(const int32_t* pos, const float* npos_in, int h, int w, const float* W, const float* b,
                         const float* gamma, const float* beta, float* npos4, float* z, int ldz,
                         float* stats, float* out, int ldo, int64_t rows, int nd, hipStream_t st);

int launch_lstm_cell_bwd(const float* dh, int lddh, float* dc, int lddc, float* gates, int ldg,
                         const float* c_prev, const float* c_new, int ldc, int64_t rows, int n,
                         hipStream_t st);
struct LstmBwdBatch;
int launch_lstm_cell_bwd_batch(LstmBwdBatch& b, int count, hipStream_t st);  // 1 or 2 cells, images optional
int launch_lstm_cell_bwd2(const float* dh0, int lddh0, float* dc0, int lddc0, float* gates0,
                          int ldg0, const float* cp0, const float* cn0, int ldc0, int n0,
                          const float* dh1, int lddh1, float* dc1, int lddc1, float* gates1,
                          int ldg1, const float* cp1, const float* cn1, int ldc1, int n1,
                          int64_t rows, hipStream_t st);

struct SampleArgs {
    const float* a_pol;  // [R, ld_a] SiLU(LN(policy hidden))
    int ld_a, nla;
    const float* w1;  // packed [nA, ldw]
    int ldw;
    const float* b1;
    const float* noise;            // [R, nA] injected Exp(1) draws (parity mode), or null
    uint64_t rng_seed, rng_ctr;    // noise == null: Exp(1) drawn in-kernel, Philox4x32-10 keyed by
                                   // rng_seed at counter (rng_ctr, row, action / 4)
    const uint64_t* rng_off_dev;   // optional device-side offset (counter block), added << 16
    int rng_on;
    const int64_t* forced;         // [R] or null
    const int32_t* pos_in;         // [R, 2]
    int32_t* pos_out;              // [R, 2]
    int64_t* step_pos;             // [R, 2] or null
    int64_t* step_actions;         // [R] or null
    float* step_logp;              // [R]
    float* probs;                  // [R, nA] saved
    int32_t* actions_i32;          // [R] saved
    int R, nA, H, W, f;
    int32_t table[MARL_MAX_ACTIONS][2];
    // optional: position embedding of the next step (written for step t+1)
    const float* pe_W;  // packed [nd, 4]; null = off
    const float* pe_b;
    const float* pe_gamma;
    const float* pe_beta;
    float* pe_npos;   // [R, 4]
    float* pe_z;      // [R, pe_ldz]
    float* pe_stats;  // [R, 2]
    float* pe_out;    // [R, pe_ldo] (the lambda slice of U[t+1])
    int pe_ldz, pe_ldo, pe_nd;
    // optional: the same values into the k16 image of U[t+1] (columns pe_col0 .., image row r + pe_row0)
    char* pe_img;
    int pe_row0, pe_steps, pe_col0;
};
int launch_sample(const SampleArgs& a, hipStream_t st);

// Perf-mode episode draws (the reference's reset draws, core/environment.py:33-43 and
// networks/models.py:148-159, from a counter-based generator): pos0[r][d] uniform in
// [0, size_d - f), h0 / c0 [R, n_b] and hc0 / cc0 [R, n_a] standard normal, optionally the
// per-step Exp(1) noise [Ns * R * nA].
int launch_draw_episode(uint64_t seed, uint64_t offset, const uint64_t* offset_dev, int64_t* pos0,
                        int R, int H, int W, int f, float* h0, float* c0, int n_b, float* hc0,
                        float* cc0, int n_a, float* noise, int64_t n_noise, hipStream_t st);

// generic permuted copies used by pack / unpack
struct PermDesc {
    const float* src;    // null: dst is zero-filled
    float* dst;
    int rows, cols;      // dst logical extent
    int dst_ld;
    int rd, rs1, rs2;    // src offset = (r / rd) * rs1 + (r % rd) * rs2
    int cd, cs1, cs2;    //            + (c / cd) * cs1 + (c % cd) * cs2
    const float* src2;   // optional: dst = src + src2 (same indexing), e.g. b_ih + b_hh
};
constexpr int kMaxPerm = 56;  // 64-byte descriptors: the batch stays under the 4 KB kernel-argument limit
struct PermBatch {
    PermDesc d[kMaxPerm];
    int count;
};
int launch_permute(const PermBatch& b, hipStream_t st);

// Device-resident per-iteration counters (marl_counters_* in marl_hip.h): what changes from one
// training iteration to the next WITHOUT host involvement, so that a captured hipGraph can be
// replayed unchanged - the generator offset of the episode draws and the Adam step with its bias
// corrections.
struct Counters {
    uint64_t rng_offset;
    uint64_t step;        // optimiser steps done so far + 1 = the step the NEXT adam launch applies
    float lr_over_bc1;    // lr / (1 - beta1^step)
    float inv_sqrt_bc2;   // 1 / sqrt(1 - beta2^step)
    float pad[2];
};
static_assert(sizeof(Counters) == MARL_COUNTERS_BYTES, "counter block layout");
int launch_counters_set(Counters* c, uint64_t rng_offset, int64_t step, float lr, float beta1,
                        float beta2, int tick, hipStream_t st);

// cnt != null: step size and bias correction come from the counter block instead
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_over_bc1,
                float inv_sqrt_bc2, float beta1, float beta2, float eps, float grad_scale,
                hipStream_t st, const Counters* cnt = nullptr);

// elementwise helpers
int launch_fill(float* p, int64_t n, float v, hipStream_t st);
// out[r][c] = a[r][c] + b[r][c]
int launch_add2d(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo,
                 int64_t rows, int cols, hipStream_t st);
int launch_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int cols,
                  hipStream_t st);
int launch_i64_to_i32(const int64_t* src, int32_t* dst, int64_t n, hipStream_t st);
// dlogits[r][j] = dlogp[r] * (1[j == a_r] - p[r][j]) into [R, ld] (cols >= nA untouched)
int launch_policy_dlogits(const float* dlogp, const float* probs, const int32_t* actions,
                          float* out, int ld, int64_t rows, int nA, hipStream_t st);
// values[r] = dot(a[r,:n], w[:n]) + b
int launch_rowdot(const float* a, int lda, const float* w, const float* b, float* out,
                  int64_t rows, int n, hipStream_t st);

// ---------------------------------------------------------------------------
// row-panel MLP kernels (panel.hip)
// ---------------------------------------------------------------------------
struct PanelLayer {
    const float* w;  // [n, ldw] packed, K-contiguous
    int ldw;
    const float* bias;
    const float* gamma;
    const float* beta;
    int n;
    float* z;  // pre-LayerNorm output [M, ldz] (null when nothing is kept for backward)
    int ldz;
    float* stats;  // [M, 2] mean, rstd (nullable)
    float* a;      // SiLU(LN(z)) [M, lda]
    int lda;
    // optional: the activations also into a k16 image (split.h): row r -> image row r + a3_row0, column
    // c -> image column c + a3_col0 (n and a3_col0 multiples of 4)
    char* a3;
    int a3_row0, a3_steps, a3_col0;
};
constexpr int kPanelMaxLayers = 4;
struct PanelFwdProb {
    const float* x;  // [M, ldx] input rows, or the message tensor when agg_na > 0
    int ldx, k0;
    int agg_na, agg_nb;  // message mean over the other agents computed while staging
    float* xbar;         // [M, ldx'] receives the aggregated rows (staging or agg_at; nullable)
    int m, nlayers;
    PanelLayer layer[kPanelMaxLayers];
    // by_batch > 0: a workgroup owns ALL agents of by_batch consecutive batch elements (local row
    // lr = a * by_batch + i is global row a * g_nb + blockIdx.x * by_batch + i) instead of
    // kPanelRows consecutive rows, so that the message mean over the other agents
    // (networks/message.py:5-17) can be taken inside the workgroup: agg_at = l > 0 applies it to
    // layer l's input panel in LDS (encoder -> mean -> decoder in ONE launch) and stores the
    // aggregated rows to xbar [M, ld_xbar].
    int by_batch, g_na, g_nb, agg_at, ld_xbar;
};
struct PanelFwdBatch {
    PanelFwdProb p[2];
    int count;
    int off_panel1, off_red, off_part, off_prm;  // LDS float offsets (filled by the launcher)
    // optional (count == 1): extra workgroups behind the panel ones run the sampling kernel's
    // rows (one wave per row) - an independent small kernel riding along in the same launch
    int has_sample, panel_blocks;
    int ln_narrow_max;  // LayerNorm widths up to this use the two-column-slot row pass (launcher: 128, knob off: 0)
    SampleArgs sample;
#ifdef MARL_KERNEL_TS
    long long* ts;  // phase timestamps of one workgroup (debug builds only)
#endif
};
// the encoder -> mean -> decoder chain in one workgroup: all agents of a batch element fit a panel
int panel_chain_supported(int na, int n_msg, int threads);
int panel_supported(int k0, int n0, int n1);
int launch_panel_fwd(PanelFwdBatch& b, hipStream_t st);

// LSTM cell backward (networks/recurrent.py:19-35 differentiated): gates holds the activated
// i,f,g,o on entry and the pre-activation gradients on exit; dc holds dL/dc_t on entry and
// dL/dc_{t-1} on exit.  One element (row, unit) per call.
struct LstmBwdArgs {
    const float* dh;
    float* dc;
    float* gates;
    const float* c_prev;
    const float* c_new;
    int lddh, lddc, ldg, ldc, n;
    // optional: the k16 image (split.h) of the gate gradients [rows, 4 n] - the A operand of the image
    // GEMMs that consume them; written by the four-units-per-thread forms only (n % 4 == 0)
    char* g3;
    int g3_row0, g3_steps;
    // 1: every consumer of the gate gradients reads the image - the fp32 copy is not written (the
    // four-units-per-thread forms only; callers set it only where the image is certain to be written)
    int skip_f32;
};
struct LstmBwdBatch {
    LstmBwdArgs a[2];
    int64_t rows;
    int vec4;  // four consecutive units per thread (16-byte accesses; required for the images)
};
inline bool lstm_bwd_vec4_ok(const LstmBwdArgs& a) {
    return (a.n & 3) == 0 && (a.lddh & 3) == 0 && (a.lddc & 3) == 0 && (a.ldg & 3) == 0 && (a.ldc & 3) == 0;
}
// element (r, u) with dL/dh given
__device__ __forceinline__ void lstm_cell_bwd_at(const LstmBwdArgs& A, int64_t r, int u, float dhv);
__device__ __forceinline__ void lstm_cell_bwd_elem(const LstmBwdArgs& A, int64_t rows, int64_t idx) {
    const int n = A.n;
    if (idx >= rows * n) return;
    const int64_t r = idx / n;
    const int u = (int)(idx % n);
    lstm_cell_bwd_at(A, r, u, A.dh[r * A.lddh + u]);
}
__device__ __forceinline__ void lstm_cell_bwd_at(const LstmBwdArgs& A, int64_t r, int u, float dhv) {
    const int n = A.n;
    float* g = A.gates + r * A.ldg + u;
    const float gi = g[0], gf = g[n], gg = g[2 * n], go = g[3 * n];
    const float tc = tanh_fast(A.c_new[r * A.ldc + u]);
    const float dcv = dhv * go * (1.0f - tc * tc) + A.dc[r * A.lddc + u];
    g[0] = dcv * gg * gi * (1.0f - gi);
    g[n] = dcv * A.c_prev[r * A.ldc + u] * gf * (1.0f - gf);
    g[2 * n] = dcv * gi * (1.0f - gg * gg);
    g[3 * n] = dhv * tc * go * (1.0f - go);
    A.dc[r * A.lddc + u] = dcv * gf;
}

// four consecutive units (row r, units u .. u + 3, u % 4 == 0) given their dL/dh; same arithmetic
__device__ __forceinline__ void lstm_cell_bwd_at4(const LstmBwdArgs& A, int64_t r, int u, const float (&dhv)[4]) {
    const int n = A.n;
    float* g = A.gates + r * A.ldg + u;
    const float4 gi4 = *reinterpret_cast<const float4*>(g), gf4 = *reinterpret_cast<const float4*>(g + n),
                 gg4 = *reinterpret_cast<const float4*>(g + 2 * n), go4 = *reinterpret_cast<const float4*>(g + 3 * n);
    const float4 cn4 = *reinterpret_cast<const float4*>(A.c_new + r * A.ldc + u),
                 cp4 = *reinterpret_cast<const float4*>(A.c_prev + r * A.ldc + u),
                 dc4 = *reinterpret_cast<const float4*>(A.dc + r * A.lddc + u);
    const float gi[4] = {gi4.x, gi4.y, gi4.z, gi4.w}, gf[4] = {gf4.x, gf4.y, gf4.z, gf4.w};
    const float gg[4] = {gg4.x, gg4.y, gg4.z, gg4.w}, go[4] = {go4.x, go4.y, go4.z, go4.w};
    const float cn[4] = {cn4.x, cn4.y, cn4.z, cn4.w}, cp[4] = {cp4.x, cp4.y, cp4.z, cp4.w};
    const float dcin[4] = {dc4.x, dc4.y, dc4.z, dc4.w};
    float o[5][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float tc = tanh_fast(cn[q]);
        const float dcv = dhv[q] * go[q] * (1.0f - tc * tc) + dcin[q];
        o[0][q] = dcv * gg[q] * gi[q] * (1.0f - gi[q]);
        o[1][q] = dcv * cp[q] * gf[q] * (1.0f - gf[q]);
        o[2][q] = dcv * gi[q] * (1.0f - gg[q] * gg[q]);
        o[3][q] = dhv[q] * tc * go[q] * (1.0f - go[q]);
        o[4][q] = dcv * gf[q];
    }
    if (!(A.g3 && A.skip_f32)) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<float4*>(g + k * n) = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
    }
    *reinterpret_cast<float4*>(A.dc + r * A.lddc + u) = make_float4(o[4][0], o[4][1], o[4][2], o[4][3]);
    if (A.g3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col = k * n + u;
            img_store4(A.g3 + img_off(A.g3_row0 + r, col >> 4, A.g3_steps), col, o[k][0], o[k][1], o[k][2], o[k][3]);
        }
    }
}
__device__ __forceinline__ void lstm_cell_bwd_elem4(const LstmBwdArgs& A, int64_t rows, int64_t idx) {
    const int n4 = A.n >> 2;
    if (idx >= rows * n4) return;
    const int64_t r = idx / n4;
    const int u = (int)(idx - r * n4) * 4;
    const float4 d = *reinterpret_cast<const float4*>(A.dh + r * A.lddh + u);
    const float dhv[4] = {d.x, d.y, d.z, d.w};
    lstm_cell_bwd_at4(A, r, u, dhv);
}

// backward of the same chain, layers listed from the LAST (output side) to the FIRST
struct PanelBwdLayer {
    const float* z;  // saved pre-LayerNorm activations [M, ldz]
    int ldz;
    const float* stats;  // [M, 2]
    const float* gamma;
    const float* beta;
    int n;      // width of this layer's output
    float* dz;  // out: d loss / d z [M, lddz] (kept for the weight gradients)
    int lddz;
    float* part;      // out: [gridDim.x][2][n] partial dgamma | dbeta of this launch
    const float* wt;  // transposed weights [k_in, ldwt] (ldwt >= pad4(n)): dX = dz * W
    int ldwt;
    int k_in;  // width of this layer's input
};
struct PanelBwdProb {
    const float* da;  // d loss / d (SiLU output of layer[0]) [M, ldda]
    int ldda;
    const float* da2;  // optional second addend of da [M, ldda2] (summed while staging)
    int ldda2;
    // agg_na > 0: da is the gradient of the message MEAN; the mean over the other agents
    // (self-adjoint, networks/message.py:5-17) is applied while staging (rows are a * agg_nb + b)
    int agg_na, agg_nb;
    int m, nlayers;
    PanelBwdLayer layer[kPanelMaxLayers];
    float* dx;  // out: d loss / d (input of the first layer) [M, lddx]
    int lddx, accumulate;
    // by_batch > 0: a workgroup owns all agents of by_batch batch elements (see PanelFwdProb), and
    // agg_at = l > 0 applies the (self-adjoint) message mean to the gradient panel entering layer
    // l: decoder backward -> mean -> encoder backward of the previous step in ONE launch
    int by_batch, g_na, g_nb, agg_at;
    // has_cellb: the rows of dx are complete after this kernel's update (dx = dL/dh of the belief
    // cell): its elementwise backward runs in the epilogue, on the value still in registers
    int has_cellb;
    LstmBwdArgs cellb;
    int off_e, off_prm, off_colp, off_part, off_rowmap;  // LDS float offsets (filled by the launcher)
    int tail_lds;  // 1: the final dX is finished row-wise from an LDS panel (filled by the launcher)
    // optional: extra workgroups behind the panel ones run one LSTM cell's elementwise backward
    // (an independent memory-bound kernel riding along with this latency-bound one)
    int has_cell, panel_blocks;
    LstmBwdArgs cell;
    int64_t cell_rows;
    int cell_vec4;  // (filled by the launcher) the riding-along cell runs four units per thread
    // reported by the launcher: the gate-gradient images asked for (cellb.g3 / cell.g3) are written
    int cellb_img_done, cell_img_done;
#ifdef MARL_KERNEL_TS
    long long* ts;  // phase timestamps of one workgroup (debug builds only)
#endif
};
int panel_bwd_blocks(int m);
int panel_chain_blocks(int na, int nb);  // workgroups of a by_batch launch
int panel_chain_by_batch(int na);        // batch elements a chained workgroup owns (panel rows / na)
int launch_panel_bwd(PanelBwdProb& p, hipStream_t st);

// ---------------------------------------------------------------------------
// CNN data movement (cnn.hip)
// ---------------------------------------------------------------------------
// Patch gather fused with the first layer's im2col: img [Nb,Cimg,H,W], pos int32 [R,2]
// -> cols [R * P, ldk], k = (kh*3+kw)*cin + ci, P = oh*ow, oh = (f-1)/2+1.
// ---------------------------------------------------------------------------
// Fused per-step CNN forward (cnn.hip): gather + [im2col -> conv -> GroupNorm -> SiLU] x L
// ---------------------------------------------------------------------------
// In-kernel phase timestamps (make EXTRA=-DMARL_KERNEL_TS): lane 0 of every wave of one
// workgroup records wall_clock64() (100 MHz) at each MARL_TS() in kernel order.
#ifdef MARL_KERNEL_TS
#define MARL_TS_DECL(ptr) long long* ts_ = (ptr); int tsi_ = 0
#define MARL_TS()                                                                         \
    if (ts_ && (threadIdx.x & 63) == 0 && blockIdx.x == 37 && blockIdx.y == 0 && tsi_ < 48) \
    ts_[(threadIdx.x >> 6) * 48 + tsi_++] = wall_clock64()
int ts_begin(long long** dev, int call);                 // returns 1 when this call records
void ts_report(const char* tag, long long* dev, int waves);
#else
#define MARL_TS_DECL(ptr)
#define MARL_TS()
#endif

// division by a launch-invariant divisor: q = umulhi(n, m), exact for n, d < 65536
struct FDiv {
    uint32_t d, m;
};
inline FDiv make_fdiv(int d) { return FDiv{(uint32_t)d, (uint32_t)(((1ull << 32) / (uint64_t)d) + 1)}; }
__device__ __forceinline__ int fdiv(int n, FDiv f) {
    return f.d == 1 ? n : (int)__umulhi((uint32_t)n, f.m);
}

struct CnnFwdLayer {
    const float *w, *bias, *gamma, *beta;  // packed conv weight [cout][ldk] (k = tap * cin + ci)
    float* cols;                           // [R * P][ldk] im2col rows kept for backward (or null)
    float* z;                              // [R * P][cout] conv output kept for backward (or null)
    float* gst;                            // [R][G][2] mean / rstd (or null)
    int cin, cout, G, hin, hout, P, K, ldk;
    // optional: the weight in MFMA-fragment order (cnn_fwd3): [cout / 16][K / 16][64 lanes][4], lane
    // (quad, l16) = row 16 nt + l16, k = 16 kk + 4 quad .. + 3 - one K step of a tile is 1 KB contiguous
    const float* wfrag;
};
struct CnnFwdArgs {
    const void* img;     // [Nb][c_img][H][W] float or uint8
    const float* obs;    // standalone step API: [R][c_img][f][f] patches (img unused)
    const int32_t* pos;  // [R][2]
    int img_u8;
    int64_t rows;
    int nb, c_img, H, W, f, L;
    CnnFwdLayer layer[MARL_MAX_CNN_LAYERS];
    float* u;  // [R][ldu], feature index c * P_last + pos (the reference's NCHW flatten)
    int ldu;
    // optional (cnn_fwd2 nets whose last layer has 4 positions): the features also into the k16 image of U
    char* u3;
    int u3_row0, u3_steps;
    // filled by the launcher
    int rb;                       // patches per workgroup
    int off_b0, off_b1, off_stat; // LDS float offsets: even / odd layers' output, statistics
    FDiv dP[MARL_MAX_CNN_LAYERS], dhout[MARL_MAX_CNN_LAYERS], dcin[MARL_MAX_CNN_LAYERS],
        dcpg[MARL_MAX_CNN_LAYERS], dG[MARL_MAX_CNN_LAYERS], dNT[MARL_MAX_CNN_LAYERS],
        dc4o[MARL_MAX_CNN_LAYERS];
    FDiv dpe, dff, df, dE;
#ifdef MARL_KERNEL_TS
    long long* ts;  // phase timestamps of one workgroup (debug builds only)
#endif
};
int cnn_fwd_supported(const CnnFwdArgs& a);
int cnn_fwd_writes_image(const CnnFwdArgs& a);  // the launch selected for these shapes honours a.u3
int launch_cnn_fwd(CnnFwdArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------
// Fused CNN layer backward (cnn.hip): dZ_l -> [transposed conv] -> dA_{l-1} -> [GroupNorm +
// SiLU backward of layer l-1] -> dZ_{l-1}, plus the per-workgroup dgamma / dbeta partials
// ---------------------------------------------------------------------------
struct CnnDgradArgs {
    const float* dz;     // [rows * P][cout]      gradient of layer l's conv output
    const float* wt;     // [9 * cin][ldwt]       layer l's weight, transposed (k = tap * cin + ci)
    const float* zin;    // [rows * Pin][cin]     layer l-1's conv output (pre-norm)
    const float* gst;    // [rows][G][2]          layer l-1's mean / rstd
    const float *gamma, *beta;  // layer l-1's affine
    float* dzin;         // [rows * Pin][cin]     out: gradient of layer l-1's conv output
    float* part;         // [workgroups][2 * cin] out: dgamma | dbeta partial sums
    int64_t rows;
    int ldwt, cin, cout, hin, hout, P, Pin, G;
    // filled by the launcher
    int rb, MT, NT, off_zin, off_da, off_perm, off_stat, off_gsum;
    int tbeg[17];  // row-tile range of wave slot s: [tbeg[s], tbeg[s + 1])
    FDiv dc4o, dc4i, dPin, dcpg, dG;
    // optional (the launch that produces dZ_0, w0_part != null): layer 0's weight / bias gradient formed in the same
    // launch from the dZ_0 panel and the raw patch - dZ_0 is then not written when dzin is null
    const void* img;       // image batch [nb][c_img][H][W] float or uint8
    const int32_t* pos;    // [rows][2] patch positions (row r reads image r % nb)
    float* w0_part;        // [workgroups][cin * K0] partial slabs (cin = layer 0's output channels here)
    float* w0_bpart;       // [workgroups][cin]
    int w0;                // 1: this launch also forms layer 0's weight gradient (decides the LDS plan)
    int img_u8, nb, c_img, H, W, cin0, f0, K0;
    int off_pix, pix_per, cs0;  // filled by the launcher
    FDiv dpe0, dff0, df0, dhin0;
#ifdef MARL_KERNEL_TS
    long long* ts;
#endif
};
int cnn_dgrad_supported(const CnnDgradArgs& a);
int cnn_dgrad_blocks(const CnnDgradArgs& a);      // partial rows the launch writes (persistent grid)
int cnn_dgrad_blocks_max(const CnnDgradArgs& a);  // its device-independent upper bound
int launch_cnn_dgrad(CnnDgradArgs& a, hipStream_t st);
int cnn_dgrad_w0_ok(const CnnDgradArgs& a, int cin0, int f0);  // (see w0_part)

// ---------------------------------------------------------------------------
// Convolution weight gradient straight from the activations (cnn.hip): for every patch,
//   dW_l[co][tap * cin + ci] += sum over output positions of dZ_l[pos][co] * A_{l-1}[in(pos, tap)][ci]
// with A_{l-1} = SiLU(GroupNorm(Z_{l-1})) recomputed from the saved pre-norm output (or the raw
// image patch for the first layer) - the im2col rows never exist in HBM.
// ---------------------------------------------------------------------------
struct CnnWgradArgs {
    const float* dz;      // [rows * P][cout]   gradient of this layer's conv output
    const void* img;      // first layer: image batch [nb][c_img][H][W] float or uint8
    const int32_t* pos;   // first layer: [rows][2] patch positions (row r reads image r % nb)
    const float* zin;     // deeper layers: [rows * hin * hin][cin] pre-norm output of the layer below
    const float* gst;     // deeper layers: [rows][G][2] mean / rstd of the layer below
    const float *gamma, *beta;  // affine of the layer below
    float* part_w;        // [blocks][cout * K] per-workgroup partial sums (k = tap * cin + ci)
    float* part_b;        // [blocks][cout]     per-workgroup partial bias gradient
    int64_t rows;
    int first, img_u8, nb, c_img, H, W;
    int cin, cout, hin, hout, P, G, K;
    // filled by the launcher
    int rb, nchunks, blocks;      // patches per chunk, chunks, persistent workgroups (grid.x)
    int hp, cs, zs, in_per;       // padded input side / channel stride / dZ row stride / floats per patch
    int off_dz, off_tab, lds_floats;
    int in_plane;                 // (bf16x6 form) bytes of one plane of the staged input
    int nct, nkt, nkt_slab, slabs;  // 16-wide tiles of cout / K, k tiles per grid.y slab
    int tgc, tgk, ms;             // wave roles: cout-tile groups x k-tile groups x row-step interleave
    int sct, skt;                 // accumulator tiles per wave (template shape of the launch)
    int pd, pi;                   // prefetch depth per thread (template shape of the launch)
    FDiv dP, dhout, dPin, dhin, dcin, dc4o, dc4i, dpe, dff, df;
};
int cnn_wgrad_supported(const CnnWgradArgs& a);
int cnn_wgrad_blocks(const CnnWgradArgs& a);   // partial slabs the launch writes
int launch_cnn_wgrad(CnnWgradArgs& a, hipStream_t st);
// out[i * ldc + j] = sum over z < splits (fixed order) of part[z * stride + i * nj + j];
// bias (nullable): same over bpart[z * ni + i]
int launch_slab_reduce(const float* part, int64_t stride, int splits, float* c, int ldc, int ni,
                       int nj, const float* bpart, float* bias, hipStream_t st);

int launch_gather_im2col(const void* img, int img_u8, const int32_t* pos, float* cols, int ldk,
                         int na, int nb, int c_img, int cin, int H, int W, int f, hipStream_t st);
// same but from pre-gathered patches obs [R, c_img, f, f] (standalone step API)
int launch_obs_im2col(const float* obs, float* cols, int ldk, int64_t rows, int c_img, int cin,
                      int f, hipStream_t st);
// plain gather (Environment.observe): obs [R, C, f, f]
int launch_patch_gather(const float* img, const int64_t* pos, float* obs, int na, int nb, int c,
                        int H, int W, int f, hipStream_t st);
// NHWC activation [rows, hin, hin, cin] -> cols [rows * hout^2, ldk]
int launch_im2col(const float* a, float* cols, int ldk, int64_t rows, int hin, int cin,
                  hipStream_t st);
// transpose of the above: dcols -> da (gather form, deterministic)
int launch_col2im(const float* dcols, int ldk, float* da, int64_t rows, int hin, int cin,
                  hipStream_t st);

// ---------------------------------------------------------------------------
// loss (loss.hip)
// ---------------------------------------------------------------------------
struct LossArgs {
    const float* preds;   // [Ns, Na, Nb, nC]
    const float* logp;    // [Ns, Na, Nb]
    const float* values;  // [Ns, Na, Nb]
    const int64_t* y;     // [Nb]
    float* g_preds;       // [Ns*R, ld_gp]
    int ld_gp;
    float* g_logp;
    float* g_values;
    float* scalars;  // [4]
    double* adv_stats;  // [3] = n, sum, sum of squares of the advantages
    float* scratch;  // returns/adv [2][Ns*R] + partial sums
    int ns, na, nb, nc;
    float gamma;
    int phase;
};
size_t loss_scratch_floats(int ns, int na, int nb);
int launch_loss(const LossArgs& a, hipStream_t st);

}  // namespace marl
