// Host-side orchestration of the episode (EpisodeSampler.__episode_impl,
// core/episode.py:32-82) and of loss.backward() through it, plus the C ABI of
// include/marl_hip.h.  Everything here only computes buffer layouts inside the two
// caller-owned workspaces and enqueues kernels on the caller's stream.
//
// Structure of one training iteration (R = Na*Nb rows, r = a*Nb + b):
//   forward, t = 0..Ns-1 (strictly sequential: positions depend on sampled actions)
//     gather+im2col -> [conv GEMM -> GroupNorm+SiLU -> im2col]*  -> b_t  into U[t]
//     message mean -> decode MLP -> d_t into U[t];  position embedding -> lambda_t into U[t]
//     belief + action LSTM: one grouped MFMA GEMM with the cell update fused in the epilogue
//     encode MLP -> msg_{t+1};  policy hidden -> output layer + softmax + sample + move
//   after the loop (batched over all Ns*R rows, better GEMM shapes):
//     critic and prediction heads on the saved h / h^ of every step
//   backward: heads batched first (their inputs' gradients do not depend on the recurrence),
//     then the reverse-time loop over only the recurrent chain (LSTM cells, message
//     decode/encode), then every weight gradient as ONE row-contraction GEMM over all
//     Ns*R rows, the full dU, and the CNN backward batched over all steps.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace marl {

// ---------------------------------------------------------------------------
// dimensions
// ---------------------------------------------------------------------------
struct Dims {
    int na, nb, ns;
    int64_t R, NR;
    int H, W, c_img, f;
    int L;
    int ch[MARL_MAX_CNN_LAYERS + 1];
    int grp[MARL_MAX_CNN_LAYERS];
    int hw[MARL_MAX_CNN_LAYERS + 1];  // spatial size entering layer l (hw[0] = f)
    int P[MARL_MAX_CNN_LAYERS];       // output positions of layer l
    int K[MARL_MAX_CNN_LAYERS];       // 9 * cin
    int ldk[MARL_MAX_CNN_LAYERS];
    int nf, nin;
    int n_b, n_a, n_m, nm2, n_mo, n_d, nA, nC, nlb, nla;
    int ld_nin, ld_nb, ld_na, ld_nm, ld_nm2, ld_nmo, ld_nd, ld_nla, ld_nlb, ld_gb, ld_ga, ld_nC,
        ld_nA;
    int ld_dbl;  // row stride of d(decoded message | position embedding): pad4(n_mo + n_d)
};

static int make_dims(const marl_config* c, Dims& d) {
    if (!c) {
        set_error("null config");
        return MARL_EINVAL;
    }
    memset(&d, 0, sizeof(d));
    d.na = c->nb_agents;
    d.nb = c->batch;
    d.ns = c->nb_steps;
    d.H = c->img_h;
    d.W = c->img_w;
    d.c_img = c->img_c;
    d.f = c->window;
    d.L = c->cnn_layers;
    if (d.na < 1 || d.nb < 1 || d.ns < 1 || d.f < 1 || d.H <= d.f || d.W <= d.f || d.c_img < 1) {
        set_error("bad episode shape na=%d nb=%d ns=%d f=%d img=%dx%dx%d", d.na, d.nb, d.ns, d.f,
                  d.c_img, d.H, d.W);
        return MARL_EINVAL;
    }
    if (d.L < 1 || d.L > MARL_MAX_CNN_LAYERS) {
        set_error("cnn_layers %d outside [1,%d]", d.L, MARL_MAX_CNN_LAYERS);
        return MARL_ELIMIT;
    }
    d.R = (int64_t)d.na * d.nb;
    d.NR = d.R * d.ns;
    d.hw[0] = d.f;
    for (int l = 0; l <= d.L; ++l) d.ch[l] = c->cnn_ch[l];
    if (d.ch[0] < 1 || d.ch[0] > d.c_img) {
        set_error("cnn input channels %d vs image channels %d", d.ch[0], d.c_img);
        return MARL_EINVAL;
    }
    for (int l = 0; l < d.L; ++l) {
        d.grp[l] = c->cnn_groups[l];
        if (d.ch[l + 1] < 1 || d.grp[l] < 1 || d.ch[l + 1] % d.grp[l] != 0 || (d.ch[l + 1] & 3)) {
            set_error("cnn layer %d: %d channels / %d groups unsupported (channels must be a "
                      "multiple of 4 and of the group count)", l, d.ch[l + 1], d.grp[l]);
            return MARL_ELIMIT;
        }
        d.hw[l + 1] = (d.hw[l] - 1) / 2 + 1;
        d.P[l] = d.hw[l + 1] * d.hw[l + 1];
        d.K[l] = 9 * d.ch[l];
        d.ldk[l] = p4(d.K[l]);
    }
    d.nf = d.ch[d.L] * d.P[d.L - 1];
    d.n_b = c->n_b;
    d.n_a = c->n_a;
    d.n_m = c->n_m;
    d.nm2 = 2 * c->n_m;
    d.n_mo = c->n_m_o;
    d.n_d = c->n_d;
    d.nA = c->nb_action;
    d.nC = c->nb_class;
    d.nlb = c->nlb;
    d.nla = c->nla;
    if (d.n_b < 1 || d.n_a < 1 || d.n_m < 1 || d.n_mo < 1 || d.n_d < 1 || d.nA < 1 || d.nC < 1 ||
        d.nlb < 1 || d.nla < 1) {
        set_error("non-positive hidden size");
        return MARL_EINVAL;
    }
    if (d.nA > MARL_MAX_ACTIONS || d.nC > 1024 || d.nlb > 2048 || d.nla > 2048 || d.nm2 > 2048 ||
        d.n_mo > 2048 || d.n_d > 2048) {
        set_error("dimension above the supported range (nA<=16, nC<=1024, LayerNorm width<=2048)");
        return MARL_ELIMIT;
    }
    d.nin = d.nf + d.n_mo + d.n_d;
    d.ld_nin = p4(d.nin);
    d.ld_nb = p4(d.n_b);
    d.ld_na = p4(d.n_a);
    d.ld_nm = p4(d.n_m);
    d.ld_nm2 = p4(d.nm2);
    d.ld_nmo = p4(d.n_mo);
    d.ld_dbl = p4(d.n_mo + d.n_d);
    d.ld_nd = p4(d.n_d);
    d.ld_nla = p4(d.nla);
    d.ld_nlb = p4(d.nlb);
    d.ld_gb = p4(4 * d.n_b);
    d.ld_ga = p4(4 * d.n_a);
    d.ld_nC = p4(d.nC);
    d.ld_nA = p4(d.nA);
    return MARL_OK;
}

// ---------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------
enum ParamKind { PK_NONE = 0, PK_MATRIX, PK_CONV, PK_VEC };
struct ParamMeta {
    ParamKind kind;
    int n, k;  // matrix [n, k]; conv [n, ci(=k/9), 3, 3]; vec [n]
};

static ParamMeta param_meta(const Dims& d, int idx) {
    auto M = [](int n, int k) { return ParamMeta{PK_MATRIX, n, k}; };
    auto V = [](int n) { return ParamMeta{PK_VEC, n, 1}; };
    if (idx < 20) {
        const int l = idx / 4, w = idx % 4;
        if (l >= d.L) return ParamMeta{PK_NONE, 0, 0};
        if (w == 0) return ParamMeta{PK_CONV, d.ch[l + 1], d.K[l]};
        return V(d.ch[l + 1]);
    }
    switch (idx) {
        case MARL_P_POS_W: return M(d.n_d, 2);
        case MARL_P_POS_B: case MARL_P_POS_LNW: case MARL_P_POS_LNB: return V(d.n_d);
        case MARL_P_ENC_W0: return M(d.nm2, d.n_b);
        case MARL_P_ENC_B0: case MARL_P_ENC_LN0W: case MARL_P_ENC_LN0B: return V(d.nm2);
        case MARL_P_ENC_W1: return M(d.n_m, d.nm2);
        case MARL_P_ENC_B1: case MARL_P_ENC_LN1W: case MARL_P_ENC_LN1B: return V(d.n_m);
        case MARL_P_DEC_W0: return M(d.nm2, d.n_m);
        case MARL_P_DEC_B0: case MARL_P_DEC_LN0W: case MARL_P_DEC_LN0B: return V(d.nm2);
        case MARL_P_DEC_W1: return M(d.n_mo, d.nm2);
        case MARL_P_DEC_B1: case MARL_P_DEC_LN1W: case MARL_P_DEC_LN1B: return V(d.n_mo);
        case MARL_P_LB_WIH: return M(4 * d.n_b, d.nin);
        case MARL_P_LB_WHH: return M(4 * d.n_b, d.n_b);
        case MARL_P_LB_BIH: case MARL_P_LB_BHH: return V(4 * d.n_b);
        case MARL_P_LA_WIH: return M(4 * d.n_a, d.nin);
        case MARL_P_LA_WHH: return M(4 * d.n_a, d.n_a);
        case MARL_P_LA_BIH: case MARL_P_LA_BHH: return V(4 * d.n_a);
        case MARL_P_POL_W0: return M(d.nla, d.n_a);
        case MARL_P_POL_B0: case MARL_P_POL_LNW: case MARL_P_POL_LNB: return V(d.nla);
        case MARL_P_POL_W1: return M(d.nA, d.nla);
        case MARL_P_POL_B1: return V(d.nA);
        case MARL_P_CRI_W0: return M(d.nla, d.n_a);
        case MARL_P_CRI_B0: case MARL_P_CRI_LNW: case MARL_P_CRI_LNB: return V(d.nla);
        case MARL_P_CRI_W1: return M(1, d.nla);
        case MARL_P_CRI_B1: return V(1);
        case MARL_P_PRE_W0: return M(d.nlb, d.n_b);
        case MARL_P_PRE_B0: case MARL_P_PRE_LNW: case MARL_P_PRE_LNB: return V(d.nlb);
        case MARL_P_PRE_W1: return M(d.nC, d.nlb);
        case MARL_P_PRE_B1: return V(d.nC);
        default: return ParamMeta{PK_NONE, 0, 0};
    }
}

// bump allocator over a workspace, in floats, 64-float (256 B) aligned
struct Bump {
    size_t off = 0;
    size_t take(size_t n) {
        const size_t o = off;
        off += (n + 63) & ~(size_t)63;
        return o;
    }
};

// weights workspace: packed (padded) copy, transposed copy, packed gradient per matrix
struct WLayout {
    size_t wp[MARL_NPARAMS];  // packed / tight copy
    int ldp[MARL_NPARAMS];
    size_t wt[MARL_NPARAMS];  // transposed copy [k, p4(n)]
    int ldt[MARL_NPARAMS];
    size_t gp[MARL_NPARAMS];  // packed gradient (same shape as wp)
    size_t wp3[MARL_NPARAMS], wt3[MARL_NPARAMS];  // bf16x3 images of wp / wt (gemm_split.hip)
    size_t wp3k[MARL_NPARAMS], wt3k[MARL_NPARAMS];  // k16 images of wp / wt (gemm3.hip, split.h), 0 = none
    size_t wf[MARL_NPARAMS];  // conv weights in MFMA-fragment order (cnn_fwd3), 0 = none
    size_t bsum_b, bsum_a;    // b_ih + b_hh
    size_t total;
};

// The image GEMMs (gemm3.hip) take over the large products when every step's row slice starts on an
// image row block (R % 32 == 0: all BASELINE shapes) and the cells' widths keep 16-byte accesses.
// (Layout-relevant: the workspaces grow by the images - marl_workspace_sizes after a knob change.)
// Measured: round 4 (128-row tile plans only) a gain from cells of >= 128 units on, the 64-unit MNIST shapes lost 4 %
// to the image writes; round 5, with the small-batch plans (gate-split LSTM launch, 64 x 64 NT tiles), the MNIST
// shapes gain too - C2 (B = 1024) 0.786 -> 0.755 ms, the same shapes at B = 32 0.614 -> 0.528 ms: on from 64 units
// (knob g3_min_units; the parity fixtures lower it further to run this path on the odd small shapes too).
// The MODEL-only part decides the weights workspace (its k16 weight images): that layout must not move
// with the batch - an epoch's last partial batch or an odd eval batch runs on the weights packed for the
// previous one.  The batch-dependent part only picks the kernels of a call.
static bool g3_model_ok(const Dims& d) {
    const int mu = tune_get("g3_min_units", 64);
    return split_mode() && tune_get("g3", 1) != 0 && (d.n_b & 3) == 0 && (d.n_a & 3) == 0 && d.n_b >= mu &&
           d.n_a >= mu;
}
static bool g3_enabled(const Dims& d) {
    return g3_model_ok(d) && d.R % 32 == 0 && d.R >= 32;
}
// the four large weight gradients on images (gemm_tn3_kernel: 256-column tiles of A, whole 256 x 256 tiles
// or column passes, g3_tn_plan): only where that form wins - >= 32768 contraction rows (at C4's 8192 rows the
// fp32-operand kernel is 25 % faster) and at most three column passes on the B side
static bool g3_tn_enabled(const Dims& d) {
    if (tune_get("g3_lstm", 1) == 0 || tune_get("g3_tn", 1) == 0) return false;
    if (tune_get("g3_tn", 1) == 2) return true;
    if (d.nin > 640 || d.n_b > 512 || d.n_a > 512) return false;
    // (round 5: the one-launch-per-cell form wins from 8 192 rows on - C4 at 32 images per GPU: 3.62 vs 3.69 ms)
    return d.NR >= 32768 || (g3_tn_cell_ok(4 * d.n_b, d.nin, d.n_b, d.NR) && g3_tn_cell_ok(4 * d.n_a, d.nin, d.n_a, d.NR));
}

// conv weights whose tiles are whole (16 output channels x 16-deep K steps) get a fragment-order copy
static bool conv_frag_ok(const ParamMeta& m) { return m.kind == PK_CONV && m.n % 16 == 0 && m.k % 16 == 0; }

static void make_wlayout(const Dims& d, WLayout& w) {
    Bump b;
    for (int i = 0; i < MARL_NPARAMS; ++i) {
        const ParamMeta m = param_meta(d, i);
        w.wp[i] = w.wt[i] = w.gp[i] = w.wp3[i] = w.wt3[i] = w.wf[i] = w.wp3k[i] = w.wt3k[i] = 0;
        w.ldp[i] = w.ldt[i] = 0;
        if (m.kind == PK_MATRIX || m.kind == PK_CONV) {
            w.ldp[i] = p4(m.k);
            w.wp[i] = b.take((size_t)m.n * w.ldp[i]);
            w.ldt[i] = p4(m.n);
            w.wt[i] = b.take((size_t)m.k * w.ldt[i]);
            w.gp[i] = b.take((size_t)m.n * w.ldp[i]);
            if (m.kind == PK_MATRIX) {
                w.wp3[i] = b.take(split_image_floats(m.n, m.k));
                w.wt3[i] = b.take(split_image_floats(m.k, m.n));
                if (g3_model_ok(d)) {
                    w.wp3k[i] = b.take(img_bytes(m.n, m.k) / sizeof(float));
                    w.wt3k[i] = b.take(img_bytes(m.k, m.n) / sizeof(float));
                }
            } else if (conv_frag_ok(m)) {
                b.take(16);  // (offset 0 means "none")
                w.wf[i] = b.take((size_t)m.n * m.k);
            }
        } else if (m.kind == PK_VEC) {
            w.ldp[i] = m.n;
            w.wp[i] = b.take((size_t)m.n);
        }
    }
    w.bsum_b = b.take((size_t)4 * d.n_b);
    w.bsum_a = b.take((size_t)4 * d.n_a);
    w.total = b.off;
}

// episode workspace
struct SBuf {  // per-step buffer: base offset + stride between steps (0 when shared)
    size_t off = 0, stride = 0;
    size_t at(int t) const { return off + stride * (size_t)t; }
};

struct ELayout {
    // all-steps state
    size_t POS;  // int32 [(Ns+1)][R][2]
    size_t H, C, HC, CC, MSG;  // [(Ns+1)][R][ld]
    size_t PROBS, ACT;         // [Ns][R][nA], int32 [Ns][R]
    // per step
    SBuf COLS[MARL_MAX_CNN_LAYERS], Z[MARL_MAX_CNN_LAYERS], GST[MARL_MAX_CNN_LAYERS],
        A[MARL_MAX_CNN_LAYERS];
    SBuf U, MBAR, ZD1, STD1, AD1, ZD2, STD2, NPOS, ZPOS, STPOS, GB, GA, ZE1, STE1, AE1, ZE2, STE2,
        ZP1, STP1, AP1;
    // batched heads
    size_t ZC1, STC1, AC1, ZQ1, STQ1, AQ1;
    // backward only
    size_t GPRED, DLOG, DVAL, DAQ1, DAC1, DAP1, DH, DHC, DC, DCC, DDBAR, DDBAR2, DAD1, DMBAR, DZE2, DAE1, DU,
        DZPOS, BTMP, PLN[4];
    size_t DZ[MARL_MAX_CNN_LAYERS], DCOLS[MARL_MAX_CNN_LAYERS], DA[MARL_MAX_CNN_LAYERS];
    size_t GB3, GA3;  // k16 images of the gate gradients [Ns*R, 4 n] (g3 only; float offsets)
    size_t U3, H3, HC3;  // images of U (train: every step, else one slice), H / H^ [(Ns+1) R]
    size_t u3_stride_rows;  // image rows between the U3 slices of consecutive steps (0: one shared slice)
    bool g3;
    size_t PART, TNS, LOSS;
    size_t RED, red_floats;  // scratch of the deferred-reduction queue (sum over every use)
    size_t part_floats, tns_bytes, loss_floats;
    size_t total;
    // which fused CNN kernels cover this shape (decides what is kept for backward)
    bool fused_fwd;                          // one launch for the whole extractor
    bool wgrad_ok[MARL_MAX_CNN_LAYERS];      // weight gradient from the activations (no im2col rows)
    bool dgrad_ok[MARL_MAX_CNN_LAYERS];      // fused transposed conv + GroupNorm backward (l >= 1)
};

static CnnFwdArgs cnn_fwd_shape(const Dims& d) {
    CnnFwdArgs a{};
    a.rows = d.R;
    a.nb = d.nb;
    a.c_img = d.c_img;
    a.H = d.H;
    a.W = d.W;
    a.f = d.f;
    a.L = d.L;
    for (int l = 0; l < d.L; ++l)
        a.layer[l] = CnnFwdLayer{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                 d.ch[l], d.ch[l + 1], d.grp[l], d.hw[l], d.hw[l + 1], d.P[l], d.K[l],
                                 d.ldk[l], nullptr};
    return a;
}
static CnnWgradArgs cnn_wgrad_shape(const Dims& d, int l) {
    CnnWgradArgs g{};
    g.rows = d.NR;
    g.first = l == 0;
    g.nb = d.nb;
    g.c_img = d.c_img;
    g.H = d.H;
    g.W = d.W;
    g.cin = d.ch[l];
    g.cout = d.ch[l + 1];
    g.hin = d.hw[l];
    g.hout = d.hw[l + 1];
    g.P = d.P[l];
    g.G = l > 0 ? d.grp[l - 1] : 1;
    g.K = d.K[l];
    return g;
}
static CnnDgradArgs cnn_dgrad_shape(const Dims& d, int l) {
    CnnDgradArgs g{};
    g.rows = d.NR;
    g.cin = d.ch[l];
    g.cout = d.ch[l + 1];
    g.hin = d.hw[l];
    g.hout = d.hw[l + 1];
    g.P = d.P[l];
    g.Pin = d.P[l - 1];
    g.G = d.grp[l - 1];
    // the launch that produces dZ_0 also forms layer 0's weight gradient when the shapes allow it (cnn.hip)
    if (l == 1 && cnn_dgrad_w0_ok(g, d.ch[0], d.f) && cnn_wgrad_supported(cnn_wgrad_shape(d, 0))) {
        g.w0 = 1;
        g.cin0 = d.ch[0];
        g.f0 = d.f;
        g.K0 = d.K[0];
        g.nb = d.nb;
        g.c_img = d.c_img;
        g.H = d.H;
        g.W = d.W;
        if (!cnn_dgrad_supported(g)) g.w0 = 0;  // (the pixel images do not fit next to the panels: separate launches)
    }
    return g;
}

static void make_elayout(const Dims& d, int train, ELayout& e) {
    Bump b;
    const size_t R = (size_t)d.R, NR = (size_t)d.NR, S1 = (size_t)d.ns + 1;
    e.POS = b.take(S1 * R * 2);
    e.H = b.take(S1 * R * d.ld_nb);
    e.C = b.take(S1 * R * d.ld_nb);
    e.HC = b.take(S1 * R * d.ld_na);
    e.CC = b.take(S1 * R * d.ld_na);
    e.MSG = b.take(S1 * R * d.ld_nm);
    e.PROBS = b.take(NR * d.nA);
    e.ACT = b.take(NR);
    const size_t steps = train ? (size_t)d.ns : 1;
    // per-step slices are packed back to back so that [Ns][rows][ld] is also [Ns*rows][ld]
    auto per = [&](SBuf& s, size_t n) {
        s.off = b.take(n * steps);
        s.stride = train ? n : 0;
    };
    auto once = [&](SBuf& s, size_t n) {  // one slice shared by every step (forward scratch)
        s.off = b.take(n);
        s.stride = 0;
    };
    e.fused_fwd = cnn_fwd_supported(cnn_fwd_shape(d)) != 0;
    for (int l = 0; l < MARL_MAX_CNN_LAYERS; ++l) e.wgrad_ok[l] = e.dgrad_ok[l] = false;
    for (int l = 0; l < d.L; ++l) {
        e.wgrad_ok[l] = cnn_wgrad_supported(cnn_wgrad_shape(d, l)) != 0;
        if (l > 0) e.dgrad_ok[l] = cnn_dgrad_supported(cnn_dgrad_shape(d, l)) != 0;
    }
    for (int l = 0; l < d.L; ++l) {
        // im2col rows: kept for every step only when the weight gradient needs them (shapes the
        // activation-based kernel does not cover); one scratch slice for the unfused forward
        if (train && !e.wgrad_ok[l])
            per(e.COLS[l], R * d.P[l] * d.ldk[l]);
        else if (!e.fused_fwd)
            once(e.COLS[l], R * d.P[l] * d.ldk[l]);
        per(e.Z[l], R * d.P[l] * d.ch[l + 1]);
        per(e.GST[l], R * d.grp[l] * 2);
        if (l + 1 < d.L && !e.fused_fwd) once(e.A[l], R * d.P[l] * d.ch[l + 1]);
    }
    per(e.U, R * d.ld_nin);
    per(e.MBAR, R * d.ld_nm);
    per(e.ZD1, R * d.ld_nm2);
    per(e.STD1, R * 2);
    per(e.AD1, R * d.ld_nm2);
    per(e.ZD2, R * d.ld_nmo);
    per(e.STD2, R * 2);
    per(e.NPOS, R * 4);
    per(e.ZPOS, R * d.ld_nd);
    per(e.STPOS, R * 2);
    per(e.GB, R * d.ld_gb);
    per(e.GA, R * d.ld_ga);
    per(e.ZE1, R * d.ld_nm2);
    per(e.STE1, R * 2);
    per(e.AE1, R * d.ld_nm2);
    per(e.ZE2, R * d.ld_nm);
    per(e.STE2, R * 2);
    per(e.ZP1, R * d.ld_nla);
    per(e.STP1, R * 2);
    per(e.AP1, R * d.ld_nla);
    e.ZC1 = b.take(NR * d.ld_nla);
    e.STC1 = b.take(NR * 2);
    e.AC1 = b.take(NR * d.ld_nla);
    e.ZQ1 = b.take(NR * d.ld_nlb);
    e.STQ1 = b.take(NR * 2);
    e.AQ1 = b.take(NR * d.ld_nlb);
    e.part_floats = e.tns_bytes = e.loss_floats = 0;
    e.g3 = g3_enabled(d);
    e.GB3 = e.GA3 = e.U3 = e.H3 = e.HC3 = 0;
    e.u3_stride_rows = 0;
    if (e.g3) {
        e.u3_stride_rows = train ? R : 0;
        e.U3 = b.take(img_bytes((int64_t)(train ? NR : R), d.nin) / sizeof(float));
        e.H3 = b.take(img_bytes((int64_t)(S1 * R), d.n_b) / sizeof(float));
        e.HC3 = b.take(img_bytes((int64_t)(S1 * R), d.n_a) / sizeof(float));
    }
    if (train && e.g3) {
        e.GB3 = b.take(img_bytes((int64_t)NR, 4 * d.n_b) / sizeof(float));
        e.GA3 = b.take(img_bytes((int64_t)NR, 4 * d.n_a) / sizeof(float));
    }
    if (train) {
        e.GPRED = b.take(NR * d.ld_nC);
        e.DLOG = b.take(NR * d.ld_nA);
        e.DVAL = b.take(NR * 4);
        e.DAQ1 = b.take(NR * d.ld_nlb);
        e.DAC1 = b.take(NR * d.ld_nla);
        e.DAP1 = b.take(NR * d.ld_nla);
        e.DH = b.take(S1 * R * d.ld_nb);
        e.DHC = b.take(S1 * R * d.ld_na);
        e.DC = b.take(R * d.ld_nb);
        e.DCC = b.take(R * d.ld_na);
        e.DDBAR = b.take(NR * d.ld_dbl);   // + the position-embedding columns of dU (belief cell)
        e.DDBAR2 = b.take(NR * d.ld_dbl);  // the action cell's share of the same columns
        e.DAD1 = b.take(NR * d.ld_nm2);
        e.DMBAR = b.take(R * d.ld_nm);
        e.DZE2 = b.take(NR * d.ld_nm);
        e.DAE1 = b.take(NR * d.ld_nm2);
        e.DU = b.take(NR * d.ld_nin);
        e.DZPOS = b.take(NR * d.ld_nd);
        int maxg = 4 * (d.n_b > d.n_a ? d.n_b : d.n_a);
        e.BTMP = b.take((size_t)maxg);
        {   // per-step LayerNorm affine partials of the panel backward kernels
            int blocks = panel_bwd_blocks((int)d.R);
            if (panel_chain_blocks(d.na, d.nb) > blocks) blocks = panel_chain_blocks(d.na, d.nb);
            const size_t nblk = (size_t)blocks * d.ns * 2;
            e.PLN[0] = b.take(nblk * d.n_mo);  // decode LN1
            e.PLN[1] = b.take(nblk * d.nm2);   // decode LN0
            e.PLN[2] = b.take(nblk * d.n_m);   // encode LN1
            e.PLN[3] = b.take(nblk * d.nm2);   // encode LN0
        }
        for (int l = 0; l < d.L; ++l) {
            e.DZ[l] = b.take(NR * d.P[l] * d.ch[l + 1]);
            // only the unfused layer backward materialises dCOLS_l and dA_{l-1}
            e.DCOLS[l] = (l > 0 && !e.dgrad_ok[l]) ? b.take(NR * d.P[l] * d.ldk[l]) : 0;
            e.DA[l] = (l + 1 < d.L && !e.dgrad_ok[l + 1]) ? b.take(NR * d.P[l] * d.ch[l + 1]) : 0;
        }
        // scratch sizes: maxima over every use
        size_t part = 0, tns = 0, red = 0;
        auto upd_part = [&](int64_t blocks, int n) {
            const size_t v = (size_t)blocks * 2 * n;
            part = v > part ? v : part;
            red += ((v + 63) & ~(size_t)63) + 64 * 2 * (size_t)n + 64;  // + stage-1 temporaries
        };
        auto upd_tn = [&](int ni, int nj, int64_t rows) {
            const size_t v = gemm_tn_scratch_bytes(ni, nj, rows);
            tns = v > tns ? v : tns;
            red += v / sizeof(float) + 64 * ((size_t)ni * nj + ni) + 192;
        };
        const int64_t nr = d.NR, r = d.R;
        upd_part(ln_bwd_blocks(nr, d.nlb), d.nlb);
        upd_part(ln_bwd_blocks(nr, d.nla), d.nla);
        upd_part(ln_bwd_blocks(nr, d.n_d), d.n_d);
        upd_part(ln_bwd_blocks(r, d.nm2), d.nm2);
        upd_part(ln_bwd_blocks(r, d.n_mo), d.n_mo);
        upd_part(ln_bwd_blocks(r, d.n_m), d.n_m);
        for (int l = 0; l < d.L; ++l) upd_part(gn_bwd_blocks(nr, d.ch[l + 1]), d.ch[l + 1]);
        for (int l = 1; l < d.L; ++l)  // fused layer backward: one partial row per workgroup
            if (e.dgrad_ok[l]) upd_part(cnn_dgrad_blocks_max(cnn_dgrad_shape(d, l)), d.ch[l]);
        upd_tn(d.nC, d.nlb, nr);
        upd_tn(d.nlb, d.n_b, nr);
        upd_tn(1, d.nla, nr);
        upd_tn(d.nla, d.n_a, nr);
        upd_tn(d.nA, d.nla, nr);
        upd_tn(d.n_mo, d.nm2, nr);
        upd_tn(d.nm2, d.n_m, nr);
        upd_tn(d.n_m, d.nm2, nr);
        upd_tn(d.nm2, d.n_b, nr);
        upd_tn(4 * d.n_b, d.nin, nr);
        upd_tn(4 * d.n_b, d.n_b, nr);
        upd_tn(4 * d.n_a, d.nin, nr);
        upd_tn(4 * d.n_a, d.n_a, nr);
        if (e.g3) {  // the same four products on images (gemm_tn3_kernel) have their own split plan
            const int tn3[4][2] = {{4 * d.n_b, d.nin}, {4 * d.n_b, d.n_b}, {4 * d.n_a, d.nin}, {4 * d.n_a, d.n_a}};
            for (const auto& q : tn3) {
                const size_t v = g3_tn_scratch_bytes(q[0], q[1], nr);
                tns = v > tns ? v : tns;
            }
            // (both products of a cell in one launch: their slabs side by side)
            const int cell[2][3] = {{4 * d.n_b, d.nin, d.n_b}, {4 * d.n_a, d.nin, d.n_a}};
            for (const auto& q : cell)
                if (g3_tn_cell_ok(q[0], q[1], q[2], nr)) {
                    const size_t v = g3_tn_cell_scratch_bytes(q[0], q[1], q[2], nr);
                    tns = v > tns ? v : tns;
                }
        }
        upd_tn(d.n_d, 2, nr);
        if (d.L > 1 && e.dgrad_ok[1]) {  // layer 0's weight gradient out of the dZ_0 launch: one small slab per workgroup
            const CnnDgradArgs g1 = cnn_dgrad_shape(d, 1);
            if (g1.w0) {
                int64_t nb_ = cnn_dgrad_blocks_max(g1);
                if (nb_ > 2048) nb_ = 2048;  // (persistent grid: occupancy <= 8 workgroups x 256 CUs)
                const size_t v = (size_t)nb_ * ((size_t)d.ch[1] * d.K[0] + d.ch[1]) * sizeof(float);
                tns = v > tns ? v : tns;
                red += v / sizeof(float) + 64 * ((size_t)d.ch[1] * (d.K[0] + 1)) + 192;
            }
        }
        for (int l = 0; l < d.L; ++l) {
            if (e.wgrad_ok[l]) {  // per-workgroup partial slabs of the activation-based kernel
                const size_t v = (size_t)cnn_wgrad_blocks(cnn_wgrad_shape(d, l)) *
                                 ((size_t)d.ch[l + 1] * d.K[l] + d.ch[l + 1]) * sizeof(float);
                tns = v > tns ? v : tns;
                red += v / sizeof(float) + 64 * ((size_t)d.ch[l + 1] * (d.K[l] + 1)) + 192;
            } else {
                upd_tn(d.ch[l + 1], d.K[l], nr * d.P[l]);
            }
        }
        e.part_floats = part;
        e.tns_bytes = tns;
        e.PART = b.take(part);
        e.TNS = b.take(tns / sizeof(float) + 16);
        {   // stage-1 temporaries of the four per-step LayerNorm reductions (their parts are PLN)
            const int pn[4] = {d.n_mo, d.nm2, d.n_m, d.nm2};
            for (int n : pn) red += 64 * 2 * (size_t)n + 64;
        }
        e.red_floats = red + 4096;
        e.RED = b.take(e.red_floats);
    }
    e.loss_floats = loss_scratch_floats(d.ns, d.na, d.nb);
    e.LOSS = b.take(e.loss_floats);
    e.total = b.off;
}

// ---------------------------------------------------------------------------
// context shared by the forward / backward drivers
// ---------------------------------------------------------------------------
struct Ctx {
    Dims d;
    WLayout w;
    ELayout e;
    float* W;  // weights workspace
    float* E;  // episode workspace
    hipStream_t st;
    int train;
    RedQueue* rq = nullptr;  // backward: deferred reductions (null: every reduction launches at once)
    TnQueue* tq = nullptr;   // backward: small weight gradients collected for one launch (needs rq)
    bool defer_slabs = false;  // also the split-K / conv weight-gradient slabs (tens of MB each)
    size_t defer_small = 0;    // ... or only those of at most this many bytes (they stay in the L2)
    bool defer_this(size_t slab_bytes) const { return defer_slabs || (defer_small && slab_bytes <= defer_small); }
    bool u3_by_producers = false;  // the kernels that write U also write its image (no separate pass)
    int u3_row(int t) const { return (int)(e.u3_stride_rows * (size_t)t); }

    const char* wp3k(int i) const { return reinterpret_cast<const char*>(W + w.wp3k[i]); }
    const char* wt3k(int i) const { return reinterpret_cast<const char*>(W + w.wt3k[i]); }
    char* img(size_t off) const { return reinterpret_cast<char*>(E + off); }
    const float* wp(int i) const { return W + w.wp[i]; }
    const float* wt(int i) const { return W + w.wt[i]; }
    float* gp(int i) const { return W + w.gp[i]; }
    float* at(size_t off) const { return E + off; }
    float* at(const SBuf& s, int t) const { return E + s.at(train ? t : 0); }
    // all-steps buffers, slice t
    float* Hs(int t) const { return E + e.H + (size_t)t * d.R * d.ld_nb; }
    float* Cs(int t) const { return E + e.C + (size_t)t * d.R * d.ld_nb; }
    float* HCs(int t) const { return E + e.HC + (size_t)t * d.R * d.ld_na; }
    float* CCs(int t) const { return E + e.CC + (size_t)t * d.R * d.ld_na; }
    float* MSGs(int t) const { return E + e.MSG + (size_t)t * d.R * d.ld_nm; }
    int32_t* POSs(int t) const { return reinterpret_cast<int32_t*>(E + e.POS) + (size_t)t * d.R * 2; }
    float* PROBSs(int t) const { return E + e.PROBS + (size_t)t * d.R * d.nA; }
    int32_t* ACTs(int t) const { return reinterpret_cast<int32_t*>(E + e.ACT) + (size_t)t * d.R; }
    float* DHs(int t) const { return E + e.DH + (size_t)t * d.R * d.ld_nb; }
    float* DHCs(int t) const { return E + e.DHC + (size_t)t * d.R * d.ld_na; }
};

static void register_split_images(const Dims& d, const WLayout& w, const float* W);
static bool use_panels(const Dims& d);
static bool use_side_stream();

static int make_ctx(const marl_config* cfg, const void* wws, size_t wbytes, void* ews, size_t ebytes,
                    int train, void* stream, Ctx& c) {
    MARL_TRY(make_dims(cfg, c.d));
    make_wlayout(c.d, c.w);
    make_elayout(c.d, train, c.e);
    if (wbytes < c.w.total * sizeof(float) || ebytes < c.e.total * sizeof(float)) {
        set_error("workspace too small for the current configuration / tuning knobs: weights %zu of %zu "
                  "bytes, episode %zu of %zu bytes (ask marl_workspace_sizes again)", wbytes,
                  c.w.total * sizeof(float), ebytes, c.e.total * sizeof(float));
        return MARL_ESIZE;
    }
    c.W = const_cast<float*>(static_cast<const float*>(wws));
    c.E = static_cast<float*>(ews);
    c.st = static_cast<hipStream_t>(stream);
    c.train = train;
    if (!c.W || !c.E) {
        set_error("null workspace");
        return MARL_EINVAL;
    }
    if ((reinterpret_cast<uintptr_t>(c.W) & 255) || (reinterpret_cast<uintptr_t>(c.E) & 255)) {
        set_error("workspaces must be 256-byte aligned");
        return MARL_EINVAL;
    }
    register_split_images(c.d, c.w, c.W);
    // the three kernels that write U[t] (fused CNN: features; decoder panel: message columns; sampling
    // launch: position embedding) also write its image when all of them are on the path and the column
    // ranges keep 8-byte image pieces whole; else (and for step 0, whose embedding comes from the
    // stand-alone kernel) one image pass over U[t] runs ahead of the LSTM launch
    CnnFwdArgs probe = cnn_fwd_shape(c.d);  // (with the fragment-order weight copies the AidCnn kernels ask for)
    for (int l = 0; l < c.d.L; ++l) probe.layer[l].wfrag = c.w.wf[4 * l] ? c.W + c.w.wf[4 * l] : nullptr;
    c.u3_by_producers = c.e.g3 && tune_get("g3_lstm", 1) != 0 && c.e.fused_fwd &&
                        cnn_fwd_writes_image(probe) && use_panels(c.d) && !use_side_stream() &&
                        ((c.d.nf | c.d.n_mo | c.d.n_d) & 3) == 0 &&
                        // (the producers write exactly nin columns: a K pad of the last 16-deep step would be
                        // whatever the workspace held - only the stand-alone image pass zeroes it)
                        c.d.nin % 16 == 0;
    return MARL_OK;
}

static int gemm1(const Ctx& c, const GemmProb& p) {
    GemmBatch b{};
    b.p[0] = p;
    b.count = 1;
    return launch_gemm_nt(b, c.st);
}
static int gemm2(const Ctx& c, const GemmProb& p0, const GemmProb& p1) {
    GemmBatch b{};
    b.p[0] = p0;
    b.p[1] = p1;
    b.count = 2;
    return launch_gemm_nt(b, c.st);
}

// ---------------------------------------------------------------------------
// one step of the network up to (not including) the policy output layer
// ---------------------------------------------------------------------------
struct StepIn {
    const float* obs = nullptr;       // standalone API: pre-gathered patches
    const float* npos = nullptr;      // standalone API: normalised positions
    const void* img = nullptr;
    int img_u8 = 0;
};

static bool use_panels(const Dims& d) {
    static int enabled = -1;
    if (enabled < 0) {
        const char* e = getenv("MARL_PANELS");
        enabled = (e && e[0] == '0') ? 0 : 1;
    }
    return enabled && panel_supported(d.n_m, d.nm2, d.n_mo) && panel_supported(d.n_b, d.nm2, d.n_m) &&
           panel_supported(d.n_a, d.nla, 0);
}

// ---- the five phases of one step; each only depends on what the comment names -----------

// observation -> CNN features b_t -> U[t][:, :nf]            (needs POS[t])
static int step_cnn(const Ctx& c, int t, const StepIn& in) {
    const Dims& d = c.d;
    hipStream_t st = c.st;
    {
        // one fused launch for the whole extractor when the shapes allow it
        const bool keep = c.train != 0;
        CnnFwdArgs a{};
        a.img = in.img;
        a.obs = in.obs;
        a.pos = c.POSs(t);
        a.img_u8 = in.img_u8;
        a.rows = d.R;
        a.nb = d.nb;
        a.c_img = d.c_img;
        a.H = d.H;
        a.W = d.W;
        a.f = d.f;
        a.L = d.L;
        for (int l = 0; l < d.L; ++l)
            a.layer[l] = CnnFwdLayer{c.wp(4 * l), c.wp(4 * l + 1), c.wp(4 * l + 2), c.wp(4 * l + 3),
                                     keep && !c.e.wgrad_ok[l] ? c.at(c.e.COLS[l], t) : nullptr,
                                     keep ? c.at(c.e.Z[l], t) : nullptr,
                                     keep ? c.at(c.e.GST[l], t) : nullptr,
                                     d.ch[l], d.ch[l + 1], d.grp[l], d.hw[l], d.hw[l + 1], d.P[l],
                                     d.K[l], d.ldk[l], c.w.wf[4 * l] ? c.W + c.w.wf[4 * l] : nullptr};
        a.u = c.at(c.e.U, t);
        a.ldu = d.ld_nin;
        if (c.u3_by_producers) {
            a.u3 = c.img(c.e.U3);
            a.u3_row0 = c.u3_row(t);
            a.u3_steps = img_steps(d.nin);
        }
        if (c.e.fused_fwd) return launch_cnn_fwd(a, st);
    }
    if (in.obs)
        MARL_TRY(launch_obs_im2col(in.obs, c.at(c.e.COLS[0], t), d.ldk[0], d.R, d.c_img, d.ch[0],
                                   d.f, st));
    else
        MARL_TRY(launch_gather_im2col(in.img, in.img_u8, c.POSs(t), c.at(c.e.COLS[0], t), d.ldk[0],
                                      d.na, d.nb, d.c_img, d.ch[0], d.H, d.W, d.f, st));
    for (int l = 0; l < d.L; ++l) {
        const int co = d.ch[l + 1];
        const int64_t rows = d.R * d.P[l];
        MARL_TRY(gemm1(c, gemm_prob(c.at(c.e.COLS[l], t), d.ldk[l], c.wp(4 * l), d.ldk[l], d.K[l],
                                    c.at(c.e.Z[l], t), co, (int)rows, co, c.wp(4 * l + 1))));
        if (l + 1 < d.L && gn_fwd_im2col_supported(d.P[l], co)) {
            // GroupNorm + SiLU + the next layer's im2col in one pass (A_l is never materialised)
            MARL_TRY(launch_gn_silu_fwd(c.at(c.e.Z[l], t), c.wp(4 * l + 2), c.wp(4 * l + 3), nullptr,
                                        0, 0, c.at(c.e.GST[l], t), d.R, d.P[l], co, d.grp[l], st,
                                        c.at(c.e.COLS[l + 1], t), d.ldk[l + 1], d.hw[l + 1]));
        } else if (l + 1 < d.L) {
            MARL_TRY(launch_gn_silu_fwd(c.at(c.e.Z[l], t), c.wp(4 * l + 2), c.wp(4 * l + 3),
                                        c.at(c.e.A[l], t), (int64_t)d.P[l] * co, 0,
                                        c.at(c.e.GST[l], t), d.R, d.P[l], co, d.grp[l], st));
            MARL_TRY(launch_im2col(c.at(c.e.A[l], t), c.at(c.e.COLS[l + 1], t), d.ldk[l + 1], d.R,
                                   d.hw[l + 1], co, st));
        } else {
            MARL_TRY(launch_gn_silu_fwd(c.at(c.e.Z[l], t), c.wp(4 * l + 2), c.wp(4 * l + 3),
                                        c.at(c.e.U, t), d.ld_nin, 1, c.at(c.e.GST[l], t), d.R,
                                        d.P[l], co, d.grp[l], st));
        }
    }
    return MARL_OK;
}

// the decoder's two layers of step t (outputs: AD1[t], U[t][:, nf:nf+n_mo])
static void fill_dec_layers(const Ctx& c, int t, PanelLayer* layer) {
    const Dims& d = c.d;
    const bool keep = c.train != 0;
    layer[0] = PanelLayer{c.wp(MARL_P_DEC_W0), p4(d.n_m), c.wp(MARL_P_DEC_B0),
                          c.wp(MARL_P_DEC_LN0W), c.wp(MARL_P_DEC_LN0B), d.nm2,
                          keep ? c.at(c.e.ZD1, t) : nullptr, d.ld_nm2,
                          keep ? c.at(c.e.STD1, t) : nullptr, c.at(c.e.AD1, t), d.ld_nm2};
    layer[1] = PanelLayer{c.wp(MARL_P_DEC_W1), d.ld_nm2, c.wp(MARL_P_DEC_B1),
                          c.wp(MARL_P_DEC_LN1W), c.wp(MARL_P_DEC_LN1B), d.n_mo,
                          keep ? c.at(c.e.ZD2, t) : nullptr, d.ld_nmo,
                          keep ? c.at(c.e.STD2, t) : nullptr, c.at(c.e.U, t) + d.nf, d.ld_nin};
    if (c.u3_by_producers) {
        layer[1].a3 = c.img(c.e.U3);
        layer[1].a3_row0 = c.u3_row(t);
        layer[1].a3_steps = img_steps(d.nin);
        layer[1].a3_col0 = d.nf;
    }
}

// message mean over the other agents + decoder -> U[t][:, nf:nf+n_mo]   (needs MSG[t])
// `sample` (panel path only): the sampling rows of the PREVIOUS step ride along in the same
// launch (they are independent of the decoder); *fused reports whether that happened
static int step_decode(const Ctx& c, int t, const SampleArgs* sample = nullptr, bool* fused = nullptr) {
    const Dims& d = c.d;
    const int R = (int)d.R;
    hipStream_t st = c.st;
    if (use_panels(d)) {
        PanelFwdBatch pb{};
        pb.count = 1;
        PanelFwdProb& p = pb.p[0];
        p.x = c.MSGs(t);
        p.ldx = d.ld_nm;
        p.k0 = d.n_m;
        p.agg_na = d.na;
        p.agg_nb = d.nb;
        p.xbar = c.at(c.e.MBAR, t);
        p.m = R;
        p.nlayers = 2;
        fill_dec_layers(c, t, p.layer);
        if (sample) {
            pb.has_sample = 1;
            pb.sample = *sample;
            if (fused) *fused = true;
        }
        return launch_panel_fwd(pb, st);
    }
    MARL_TRY(launch_agg_msg(c.MSGs(t), c.at(c.e.MBAR, t), d.ld_nm, d.na, d.nb, d.n_m, st));
    MARL_TRY(gemm1(c, gemm_prob(c.at(c.e.MBAR, t), d.ld_nm, c.wp(MARL_P_DEC_W0), p4(d.n_m), d.n_m,
                                c.at(c.e.ZD1, t), d.ld_nm2, R, d.nm2, c.wp(MARL_P_DEC_B0))));
    MARL_TRY(launch_ln_silu_fwd(c.at(c.e.ZD1, t), d.ld_nm2, c.wp(MARL_P_DEC_LN0W),
                                c.wp(MARL_P_DEC_LN0B), c.at(c.e.AD1, t), d.ld_nm2,
                                c.at(c.e.STD1, t), d.R, d.nm2, st));
    MARL_TRY(gemm1(c, gemm_prob(c.at(c.e.AD1, t), d.ld_nm2, c.wp(MARL_P_DEC_W1), d.ld_nm2, d.nm2,
                                c.at(c.e.ZD2, t), d.ld_nmo, R, d.n_mo, c.wp(MARL_P_DEC_B1))));
    return launch_ln_silu_fwd(c.at(c.e.ZD2, t), d.ld_nmo, c.wp(MARL_P_DEC_LN1W),
                              c.wp(MARL_P_DEC_LN1B), c.at(c.e.U, t) + d.nf, d.ld_nin,
                              c.at(c.e.STD2, t), d.R, d.n_mo, st);
}

// position embedding + both LSTM cells -> H/C/H^/C^[t+1]     (needs all of U[t])
static int step_pos_lstm(const Ctx& c, int t, const StepIn& in, bool pos_done = false) {
    const Dims& d = c.d;
    const int R = (int)d.R;
    hipStream_t st = c.st;
    if (!pos_done)
    MARL_TRY(launch_pos_embed_fwd(c.POSs(t), in.npos, d.H, d.W, c.wp(MARL_P_POS_W),
                                  c.wp(MARL_P_POS_B), c.wp(MARL_P_POS_LNW), c.wp(MARL_P_POS_LNB),
                                  c.at(c.e.NPOS, t), c.at(c.e.ZPOS, t), d.ld_nd, c.at(c.e.STPOS, t),
                                  c.at(c.e.U, t) + d.nf + d.n_mo, d.ld_nin, d.R, d.n_d, st));
    if (c.e.g3 && tune_get("g3_lstm", 1) != 0) {
        // image form: U[t] is complete here -> its k16 image (until its three producers write it
        // themselves), then both cells from images; the epilogue writes the images of h' / h^'
        const int u_row0 = c.u3_row(t);
        if (!c.u3_by_producers || t == 0 || in.obs) {
            ImgBatch ib{};
            ib.d[0] = ImgDesc{c.at(c.e.U, t), c.img(c.e.U3) + img_off(u_row0, 0, img_steps(d.nin)), d.R, d.nin, d.ld_nin};
            ib.count = 1;
            MARL_TRY(launch_images(ib, st));
        }
        const int r0 = (int)((int64_t)t * d.R), r1 = r0 + R;
        G3Batch g{};
        g.count = 2;
        G3Prob& qb = g.p[0];
        qb = g3_prob(c.img(c.e.U3), u_row0, c.wp3k(MARL_P_LB_WIH), 0, d.nin, nullptr, 0, R, d.n_b, c.W + c.w.bsum_b);
        g3_add_seg(qb, c.img(c.e.H3), r0, c.wp3k(MARL_P_LB_WHH), 0, d.n_b);
        qb.c_prev = c.Cs(t);
        qb.h_next = c.Hs(t + 1);
        qb.c_next = c.Cs(t + 1);
        qb.gates = c.train ? c.at(c.e.GB, t) : nullptr;
        qb.ld_state = d.ld_nb;
        qb.ld_gates = d.ld_gb;
        qb.h3 = c.img(c.e.H3);
        qb.h3_row0 = r1;
        qb.h3_steps = img_steps(d.n_b);
        G3Prob& qa = g.p[1];
        qa = g3_prob(c.img(c.e.U3), u_row0, c.wp3k(MARL_P_LA_WIH), 0, d.nin, nullptr, 0, R, d.n_a, c.W + c.w.bsum_a);
        g3_add_seg(qa, c.img(c.e.HC3), r0, c.wp3k(MARL_P_LA_WHH), 0, d.n_a);
        qa.c_prev = c.CCs(t);
        qa.h_next = c.HCs(t + 1);
        qa.c_next = c.CCs(t + 1);
        qa.gates = c.train ? c.at(c.e.GA, t) : nullptr;
        qa.ld_state = d.ld_na;
        qa.ld_gates = d.ld_ga;
        qa.h3 = c.img(c.e.HC3);
        qa.h3_row0 = r1;
        qa.h3_steps = img_steps(d.n_a);
        return launch_gemm_lstm3(g, st);
    }
    GemmBatch b{};
    b.count = 2;
    GemmProb& pb = b.p[0];
    pb = gemm_prob(c.at(c.e.U, t), d.ld_nin, c.wp(MARL_P_LB_WIH), d.ld_nin, d.nin, nullptr, 0, R,
                   d.n_b, c.W + c.w.bsum_b);
    gemm_add_seg(pb, c.Hs(t), d.ld_nb, c.wp(MARL_P_LB_WHH), d.ld_nb, d.n_b);
    pb.c_prev = c.Cs(t);
    pb.h_next = c.Hs(t + 1);
    pb.c_next = c.Cs(t + 1);
    pb.gates = c.train ? c.at(c.e.GB, t) : nullptr;
    pb.ld_state = d.ld_nb;
    pb.ld_gates = d.ld_gb;
    GemmProb& pa = b.p[1];
    pa = gemm_prob(c.at(c.e.U, t), d.ld_nin, c.wp(MARL_P_LA_WIH), d.ld_nin, d.nin, nullptr, 0, R,
                   d.n_a, c.W + c.w.bsum_a);
    gemm_add_seg(pa, c.HCs(t), d.ld_na, c.wp(MARL_P_LA_WHH), d.ld_na, d.n_a);
    pa.c_prev = c.CCs(t);
    pa.h_next = c.HCs(t + 1);
    pa.c_next = c.CCs(t + 1);
    pa.gates = c.train ? c.at(c.e.GA, t) : nullptr;
    pa.ld_state = d.ld_na;
    pa.ld_gates = d.ld_ga;
    return launch_gemm_lstm(b, st);
}

static void fill_enc_prob(const Ctx& c, int t, PanelFwdProb& pe) {
    const Dims& d = c.d;
    const bool keep = c.train != 0;
    pe.x = c.Hs(t + 1);
    pe.ldx = d.ld_nb;
    pe.k0 = d.n_b;
    pe.m = (int)d.R;
    pe.nlayers = 2;
    pe.layer[0] = PanelLayer{c.wp(MARL_P_ENC_W0), d.ld_nb, c.wp(MARL_P_ENC_B0),
                             c.wp(MARL_P_ENC_LN0W), c.wp(MARL_P_ENC_LN0B), d.nm2,
                             keep ? c.at(c.e.ZE1, t) : nullptr, d.ld_nm2,
                             keep ? c.at(c.e.STE1, t) : nullptr, c.at(c.e.AE1, t), d.ld_nm2};
    pe.layer[1] = PanelLayer{c.wp(MARL_P_ENC_W1), d.ld_nm2, c.wp(MARL_P_ENC_B1),
                             c.wp(MARL_P_ENC_LN1W), c.wp(MARL_P_ENC_LN1B), d.n_m,
                             keep ? c.at(c.e.ZE2, t) : nullptr, d.ld_nm,
                             keep ? c.at(c.e.STE2, t) : nullptr, c.MSGs(t + 1), d.ld_nm};
}
static void fill_pol_prob(const Ctx& c, int t, PanelFwdProb& pp) {
    const Dims& d = c.d;
    const bool keep = c.train != 0;
    pp.x = c.HCs(t + 1);
    pp.ldx = d.ld_na;
    pp.k0 = d.n_a;
    pp.m = (int)d.R;
    pp.nlayers = 1;
    pp.layer[0] = PanelLayer{c.wp(MARL_P_POL_W0), d.ld_na, c.wp(MARL_P_POL_B0),
                             c.wp(MARL_P_POL_LNW), c.wp(MARL_P_POL_LNB), d.nla,
                             keep ? c.at(c.e.ZP1, t) : nullptr, d.ld_nla,
                             keep ? c.at(c.e.STP1, t) : nullptr, c.at(c.e.AP1, t), d.ld_nla};
}

// message encoder -> MSG[t+1] (which = 1), policy hidden layer -> AP1[t] (which = 2), or both
// in one launch (which = 3).                                   (needs H / H^[t+1])
static int step_encode_policy(const Ctx& c, int t, int which) {
    const Dims& d = c.d;
    const int R = (int)d.R;
    hipStream_t st = c.st;
    if (use_panels(d)) {
        PanelFwdBatch pb{};
        pb.count = 0;
        if (which & 1) fill_enc_prob(c, t, pb.p[pb.count++]);
        if (which & 2) fill_pol_prob(c, t, pb.p[pb.count++]);
        return launch_panel_fwd(pb, st);
    }
    if (which == 3) {
        MARL_TRY(gemm2(c,
                       gemm_prob(c.Hs(t + 1), d.ld_nb, c.wp(MARL_P_ENC_W0), d.ld_nb, d.n_b,
                                 c.at(c.e.ZE1, t), d.ld_nm2, R, d.nm2, c.wp(MARL_P_ENC_B0)),
                       gemm_prob(c.HCs(t + 1), d.ld_na, c.wp(MARL_P_POL_W0), d.ld_na, d.n_a,
                                 c.at(c.e.ZP1, t), d.ld_nla, R, d.nla, c.wp(MARL_P_POL_B0))));
    } else if (which == 1) {
        MARL_TRY(gemm1(c, gemm_prob(c.Hs(t + 1), d.ld_nb, c.wp(MARL_P_ENC_W0), d.ld_nb, d.n_b,
                                    c.at(c.e.ZE1, t), d.ld_nm2, R, d.nm2, c.wp(MARL_P_ENC_B0))));
    } else {
        MARL_TRY(gemm1(c, gemm_prob(c.HCs(t + 1), d.ld_na, c.wp(MARL_P_POL_W0), d.ld_na, d.n_a,
                                    c.at(c.e.ZP1, t), d.ld_nla, R, d.nla, c.wp(MARL_P_POL_B0))));
    }
    if (which & 2)
        MARL_TRY(launch_ln_silu_fwd(c.at(c.e.ZP1, t), d.ld_nla, c.wp(MARL_P_POL_LNW),
                                    c.wp(MARL_P_POL_LNB), c.at(c.e.AP1, t), d.ld_nla,
                                    c.at(c.e.STP1, t), d.R, d.nla, st));
    if (which & 1) {
        MARL_TRY(launch_ln_silu_fwd(c.at(c.e.ZE1, t), d.ld_nm2, c.wp(MARL_P_ENC_LN0W),
                                    c.wp(MARL_P_ENC_LN0B), c.at(c.e.AE1, t), d.ld_nm2,
                                    c.at(c.e.STE1, t), d.R, d.nm2, st));
        MARL_TRY(gemm1(c, gemm_prob(c.at(c.e.AE1, t), d.ld_nm2, c.wp(MARL_P_ENC_W1), d.ld_nm2,
                                    d.nm2, c.at(c.e.ZE2, t), d.ld_nm, R, d.n_m,
                                    c.wp(MARL_P_ENC_B1))));
        MARL_TRY(launch_ln_silu_fwd(c.at(c.e.ZE2, t), d.ld_nm, c.wp(MARL_P_ENC_LN1W),
                                    c.wp(MARL_P_ENC_LN1B), c.MSGs(t + 1), d.ld_nm,
                                    c.at(c.e.STE2, t), d.R, d.n_m, st));
    }
    return MARL_OK;
}

// All agents of a batch element in one workgroup (rows a * nb + b for every a), so that the
// message mean over the other agents needs no second launch:
//   forward : encoder(t) -> MSG[t+1] -> mean -> decoder(t+1) -> U[t+1]  ||  policy hidden layer(t)
//   backward: decoder(t) -> mean -> encoder(t-1) -> dh_t complete -> belief cell(t-1)
static bool use_chain(const Dims& d) {
    return use_panels(d) && panel_chain_supported(d.na, d.n_m, 256) && d.n_mo <= 384 && d.nm2 <= 384 &&
           d.n_m <= 384 && tune_get("panel_chain", 1) != 0;
}
static int step_chain(const Ctx& c, int t) {
    const Dims& d = c.d;
    PanelFwdBatch pb{};
    pb.count = 2;
    PanelFwdProb& pe = pb.p[0];
    fill_enc_prob(c, t, pe);
    pe.by_batch = panel_chain_by_batch(d.na);
    pe.g_na = d.na;
    pe.g_nb = d.nb;
    if (t + 1 < d.ns) {
        fill_dec_layers(c, t + 1, pe.layer + 2);
        pe.nlayers = 4;
        pe.agg_at = 2;
        pe.xbar = c.at(c.e.MBAR, t + 1);
        pe.ld_xbar = d.ld_nm;
    }
    fill_pol_prob(c, t, pb.p[1]);
    return launch_panel_fwd(pb, c.st);
}

// whole step on one stream (standalone step API)
static int step_core(const Ctx& c, int t, const StepIn& in) {
    MARL_TRY(step_cnn(c, t, in));
    MARL_TRY(step_decode(c, t));
    MARL_TRY(step_pos_lstm(c, t, in));
    return step_encode_policy(c, t, 3);
}

// ---------------------------------------------------------------------------
// side stream: chains that are independent within a step run concurrently
//   forward : message encoder(t) -> decoder(t+1)   ||  policy(t) -> sample(t) -> CNN(t+1)
//   backward: W_hh recurrent GEMM(t)               ||  decoder / encoder backward chain(t)
// ---------------------------------------------------------------------------
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t ev[64];
    int next = 0;
    int init() {
        if (s) return MARL_OK;
        MARL_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (auto& e : ev) MARL_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return MARL_OK;
    }
    // everything enqueued on `from` so far happens before what is enqueued on `to` afterwards
    int order(hipStream_t from, hipStream_t to) {
        hipEvent_t e = ev[next];
        next = (next + 1) % 64;
        MARL_HIP_CHECK(hipEventRecord(e, from));
        MARL_HIP_CHECK(hipStreamWaitEvent(to, e, 0));
        return MARL_OK;
    }
};
static SideStream g_side;

static bool use_side_stream() {
    static int enabled = -1;
    // measured on MI355X (C3, rounds 1-2): no gain (16.38 vs 16.28 ms / iteration; round 2: 11.15 vs 10.23) - the
    // cross-stream event waits cost what the overlap buys.  The environment switch is gone; the form stays for reference.
    (void)enabled;
    return false;
}

// critic + prediction heads on `rows` rows starting at state slice t0 (+1)
static int heads_batched(const Ctx& c, int t0, int64_t rows, float* values, float* preds) {
    const Dims& d = c.d;
    hipStream_t st = c.st;
    if (c.e.g3 && tune_get("g3_lstm", 1) != 0) {  // A = the images the LSTM epilogues wrote
        const int r1 = (int)((int64_t)(t0 + 1) * d.R);
        G3Batch g{};
        g.p[0] = g3_prob(c.img(c.e.HC3), r1, c.wp3k(MARL_P_CRI_W0), 0, d.n_a, c.at(c.e.ZC1), d.ld_nla, (int)rows, d.nla,
                         c.wp(MARL_P_CRI_B0));
        g.p[1] = g3_prob(c.img(c.e.H3), r1, c.wp3k(MARL_P_PRE_W0), 0, d.n_b, c.at(c.e.ZQ1), d.ld_nlb, (int)rows, d.nlb,
                         c.wp(MARL_P_PRE_B0));
        g.count = 2;
        MARL_TRY(launch_gemm_nt3(g, st));
    } else
    MARL_TRY(gemm2(c,
                   gemm_prob(c.HCs(t0 + 1), d.ld_na, c.wp(MARL_P_CRI_W0), d.ld_na, d.n_a,
                             c.at(c.e.ZC1), d.ld_nla, (int)rows, d.nla, c.wp(MARL_P_CRI_B0)),
                   gemm_prob(c.Hs(t0 + 1), d.ld_nb, c.wp(MARL_P_PRE_W0), d.ld_nb, d.n_b,
                             c.at(c.e.ZQ1), d.ld_nlb, (int)rows, d.nlb, c.wp(MARL_P_PRE_B0))));
    const bool vdot = d.nla <= 384;  // the critic's output layer rides in the LayerNorm kernel
    MARL_TRY(launch_ln_silu_fwd(c.at(c.e.ZC1), d.ld_nla, c.wp(MARL_P_CRI_LNW), c.wp(MARL_P_CRI_LNB),
                                c.at(c.e.AC1), d.ld_nla, c.at(c.e.STC1), rows, d.nla, st,
                                vdot ? c.wp(MARL_P_CRI_W1) : nullptr, c.wp(MARL_P_CRI_B1), values));
    MARL_TRY(launch_ln_silu_fwd(c.at(c.e.ZQ1), d.ld_nlb, c.wp(MARL_P_PRE_LNW), c.wp(MARL_P_PRE_LNB),
                                c.at(c.e.AQ1), d.ld_nlb, c.at(c.e.STQ1), rows, d.nlb, st));
    if (!vdot)
        MARL_TRY(launch_rowdot(c.at(c.e.AC1), d.ld_nla, c.wp(MARL_P_CRI_W1), c.wp(MARL_P_CRI_B1),
                               values, rows, d.nla, st));
    MARL_TRY(gemm1(c, gemm_prob(c.at(c.e.AQ1), d.ld_nlb, c.wp(MARL_P_PRE_W1), d.ld_nlb, d.nlb,
                                preds, d.nC, (int)rows, d.nC, c.wp(MARL_P_PRE_B1))));
    return MARL_OK;
}

static void fill_sample_args(const Ctx& c, const marl_config* cfg, int t, SampleArgs& a) {
    const Dims& d = c.d;
    memset(&a, 0, sizeof(a));
    a.a_pol = c.at(c.e.AP1, t);
    a.ld_a = d.ld_nla;
    a.nla = d.nla;
    a.w1 = c.wp(MARL_P_POL_W1);
    a.ldw = d.ld_nla;
    a.b1 = c.wp(MARL_P_POL_B1);
    a.pos_in = c.POSs(t);
    a.pos_out = c.POSs(t + 1);
    a.probs = c.PROBSs(t);
    a.actions_i32 = c.ACTs(t);
    a.R = (int)d.R;
    a.nA = d.nA;
    a.H = d.H;
    a.W = d.W;
    a.f = d.f;
    for (int j = 0; j < d.nA; ++j) {
        a.table[j][0] = cfg->actions[j][0];
        a.table[j][1] = cfg->actions[j][1];
    }
}

// ---------------------------------------------------------------------------
// pack / unpack
// ---------------------------------------------------------------------------
struct PermQueue {
    PermBatch b{};
    hipStream_t st;
    int rc = MARL_OK;
    explicit PermQueue(hipStream_t s) : st(s) { b.count = 0; }
    void push(const PermDesc& p) {
        if (rc != MARL_OK) return;
        b.d[b.count++] = p;
        if (b.count == kMaxPerm) flush();
    }
    void flush() {
        if (rc == MARL_OK && b.count > 0) rc = launch_permute(b, st);
        b.count = 0;
    }
};

static PermDesc perm(const float* src, float* dst, int rows, int cols, int dst_ld, int rd, int rs1,
                     int rs2, int cd, int cs1, int cs2, const float* src2 = nullptr) {
    PermDesc p;
    p.src = src;
    p.dst = dst;
    p.rows = rows;
    p.cols = cols;
    p.dst_ld = dst_ld;
    p.rd = rd;
    p.rs1 = rs1;
    p.rs2 = rs2;
    p.cd = cd;
    p.cs1 = cs1;
    p.cs2 = cs2;
    p.src2 = src2;
    return p;
}

// plain strided copy (src null: zero fill) as a descriptor of the batched permute kernel
static PermDesc copy_desc(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int cols) {
    return perm(src, dst, (int)rows, cols, (int)ldd, 1, (int)lds, 0, 1, 1, 0);
}

// initial recurrent state -> slice 0 of the all-steps buffers, one launch
static int load_state(const Ctx& c, const float* h, const float* cc_, const float* hc,
                      const float* cca, const float* msg) {
    const Dims& d = c.d;
    PermQueue q(c.st);
    q.push(copy_desc(h, d.n_b, c.Hs(0), d.ld_nb, d.R, d.n_b));
    q.push(copy_desc(cc_, d.n_b, c.Cs(0), d.ld_nb, d.R, d.n_b));
    q.push(copy_desc(hc, d.n_a, c.HCs(0), d.ld_na, d.R, d.n_a));
    q.push(copy_desc(cca, d.n_a, c.CCs(0), d.ld_na, d.R, d.n_a));
    // (no message: zeros, models.py:161-162 - the whole padded row)
    q.push(copy_desc(msg, d.n_m, c.MSGs(0), d.ld_nm, d.R, msg ? d.n_m : d.ld_nm));
    q.flush();
    MARL_TRY(q.rc);
    if (c.e.g3) {  // the image GEMMs read the state through its images
        ImgBatch ib{};
        ib.d[0] = ImgDesc{h, c.img(c.e.H3), d.R, d.n_b, d.n_b};
        ib.d[1] = ImgDesc{hc, c.img(c.e.HC3), d.R, d.n_a, d.n_a};
        ib.count = 2;
        MARL_TRY(launch_images(ib, c.st));
    }
    return MARL_OK;
}


static int pack_weights(const Dims& d, const WLayout& w, const float* const* params, float* W,
                        hipStream_t st) {
    PermQueue q(st);
    for (int i = 0; i < MARL_NPARAMS; ++i) {
        const ParamMeta m = param_meta(d, i);
        if (m.kind == PK_NONE) continue;
        const float* src = params[i];
        if (!src) {
            set_error("parameter %d is null", i);
            return MARL_EINVAL;
        }
        if (m.kind == PK_MATRIX) {
            q.push(perm(src, W + w.wp[i], m.n, m.k, w.ldp[i], 1, m.k, 0, 1, 1, 0));
            q.push(perm(src, W + w.wt[i], m.k, m.n, w.ldt[i], 1, 1, 0, 1, m.k, 0));
        } else if (m.kind == PK_CONV) {
            const int ci = m.k / 9;
            // packed [co][tap*ci + c]  <- src [co][c][tap]
            q.push(perm(src, W + w.wp[i], m.n, m.k, w.ldp[i], 1, m.k, 0, ci, 1, 9));
            // transposed [tap*ci + c][co]
            q.push(perm(src, W + w.wt[i], m.k, m.n, w.ldt[i], ci, 1, 9, 1, m.k, 0));
        } else {
            q.push(perm(src, W + w.wp[i], 1, m.n, m.n, 1, 0, 0, 1, 1, 0));
        }
    }
    q.push(perm(params[MARL_P_LB_BIH], W + w.bsum_b, 1, 4 * d.n_b, 4 * d.n_b, 1, 0, 0, 1, 1, 0,
                params[MARL_P_LB_BHH]));
    q.push(perm(params[MARL_P_LA_BIH], W + w.bsum_a, 1, 4 * d.n_a, 4 * d.n_a, 1, 0, 0, 1, 1, 0,
                params[MARL_P_LA_BHH]));
    q.flush();
    MARL_TRY(q.rc);
    // fragment-order copies of the deeper conv weights (read back from the packed copies just written):
    // dst row (nt * steps + kk) * 4 + quad, column l16 * 4 + j  <-  packed [nt * 16 + l16][kk * 16 + quad * 4 + j]
    for (int i = 0; i < MARL_NPARAMS; ++i) {
        const ParamMeta m = param_meta(d, i);
        if (!w.wf[i]) continue;
        const int steps = m.k / 16;
        q.push(perm(W + w.wp[i], W + w.wf[i], (m.n / 16) * steps * 4, 64, 64, steps * 4, 16 * w.ldp[i], 4, 4,
                    w.ldp[i], 1));
    }
    q.flush();
    MARL_TRY(q.rc);
    // bf16x3 images of every matrix copy (read back from the fp32 copies just written)
    SplitBatch sb{};
    sb.count = 0;
    for (int i = 0; i < MARL_NPARAMS; ++i) {
        const ParamMeta m = param_meta(d, i);
        if (m.kind != PK_MATRIX) continue;
        sb.d[sb.count++] = SplitDesc{W + w.wp[i], W + w.wp3[i], m.n, m.k, w.ldp[i], (m.k + 31) / 32};
        sb.d[sb.count++] = SplitDesc{W + w.wt[i], W + w.wt3[i], m.k, m.n, w.ldt[i], (m.n + 31) / 32};
    }
    MARL_TRY(launch_split_weights(sb, st));
    ImgBatch ib{};
    ib.count = 0;
    for (int i = 0; i < MARL_NPARAMS; ++i) {
        const ParamMeta m = param_meta(d, i);
        if (m.kind != PK_MATRIX || !w.wp3k[i]) continue;
        if (ib.count + 2 > kMaxImgDesc) {
            MARL_TRY(launch_images(ib, st));
            ib.count = 0;
        }
        ib.d[ib.count++] = ImgDesc{W + w.wp[i], W + w.wp3k[i], m.n, m.k, w.ldp[i]};
        ib.d[ib.count++] = ImgDesc{W + w.wt[i], W + w.wt3k[i], m.k, m.n, w.ldt[i]};
    }
    return launch_images(ib, st);
}

// tells the bf16x6 launchers where the image of each fp32 weight copy lives
static void register_split_images(const Dims& d, const WLayout& w, const float* W) {
    split_registry_reset();
    for (int i = 0; i < MARL_NPARAMS; ++i) {
        const ParamMeta m = param_meta(d, i);
        if (m.kind != PK_MATRIX) continue;
        split_registry_add(W + w.wp[i], m.n, w.ldp[i], m.k, W + w.wp3[i]);
        split_registry_add(W + w.wt[i], m.k, w.ldt[i], m.n, W + w.wt3[i]);
    }
}

// which: 0 every parameter, 1 only the heads' matrices (MARL_P_POL_W0 .. MARL_P_PRE_B1: final before the reverse loop -
// marl_backward_heads_event), 2 everything but those
static int unpack_grads(const Ctx& c, float* const* grads, int which = 0) {
    PermQueue q(c.st);
    for (int i = 0; i < MARL_NPARAMS; ++i) {
        const bool head = i >= MARL_P_POL_W0 && i <= MARL_P_PRE_B1;
        if ((which == 1 && !head) || (which == 2 && head)) continue;
        const ParamMeta m = param_meta(c.d, i);
        if (m.kind == PK_MATRIX) {
            q.push(perm(c.gp(i), grads[i], m.n, m.k, m.k, 1, c.w.ldp[i], 0, 1, 1, 0));
        } else if (m.kind == PK_CONV) {
            const int ci = m.k / 9;
            // dst [co][c*9 + tap] <- packed [co][tap*ci + c]
            q.push(perm(c.gp(i), grads[i], m.n, m.k, m.k, 1, c.w.ldp[i], 0, 9, 1, ci));
        }
    }
    if (which != 1) {
        // b_hh enters every gate sum exactly like b_ih: same gradient
        q.push(perm(grads[MARL_P_LB_BIH], grads[MARL_P_LB_BHH], 1, 4 * c.d.n_b, 4 * c.d.n_b, 1, 0, 0, 1, 1, 0));
        q.push(perm(grads[MARL_P_LA_BIH], grads[MARL_P_LA_BHH], 1, 4 * c.d.n_a, 4 * c.d.n_a, 1, 0, 0, 1, 1, 0));
    }
    q.flush();
    return q.rc;
}

// ---------------------------------------------------------------------------
// backward helpers
// ---------------------------------------------------------------------------
// weight gradient gp(pidx) = A^T B over `rows` rows; bias (nullable) = column sums of A
static int tn(const Ctx& c, const float* a, int lda, const float* b, int ldb, int pidx, int ni,
              int nj, int64_t rows, float* bias = nullptr) {
    RedQueue* q = c.defer_this(gemm_tn_scratch_bytes(ni, nj, rows)) ? c.rq : nullptr;
    return launch_gemm_tn(a, lda, b, ldb, c.gp(pidx), c.w.ldp[pidx], ni, nj, rows, c.at(c.e.TNS),
                          c.e.tns_bytes, c.st, bias, q, q ? c.tq : nullptr);
}
// the same from images (rows = contraction index; A [rows, ni], B [rows, nj], both starting at image row 0)
static int tn3(const Ctx& c, const char* a3, int ni, const char* b3, int nj, int pidx, int64_t rows, float* bias) {
    const G3TnPlan plan = g3_tn_plan(ni, nj, rows);
    if (g3_tn_scratch_bytes(ni, nj, rows) > c.e.tns_bytes) {
        set_error("tn3: scratch too small");
        return MARL_ESIZE;
    }
    float* scratch = c.at(c.e.TNS);
    G3TnArgs a{};
    a.a3 = a3;
    a.b3 = b3;
    a.a_steps = img_steps(ni);
    a.b_steps = img_steps(nj);
    a.out = scratch;
    a.ldo = nj;
    a.out_split_stride = (int64_t)ni * nj;
    a.ni = ni;
    a.nj = nj;
    a.rows = rows;
    a.csum = bias ? scratch + (size_t)plan.splits * ni * nj : nullptr;
    MARL_TRY(launch_gemm_tn3(a, plan, c.st));
    return launch_slab_reduce(scratch, (int64_t)ni * nj, plan.splits, c.gp(pidx), c.w.ldp[pidx], ni, nj, a.csum, bias, c.st);
}
// both weight gradients of one LSTM cell from one launch (gemm3.hip, gemm_tn3_cell_kernel): G read once
static int tn3_cell(const Ctx& c, const char* g3, int ni, const char* u3, int nin, int p_ih, const char* h3, int nh,
                    int p_hh, int64_t rows, float* bias) {
    const G3TnPlan plan = g3_tn_cell_plan(ni, nin, nh, rows);
    if (g3_tn_cell_scratch_bytes(ni, nin, nh, rows) > c.e.tns_bytes) {
        set_error("tn3_cell: scratch too small");
        return MARL_ESIZE;
    }
    float* s_ih = c.at(c.e.TNS);
    float* s_hh = s_ih + (size_t)plan.splits * ni * nin;
    float* s_cs = s_hh + (size_t)plan.splits * ni * nh;
    G3TnArgs ih{}, hh{};
    ih.a3 = hh.a3 = g3;
    ih.a_steps = hh.a_steps = img_steps(ni);
    ih.ni = hh.ni = ni;
    ih.rows = hh.rows = rows;
    ih.b3 = u3;
    ih.b_steps = img_steps(nin);
    ih.nj = ih.ldo = nin;
    ih.out = s_ih;
    ih.out_split_stride = (int64_t)ni * nin;
    ih.csum = bias ? s_cs : nullptr;  // (the column sums of G ride on the first tile of the launch)
    hh.b3 = h3;
    hh.b_steps = img_steps(nh);
    hh.nj = hh.ldo = nh;
    hh.out = s_hh;
    hh.out_split_stride = (int64_t)ni * nh;
    MARL_TRY(launch_gemm_tn3_cell(ih, hh, plan, c.st));
    MARL_TRY(launch_slab_reduce(s_ih, (int64_t)ni * nin, plan.splits, c.gp(p_ih), c.w.ldp[p_ih], ni, nin, nullptr, nullptr, c.st));
    return launch_slab_reduce(s_hh, (int64_t)ni * nh, plan.splits, c.gp(p_hh), c.w.ldp[p_hh], ni, nh, ih.csum, bias, c.st);
}
// scratch for `blocks` affine partial rows of width 2n: the queue's when the reduction can wait
static float* part_scratch(const Ctx& c, int64_t blocks, int n, int acc, RedQueue*& q) {
    q = acc ? nullptr : c.rq;
    if (!q) return c.at(c.e.PART);
    float* p = q->take((size_t)blocks * 2 * n);
    if (q->rc != MARL_OK) q = nullptr;  // cannot happen with the layout's sizes; fall back
    return q ? p : c.at(c.e.PART);
}
// LayerNorm+SiLU backward in place (da -> dz) with affine gradients (accumulated when acc)
static int ln_bwd(const Ctx& c, float* da, int ldda, const float* z, int ldz, const float* stats,
                  int pw, int pb, int64_t rows, int n, float* const* grads, int acc,
                  float* dz = nullptr, int lddz = 0) {
    if (!dz) {
        dz = da;
        lddz = ldda;
    }
    RedQueue* q;
    float* part = part_scratch(c, ln_bwd_blocks(rows, n), n, acc, q);
    MARL_TRY(launch_ln_silu_bwd(da, ldda, z, ldz, stats, c.wp(pw), c.wp(pb), dz, lddz, part, rows, n,
                                c.st));
    return launch_reduce_affine(part, ln_bwd_blocks(rows, n), n, grads[pw], grads[pb], acc, c.st, q);
}

// LayerNorm+SiLU backward of a hidden layer whose successor has only kin <= 4 outputs: the
// incoming gradient g * W1 is formed inside the kernel instead of by a K = kin GEMM
static int ln_bwd_rank(const Ctx& c, const float* g, int ldg, int kin, int w1, const float* z, int ldz,
                       const float* stats, int pw, int pb, int64_t rows, int n, float* const* grads,
                       float* dz, int lddz) {
    RedQueue* q;
    float* part = part_scratch(c, ln_bwd_blocks(rows, n), n, 0, q);
    MARL_TRY(launch_ln_silu_bwd_rank(g, ldg, kin, c.wt(w1), p4(kin), z, ldz, stats, c.wp(pw),
                                     c.wp(pb), dz, lddz, part, rows, n, c.st));
    return launch_reduce_affine(part, ln_bwd_blocks(rows, n), n, grads[pw], grads[pb], 0, c.st, q);
}

// the panel launch could not write a gate-gradient image it was asked for (shapes without the row-wise
// tail): build it from the fp32 gradients - correct, one extra pass
static int gate_image_fallback(const Ctx& c, const PanelBwdProb& pd, int t) {
    const Dims& d = c.d;
    ImgBatch ib{};
    ib.count = 0;
    if (pd.has_cellb && pd.cellb.g3 && !pd.cellb_img_done)
        ib.d[ib.count++] = ImgDesc{c.at(c.e.GB, t), c.img(c.e.GB3) + img_off((int64_t)t * d.R, 0, img_steps(4 * d.n_b)),
                                   d.R, 4 * d.n_b, d.ld_gb};
    if (pd.has_cell && pd.cell.g3 && !pd.cell_img_done)
        ib.d[ib.count++] = ImgDesc{c.at(c.e.GA, t), c.img(c.e.GA3) + img_off((int64_t)t * d.R, 0, img_steps(4 * d.n_a)),
                                   d.R, 4 * d.n_a, d.ld_ga};
    return launch_images(ib, c.st);
}

// marl_backward_heads_event: recorded by episode_backward once the heads' parameter gradients are final
static hipEvent_t g_heads_event = nullptr;

static int episode_backward(const Ctx& c0, const void* img, int img_u8, const float* g_preds,
                            const float* g_logp, const float* g_values, float* const* grads) {
    Ctx c = c0;
    RedQueue rq;
    rq.reset(c.at(c.e.RED), c.e.red_floats, c.st);
    // 1: the small LayerNorm / GroupNorm affine partials wait for one launch at the end; the
    // weight-gradient slabs (~0.4 GB per iteration in all) are reduced at once, while the
    // Infinity Cache still holds them (2: defer those too - measured slower at C3 / C4: 0.43 GB of
    // slabs come back from HBM).  3 (default): as 1, plus the slabs of at most red_defer_kb KB
    // each (8 MB: larger thresholds lose again at C4) - on small problems (C2: 22 weight gradients) one batched
    // reduction replaces 22 launches: 0.953 -> 0.897 ms per iteration.
    const int defer = tune_get("red_defer", 3);
    TnQueue tq;
    if (defer) c.rq = &rq;
    if (defer) c.tq = &tq;
    c.defer_slabs = defer == 2;
    c.defer_small = defer == 3 ? (size_t)8192 * 1024 : 0;
    const Dims& d = c.d;
    hipStream_t st = c.st;
    const int64_t NR = d.NR;
    const int R = (int)d.R, ns = d.ns;
    for (int i = 0; i < MARL_NPARAMS; ++i)
        if (param_meta(d, i).kind != PK_NONE && !grads[i]) {
            set_error("gradient buffer %d is null", i);
            return MARL_EINVAL;
        }
    // ---- heads, batched over all steps -----------------------------------------------
    // prediction head (networks/prediction.py:11-14)
    {   // output gradients into the padded layouts + the zero initial state gradients: one launch
        PermQueue q(st);
        q.push(copy_desc(g_preds, d.nC, c.at(c.e.GPRED), d.ld_nC, NR, g_preds ? d.nC : d.ld_nC));
        q.push(copy_desc(g_values, 1, c.at(c.e.DVAL), 4, NR, g_values ? 1 : 4));
        q.push(copy_desc(nullptr, 0, c.DHs(0), d.ld_nb, d.R, d.ld_nb));
        q.push(copy_desc(nullptr, 0, c.DHCs(0), d.ld_na, d.R, d.ld_na));
        q.push(copy_desc(nullptr, 0, c.at(c.e.DC), d.ld_nb, d.R, d.ld_nb));
        q.push(copy_desc(nullptr, 0, c.at(c.e.DCC), d.ld_na, d.R, d.ld_na));
        q.flush();
        MARL_TRY(q.rc);
    }
    MARL_TRY(gemm1(c, gemm_prob(c.at(c.e.GPRED), d.ld_nC, c.wt(MARL_P_PRE_W1), p4(d.nC), d.nC,
                                c.at(c.e.DAQ1), d.ld_nlb, (int)NR, d.nlb)));
    MARL_TRY(tn(c, c.at(c.e.GPRED), d.ld_nC, c.at(c.e.AQ1), d.ld_nlb, MARL_P_PRE_W1, d.nC, d.nlb, NR, grads[MARL_P_PRE_B1]));
    MARL_TRY(ln_bwd(c, c.at(c.e.DAQ1), d.ld_nlb, c.at(c.e.ZQ1), d.ld_nlb, c.at(c.e.STQ1),
                    MARL_P_PRE_LNW, MARL_P_PRE_LNB, NR, d.nlb, grads, 0));
    MARL_TRY(tn(c, c.at(c.e.DAQ1), d.ld_nlb, c.Hs(1), d.ld_nb, MARL_P_PRE_W0, d.nlb, d.n_b, NR, grads[MARL_P_PRE_B0]));
    // critic head (networks/policy.py:23-27)
    MARL_TRY(tn(c, c.at(c.e.DVAL), 4, c.at(c.e.AC1), d.ld_nla, MARL_P_CRI_W1, 1, d.nla, NR, grads[MARL_P_CRI_B1]));
    if (d.nla <= 384) {
        MARL_TRY(ln_bwd_rank(c, c.at(c.e.DVAL), 4, 1, MARL_P_CRI_W1, c.at(c.e.ZC1), d.ld_nla,
                             c.at(c.e.STC1), MARL_P_CRI_LNW, MARL_P_CRI_LNB, NR, d.nla, grads,
                             c.at(c.e.DAC1), d.ld_nla));
    } else {
        MARL_TRY(gemm1(c, gemm_prob(c.at(c.e.DVAL), 4, c.wt(MARL_P_CRI_W1), 4, 1, c.at(c.e.DAC1),
                                    d.ld_nla, (int)NR, d.nla)));
        MARL_TRY(ln_bwd(c, c.at(c.e.DAC1), d.ld_nla, c.at(c.e.ZC1), d.ld_nla, c.at(c.e.STC1),
                        MARL_P_CRI_LNW, MARL_P_CRI_LNB, NR, d.nla, grads, 0));
    }
    MARL_TRY(tn(c, c.at(c.e.DAC1), d.ld_nla, c.HCs(1), d.ld_na, MARL_P_CRI_W0, d.nla, d.n_a, NR, grads[MARL_P_CRI_B0]));
    // policy head: logp = log softmax(logits)[a]  (networks/policy.py:12-16, core/agent.py:57-61)
    MARL_TRY(launch_policy_dlogits(g_logp, c.PROBSs(0), c.ACTs(0), c.at(c.e.DLOG), d.ld_nA, NR,
                                   d.nA, st));
    MARL_TRY(tn(c, c.at(c.e.DLOG), d.ld_nA, c.at(c.e.AP1, 0), d.ld_nla, MARL_P_POL_W1, d.nA, d.nla, NR, grads[MARL_P_POL_B1]));
    if (d.nA <= 4 && d.nla <= 384) {
        MARL_TRY(ln_bwd_rank(c, c.at(c.e.DLOG), d.ld_nA, d.nA, MARL_P_POL_W1, c.at(c.e.ZP1, 0),
                             d.ld_nla, c.at(c.e.STP1, 0), MARL_P_POL_LNW, MARL_P_POL_LNB, NR, d.nla,
                             grads, c.at(c.e.DAP1), d.ld_nla));
    } else {
        MARL_TRY(gemm1(c, gemm_prob(c.at(c.e.DLOG), d.ld_nA, c.wt(MARL_P_POL_W1), p4(d.nA), d.nA,
                                    c.at(c.e.DAP1), d.ld_nla, (int)NR, d.nla)));
        MARL_TRY(ln_bwd(c, c.at(c.e.DAP1), d.ld_nla, c.at(c.e.ZP1, 0), d.ld_nla, c.at(c.e.STP1, 0),
                        MARL_P_POL_LNW, MARL_P_POL_LNB, NR, d.nla, grads, 0));
    }
    MARL_TRY(tn(c, c.at(c.e.DAP1), d.ld_nla, c.HCs(1), d.ld_na, MARL_P_POL_W0, d.nla, d.n_a, NR, grads[MARL_P_POL_B0]));
    // gradients reaching h_t / h^_t from the heads, all steps at once
    {
        GemmProb ph = gemm_prob(c.at(c.e.DAQ1), d.ld_nlb, c.wt(MARL_P_PRE_W0), d.ld_nlb, d.nlb,
                                c.DHs(1), d.ld_nb, (int)NR, d.n_b);
        GemmProb pa = gemm_prob(c.at(c.e.DAP1), d.ld_nla, c.wt(MARL_P_POL_W0), d.ld_nla, d.nla,
                                c.DHCs(1), d.ld_na, (int)NR, d.n_a);
        gemm_add_seg(pa, c.at(c.e.DAC1), d.ld_nla, c.wt(MARL_P_CRI_W0), d.ld_nla, d.nla);
        MARL_TRY(gemm2(c, ph, pa));
    }

    // Data parallelism (parallel.py, BucketedGradAllReduce): every gradient of the three heads' parameters (POL_*,
    // CRI_*, PRE_*) is complete here, ahead of the ~1.4 ms reverse loop - flush what they queued and mark the point.
    const bool heads_early = g_heads_event != nullptr;
    if (heads_early) {
        if (c.tq) MARL_TRY(launch_tn_queue(tq, c.rq, st));
        MARL_TRY(rq.flush());
        MARL_TRY(unpack_grads(c, grads, 1));  // (their matrices leave the packed layout now; the end skips them)
        MARL_HIP_CHECK(hipEventRecord(g_heads_event, st));
    }

    // ---- reverse-time loop over the recurrent chain ----------------------------------
    const size_t s_nmo = (size_t)d.R * d.ld_dbl, s_nm2 = (size_t)d.R * d.ld_nm2,
                 s_nm = (size_t)d.R * d.ld_nm;
    const bool panels = use_panels(d) && d.n_mo <= 384 && d.nm2 <= 384 && d.n_m <= 384;
    // the per-step LayerNorm reductions of the unfused path accumulate into the gradients step
    // by step: nothing of a parameter may still be queued then, so that path does not defer
    if (!panels) {
        MARL_TRY(launch_tn_queue(tq, c.rq, st));  // (what the heads queued so far)
        c.rq = nullptr;
        c.tq = nullptr;
    }
    // The action cell's backward of step t-1 only needs dh^_t, which is complete after step t's
    // W_hh product; it rides along (extra workgroups) with step t's decoder-panel launch, so
    // that from the second iteration on only the belief cell is left for the separate launch.
    const bool ride = panels && !use_side_stream();
    const bool dl_in_loop = ride;  // dU[:, nf:] (message + embedding columns) comes out of the loop
    bool action_done = false;  // the action cell of this step was handled by the ride-along
    bool belief_done = false;  // the belief cell: by the epilogue of the chained panel launch
    const bool chain = ride && use_chain(d);
    const bool g3 = c.e.g3;  // the gate gradients also leave as k16 images; the products below read those
    // every consumer on images (in-loop batch, dU, the four weight gradients): no fp32 copy of them at all
    const int skip_f32 = g3 && dl_in_loop && g3_tn_enabled(d);
    const int pln_blocks = chain ? panel_chain_blocks(d.na, d.nb) : panel_bwd_blocks(R);
    for (int t = ns - 1; t >= 0; --t) {
        const int first = (t == ns - 1);
        {   // cells of step t not yet handled inside the previous iteration's panel launch
            LstmBwdBatch lb{};
            int nc = 0;
            if (!belief_done)
                lb.a[nc++] = LstmBwdArgs{c.DHs(t + 1), c.at(c.e.DC), c.at(c.e.GB, t), c.Cs(t), c.Cs(t + 1), d.ld_nb,
                                         d.ld_nb, d.ld_gb, d.ld_nb, d.n_b, g3 ? c.img(c.e.GB3) : nullptr,
                                         (int)((int64_t)t * d.R), img_steps(4 * d.n_b), skip_f32};
            if (!action_done)
                lb.a[nc++] = LstmBwdArgs{c.DHCs(t + 1), c.at(c.e.DCC), c.at(c.e.GA, t), c.CCs(t), c.CCs(t + 1), d.ld_na,
                                         d.ld_na, d.ld_ga, d.ld_na, d.n_a, g3 ? c.img(c.e.GA3) : nullptr,
                                         (int)((int64_t)t * d.R), img_steps(4 * d.n_a), skip_f32};
            lb.rows = d.R;
            if (nc) MARL_TRY(launch_lstm_cell_bwd_batch(lb, nc, st));
        }
        action_done = belief_done = false;
        // The W_hh recurrent GEMM (main stream) and the decoder / encoder backward chain (side
        // stream) only meet at DH[t]: the chain's last kernel waits for the GEMM.
        const bool side = panels && use_side_stream();
        Ctx cs = c;
        if (side) {
            MARL_TRY(g_side.init());
            cs.st = g_side.s;
            MARL_TRY(g_side.order(c.st, cs.st));
        }
        hipStream_t st = cs.st;  // stream of the message chain below
        // recurrent paths dh_{t-1} += dgates * W_hh, and d(decoded message) = columns
        // [nf, nf + n_mo) of dU - three independent products of the same gate gradients
        float* ddbar = c.at(c.e.DDBAR) + (size_t)t * s_nmo;
        {
            GemmBatch gb{};
            gb.p[0] = gemm_prob(c.at(c.e.GB, t), d.ld_gb, c.wt(MARL_P_LB_WHH), d.ld_gb, 4 * d.n_b,
                                c.DHs(t), d.ld_nb, R, d.n_b, nullptr, 1);
            gb.p[1] = gemm_prob(c.at(c.e.GA, t), d.ld_ga, c.wt(MARL_P_LA_WHH), d.ld_ga, 4 * d.n_a,
                                c.DHCs(t), d.ld_na, R, d.n_a, nullptr, 1);
            GemmProb p = gemm_prob(c.at(c.e.GB, t), d.ld_gb,
                                   c.wt(MARL_P_LB_WIH) + (size_t)d.nf * d.ld_gb, d.ld_gb, 4 * d.n_b,
                                   ddbar, d.ld_dbl, R, d.n_mo);
            gemm_add_seg(p, c.at(c.e.GA, t), d.ld_ga, c.wt(MARL_P_LA_WIH) + (size_t)d.nf * d.ld_ga,
                         d.ld_ga, 4 * d.n_a);
            if (side) {  // W_hh products on the main stream, the message chain on the side stream
                gb.count = 2;
                MARL_TRY(launch_gemm_nt(gb, c.st));
                MARL_TRY(gemm1(cs, p));
            } else if (dl_in_loop) {
                // four products of EQUAL depth (a two-segment one would run twice as long as the
                // others and finish the launch alone): each cell's share of dU[:, nf:] - the
                // decoded-message AND the position-embedding columns, which sit next to each
                // other and fit the same two 64-wide tiles - goes to its own buffer; the decoder
                // panel sums the message halves while staging, the embedding halves are summed
                // once after the loop.  The big dU product then only covers the CNN features.
                gb.p[2] = gemm_prob(c.at(c.e.GB, t), d.ld_gb,
                                    c.wt(MARL_P_LB_WIH) + (size_t)d.nf * d.ld_gb, d.ld_gb, 4 * d.n_b,
                                    ddbar, d.ld_dbl, R, d.n_mo + d.n_d);
                gb.p[3] = gemm_prob(c.at(c.e.GA, t), d.ld_ga,
                                    c.wt(MARL_P_LA_WIH) + (size_t)d.nf * d.ld_ga, d.ld_ga, 4 * d.n_a,
                                    c.at(c.e.DDBAR2) + (size_t)t * s_nmo, d.ld_dbl, R, d.n_mo + d.n_d);
                gb.count = 4;
                if (g3) {
                    const int r0 = (int)((int64_t)t * d.R);
                    G3Batch g{};
                    g.p[0] = g3_prob(c.img(c.e.GB3), r0, c.wt3k(MARL_P_LB_WHH), 0, 4 * d.n_b, c.DHs(t), d.ld_nb, R,
                                     d.n_b, nullptr, 1);
                    g.p[1] = g3_prob(c.img(c.e.GA3), r0, c.wt3k(MARL_P_LA_WHH), 0, 4 * d.n_a, c.DHCs(t), d.ld_na, R,
                                     d.n_a, nullptr, 1);
                    g.p[2] = g3_prob(c.img(c.e.GB3), r0, c.wt3k(MARL_P_LB_WIH), d.nf, 4 * d.n_b, ddbar, d.ld_dbl, R,
                                     d.n_mo + d.n_d);
                    g.p[3] = g3_prob(c.img(c.e.GA3), r0, c.wt3k(MARL_P_LA_WIH), d.nf, 4 * d.n_a,
                                     c.at(c.e.DDBAR2) + (size_t)t * s_nmo, d.ld_dbl, R, d.n_mo + d.n_d);
                    g.count = 4;
                    MARL_TRY(launch_gemm_nt3(g, c.st));
                } else {
                    MARL_TRY(launch_gemm_nt(gb, c.st));
                }
            } else {
                gb.p[2] = p;
                gb.count = 3;
                MARL_TRY(launch_gemm_nt(gb, c.st));
            }
        }
        float* dad1 = c.at(c.e.DAD1) + (size_t)t * s_nm2;
        if (panels) {
            const size_t pblk = (size_t)pln_blocks * 2;
            PanelBwdProb pd{};
            pd.da = ddbar;
            pd.ldda = d.ld_dbl;
            if (dl_in_loop) {
                pd.da2 = c.at(c.e.DDBAR2) + (size_t)t * s_nmo;
                pd.ldda2 = d.ld_dbl;
            }
            pd.m = R;
            pd.nlayers = 2;
            pd.layer[0] = PanelBwdLayer{c.at(c.e.ZD2, t), d.ld_nmo, c.at(c.e.STD2, t),
                                        c.wp(MARL_P_DEC_LN1W), c.wp(MARL_P_DEC_LN1B), d.n_mo, ddbar,
                                        d.ld_dbl, c.at(c.e.PLN[0]) + (size_t)t * pblk * d.n_mo,
                                        c.wt(MARL_P_DEC_W1), p4(d.n_mo), d.nm2};
            pd.layer[1] = PanelBwdLayer{c.at(c.e.ZD1, t), d.ld_nm2, c.at(c.e.STD1, t),
                                        c.wp(MARL_P_DEC_LN0W), c.wp(MARL_P_DEC_LN0B), d.nm2, dad1,
                                        d.ld_nm2, c.at(c.e.PLN[1]) + (size_t)t * pblk * d.nm2,
                                        c.wt(MARL_P_DEC_W0), p4(d.nm2), d.n_m};
            pd.dx = c.at(c.e.DMBAR);
            pd.lddx = d.ld_nm;
            pd.accumulate = 0;
            if (ride && t > 0) {  // action cell of step t-1: gates GA[t-1], dh^ = DHC[t]
                pd.has_cell = 1;
                pd.cell = LstmBwdArgs{c.DHCs(t), c.at(c.e.DCC), c.at(c.e.GA, t - 1), c.CCs(t - 1),
                                      c.CCs(t), d.ld_na, d.ld_na, d.ld_ga, d.ld_na, d.n_a,
                                      g3 ? c.img(c.e.GA3) : nullptr, (int)((int64_t)(t - 1) * d.R), img_steps(4 * d.n_a),
                                      skip_f32};
                pd.cell_rows = d.R;
                action_done = true;
            }
            if (chain) {
                // decoder(t) -> mean -> encoder(t-1) -> dh_t -> belief cell(t-1), one launch
                pd.by_batch = panel_chain_by_batch(d.na);
                pd.g_na = d.na;
                pd.g_nb = d.nb;
                if (t > 0) {
                    pd.nlayers = 4;
                    pd.agg_at = 2;
                    pd.layer[2] = PanelBwdLayer{c.at(c.e.ZE2, t - 1), d.ld_nm, c.at(c.e.STE2, t - 1),
                                                c.wp(MARL_P_ENC_LN1W), c.wp(MARL_P_ENC_LN1B), d.n_m,
                                                c.at(c.e.DZE2) + (size_t)(t - 1) * s_nm, d.ld_nm,
                                                c.at(c.e.PLN[2]) + (size_t)(t - 1) * pblk * d.n_m,
                                                c.wt(MARL_P_ENC_W1), p4(d.n_m), d.nm2};
                    pd.layer[3] = PanelBwdLayer{c.at(c.e.ZE1, t - 1), d.ld_nm2, c.at(c.e.STE1, t - 1),
                                                c.wp(MARL_P_ENC_LN0W), c.wp(MARL_P_ENC_LN0B), d.nm2,
                                                c.at(c.e.DAE1) + (size_t)(t - 1) * s_nm2, d.ld_nm2,
                                                c.at(c.e.PLN[3]) + (size_t)(t - 1) * pblk * d.nm2,
                                                c.wt(MARL_P_ENC_W0), p4(d.nm2), d.n_b};
                    pd.dx = c.DHs(t);
                    pd.lddx = d.ld_nb;
                    pd.accumulate = 1;
                    pd.has_cellb = 1;
                    pd.cellb = LstmBwdArgs{c.DHs(t), c.at(c.e.DC), c.at(c.e.GB, t - 1), c.Cs(t - 1), c.Cs(t),
                                           d.ld_nb, d.ld_nb, d.ld_gb, d.ld_nb, d.n_b, g3 ? c.img(c.e.GB3) : nullptr,
                                           (int)((int64_t)(t - 1) * d.R), img_steps(4 * d.n_b), skip_f32};
                    belief_done = true;
                }
                MARL_TRY(launch_panel_bwd(pd, st));
                MARL_TRY(gate_image_fallback(c, pd, t - 1));
                continue;
            }
            MARL_TRY(launch_panel_bwd(pd, st));
            MARL_TRY(gate_image_fallback(c, pd, t - 1));
            if (t > 0) {
                float* dze2 = c.at(c.e.DZE2) + (size_t)(t - 1) * s_nm;
                float* dae1 = c.at(c.e.DAE1) + (size_t)(t - 1) * s_nm2;
                PanelBwdProb pe{};
                pe.da = c.at(c.e.DMBAR);  // message mean applied while staging
                pe.ldda = d.ld_nm;
                pe.agg_na = d.na;
                pe.agg_nb = d.nb;
                pe.m = R;
                pe.nlayers = 2;
                pe.layer[0] = PanelBwdLayer{c.at(c.e.ZE2, t - 1), d.ld_nm, c.at(c.e.STE2, t - 1),
                                            c.wp(MARL_P_ENC_LN1W), c.wp(MARL_P_ENC_LN1B), d.n_m,
                                            dze2, d.ld_nm,
                                            c.at(c.e.PLN[2]) + (size_t)(t - 1) * pblk * d.n_m,
                                            c.wt(MARL_P_ENC_W1), p4(d.n_m), d.nm2};
                pe.layer[1] = PanelBwdLayer{c.at(c.e.ZE1, t - 1), d.ld_nm2, c.at(c.e.STE1, t - 1),
                                            c.wp(MARL_P_ENC_LN0W), c.wp(MARL_P_ENC_LN0B), d.nm2,
                                            dae1, d.ld_nm2,
                                            c.at(c.e.PLN[3]) + (size_t)(t - 1) * pblk * d.nm2,
                                            c.wt(MARL_P_ENC_W0), p4(d.nm2), d.n_b};
                pe.dx = c.DHs(t);
                pe.lddx = d.ld_nb;
                pe.accumulate = 1;
                if (side) MARL_TRY(g_side.order(c.st, cs.st));  // after the W_hh GEMM's DH[t] update
                MARL_TRY(launch_panel_bwd(pe, st));
            }
            if (side) MARL_TRY(g_side.order(cs.st, c.st));  // join before step t-1
            continue;
        }
        MARL_TRY(ln_bwd(c, ddbar, d.ld_dbl, c.at(c.e.ZD2, t), d.ld_nmo, c.at(c.e.STD2, t),
                        MARL_P_DEC_LN1W, MARL_P_DEC_LN1B, d.R, d.n_mo, grads, !first));
        MARL_TRY(gemm1(c, gemm_prob(ddbar, d.ld_dbl, c.wt(MARL_P_DEC_W1), p4(d.n_mo), d.n_mo, dad1,
                                    d.ld_nm2, R, d.nm2)));
        MARL_TRY(ln_bwd(c, dad1, d.ld_nm2, c.at(c.e.ZD1, t), d.ld_nm2, c.at(c.e.STD1, t),
                        MARL_P_DEC_LN0W, MARL_P_DEC_LN0B, d.R, d.nm2, grads, !first));
        if (t > 0) {
            // through the message mean (self-adjoint) into the encoder of step t-1
            MARL_TRY(gemm1(c, gemm_prob(dad1, d.ld_nm2, c.wt(MARL_P_DEC_W0), p4(d.nm2), d.nm2,
                                        c.at(c.e.DMBAR), d.ld_nm, R, d.n_m)));
            float* dze2 = c.at(c.e.DZE2) + (size_t)(t - 1) * s_nm;
            MARL_TRY(launch_agg_msg(c.at(c.e.DMBAR), dze2, d.ld_nm, d.na, d.nb, d.n_m, st));
            const int efirst = (t == ns - 1);
            MARL_TRY(ln_bwd(c, dze2, d.ld_nm, c.at(c.e.ZE2, t - 1), d.ld_nm, c.at(c.e.STE2, t - 1),
                            MARL_P_ENC_LN1W, MARL_P_ENC_LN1B, d.R, d.n_m, grads, !efirst));
            float* dae1 = c.at(c.e.DAE1) + (size_t)(t - 1) * s_nm2;
            MARL_TRY(gemm1(c, gemm_prob(dze2, d.ld_nm, c.wt(MARL_P_ENC_W1), p4(d.n_m), d.n_m, dae1,
                                        d.ld_nm2, R, d.nm2)));
            MARL_TRY(ln_bwd(c, dae1, d.ld_nm2, c.at(c.e.ZE1, t - 1), d.ld_nm2,
                            c.at(c.e.STE1, t - 1), MARL_P_ENC_LN0W, MARL_P_ENC_LN0B, d.R, d.nm2,
                            grads, !efirst));
            MARL_TRY(gemm1(c, gemm_prob(dae1, d.ld_nm2, c.wt(MARL_P_ENC_W0), p4(d.nm2), d.nm2,
                                        c.DHs(t), d.ld_nb, R, d.n_b, nullptr, 1)));
        }
    }
    if (panels) {  // LayerNorm affine gradients of the in-loop layers: one reduction each
        const int64_t nblk = pln_blocks;
        MARL_TRY(launch_reduce_affine(c.at(c.e.PLN[0]), nblk * ns, d.n_mo, grads[MARL_P_DEC_LN1W],
                                      grads[MARL_P_DEC_LN1B], 0, st, c.rq));
        MARL_TRY(launch_reduce_affine(c.at(c.e.PLN[1]), nblk * ns, d.nm2, grads[MARL_P_DEC_LN0W],
                                      grads[MARL_P_DEC_LN0B], 0, st, c.rq));
        if (ns > 1) {
            MARL_TRY(launch_reduce_affine(c.at(c.e.PLN[2]), nblk * (ns - 1), d.n_m,
                                          grads[MARL_P_ENC_LN1W], grads[MARL_P_ENC_LN1B], 0, st, c.rq));
            MARL_TRY(launch_reduce_affine(c.at(c.e.PLN[3]), nblk * (ns - 1), d.nm2,
                                          grads[MARL_P_ENC_LN0W], grads[MARL_P_ENC_LN0B], 0, st, c.rq));
        }
    }

    // ---- weight gradients of the recurrent chain: one contraction over all steps -------
    MARL_TRY(tn(c, c.at(c.e.DDBAR), d.ld_dbl, c.at(c.e.AD1, 0), d.ld_nm2, MARL_P_DEC_W1, d.n_mo, d.nm2, NR, grads[MARL_P_DEC_B1]));
    MARL_TRY(tn(c, c.at(c.e.DAD1), d.ld_nm2, c.at(c.e.MBAR, 0), d.ld_nm, MARL_P_DEC_W0, d.nm2, d.n_m, NR, grads[MARL_P_DEC_B0]));
    if (ns > 1) {
        const int64_t er = (int64_t)(ns - 1) * d.R;  // the last step's message is never read
        MARL_TRY(tn(c, c.at(c.e.DZE2), d.ld_nm, c.at(c.e.AE1, 0), d.ld_nm2, MARL_P_ENC_W1, d.n_m, d.nm2, er, grads[MARL_P_ENC_B1]));
        MARL_TRY(tn(c, c.at(c.e.DAE1), d.ld_nm2, c.Hs(1), d.ld_nb, MARL_P_ENC_W0, d.nm2, d.n_b, er, grads[MARL_P_ENC_B0]));
    } else {
        const int enc[] = {MARL_P_ENC_B0, MARL_P_ENC_LN0W, MARL_P_ENC_LN0B, MARL_P_ENC_B1,
                           MARL_P_ENC_LN1W, MARL_P_ENC_LN1B};
        for (int i : enc) MARL_TRY(launch_fill(grads[i], param_meta(d, i).n, 0.f, st));
        MARL_TRY(launch_fill(c.gp(MARL_P_ENC_W0), (int64_t)d.nm2 * c.w.ldp[MARL_P_ENC_W0], 0.f, st));
        MARL_TRY(launch_fill(c.gp(MARL_P_ENC_W1), (int64_t)d.n_m * c.w.ldp[MARL_P_ENC_W1], 0.f, st));
    }
    if (g3 && g3_tn_enabled(d)) {
        // contraction over the rows of the images written by the cell-backward kernels (gate gradients),
        // the forward kernels (U) and the LSTM epilogues (h, h^); bias gradients = column sums of A
        if (g3_tn_cell_ok(4 * d.n_b, d.nin, d.n_b, NR)) {
            MARL_TRY(tn3_cell(c, c.img(c.e.GB3), 4 * d.n_b, c.img(c.e.U3), d.nin, MARL_P_LB_WIH, c.img(c.e.H3), d.n_b,
                              MARL_P_LB_WHH, NR, grads[MARL_P_LB_BIH]));
        } else {
            MARL_TRY(tn3(c, c.img(c.e.GB3), 4 * d.n_b, c.img(c.e.U3), d.nin, MARL_P_LB_WIH, NR, nullptr));
            MARL_TRY(tn3(c, c.img(c.e.GB3), 4 * d.n_b, c.img(c.e.H3), d.n_b, MARL_P_LB_WHH, NR, grads[MARL_P_LB_BIH]));
        }
        if (g3_tn_cell_ok(4 * d.n_a, d.nin, d.n_a, NR)) {
            MARL_TRY(tn3_cell(c, c.img(c.e.GA3), 4 * d.n_a, c.img(c.e.U3), d.nin, MARL_P_LA_WIH, c.img(c.e.HC3), d.n_a,
                              MARL_P_LA_WHH, NR, grads[MARL_P_LA_BIH]));
        } else {
            MARL_TRY(tn3(c, c.img(c.e.GA3), 4 * d.n_a, c.img(c.e.U3), d.nin, MARL_P_LA_WIH, NR, nullptr));
            MARL_TRY(tn3(c, c.img(c.e.GA3), 4 * d.n_a, c.img(c.e.HC3), d.n_a, MARL_P_LA_WHH, NR, grads[MARL_P_LA_BIH]));
        }
    } else {
    MARL_TRY(tn(c, c.at(c.e.GB, 0), d.ld_gb, c.at(c.e.U, 0), d.ld_nin, MARL_P_LB_WIH, 4 * d.n_b, d.nin, NR));
    MARL_TRY(tn(c, c.at(c.e.GB, 0), d.ld_gb, c.Hs(0), d.ld_nb, MARL_P_LB_WHH, 4 * d.n_b, d.n_b, NR, grads[MARL_P_LB_BIH]));
    MARL_TRY(tn(c, c.at(c.e.GA, 0), d.ld_ga, c.at(c.e.U, 0), d.ld_nin, MARL_P_LA_WIH, 4 * d.n_a, d.nin, NR));
    MARL_TRY(tn(c, c.at(c.e.GA, 0), d.ld_ga, c.HCs(0), d.ld_na, MARL_P_LA_WHH, 4 * d.n_a, d.n_a, NR, grads[MARL_P_LA_BIH]));
    }

    // ---- dU for all steps, then position embedding and CNN backward -------------------
    {
        // with the [nf, nin) columns already produced step by step, only the CNN features remain
        if (g3 && dl_in_loop) {
            G3Batch g{};
            g.p[0] = g3_prob(c.img(c.e.GB3), 0, c.wt3k(MARL_P_LB_WIH), 0, 4 * d.n_b, c.at(c.e.DU), d.ld_nin, (int)NR,
                             d.nf);
            g3_add_seg(g.p[0], c.img(c.e.GA3), 0, c.wt3k(MARL_P_LA_WIH), 0, 4 * d.n_a);
            g.count = 1;
            MARL_TRY(launch_gemm_nt3(g, st));
        } else {
        GemmProb p = gemm_prob(c.at(c.e.GB, 0), d.ld_gb, c.wt(MARL_P_LB_WIH), d.ld_gb, 4 * d.n_b,
                               c.at(c.e.DU), d.ld_nin, (int)NR, dl_in_loop ? d.nf : d.nin);
        gemm_add_seg(p, c.at(c.e.GA, 0), d.ld_ga, c.wt(MARL_P_LA_WIH), d.ld_ga, 4 * d.n_a);
        MARL_TRY(gemm1(c, p));
        }
        if (dl_in_loop)  // d(position embedding) = belief share + action share
            MARL_TRY(launch_add2d(c.at(c.e.DDBAR) + d.n_mo, d.ld_dbl, c.at(c.e.DDBAR2) + d.n_mo, d.ld_dbl,
                                  c.at(c.e.DU) + d.nf + d.n_mo, d.ld_nin, NR, d.n_d, st));
    }
    MARL_TRY(ln_bwd(c, c.at(c.e.DU) + d.nf + d.n_mo, d.ld_nin, c.at(c.e.ZPOS, 0), d.ld_nd,
                    c.at(c.e.STPOS, 0), MARL_P_POS_LNW, MARL_P_POS_LNB, NR, d.n_d, grads, 0,
                    c.at(c.e.DZPOS), d.ld_nd));
    MARL_TRY(tn(c, c.at(c.e.DZPOS), d.ld_nd, c.at(c.e.NPOS, 0), 4, MARL_P_POS_W, d.n_d, 2, NR, grads[MARL_P_POS_B]));
    {
        const float* da = c.at(c.e.DU);
        int64_t ldda = d.ld_nin;
        int chw = 1;
        bool have_dz = false;  // DZ[l] already produced by the fused layer backward of layer l+1
        bool w0_done = false;  // ... which also formed layer 0's weight gradient (dZ_0 never left LDS)
        for (int l = d.L - 1; l >= 0; --l) {
            const int co = d.ch[l + 1];
            const int64_t rows = NR * d.P[l];
            float* dz = c.at(c.e.DZ[l]);
            if (!have_dz) {
                RedQueue* q;
                float* part = part_scratch(c, gn_bwd_blocks(NR, co), co, 0, q);
                MARL_TRY(launch_gn_silu_bwd(da, ldda, chw, c.at(c.e.Z[l], 0), c.at(c.e.GST[l], 0),
                                            c.wp(4 * l + 2), c.wp(4 * l + 3), dz, part, NR, d.P[l], co,
                                            d.grp[l], st));
                MARL_TRY(launch_reduce_affine(part, gn_bwd_blocks(NR, co), co, grads[4 * l + 2],
                                              grads[4 * l + 3], 0, st, q));
            }
            have_dz = false;
            if (l == 0 && w0_done) {
                // (layer 0's weight and bias gradient came out of the launch that produced dZ_0)
            } else if (c.e.wgrad_ok[l]) {
                // dW_l (and db_l) from dZ_l and the layer's input, recomputed from what forward
                // kept (Z_{l-1} + statistics, or the image patch): no im2col rows in HBM
                CnnWgradArgs w = cnn_wgrad_shape(d, l);
                w.dz = dz;
                w.img = img;
                w.img_u8 = img_u8;
                w.pos = c.POSs(0);
                if (l > 0) {
                    w.zin = c.at(c.e.Z[l - 1], 0);
                    w.gst = c.at(c.e.GST[l - 1], 0);
                    w.gamma = c.wp(4 * (l - 1) + 2);
                    w.beta = c.wp(4 * (l - 1) + 3);
                } else if (!img) {
                    set_error("episode_backward: the image batch of the forward call is needed");
                    return MARL_EINVAL;
                }
                // per-workgroup slabs; the launcher places part_b behind the weight slabs
                w.part_w = c.at(c.e.TNS);
                if (c.rq && c.defer_this((size_t)cnn_wgrad_blocks(w) * ((size_t)co * d.K[l] + co) * sizeof(float))) {
                    float* p = c.rq->take((size_t)cnn_wgrad_blocks(w) * ((size_t)co * d.K[l] + co));
                    if (c.rq->rc == MARL_OK) w.part_w = p;
                }
                MARL_TRY(launch_cnn_wgrad(w, st));
                if (c.rq && w.part_w != c.at(c.e.TNS)) {
                    c.rq->push(w.part_w, (int64_t)co * d.K[l], w.blocks, co * d.K[l], c.gp(4 * l),
                               co * d.K[l], d.K[l], c.w.ldp[4 * l], nullptr, 0);
                    c.rq->push(w.part_b, co, w.blocks, co, grads[4 * l + 1], co, co, co, nullptr, 0);
                } else {
                    MARL_TRY(launch_slab_reduce(w.part_w, (int64_t)co * d.K[l], w.blocks, c.gp(4 * l),
                                                c.w.ldp[4 * l], co, d.K[l], w.part_b, grads[4 * l + 1], st));
                }
            } else {
                MARL_TRY(tn(c, dz, co, c.at(c.e.COLS[l], 0), d.ldk[l], 4 * l, co, d.K[l], rows, grads[4 * l + 1]));
            }
            if (l > 0) {
                // dZ_l -> dZ_{l-1} in one launch (transposed conv + GroupNorm/SiLU backward)
                CnnDgradArgs g = cnn_dgrad_shape(d, l);
                g.dz = dz;
                g.wt = c.wt(4 * l);
                g.ldwt = p4(co);
                g.zin = c.at(c.e.Z[l - 1], 0);
                g.gst = c.at(c.e.GST[l - 1], 0);
                g.gamma = c.wp(4 * (l - 1) + 2);
                g.beta = c.wp(4 * (l - 1) + 3);
                g.dzin = c.at(c.e.DZ[l - 1]);
                if (g.w0 && !img) g.w0 = 0;  // (the step API has no image batch: layer 0 takes the separate launch)
                if (c.e.dgrad_ok[l] &&
                    (size_t)cnn_dgrad_blocks(g) * 2 * d.ch[l] <= c.e.part_floats) {
                    RedQueue* q;
                    const int nblk = cnn_dgrad_blocks(g);
                    g.part = part_scratch(c, nblk, d.ch[l], 0, q);
                    const int co0 = d.ch[l], k0 = d.K[0];
                    if (g.w0) {  // + layer 0's weight gradient: per-workgroup slabs (tiny: 448 floats each at RESISC)
                        const size_t fl = (size_t)nblk * ((size_t)co0 * k0 + co0);
                        g.w0_part = c.at(c.e.TNS);
                        if (c.rq && c.defer_this(fl * sizeof(float))) {
                            float* p = c.rq->take(fl);
                            if (c.rq->rc == MARL_OK) g.w0_part = p;
                        }
                        g.w0_bpart = g.w0_part + (size_t)nblk * co0 * k0;
                        g.img = img;
                        g.img_u8 = img_u8;
                        g.pos = c.POSs(0);
                        g.dzin = nullptr;  // dZ_0 stays in LDS
                    }
                    MARL_TRY(launch_cnn_dgrad(g, st));
                    MARL_TRY(launch_reduce_affine(g.part, nblk, d.ch[l],
                                                  grads[4 * (l - 1) + 2], grads[4 * (l - 1) + 3], 0, st, q));
                    if (g.w0) {
                        if (c.rq && g.w0_part != c.at(c.e.TNS)) {
                            c.rq->push(g.w0_part, (int64_t)co0 * k0, nblk, co0 * k0, c.gp(0), co0 * k0, k0, c.w.ldp[0], nullptr, 0);
                            c.rq->push(g.w0_bpart, co0, nblk, co0, grads[1], co0, co0, co0, nullptr, 0);
                        } else {
                            MARL_TRY(launch_slab_reduce(g.w0_part, (int64_t)co0 * k0, nblk, c.gp(0), c.w.ldp[0], co0, k0,
                                                        g.w0_bpart, grads[1], st));
                        }
                        w0_done = true;
                    }
                    have_dz = true;
                    continue;
                }
                g.w0 = 0;
                MARL_TRY(gemm1(c, gemm_prob(dz, co, c.wt(4 * l), p4(co), co, c.at(c.e.DCOLS[l]),
                                            d.ldk[l], (int)rows, d.K[l])));
                MARL_TRY(launch_col2im(c.at(c.e.DCOLS[l]), d.ldk[l], c.at(c.e.DA[l - 1]), NR,
                                       d.hw[l], d.ch[l], st));
                da = c.at(c.e.DA[l - 1]);
                ldda = (int64_t)d.P[l - 1] * d.ch[l];
                chw = 0;
            }
        }
    }
    if (c.tq) MARL_TRY(launch_tn_queue(tq, c.rq, st));  // the small weight gradients, one launch
    MARL_TRY(rq.flush());
    return unpack_grads(c, grads, heads_early ? 2 : 0);
}

}  // namespace marl

// ===========================================================================
// C ABI
// ===========================================================================
using namespace marl;

extern "C" {

int marl_abi_version(void) { return MARL_ABI_VERSION; }

int64_t marl_param_numel(const marl_config* cfg, int index) {
    Dims d;
    if (make_dims(cfg, d) != MARL_OK || index < 0 || index >= MARL_NPARAMS) return -1;
    const ParamMeta m = param_meta(d, index);
    if (m.kind == PK_NONE) return 0;
    return m.kind == PK_VEC ? m.n : (int64_t)m.n * m.k;
}

int marl_workspace_sizes(const marl_config* cfg, int train, size_t* wbytes, size_t* ebytes) {
    Dims d;
    MARL_TRY(make_dims(cfg, d));
    WLayout w;
    make_wlayout(d, w);
    ELayout e;
    make_elayout(d, train, e);
    if (wbytes) *wbytes = w.total * sizeof(float);
    if (ebytes) *ebytes = e.total * sizeof(float);
    return MARL_OK;
}

int marl_pack_weights(const marl_config* cfg, const float* const* params_host, void* weights_ws,
                      size_t weights_ws_bytes, void* stream) {
    Dims d;
    MARL_TRY(make_dims(cfg, d));
    if (!params_host || !weights_ws) {
        set_error("null argument");
        return MARL_EINVAL;
    }
    WLayout w;
    make_wlayout(d, w);
    if (weights_ws_bytes < w.total * sizeof(float)) {
        set_error("pack_weights: weights workspace too small (%zu of %zu bytes)", weights_ws_bytes,
                  w.total * sizeof(float));
        return MARL_ESIZE;
    }
    return pack_weights(d, w, params_host, static_cast<float*>(weights_ws),
                        static_cast<hipStream_t>(stream));
}

int marl_patch_gather(const float* img, const int64_t* pos, float* obs, int nb_agents, int batch,
                      int c, int h, int w, int f, void* stream) {
    if (!img || !pos || !obs || nb_agents < 1 || batch < 1 || c < 1 || f < 1 || h < f || w < f) {
        set_error("patch_gather: bad argument");
        return MARL_EINVAL;
    }
    return launch_patch_gather(img, pos, obs, nb_agents, batch, c, h, w, f,
                               static_cast<hipStream_t>(stream));
}

}  // extern "C"

namespace marl {
__global__ void transition_kernel(const int64_t* __restrict__ pos_in,
                                  const int64_t* __restrict__ actions,
                                  int64_t* __restrict__ pos_out, SampleArgs tbl, int rows, int nA,
                                  int H, int W, int f) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    int a = (int)actions[r];
    a = a < 0 ? 0 : (a >= nA ? nA - 1 : a);
    const int64_t p0 = pos_in[r * 2], p1 = pos_in[r * 2 + 1];
    const int64_t q0 = p0 + tbl.table[a][0], q1 = p1 + tbl.table[a][1];
    const bool ok = q0 >= 0 && q0 + f < H && q1 >= 0 && q1 + f < W;
    pos_out[r * 2] = ok ? q0 : p0;
    pos_out[r * 2 + 1] = ok ? q1 : p1;
}
}  // namespace marl

extern "C" {

int marl_transition(const int64_t* pos_in, const int64_t* actions, int64_t* pos_out,
                    const int32_t* table_host, int nb_action, int rows, int h, int w, int f,
                    void* stream) {
    if (!pos_in || !actions || !pos_out || !table_host || nb_action < 1 ||
        nb_action > MARL_MAX_ACTIONS || rows < 1) {
        set_error("transition: bad argument");
        return MARL_EINVAL;
    }
    SampleArgs tbl;
    memset(&tbl, 0, sizeof(tbl));
    for (int j = 0; j < nb_action; ++j) {
        tbl.table[j][0] = table_host[2 * j];
        tbl.table[j][1] = table_host[2 * j + 1];
    }
    hipLaunchKernelGGL(transition_kernel, dim3((unsigned)cdiv(rows, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), pos_in, actions, pos_out, tbl, rows,
                       nb_action, h, w, f);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

int marl_episode_forward(const marl_config* cfg, const void* weights_ws, size_t weights_ws_bytes,
                         void* episode_ws, size_t episode_ws_bytes, const void* img, const int64_t* pos0, const float* h0, const float* c0,
                         const float* hc0, const float* cc0, const float* noise,
                         const int64_t* forced_actions, uint64_t rng_seed, uint64_t rng_offset,
                         const void* counters, float* step_preds, float* step_logp,
                         float* step_values, int64_t* step_pos, int64_t* step_actions, int train,
                         void* stream) {
    Ctx c;
    SplitRegistryScope reg_scope;
    MARL_TRY(make_ctx(cfg, weights_ws, weights_ws_bytes, episode_ws, episode_ws_bytes, train, stream, c));
    if (!img || !pos0 || !h0 || !c0 || !hc0 || !cc0 || !step_preds || !step_logp || !step_values) {
        set_error("episode_forward: null argument");
        return MARL_EINVAL;
    }
    if (c.d.ns >= 65536) {
        set_error("episode_forward: more than 65535 steps");
        return MARL_ELIMIT;
    }
    const Dims& d = c.d;
    MARL_TRY(launch_i64_to_i32(pos0, c.POSs(0), d.R * 2, c.st));
    MARL_TRY(load_state(c, h0, c0, hc0, cc0, nullptr));
    StepIn in;
    in.img = img;
    in.img_u8 = cfg->img_u8 != 0;
    const bool side = use_side_stream();
    Ctx c2 = c;
    if (side) {
        MARL_TRY(g_side.init());
        c2.st = g_side.s;
    }
    bool decoded_ahead = false;  // decoder(t) already ran with the sampling of step t-1
    const bool chain = !side && use_chain(d);
    for (int t = 0; t < d.ns; ++t) {
        MARL_TRY(step_cnn(c, t, in));
        if (side && t > 0)
            MARL_TRY(g_side.order(c2.st, c.st));  // decoder(t) ran on the side stream
        else if (!decoded_ahead)
            MARL_TRY(step_decode(c, t));
        MARL_TRY(step_pos_lstm(c, t, in, t > 0));  // lambda_t (t > 0) came from sample(t-1)
        if (chain) {
            MARL_TRY(step_chain(c, t));  // encoder(t) -> decoder(t+1) || policy layer(t)
        } else if (side) {
            MARL_TRY(g_side.order(c.st, c2.st));  // side stream: after the LSTM of step t
            MARL_TRY(step_encode_policy(c2, t, 1));
            if (t + 1 < d.ns) MARL_TRY(step_decode(c2, t + 1));
            MARL_TRY(step_encode_policy(c, t, 2));
        } else {
            MARL_TRY(step_encode_policy(c, t, 3));
        }
        SampleArgs a;
        fill_sample_args(c, cfg, t, a);
        a.noise = noise ? noise + (size_t)t * d.R * d.nA : nullptr;
        a.rng_on = !noise && !forced_actions;  // perf mode: Exp(1) drawn inside the kernel
        a.rng_seed = rng_seed ^ 0x9E3779B97F4A7C15ull;  // a key of its own (marl_draw_episode uses rng_seed)
        a.rng_ctr = (rng_offset << 16) + (uint64_t)t;
        a.rng_off_dev = counters ? &static_cast<const Counters*>(counters)->rng_offset : nullptr;
        a.forced = forced_actions ? forced_actions + (size_t)t * d.R : nullptr;
        a.step_pos = step_pos ? step_pos + (size_t)t * d.R * 2 : nullptr;
        a.step_actions = step_actions ? step_actions + (size_t)t * d.R : nullptr;
        a.step_logp = step_logp + (size_t)t * d.R;
        if (t + 1 < d.ns) {  // position embedding of step t+1 rides along with the move
            a.pe_W = c.wp(MARL_P_POS_W);
            a.pe_b = c.wp(MARL_P_POS_B);
            a.pe_gamma = c.wp(MARL_P_POS_LNW);
            a.pe_beta = c.wp(MARL_P_POS_LNB);
            a.pe_npos = c.at(c.e.NPOS, t + 1);
            a.pe_z = c.at(c.e.ZPOS, t + 1);
            a.pe_stats = c.at(c.e.STPOS, t + 1);
            a.pe_out = c.at(c.e.U, t + 1) + d.nf + d.n_mo;
            a.pe_ldz = d.ld_nd;
            a.pe_ldo = d.ld_nin;
            a.pe_nd = d.n_d;
            if (c.u3_by_producers) {
                a.pe_img = c.img(c.e.U3);
                a.pe_row0 = c.u3_row(t + 1);
                a.pe_steps = img_steps(d.nin);
                a.pe_col0 = d.nf + d.n_mo;
            }
        }
        // sample(t) and decoder(t+1) are independent (the decoder needs MSG[t+1], written by the
        // encoder above): one launch runs both
        decoded_ahead = chain && t + 1 < d.ns;
        if (!side && !chain && t + 1 < d.ns) MARL_TRY(step_decode(c, t + 1, &a, &decoded_ahead));
        if (chain || !decoded_ahead) MARL_TRY(launch_sample(a, c.st));
    }
    if (side) MARL_TRY(g_side.order(c2.st, c.st));  // join before the caller's stream continues
    return heads_batched(c, 0, d.NR, step_values, step_preds);
}

int marl_episode_backward(const marl_config* cfg, void* weights_ws, size_t weights_ws_bytes,
                          void* episode_ws, size_t episode_ws_bytes, const void* img, const float* g_preds, const float* g_logp,
                          const float* g_values, float* const* grads_host, void* stream) {
    Ctx c;
    SplitRegistryScope reg_scope;
    MARL_TRY(make_ctx(cfg, weights_ws, weights_ws_bytes, episode_ws, episode_ws_bytes, 1, stream, c));
    if (!grads_host) {
        set_error("episode_backward: null gradient table");
        return MARL_EINVAL;
    }
    return episode_backward(c, img, cfg->img_u8 != 0, g_preds, g_logp, g_values, grads_host);
}

int marl_backward_heads_event(void* hip_event) {
    g_heads_event = static_cast<hipEvent_t>(hip_event);
    return MARL_OK;
}

int marl_draw_episode(const marl_config* cfg, uint64_t seed, uint64_t offset, const void* counters,
                      int64_t* pos0, float* h0, float* c0, float* hc0, float* cc0, float* noise,
                      void* stream) {
    Dims d;
    MARL_TRY(make_dims(cfg, d));
    if (!pos0 || !h0 || !c0 || !hc0 || !cc0) {
        set_error("draw_episode: null argument");
        return MARL_EINVAL;
    }
    return launch_draw_episode(seed, offset,
                               counters ? &static_cast<const Counters*>(counters)->rng_offset : nullptr,
                               pos0, (int)d.R, d.H, d.W, d.f, h0, c0, d.n_b, hc0, cc0, d.n_a, noise,
                               noise ? d.NR * d.nA : 0, static_cast<hipStream_t>(stream));
}

int marl_counters_set(void* counters, uint64_t rng_offset, int64_t step, float lr, float beta1,
                      float beta2, void* stream) {
    if (!counters || step < 1) {
        set_error("counters_set: bad argument");
        return MARL_EINVAL;
    }
    return launch_counters_set(static_cast<Counters*>(counters), rng_offset, step, lr, beta1, beta2, 0,
                               static_cast<hipStream_t>(stream));
}

int marl_counters_tick(void* counters, float lr, float beta1, float beta2, void* stream) {
    if (!counters) {
        set_error("counters_tick: null block");
        return MARL_EINVAL;
    }
    return launch_counters_set(static_cast<Counters*>(counters), 0, 0, lr, beta1, beta2, 1,
                               static_cast<hipStream_t>(stream));
}

// ---- hipGraph capture / replay -------------------------------------------------------------
int marl_graph_begin(void* stream) {
    if (!stream) {
        set_error("graph_begin: the NULL stream cannot be captured");
        return MARL_EINVAL;
    }
    MARL_HIP_CHECK(hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
    return MARL_OK;
}

int marl_graph_end(void* stream, void** graph_exec_out) {
    if (!stream || !graph_exec_out) return MARL_EINVAL;
    hipGraph_t graph = nullptr;
    MARL_HIP_CHECK(hipStreamEndCapture(static_cast<hipStream_t>(stream), &graph));
    hipGraphExec_t exec = nullptr;
    const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return MARL_EHIP;
    }
    *graph_exec_out = exec;
    return MARL_OK;
}

int marl_graph_launch(void* graph_exec, void* stream) {
    if (!graph_exec) return MARL_EINVAL;
    MARL_HIP_CHECK(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipStream_t>(stream)));
    return MARL_OK;
}

int marl_graph_destroy(void* graph_exec) {
    if (graph_exec) MARL_HIP_CHECK(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
    return MARL_OK;
}

int marl_a2c_loss_fwd_bwd(const marl_config* cfg, void* episode_ws, size_t episode_ws_bytes,
                          const float* step_preds,
                          const float* step_logp, const float* step_values, const int64_t* y,
                          float gamma, float* g_preds, float* g_logp, float* g_values,
                          float* scalars_out, double* adv_stats, int phase, void* stream) {
    Dims d;
    MARL_TRY(make_dims(cfg, d));
    if (!episode_ws || !step_preds || !step_logp || !step_values || !y || !scalars_out ||
        !adv_stats || phase < 0 || phase > 2) {
        set_error("a2c_loss: bad argument");
        return MARL_EINVAL;
    }
    // the loss scratch is the last buffer of either episode-workspace layout
    ELayout e;
    make_elayout(d, 1, e);
    if (episode_ws_bytes < e.total * sizeof(float)) {
        set_error("a2c_loss: episode workspace too small (%zu of %zu bytes, training layout)", episode_ws_bytes,
                  e.total * sizeof(float));
        return MARL_ESIZE;
    }
    LossArgs a;
    a.preds = step_preds;
    a.logp = step_logp;
    a.values = step_values;
    a.y = y;
    a.g_preds = g_preds;
    a.ld_gp = d.nC;
    a.g_logp = g_logp;
    a.g_values = g_values;
    a.scalars = scalars_out;
    a.adv_stats = adv_stats;
    a.scratch = static_cast<float*>(episode_ws) + e.LOSS;
    a.ns = d.ns;
    a.na = d.na;
    a.nb = d.nb;
    a.nc = d.nC;
    a.gamma = gamma;
    a.phase = phase;
    return launch_loss(a, static_cast<hipStream_t>(stream));
}

int marl_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                   int64_t step, float lr, float beta1, float beta2, float eps, float grad_scale,
                   const void* counters, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || step < 1) {
        set_error("adam: bad argument");
        return MARL_EINVAL;
    }
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    return launch_adam(params, grads, exp_avg, exp_avg_sq, n, (float)((double)lr / bc1),
                       (float)(1.0 / sqrt(bc2)), beta1, beta2, eps, grad_scale,
                       static_cast<hipStream_t>(stream), static_cast<const Counters*>(counters));
}

int marl_step_forward(const marl_config* cfg, const void* weights_ws, size_t weights_ws_bytes,
                      void* episode_ws, size_t episode_ws_bytes, const float* obs, const float* msg, const float* norm_pos, const float* h,
                      const float* cc_, const float* hc, const float* cca, float* probs,
                      float* values, float* preds, float* new_msg, float* h_out, float* c_out,
                      float* hc_out, float* cc_out, const float* noise, uint64_t rng_seed,
                      uint64_t rng_offset, int64_t* actions_out, float* logp_out, void* stream) {
    Ctx c;
    SplitRegistryScope reg_scope;
    MARL_TRY(make_ctx(cfg, weights_ws, weights_ws_bytes, episode_ws, episode_ws_bytes, 0, stream, c));
    if (!obs || !msg || !norm_pos || !h || !cc_ || !hc || !cca || !probs || !values || !preds ||
        !new_msg || !h_out || !c_out || !hc_out || !cc_out) {
        set_error("step_forward: null argument");
        return MARL_EINVAL;
    }
    const Dims& d = c.d;
    MARL_TRY(load_state(c, h, cc_, hc, cca, msg));
    StepIn in;
    in.obs = obs;
    in.npos = norm_pos;
    MARL_TRY(step_core(c, 0, in));
    SampleArgs a;
    fill_sample_args(c, cfg, 0, a);
    a.probs = probs;
    if (actions_out && logp_out) {  // also sample (positions of the scratch slot are unused)
        a.noise = noise;
        a.rng_on = !noise;
        a.rng_seed = rng_seed ^ 0x9E3779B97F4A7C15ull;
        a.rng_ctr = rng_offset << 16;
        a.step_actions = actions_out;
        a.step_logp = logp_out;
    } else {
        a.step_logp = nullptr;  // probabilities only
    }
    MARL_TRY(launch_sample(a, c.st));
    MARL_TRY(heads_batched(c, 0, d.R, values, preds));
    MARL_TRY(launch_copy2d(c.MSGs(1), d.ld_nm, new_msg, d.n_m, d.R, d.n_m, c.st));
    MARL_TRY(launch_copy2d(c.Hs(1), d.ld_nb, h_out, d.n_b, d.R, d.n_b, c.st));
    MARL_TRY(launch_copy2d(c.Cs(1), d.ld_nb, c_out, d.n_b, d.R, d.n_b, c.st));
    MARL_TRY(launch_copy2d(c.HCs(1), d.ld_na, hc_out, d.n_a, d.R, d.n_a, c.st));
    MARL_TRY(launch_copy2d(c.CCs(1), d.ld_na, cc_out, d.n_a, d.R, d.n_a, c.st));
    return MARL_OK;
}

// test hook: where a named activation of step t lives inside episode_ws (float offset, ld)
int marl_debug_buffer(const marl_config* cfg, int train, const char* name, int t,
                      int64_t* offset_floats, int* ld) {
    Dims d;
    MARL_TRY(make_dims(cfg, d));
    ELayout e;
    make_elayout(d, train, e);
    if (!name || !offset_floats || !ld || t < 0 || t > d.ns) return MARL_EINVAL;
    const size_t R = (size_t)d.R;
    const int ts = train ? t : 0;
    auto set = [&](size_t off, int l) {
        *offset_floats = (int64_t)off;
        *ld = l;
        return MARL_OK;
    };
    if (!strncmp(name, "WP", 2) || !strncmp(name, "WT", 2)) {  // weights workspace: packed / transposed copy of parameter <idx>
        const int idx = atoi(name + 2);
        WLayout w;
        make_wlayout(d, w);
        if (idx < 0 || idx >= MARL_NPARAMS || param_meta(d, idx).kind == PK_NONE) {
            set_error("no parameter %d", idx);
            return MARL_EINVAL;
        }
        return name[1] == 'P' ? set(w.wp[idx], w.ldp[idx]) : set(w.wt[idx], w.ldt[idx]);
    }
    if (!strcmp(name, "U")) return set(e.U.at(ts), d.ld_nin);
    if (!strcmp(name, "H")) return set(e.H + (size_t)t * R * d.ld_nb, d.ld_nb);
    if (!strcmp(name, "C")) return set(e.C + (size_t)t * R * d.ld_nb, d.ld_nb);
    if (!strcmp(name, "HC")) return set(e.HC + (size_t)t * R * d.ld_na, d.ld_na);
    if (!strcmp(name, "CC")) return set(e.CC + (size_t)t * R * d.ld_na, d.ld_na);
    if (!strcmp(name, "MSG")) return set(e.MSG + (size_t)t * R * d.ld_nm, d.ld_nm);
    if (!strcmp(name, "PROBS")) return set(e.PROBS + (size_t)t * R * d.nA, d.nA);
    if (!strcmp(name, "COLS0")) {
        if (e.fused_fwd && (!train || e.wgrad_ok[0])) {
            set_error("COLS0 is not materialised for this shape (fused CNN kernels)");
            return MARL_EINVAL;
        }
        return set(e.COLS[0].at(e.COLS[0].stride ? ts : 0), d.ldk[0]);
    }
    if (!strcmp(name, "Z0")) return set(e.Z[0].at(ts), d.ch[1]);
    if (!strcmp(name, "GB")) return set(e.GB.at(ts), d.ld_gb);
    if (train) {
        if (!strcmp(name, "DU")) return set(e.DU + (size_t)t * R * d.ld_nin, d.ld_nin);
        if (!strcmp(name, "DH")) return set(e.DH + (size_t)t * R * d.ld_nb, d.ld_nb);
        if (!strcmp(name, "DHC")) return set(e.DHC + (size_t)t * R * d.ld_na, d.ld_na);
    }
    set_error("unknown debug buffer %s", name);
    return MARL_EINVAL;
}

int marl_plan_query(const marl_config* cfg, int train, const char* key, int* value) {
    Dims d;
    MARL_TRY(make_dims(cfg, d));
    (void)train;
    if (!key || !value) return MARL_EINVAL;
    if (!strcmp(key, "g3")) *value = g3_enabled(d);
    else if (!strcmp(key, "g3_model")) *value = g3_model_ok(d);
    else if (!strcmp(key, "g3_lstm")) *value = g3_enabled(d) && tune_get("g3_lstm", 1) != 0;
    else if (!strcmp(key, "g3_tn")) *value = g3_enabled(d) && g3_tn_enabled(d);
    else if (!strcmp(key, "lstm_plan") || !strcmp(key, "small_r")) {
        // the tile plan the fused two-cell LSTM launch takes (gemm3.hip: 2 = 128-row tiles, 3 / 4 = the 32- / 64-row
        // gate-split plans, 1 / 5 / 6 by knob) - asked of the launcher's own rule with the shapes of the launch (ADVICE
        // r5: this used to re-derive the threshold by hand); "small_r" = one of the gate-split plans
        int plan = 0;
        if (g3_enabled(d) && tune_get("g3_lstm", 1) != 0) {
            G3Batch b{};
            b.count = 2;
            b.p[0].m = b.p[1].m = (int)d.R;
            b.p[0].n = d.n_b;
            b.p[1].n = d.n_a;
            plan = tune_get("g3_lstm_variant", 0);
            if (!plan) plan = g3_lstm_plan(b);
        }
        *value = !strcmp(key, "small_r") ? (plan == 3 || plan == 4) : plan;
    }
    else if (!strcmp(key, "g3_tn_cell"))  // both weight gradients of a cell from one launch (gemm_tn3_cell_kernel)
        *value = g3_enabled(d) && g3_tn_enabled(d) && g3_tn_cell_ok(4 * d.n_b, d.nin, d.n_b, d.NR) &&
                 g3_tn_cell_ok(4 * d.n_a, d.nin, d.n_a, d.NR);
    else if (!strcmp(key, "g3_tn_pipe")) *value = tune_get("g3_tn_pipe", 1) != 0;  // phase-pipelined row contractions
    else if (!strcmp(key, "wgrad3")) *value = tune_get("wgrad3", 1) != 0;  // conv weight gradients on the bf16 pipe (cin >= 16)
    else {
        set_error("unknown plan key %s", key);
        return MARL_EINVAL;
    }
    return MARL_OK;
}

}  // extern "C"
namespace marl {
__global__ void normalize_positions_kernel(const int64_t* __restrict__ pos, float* __restrict__ out,
                                           int rows, int H, int W) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    out[r * 2] = (float)pos[r * 2] / (float)H;
    out[r * 2 + 1] = (float)pos[r * 2 + 1] / (float)W;
}
}  // namespace marl
extern "C" {

int marl_normalize_positions(const int64_t* pos, float* out, int rows, int h, int w, void* stream) {
    if (!pos || !out || rows < 1 || h < 1 || w < 1) {
        set_error("normalize_positions: bad argument");
        return MARL_EINVAL;
    }
    hipLaunchKernelGGL(normalize_positions_kernel, dim3((unsigned)cdiv(rows, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), pos, out, rows, h, w);
    MARL_LAUNCH_CHECK();
    return MARL_OK;
}

int marl_profile_begin(int kernel_class, int max_launches) {
    if (kernel_class < 0 || kernel_class >= kProfClasses || max_launches < 1) return MARL_EINVAL;
    return profile_begin(kernel_class, max_launches);
}
int marl_profile_end(double* total_ms, int* launches) { return profile_end(total_ms, launches); }

// ---- kernel-level entry points -------------------------------------------------------
int marl_gemm_nt(const float* a, int lda, const float* b, int ldb, const float* bias, float* c,
                 int ldc, int m, int n, int k, int accumulate, void* stream) {
    split_registry_reset();  // plain fp32 operands: no image of B is known to this call
    GemmBatch bt{};
    bt.p[0] = gemm_prob(a, lda, b, ldb, k, c, ldc, m, n, bias, accumulate);
    bt.count = 1;
    return launch_gemm_nt(bt, static_cast<hipStream_t>(stream));
}

size_t marl_gemm_weight_image_bytes(int n, int k) { return split_image_floats(n, k) * sizeof(float); }

int marl_gemm_nt_weights(const float* a, int lda, const float* b, int ldb, const float* bias, float* c,
                         int ldc, int m, int n, int k, int accumulate, void* image_scratch, void* stream) {
    if (!b || !image_scratch || n < 1 || k < 1 || ldb < k) {
        set_error("gemm_nt_weights: bad argument");
        return MARL_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    SplitBatch sb{};
    sb.d[0] = SplitDesc{b, image_scratch, n, k, ldb, (k + 31) / 32};
    sb.count = 1;
    MARL_TRY(launch_split_weights(sb, st));
    split_registry_reset();
    split_registry_add(b, n, ldb, k, image_scratch);
    GemmBatch bt{};
    bt.p[0] = gemm_prob(a, lda, b, ldb, k, c, ldc, m, n, bias, accumulate);
    bt.count = 1;
    const int rc = launch_gemm_nt(bt, st);
    split_registry_reset();
    return rc;
}

size_t marl_gemm_tn_scratch(int ni, int nj, int64_t rows) { return gemm_tn_scratch_bytes(ni, nj, rows); }

int marl_gemm_tn(const float* a, int lda, const float* b, int ldb, float* c, int ldc, int ni, int nj,
                 int64_t rows, float* scratch, size_t scratch_bytes, void* stream) {
    return launch_gemm_tn(a, lda, b, ldb, c, ldc, ni, nj, rows, scratch, scratch_bytes,
                          static_cast<hipStream_t>(stream));
}

static CnnWgradArgs wgrad_api_args(int64_t rows, int nb, int c_img, int h, int w, int cin, int cout,
                                   int hin, int groups, int first) {
    CnnWgradArgs g{};
    g.rows = rows;
    g.first = first;
    g.nb = nb;
    g.c_img = c_img;
    g.H = h;
    g.W = w;
    g.cin = cin;
    g.cout = cout;
    g.hin = hin;
    g.hout = (hin - 1) / 2 + 1;
    g.P = g.hout * g.hout;
    g.G = groups > 0 ? groups : 1;
    g.K = 9 * cin;
    return g;
}

size_t marl_cnn_wgrad_scratch(int64_t rows, int cin, int cout, int hin, int groups, int first) {
    const CnnWgradArgs g = wgrad_api_args(rows, 1, cin, hin + 1, hin + 1, cin, cout, hin, groups, first);
    return (size_t)cnn_wgrad_blocks(g) * ((size_t)cout * 9 * cin + cout) * sizeof(float);
}

int marl_cnn_wgrad(const float* dz, const void* img, int img_u8, const int32_t* pos,
                   const float* zin, const float* gst, const float* gamma, const float* beta,
                   int64_t rows, int nb, int c_img, int h, int w, int cin, int cout, int hin,
                   int groups, float* dw, float* db, float* scratch, size_t scratch_bytes,
                   void* stream) {
    const int first = zin == nullptr;
    if (!dz || !dw || !db || !scratch || rows < 1 || cin < 1 || cout < 1 || hin < 1 ||
        (first ? (!img || !pos || nb < 1) : (!gst || !gamma || !beta))) {
        set_error("cnn_wgrad: bad argument");
        return MARL_EINVAL;
    }
    CnnWgradArgs g = wgrad_api_args(rows, nb, c_img, h, w, cin, cout, hin, groups, first);
    if (!cnn_wgrad_supported(g)) {
        set_error("cnn_wgrad: shape outside the kernel's range (cin %d cout %d hin %d)", cin, cout, hin);
        return MARL_ELIMIT;
    }
    if (scratch_bytes < marl_cnn_wgrad_scratch(rows, cin, cout, hin, groups, first)) {
        set_error("cnn_wgrad: scratch too small");
        return MARL_ESIZE;
    }
    g.dz = dz;
    g.img = img;
    g.img_u8 = img_u8;
    g.pos = pos;
    g.zin = zin;
    g.gst = gst;
    g.gamma = gamma;
    g.beta = beta;
    g.part_w = scratch;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MARL_TRY(launch_cnn_wgrad(g, st));
    return launch_slab_reduce(g.part_w, (int64_t)cout * g.K, g.blocks, dw, g.K, cout, g.K, g.part_b, db, st);
}

int marl_tune(const char* key, int value) {
    if (!key) return MARL_EINVAL;
    return tune_set(key, value);
}

int marl_tune_get(const char* key, int dflt) { return key ? tune_get(key, dflt) : dflt; }

int marl_ln_silu_fwd(const float* z, int ldz, const float* gamma, const float* beta, float* out,
                     int ldo, float* stats, int m, int n, void* stream) {
    return launch_ln_silu_fwd(z, ldz, gamma, beta, out, ldo, stats, m, n,
                              static_cast<hipStream_t>(stream));
}

}  // extern "C"
