/*
 * marl_hip.h - C ABI of libmarl_hip.so: the MI355X (gfx950) implementation of
 * MARLClassification's hot path (multi-agent episode rollout + A2C update).
 *
 * The reference (Ipsedo/MARLClassification) has no FFI: its hot path is Python
 * over torch tensors.  Each entry point below therefore names the reference
 * function(s) it replaces (paths relative to /root/reference/marl_classification).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host;
 *     fp32 tensors are contiguous, int64 where the reference uses int64;
 *   - rows are r = a * nb + b, i.e. [Na, Nb, ...] tensors flattened over their
 *     first two dimensions (networks/models.py:93);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     call only enqueues work on it, nothing synchronises the device;
 *   - no hidden device allocations: the caller owns every buffer, including the
 *     two workspaces whose sizes marl_workspace_sizes() reports.  Both must be
 *     ZERO-FILLED once by the caller before first use (padding columns inside
 *     them are never written and are read as zeros by the matrix kernels);
 *   - the WEIGHTS workspace (size and the offset of everything inside it) is a function of the model fields of
 *     marl_config and of the tuning knobs only - never of batch, nb_agents, nb_steps or the image size: one
 *     marl_pack_weights serves every batch size until the weights change (an epoch's last partial batch, an
 *     evaluation batch).  The EPISODE workspace depends on all of them: re-query marl_workspace_sizes;
 *   - return value: 0 on success, a negative MARL_E* code otherwise; nothing
 *     throws across the ABI.  marl_last_error() gives a thread-local message;
 *   - one process drives one GPU (the launch contract of bench.py / train.py): the one-off
 *     kernel attributes (LDS opt-in), the profiling hook and the tuning knobs are per-process
 *     state, not per-device or per-thread;
 *   - data parallelism needs no entry point here: the gradients land in ONE flat fp32 buffer
 *     (the caller's, see marl_adam_step) and the caller all-reduces that buffer with RCCL
 *     (torch.distributed backend "nccl" in parallel.py) between marl_episode_backward and
 *     marl_adam_step, passing 1 / world_size as grad_scale.  SURVEY 8(b) sketched a
 *     "marl_allreduce_grads" wrapper taking a communicator; it would only forward to ncclAllReduce, so the
 *     collective stays with the host framework that owns the communicator;
 *   - size limits: GEMM operands are addressed with 32-bit byte offsets from a 64-bit base,
 *     so rows * leading_dimension of any activation matrix must stay below 2^30 floats
 *     (e.g. Ns * Na * Nb < 2^20 rows at 4 * n_b = 1024 gate columns: 4095 images per GPU at
 *     the RESISC45 16-agent / 16-step configuration).
 */
#ifndef MARL_HIP_H
#define MARL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MARL_ABI_VERSION 3

#define MARL_OK 0
#define MARL_EINVAL (-1)   /* bad configuration / null pointer          */
#define MARL_ELIMIT (-2)   /* dimension outside the supported range     */
#define MARL_EHIP (-3)     /* a HIP runtime call failed                 */
#define MARL_ESIZE (-4)    /* workspace too small                       */

#define MARL_COUNTERS_BYTES 32 /* device-side counter block, see marl_counters_set */

#define MARL_MAX_CNN_LAYERS 5
#define MARL_MAX_ACTIONS 16

/* Shape of one ModelsWrapper + Environment + EpisodeSampler triple
 * (networks/models.py:37-50, core/environment.py:14-16, core/episode.py:24-30). */
typedef struct marl_config {
    int32_t nb_agents;   /* Na */
    int32_t batch;       /* Nb */
    int32_t nb_steps;    /* Ns */
    int32_t img_c, img_h, img_w;  /* image batch [Nb, img_c, img_h, img_w] fp32 */
    int32_t window;      /* f */
    int32_t cnn_layers;  /* number of Conv-GroupNorm-SiLU blocks (vision.py:23-52) */
    int32_t cnn_ch[MARL_MAX_CNN_LAYERS + 1]; /* cnn_ch[0] = channels the CNN reads
                            (1 for MnistCnn: channel 0 only, vision.py:63-65), then
                            the output channels of every layer */
    int32_t cnn_groups[MARL_MAX_CNN_LAYERS];
    int32_t n_b, n_a, n_m, n_m_o, n_d; /* belief / action / message / decoded / pos */
    int32_t nb_action, nb_class;
    int32_t nlb, nla;    /* hidden_size_belief / hidden_size_action of the heads */
    int32_t actions[MARL_MAX_ACTIONS][2]; /* movement per action index (dim 0 = H) */
    int32_t img_u8;      /* 1: the image batch is uint8 [Nb,C,H,W]; the gather kernel applies
                            ToTensor (x / 255, registry.py:56-57) on the fly - 4x less HBM and
                            PCIe traffic than uploading fp32 */
} marl_config;

/* Parameter table: an array of MARL_NPARAMS device pointers in this order.
 * Shapes are those of the reference's state_dict() (tight, row-major):
 *   4*l+0 conv_l.weight [co,ci,3,3]  4*l+1 conv_l.bias [co]
 *   4*l+2 gn_l.weight [co]           4*l+3 gn_l.bias [co]          (l < 5, unused = NULL)
 * then the indices below. */
enum {
    MARL_P_POS_W = 20, MARL_P_POS_B, MARL_P_POS_LNW, MARL_P_POS_LNB,          /* map_pos   */
    MARL_P_ENC_W0, MARL_P_ENC_B0, MARL_P_ENC_LN0W, MARL_P_ENC_LN0B,
    MARL_P_ENC_W1, MARL_P_ENC_B1, MARL_P_ENC_LN1W, MARL_P_ENC_LN1B,          /* encode_msg */
    MARL_P_DEC_W0, MARL_P_DEC_B0, MARL_P_DEC_LN0W, MARL_P_DEC_LN0B,
    MARL_P_DEC_W1, MARL_P_DEC_B1, MARL_P_DEC_LN1W, MARL_P_DEC_LN1B,          /* decode_msg */
    MARL_P_LB_WIH, MARL_P_LB_WHH, MARL_P_LB_BIH, MARL_P_LB_BHH,              /* belief LSTM */
    MARL_P_LA_WIH, MARL_P_LA_WHH, MARL_P_LA_BIH, MARL_P_LA_BHH,              /* action LSTM */
    MARL_P_POL_W0, MARL_P_POL_B0, MARL_P_POL_LNW, MARL_P_POL_LNB, MARL_P_POL_W1, MARL_P_POL_B1,
    MARL_P_CRI_W0, MARL_P_CRI_B0, MARL_P_CRI_LNW, MARL_P_CRI_LNB, MARL_P_CRI_W1, MARL_P_CRI_B1,
    MARL_P_PRE_W0, MARL_P_PRE_B0, MARL_P_PRE_LNW, MARL_P_PRE_LNB, MARL_P_PRE_W1, MARL_P_PRE_B1,
    MARL_NPARAMS
};

int marl_abi_version(void);
const char* marl_last_error(void);

/* Number of fp32 elements of parameter `index` for `cfg` (0 for unused slots). */
int64_t marl_param_numel(const marl_config* cfg, int index);

/* Byte sizes of the two caller-owned workspaces:
 *   weights_ws : padded / transposed copies of the parameters + packed gradients
 *   episode_ws : every per-step activation of one episode (saved for backward)
 * `train` = 0 sizes episode_ws for rollout only (nothing kept for backward).
 * ABI 3: every entry point that takes a workspace also takes its size in bytes and returns
 * MARL_ESIZE (nothing enqueued) when the layout for the CURRENT configuration and tuning knobs does
 * not fit - e.g. a buffer allocated before a layout-affecting marl_tune() call.  (marl_a2c_loss_fwd_bwd
 * wants the TRAINING size: its scratch is the tail of that layout.) */
int marl_workspace_sizes(const marl_config* cfg, int train,
                         size_t* weights_ws_bytes, size_t* episode_ws_bytes);

/* Re-packs the parameters into weights_ws; call after every optimiser step and
 * after load_state_dict (replaces nothing in the reference - layout plumbing). */
int marl_pack_weights(const marl_config* cfg, const float* const* params_host,
                      void* weights_ws, size_t weights_ws_bytes, void* stream);

/* Environment.__observation (core/environment.py:95-126): coalesced patch gather.
 * img [Nb,C,H,W] fp32, pos int64 [Na*Nb,2] -> obs [Na*Nb, C, f, f] fp32. */
int marl_patch_gather(const float* img, const int64_t* pos, float* obs,
                      int nb_agents, int batch, int c, int h, int w, int f, void* stream);

/* Environment.step / __transition (core/environment.py:56-66,128-150):
 * pos' = pos + table[a] if the move stays inside [0, size - f) in every dim, else pos.
 * table_host: nb_action x 2 int32 on the HOST. pos_in/pos_out int64 [rows,2] (may alias). */
int marl_transition(const int64_t* pos_in, const int64_t* actions, int64_t* pos_out,
                    const int32_t* table_host, int nb_action, int rows,
                    int h, int w, int f, void* stream);

/* Perf-mode draws of one episode inside the library (counter-based Philox4x32-10; same
 * distributions as the reference's draws - core/environment.py:33-43: pos0[r][d] uniform in
 * [0, size_d - f); networks/models.py:148-159: h0,c0 [R,n_b], hc0,cc0 [R,n_a] ~ N(0,1) - but not
 * the reference's mt19937 stream: parity runs inject host-drawn tensors instead).  `noise`
 * (nullable, [Ns,R,nA]) additionally receives Exp(1) draws.  (seed, offset) selects the stream:
 * the same pair always produces the same tensors. */
int marl_draw_episode(const marl_config* cfg, uint64_t seed, uint64_t offset,
                      const void* counters, int64_t* pos0, float* h0, float* c0, float* hc0,
                      float* cc0, float* noise, void* stream);

/* Device-resident iteration counters (MARL_COUNTERS_BYTES, caller-owned device memory): the
 * generator offset of the episode draws and the Adam step with its bias corrections.  Calls that
 * take `counters` (nullable) ADD its generator offset to their rng_offset argument / take the
 * Adam step scalars from it, so one iteration (draw -> rollout -> loss -> backward -> Adam ->
 * re-pack -> marl_counters_tick) can be captured once (marl_graph_*) and replayed with no
 * host-side change between replays.  set: offset / step (1-based, the step the next Adam applies)
 * as given; tick: both + 1. */
int marl_counters_set(void* counters, uint64_t rng_offset, int64_t step, float lr, float beta1,
                      float beta2, void* stream);
int marl_counters_tick(void* counters, float lr, float beta1, float beta2, void* stream);

/* hipGraph capture of whatever the library enqueues on `stream` between begin and end (the
 * stream must not be the NULL stream; nothing may synchronise or allocate in between).
 * marl_graph_end instantiates the captured graph; marl_graph_launch replays it.  The captured
 * calls keep their pointer arguments: replays read / write the same buffers. */
int marl_graph_begin(void* stream);
int marl_graph_end(void* stream, void** graph_exec_out);
int marl_graph_launch(void* graph_exec, void* stream);
int marl_graph_destroy(void* graph_exec);

/* EpisodeSampler.__episode_impl (core/episode.py:32-82) with the reference's random
 * draws as INPUTS: img fp32 (or uint8 when cfg->img_u8) [Nb,C,H,W]; pos0 int64 [R,2]; h0,c0 [R,n_b]; hc0,cc0 [R,n_a];
 * noise [Ns,R,nA] ~ Exp(1) (th.multinomial == argmax(p / noise)).
 * noise == NULL (and no forced_actions): the Exp(1) variates are drawn inside the sampling
 * kernel from Philox4x32-10 keyed by (rng_seed, rng_offset, step, row) - perf mode, nothing to
 * generate, store or read back per step.
 * forced_actions (int64 [Ns,R]) may be NULL; if given it replaces sampling.
 * Outputs: step_preds [Ns,R,nC], step_logp [Ns,R], step_values [Ns,R],
 * step_pos int64 [Ns,R,2] (after move t), step_actions int64 [Ns,R] (may be NULL).
 * With train != 0 the activations needed by marl_episode_backward stay in episode_ws. */
int marl_episode_forward(const marl_config* cfg, const void* weights_ws, size_t weights_ws_bytes,
                         void* episode_ws, size_t episode_ws_bytes,
                         const void* img, const int64_t* pos0,
                         const float* h0, const float* c0, const float* hc0, const float* cc0,
                         const float* noise, const int64_t* forced_actions,
                         uint64_t rng_seed, uint64_t rng_offset, const void* counters,
                         float* step_preds, float* step_logp, float* step_values,
                         int64_t* step_pos, int64_t* step_actions,
                         int train, void* stream);

/* loss.backward() through the episode (training/trainer.py:115): given dL/d(step_preds)
 * [Ns,R,nC], dL/d(step_logp) [Ns,R], dL/d(step_values) [Ns,R] (any may be NULL = zero),
 * writes dL/d(param) for every parameter into grads_host[i] (tight reference shapes,
 * overwritten, not accumulated).  `img` is the image batch of the matching
 * marl_episode_forward(train = 1) call: the first convolution's weight gradient re-gathers
 * the patches at the saved positions instead of keeping im2col rows in episode_ws (autograd
 * would keep the observation tensors alive the same way). */
int marl_episode_backward(const marl_config* cfg, void* weights_ws, size_t weights_ws_bytes,
                          void* episode_ws, size_t episode_ws_bytes,
                          const void* img, const float* g_preds, const float* g_logp,
                          const float* g_values, float* const* grads_host, void* stream);

/* Data-parallel overlap (no reference counterpart: training/trainer.py is single-device).  While an event
 * is installed (per-process state, NULL clears it), every marl_episode_backward records it on its stream at the
 * point where the gradients of the three heads' parameters (MARL_P_POL_* .. MARL_P_PRE_*: Policy / Critic /
 * Prediction, networks/policy.py, prediction.py) are final - before the reverse-time loop and the batched weight
 * gradients behind it.  The caller's side stream waits for it and all-reduces that slice of the flat gradient buffer
 * while the rest of the backward pass runs (parallel.py, BucketedGradAllReduce). */
int marl_backward_heads_event(void* hip_event);

/* Loss of Trainer.train_epoch (training/trainer.py:76-111; training/functions.py:7-55)
 * and its gradient w.r.t. the episode outputs in one pass.
 * y int64 [Nb].  scalars_out[4] = {loss, path, error, critic} (trainer.py:111,119-122).
 * stats_inout: optional (n, sum, sumsq) hook for exact multi-GPU standardize:
 *   phase 0 = everything local (single GPU); phase 1 = only write local
 *   (n, sum x, sum x^2) of the advantages to adv_stats[3] (caller all-reduces them);
 *   phase 2 = finish using the all-reduced adv_stats. */
int marl_a2c_loss_fwd_bwd(const marl_config* cfg, void* episode_ws, size_t episode_ws_bytes,
                          const float* step_preds, const float* step_logp,
                          const float* step_values, const int64_t* y, float gamma,
                          float* g_preds, float* g_logp, float* g_values,
                          float* scalars_out, double* adv_stats, int phase, void* stream);

/* th.optim.Adam.step (training/trainer.py:33,116) on one flat buffer:
 * betas (0.9, 0.999), eps 1e-8, no weight decay; `step` is 1-based. grad_scale
 * multiplies the gradient first (1/world_size after an all-reduce sum).  counters != NULL:
 * `step` and the bias corrections come from the device-side counter block (graph replay). */
int marl_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                   int64_t n, int64_t step, float lr, float beta1, float beta2, float eps,
                   float grad_scale, const void* counters, void* stream);

/* ModelsWrapper.forward + MultiAgent.act for ONE step used standalone
 * (networks/models.py:78-138, core/agent.py:40-68): obs [R,C,f,f] is gathered by the
 * caller (marl_patch_gather); msg [R,n_m]; norm_pos [R,2]; state in h,c,hc,cc.
 * Outputs: probs [R,nA], values [R], preds [R,nC], new_msg [R,n_m], new state. */
int marl_step_forward(const marl_config* cfg, const void* weights_ws, size_t weights_ws_bytes,
                      void* episode_ws, size_t episode_ws_bytes,
                      const float* obs, const float* msg, const float* norm_pos,
                      const float* h, const float* c, const float* hc, const float* cc,
                      float* probs, float* values, float* preds, float* new_msg,
                      float* h_out, float* c_out, float* hc_out, float* cc_out,
                      const float* noise, uint64_t rng_seed, uint64_t rng_offset,
                      int64_t* actions_out, float* logp_out, void* stream);
/* (actions_out int64 [R], logp_out [R]: optional - when both are given the call also samples the
 * action and its log-probability, core/agent.py:53-61, from noise [R,nA] ~ Exp(1) if given, else
 * from the library's generator at (rng_seed, rng_offset).) */

/* Environment.normalized_positions (core/environment.py:74-81): out[r,d] = pos[r,d] / size_d. */
int marl_normalize_positions(const int64_t* pos, float* out, int rows, int h, int w, void* stream);

/* ---- kernel-level entry points (used by tests/ and profiling; same kernels) ---- */
/* C[M,N] (+)= A[M,K] * B[N,K]^T + bias ; lda/ldb multiples of 4, 16-byte aligned. */
int marl_gemm_nt(const float* a, int lda, const float* b, int ldb, const float* bias,
                 float* c, int ldc, int m, int n, int k, int accumulate, void* stream);
/* Same product with B treated as a WEIGHT matrix, the way the episode calls it: its bf16x3 image
 * (what marl_pack_weights builds inside the weights workspace) is built in image_scratch
 * (>= marl_gemm_weight_image_bytes(n, k), 256-byte aligned, must stay valid until the product ran)
 * and the bf16x6 kernel copies its tiles instead of splitting them.  With the knob mfma_split = 0
 * this is marl_gemm_nt. */
int marl_gemm_nt_weights(const float* a, int lda, const float* b, int ldb, const float* bias,
                         float* c, int ldc, int m, int n, int k, int accumulate,
                         void* image_scratch, void* stream);
size_t marl_gemm_weight_image_bytes(int n, int k);
/* C[NI,NJ] = sum_r A[r,i] * B[r,j] over `rows` rows; scratch >= marl_gemm_tn_scratch(). */
int marl_gemm_tn(const float* a, int lda, const float* b, int ldb, float* c, int ldc,
                 int ni, int nj, int64_t rows, float* scratch, size_t scratch_bytes, void* stream);
size_t marl_gemm_tn_scratch(int ni, int nj, int64_t rows);
/* ---- image GEMMs (csrc/gemm3.hip): the form the episode uses for its large products ----
 * A "k16 image" of an fp32 matrix [rows][k] is [rows][ceil(k/16)][3][16] bf16: every element split
 * into three bf16 terms x0 + x1 + x2 == x, K zero-padded to a whole 16-deep step (96 bytes per row
 * and step).  marl_image_build makes one from an fp32 matrix (ld >= k); inside the episode the
 * PRODUCER kernels write the images of their outputs directly.  16-byte aligned. */
size_t marl_image_bytes(int64_t rows, int k);
int marl_image_build(const float* src, int ld, int64_t rows, int k, void* image, void* stream);
/* C[M,N] (+)= A * B^T + bias from the images of A [M,K] and B [N,K] (six bf16 MFMA products per fp32
 * product, fp32 accumulation - same arithmetic as marl_gemm_nt with mfma_split = 1).
 * variant: 0 = automatic tile plan (1 / 2 / 3 force 256x128, 128x128, 128x64 tiles - tools only). */
int marl_gemm_nt_images(const void* a3, const void* b3, const float* bias, float* c, int ldc, int m, int n,
                        int k, int accumulate, int variant, void* stream);
int marl_gemm_nt_images_batch(int count, const void* const* a3, const void* const* b3, float* const* c,
                              const int* n, const int* ldc, int m, int k, int accumulate, int variant,
                              void* stream);
/* nn.LSTMCell (networks/recurrent.py:19-35) from images: gates = U * W_ih^T + H * W_hh^T + bias (rows of
 * the weight images g * n + unit, g = i,f,g,o; bias = b_ih + b_hh), c' = s(f) c + s(i) tanh(g),
 * h' = s(o) tanh(c'); gates (nullable) receives the activated gates [M, ld_gates]; h3_next (nullable)
 * the image of h'. */
int marl_lstm_images(const void* u3, int nin, const void* h3, const void* wih3, const void* whh3,
                     const float* bias, const float* c_prev, float* h_next, float* c_next, float* gates,
                     void* h3_next, int m, int n, int ld_state, int ld_gates, int variant, int cells,
                     void* stream);
/* C[NI,NJ] = sum_r A[r,i] * B[r,j] from the images of A [rows,NI] and B [rows,NJ] (rows % 32 == 0);
 * colsum (nullable) [NI] = column sums of A (the matching bias gradient). */
size_t marl_gemm_tn_images_scratch(int ni, int nj, int64_t rows);
int marl_gemm_tn_images(const void* a3, const void* b3, float* c, int ldc, int ni, int nj, int64_t rows,
                        float* colsum, float* scratch, size_t scratch_bytes, void* stream);
/* Both weight gradients of one LSTM cell (backward of networks/recurrent.py:19-35 through loss.backward(),
 * training/trainer.py:115): C_ih [NI, NIH] = G^T U and C_hh [NI, NHH] = G^T H from the images of the gate gradients
 * G [rows, NI], of U [rows, NIH] and of H [rows, NHH]; colsum (nullable) [NI] = column sums of G (the bias
 * gradient).  One launch whose workgroups share G's row slabs through an XCD's L2: G leaves HBM once.
 * marl_gemm_tn_images_cell_scratch returns 0 for shapes outside this plan (NI < 256, rows < 8192, more than four
 * column tiles): use marl_gemm_tn_images for those. */
size_t marl_gemm_tn_images_cell_scratch(int ni, int nih, int nhh, int64_t rows);
int marl_gemm_tn_images_cell(const void* g3, int ni, const void* u3, int nih, const void* h3, int nhh, int64_t rows,
                             float* c_ih, int ld_ih, float* c_hh, int ld_hh, float* colsum, float* scratch,
                             size_t scratch_bytes, void* stream);
int marl_ln_silu_fwd(const float* z, int ldz, const float* gamma, const float* beta,
                     float* out, int ldo, float* stats, int m, int n, void* stream);

/* Convolution weight gradient from the activations (the kernel behind the conv layers'
 * dW in marl_episode_backward; backward of networks/vision.py:33-35 for Conv2d(k3,s2,p1)):
 *   dw[co][tap*cin+ci] = sum over rows r, output positions of dz[r][pos][co] * in[r][window(pos,tap)][ci]
 * in = the raw patch img[r % nb][:cin][pos[r] + ...] when zin == NULL (first layer), else
 * SiLU(GroupNorm(zin[r])) with the saved statistics gst[r][G][2] (mean, rstd) and affine.
 * dz [rows][hout*hout][cout]; zin [rows][hin*hin][cin]; dw [cout][9*cin]; db [cout] = column
 * sums of dz.  scratch >= marl_cnn_wgrad_scratch() bytes. */
int marl_cnn_wgrad(const float* dz, const void* img, int img_u8, const int32_t* pos,
                   const float* zin, const float* gst, const float* gamma, const float* beta,
                   int64_t rows, int nb, int c_img, int h, int w, int cin, int cout, int hin,
                   int groups, float* dw, float* db, float* scratch, size_t scratch_bytes,
                   void* stream);
size_t marl_cnn_wgrad_scratch(int64_t rows, int cin, int cout, int hin, int groups, int first);

/* Perf-experiment hook: overrides an internal tuning knob (same names as the MARL_<KEY>
 * environment variables, lower case, e.g. "wgrad_rb", "mfma_split"); tools/ and tests only.
 * Several knobs change the LAYOUT of the episode workspace (tile plans, split-K targets,
 * mfma_split ...): marl_workspace_sizes() must be asked again and the workspace re-allocated
 * after such a call; a stale buffer is refused with MARL_ESIZE (ABI 3: sizes are passed in).  (The
 * Python wrapper marlclassification_amd.engine.tune() drops every cached workspace.) */
int marl_tune(const char* key, int value);
/* current value of a knob (marl_tune value, else MARL_<KEY>, else dflt) */
int marl_tune_get(const char* key, int dflt);

/* Measurement hook (bench.py roofline): time every launch of one kernel class with HIP
 * events recorded on the launch stream.  class 0 = fused LSTM-cell GEMM, 1 = plain NT GEMM,
 * 2 = row-contraction (weight-gradient) GEMM, 3 = fused CNN forward, 4 = row-panel MLP kernels,
 * 5 = CNN backward (layer backward + weight gradients).  marl_profile_end synchronises those events
 * and returns the summed kernel time and the number of launches seen. */
int marl_profile_begin(int kernel_class, int max_launches);
int marl_profile_end(double* total_ms, int* launches);

/* test hook: float offset and leading dimension of a named per-step activation inside
 * episode_ws ("U","H","C","HC","CC","MSG","PROBS","COLS0","Z0","GB","DU","DH","DHC"), or - "WP<i>" /
 * "WT<i>" - of the packed / transposed fp32 copy of parameter slot i inside weights_ws. */
int marl_debug_buffer(const marl_config* cfg, int train, const char* name, int t,
                      int64_t* offset_floats, int* ld);

/* Which kernel family the library picks for `cfg` under the current knobs (bench.py's `matrix_products`
 * note, tests): key "g3" = the large matrix products of this batch run on operand images (gemm3.hip),
 * "g3_model" = the weights workspace holds the k16 weight images (a property of the MODEL and the knobs only:
 * the weights layout never moves with the batch), "g3_tn" = the four large weight gradients run on images,
 * "g3_lstm" = the fused LSTM launch does, "lstm_plan" = the tile plan that launch takes (the launcher's own rule: 2 =
 * 128-row tiles, 3 / 4 = the gate-split small-batch plans, 1 / 5 / 6 by knob), "small_r" = lstm_plan is 3 or 4,
 * "g3_tn_cell" = both weight gradients of an LSTM cell come from one launch, "g3_tn_pipe" = the row contractions run
 * the phase-pipelined step, "wgrad3" = conv weight gradients (cin >= 16) on the bf16 pipe.  *value = 0 / 1 (a plan
 * number for "lstm_plan"). */
int marl_plan_query(const marl_config* cfg, int train, const char* key, int* value);

#ifdef __cplusplus
}
#endif
#endif /* MARL_HIP_H */
