/*
 * genie_hip.h -- C ABI of libgenie_hip.so: the MI355X (gfx950) denoising path
 * of Genie 2 (marvinli00/genie2), hand-written HIP behind plain pointers.
 *
 * The reference has no FFI of its own: its seam is the Python
 * call `self.model.model(ts, timesteps, features)['z']` inside the reverse
 * loop.  Each entry point below names the reference code it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - return 0 = ok, < 0 = error (GENIE_E_*); text via genie_last_error().
 *     No C++ exception crosses this boundary.
 *   - the caller owns every tensor buffer passed in (device memory on the
 *     handle's device unless marked HOST; contiguous; fp32 / int32 / uint8);
 *     the library owns the handle, packed weights, tables and workspace.
 *   - all device work is enqueued on the caller's stream and is asynchronous;
 *     no entry point except genie_create / genie_load_weights /
 *     genie_set_tables / genie_profile_read synchronises.
 *   - a handle is single-threaded: one per (process, device), matching the
 *     reference's one-process-per-device launcher
 *     (genie/utils/multiprocessor.py:84-96).
 */
#ifndef GENIE_HIP_H
#define GENIE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GENIE_OK            0
#define GENIE_E_ARG        -1   /* bad argument / unsupported shape */
#define GENIE_E_STATE      -2   /* call order (weights/tables/features missing) */
#define GENIE_E_HIP        -3   /* HIP runtime error */
#define GENIE_E_NOMEM      -4

typedef struct genie_ctx* genie_handle_t;
typedef void* genie_stream_t;           /* hipStream_t */

/* Mirrors the keys of Config.model / .diffusion / .io that
 * Denoiser(**config.model, n_timestep, max_n_res, max_n_chain) consumes
 * (genie/config.py:36-80, genie/diffusion/ddpm.py:26-31). */
typedef struct {
    int32_t c_s, c_p;
    int32_t c_pos_emb, c_chain_emb, c_timestep_emb;
    int32_t relpos_k;
    int32_t template_dist_n_bin;
    float   template_dist_min, template_dist_step;
    int32_t n_pair_transform_layer, c_hidden_mul, pair_transition_n;
    int32_t n_structure_layer, n_structure_block;
    int32_t c_hidden_ipa, n_head_ipa, n_qk_point, n_v_point;
    float   rescale;
    int32_t n_timestep, max_n_res, max_n_chain;
} genie_dims_t;

/* The tensors of the reference's 12-key feature dict the denoiser reads
 * (genie/utils/feat_utils.py:304-321; model/ *.py).  bool tensors are passed
 * as their 1-byte storage. */
typedef struct {
    const int32_t* aatype;               /* [B,N,20] */
    const float*   atom_positions;       /* [B,N,3]  motif coordinates */
    const int32_t* residue_mask;         /* [B,N]    */
    const int32_t* residue_index;        /* [B,N]    */
    const int32_t* chain_index;          /* [B,N]    */
    const uint8_t* fixed_sequence_mask;  /* [B,N]    */
    const uint8_t* fixed_structure_mask; /* [B,N,N]  */
    const uint8_t* interface_mask;       /* [B,N]    */
} genie_features_t;

/* Optional stage outputs of one denoiser call (any pointer may be NULL).
 * They are the other entries of the dict Denoiser.forward returns
 * (genie/model/model.py:186-192) plus two intermediate taps for tests. */
typedef struct {
    float* s;          /* [B,N,c_s]      's'  (single feature net output)      */
    float* p;          /* [B,N,N,c_p]    'p'  (pair transform net output)      */
    float* s_final;    /* [B,N,c_s]      'states'[-1]                          */
    float* rots_out;   /* [B,N,3,3]      'ts'.rots                             */
    float* trans_out;  /* [B,N,3]        'ts'.trans                            */
    float* p_init;     /* [B,N,N,c_p]    pair feature net output (test tap)    */
    float* p_layer0;   /* [B,N,N,c_p]    after pair transform layer 0 (tap)    */
    float* states;     /* [1+blocks*layers,B,N,c_s] 'states' (structure_net.py:236-243) */
    float* p_trimul_out0; /* [B,N,N,c_p] p after layer 0's outgoing triangle multiplication (pair_transform_net.py:109-110; tap) */
    float* ipa_cat0;   /* [B,N,H*(c_hidden+4*Pv+c_p)] input of layer 0's IPA linear_out (invariant_point_attention.py:251-258; tap) */
} genie_taps_t;

/* ---- lifetime ---------------------------------------------------------- */

/* Replaces Denoiser.__init__ (genie/model/model.py:20-123). */
int genie_create(const genie_dims_t* dims, int device, genie_handle_t* out);
void genie_destroy(genie_handle_t h);
const char* genie_last_error(genie_handle_t h);   /* h may be NULL: last create error */

/* Number of fp32 values genie_load_weights expects for these dims. */
size_t genie_weight_count(const genie_dims_t* dims);

/* Replaces load_state_dict: `blob` (HOST) is every tensor of
 * Denoiser.state_dict() in its own order (checkpoint
 * keys are these with a 'model.' prefix, genie/utils/model_io.py:159-173),
 * each [out,in] row-major, concatenated.  Repacked into MFMA fragment order
 * and uploaded. */
int genie_load_weights(genie_handle_t h, const float* blob, size_t n_floats);

/* Sinusoidal tables and the diffusion schedule, computed by the host with
 * the reference's own expressions (genie/utils/encoding.py:5-25,
 * genie/diffusion/ddpm.py:40-56) and uploaded once.  All HOST pointers.
 *   pos_tab   [n_pos][c_pos_emb]      row v = encoding of residue_index v
 *   chain_tab [n_chain][c_chain_emb]  row v = encoding of chain_index v
 *   t_tab     [n_timestep+1][c_timestep_emb]
 *   sched     [4][n_timestep+1] = alphas, sqrt_alphas,
 *             sqrt_one_minus_alphas_cumprod, sqrt_betas */
int genie_set_tables(genie_handle_t h, const float* pos_tab, int n_pos,
                     const float* chain_tab, int n_chain,
                     const float* t_tab, const float* sched);

/* ---- per batch --------------------------------------------------------- */

/* Binds a batch of features (device pointers are read now and copied into
 * library-owned buffers) and computes the step-invariant pair terms
 * (_relpos and the motif template: genie/model/pair_feature_net.py:134,
 * 149-158,166-221).  Sizes the workspace for (B, N). */
int genie_prepare_features(genie_handle_t h, genie_stream_t stream, int B, int N,
                           const genie_features_t* feats);

/* compute_frenet_frames (genie/utils/geo_utils.py:21-85) for the bound batch. */
int genie_frenet(genie_handle_t h, genie_stream_t stream, const float* trans /*[B,N,3]*/,
                 float* rots_out /*[B,N,3,3]*/);

/* The same without a handle, as the reference's callers use it: compute_frenet_frames(coords, chains, mask)
 * (genie/utils/geo_utils.py:21; callers sampler/base.py:228,282, diffusion/genie.py:86).  chains / mask: int32 [B,N]. */
int genie_frenet_frames(genie_stream_t stream, int B, int N, const float* coords /*[B,N,3]*/, const int32_t* chains,
                        const int32_t* mask, float* rots_out /*[B,N,3,3]*/);

/* Denoiser.forward (genie/model/model.py:125-192): z_out[B,N,3].
 * timesteps: device int32 [B].  quat_codes: optional device int8 [B,N,N]
 * pinning the sign of each pair quaternion to the reference's eigh output
 * (SURVEY.md hazard 1): 0 = canonical, else 1 + 2*m + neg = "component m has
 * sign (neg ? - : +)". */
int genie_denoise(genie_handle_t h, genie_stream_t stream,
                  const float* trans, const float* rots, const int32_t* timesteps,
                  const int8_t* quat_codes, float* z_out, const genie_taps_t* taps);

/* One ancestral step of BaseSampler._sample (genie/sampler/base.py:249-282):
 * trans <- ((trans - w_z z)/sqrt(alpha_t) * mask [+ scale sqrt(beta_t) eps]) * mask,
 * then Frenet frames.  eps == NULL for step 1. */
int genie_p_sample(genie_handle_t h, genie_stream_t stream, int step, float scale,
                   float* trans_inout, float* rots_out, const float* z, const float* eps);

/* The whole reverse loop (genie/sampler/base.py:227-282), device resident.
 * noise [n_timestep][B,N,3]: noise[0] is the initial trans (base.py:227),
 * noise[k] the draw of iteration k (steps T..2).  first_step/last_step allow a
 * partial trajectory (T..1 for the full one); when first_step < T,
 * trans_io/rots_io carry the state in.  quat_codes: NULL or
 * [n_iterations][B,N,N].  record: NULL or [n_iterations][B,N,3], x after
 * every iteration. */
int genie_sample_loop(genie_handle_t h, genie_stream_t stream, float scale,
                      const float* noise, const int8_t* quat_codes,
                      int first_step, int last_step,
                      float* trans_io, float* rots_io, float* record);

/* ---- arithmetic -------------------------------------------------------- */

/* How the pair-stack GEMMs (triangle multiplication, pair transition: 95 % of the FLOPs of
 * genie/model/model.py:125-192) are carried out.  Both give f32 results to the stated tolerance;
 * the reference computes them with f32 torch.matmul / nn.Linear.
 *   GENIE_MATH_HX  (default) every f32 operand is split in two f16 halves (22 significand bits) and
 *                  a product is three f16 MFMAs accumulated in f32 (csrc/hx.h);
 *   GENIE_MATH_F32 exact f32 MFMA (v_mfma_f32_32x32x2_f32), 1/16 of the matrix rate.
 * May be switched at any time between calls (both weight images are kept).  The environment
 * variable GENIE_MATH=f32|hx sets the initial mode of new handles. */
#define GENIE_MATH_F32 0
#define GENIE_MATH_HX 1
int genie_set_math(genie_handle_t h, int mode);
int genie_get_math(genie_handle_t h);

/* ---- training step (genie/diffusion/genie.py:60-120) ---------------------
 * The two ends around the denoiser first (noising, loss), then the optimizer and the forward + backward pass through the
 * Denoiser itself (genie_train_forward_backward).  All work on the batch bound with genie_prepare_features. */

/* Forward noising + frames (genie.py:80-87):
 *   trans_s = c_x0[b] x0 + c_z[b] z,   rots_s = compute_frenet_frames(trans_s)
 * with c_x0 = sqrt_alphas_cumprod[s], c_z = sqrt_one_minus_alphas_cumprod[s] (device, [B]; ddpm.py:54-56) and z
 * already masked by the caller (genie.py:77). */
int genie_q_sample(genie_handle_t h, genie_stream_t stream, const float* x0 /*[B,N,3]*/, const float* z /*[B,N,3]*/,
                   const float* c_x0 /*[B]*/, const float* c_z /*[B]*/, float* trans_out /*[B,N,3]*/,
                   float* rots_out /*[B,N,3,3]*/);

/* The loss training_step returns (genie.py:90-105, utils/loss.py:4-36) and its gradient:
 *   losses_out (device) [2 + 2B] = unweighted_loss, weighted_loss, condition_losses[B], infill_losses[B];
 *   grad_out [B,N,3] = d weighted_loss / d z_pred, or NULL.
 * Masks are the bound batch's residue_mask and fixed_sequence_mask; num_residues is taken as the mask's
 * row sum (feat_utils.py:17-65 builds them so). */
int genie_training_loss(genie_handle_t h, genie_stream_t stream, const float* z_pred /*[B,N,3]*/, const float* z /*[B,N,3]*/,
                        float condition_loss_weight, float* losses_out, float* grad_out);

/* One Adam update over a flat fp32 parameter blob (torch.optim.Adam as configured by ddpm.py:73-77: betas, eps,
 * no weight decay, no amsgrad), all arrays on the device, `step` = 1 for the first update; the scalars are doubles because
 * torch derives 1 - beta and the bias corrections in double before it rounds them:
 *   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^step) * m / (sqrt(v) / sqrt(1 - b2^step) + eps). */
int genie_adam_step(genie_stream_t stream, size_t n, float* p, const float* g, float* m, float* v, double lr, double beta1,
                    double beta2, double eps, int step);

/* The training step's forward and backward pass through the Denoiser (genie/diffusion/genie.py:88-105; what Lightning's
 * automatic optimisation runs between training_step and the optimizer, train.py:54-65), for the batch bound with
 * genie_prepare_features:
 *   output = Denoiser(T(rots, trans), timesteps, features)            (train mode: dropout live, see opts)
 *   loss   = the weighted loss of genie_training_loss(output['z'], z_target)
 *   grads  = d loss / d weights
 * `weights` and `grads` are DEVICE blobs owned by the caller in Denoiser.state_dict() order (the layout genie_load_weights takes
 * from the host; genie_weight_count floats): the optimizer (genie_adam_step) and the DDP gradient all-reduce (torch.distributed
 * over RCCL, train.py:57-59) act on them directly.  losses_out as genie_training_loss; z_pred_out [B,N,3] optional.
 * Dropout (modules/dropout.py:23-76 row-shared on both triangle multiplications, rate tri_dropout; nn.Dropout on s + ipa(s) and
 * inside StructureTransition, structure_net.py:109, structure_transition.py:66) uses counter-based masks derived from `seed`;
 * train_mode = 0 gives the eval-mode forward (what tests/golden/train_grads_n16_b2.npz was recorded in).
 * fast_math: how the GEMMs' f32 operands reach the bf16 matrix pipe -- 0: split in three bf16 pieces (24 significand bits, six MFMAs
 * per product: f32-grade -- the reference's arithmetic, its Trainer sets no `precision=` (train.py:54-65) and trains in fp32; the
 * mode every parity test and every quoted figure uses), 2: two pieces (16 bits, three MFMAs), 1: plain bf16 operands (one MFMA;
 * what a bf16-autocast run would compute -- NARROWER than the reference, offered for BASELINE config 5's "bf16" wording only). */
typedef struct {
    float tri_dropout, ipa_dropout, transition_dropout;
    uint32_t seed;
    int32_t train_mode, fast_math;
    void* struct_done_event;   /* hipEvent_t or NULL: recorded on `stream` once the gradients of every structure_net.* tensor (the tail
                                  of the blob) are final, so that their all-reduce can overlap the pair stack's backward pass */
} genie_train_opts_t;
int genie_train_forward_backward(genie_handle_t h, genie_stream_t stream, const float* weights, float* grads, const float* trans /*[B,N,3]*/,
                                 const float* rots /*[B,N,3,3]*/, const int32_t* timesteps /*[B]*/, const float* z_target /*[B,N,3]*/,
                                 const int8_t* quat_codes /*[B,N,N] or NULL*/, float condition_loss_weight, const genie_train_opts_t* opts,
                                 float* losses_out /*[2 + 2B]*/, float* z_pred_out);
/* Vector-Jacobian product of the denoiser wrt its input translations, frames held fixed: what the fork's twisted-diffusion / SMC
 * samplers take with torch.autograd.grad(log_prob, ts.trans) after ts = T(rots.detach(), trans.detach())
 * (genie/sampler/unconditional_smc.py:465-482, 570-576):
 *   z_out      [B,N,3] = Denoiser(T(rots, trans), timesteps, features)['z']   (eval mode; may be NULL)
 *   dtrans_out [B,N,3] = sum over z entries of dz * d z / d trans
 * through the direct term, the frame translations entering the structure net and every IPA layer, and the template distance
 * bins of the pair feature net.  `weights`: device blob as for genie_train_forward_backward.  No weight gradients are formed. */
int genie_denoise_vjp(genie_handle_t h, genie_stream_t stream, const float* weights, const float* trans, const float* rots,
                      const int32_t* timesteps, const int8_t* quat_codes, const float* dz, float* z_out, float* dtrans_out);
/* Bytes of activations + scratch the last training call holds; the kept-for-backward part alone; the algorithmic FLOP (2 M N K per
 * product) of the GEMMs the last genie_train_forward_backward launched (bench.py's train_step leg prices them with it). */
size_t genie_train_workspace_bytes(genie_handle_t h);
size_t genie_train_kept_bytes(genie_handle_t h);
double genie_train_gemm_flop(genie_handle_t h);

/* The training path's building block, exposed so that it can be checked on its own (tests/test_train_gemm.py; the reference has no
 * counterpart: there these are torch.nn.Linear / einsum calls inside autograd, the files under genie/model).  Device pointers, element strides:
 *   C[z][m][n] (op)= alpha * sum_k A[z][m][k] B[z][k][n] (+ bias[n]),   z = z1 * nb2 + z2,  z1 < batch / nb2
 *   A at a + z1 a1 + z2 a2 + m am + k ak;  B at b + z1 b1 + z2 b2 + k bk + n bn;  C at c + z1 c1 + z2 c2 + m cm + n cn
 * mode 0 store (then optionally relu, and / or zero where gate <= 0, gate laid out like C), 1 add, 2 atomic add (required for
 * nsplit > 1 or batches sharing C).  terms: bf16 pieces per f32 operand (1 = bf16 products, 2 = 16 bits, 3 = f32-grade).
 * asum (optional, batch 1, am == 1, ak == M): asum[m] += sum_k A[m][k].
 * cblk > 0 (batch 1, at most 8 blocks): C is a row of separately placed blocks of cblk output rows (cblk_m = 1) or columns (0), block t
 * at c + ctab[t] with (m, n) at local index m - t cblk resp. n - t cblk, its asum entries at asum + atab[t] + local m -- how several
 * Linears sharing an input run as one GEMM (genie_train.hip lin_fwd_ln_cat / lin_bwd_w_ln_cat).  Returns 0, or -1 for an impossible shape. */
typedef struct {
    int32_t M, N, K, batch, nb2, nsplit, mode, terms, relu;
    int64_t am, ak, bk, bn, cm, cn, a1, a2, b1, b2, c1, c2;
    float alpha;
    int32_t cblk, cblk_m;
    int64_t ctab[8], atab[8];
} genie_gemm_desc_t;
int genie_train_gemm(genie_stream_t stream, const genie_gemm_desc_t* desc, const float* a, const float* b, float* c, const float* bias,
                     const float* gate, float* asum);

/* ---- measurement ------------------------------------------------------- */

/* Per-kernel-class HIP-event timing on the launch stream (bench.py roofline
 * leg).  enable != 0 starts recording (adds event overhead: not for timed
 * regions).  genie_profile_read synchronises and returns up to `cap` entries;
 * returns the number of classes, names are static strings. */
int genie_profile_enable(genie_handle_t h, int enable);
int genie_profile_read(genie_handle_t h, const char** names, double* total_ms,
                       int64_t* launches, int cap);

/* What dense f16 MFMA work the device sustains, measured live (csrc/probe_kernels.hip): the instruction stream of one transition stage
 * of the fused pair chain on every CU for about ms_target milliseconds (<= 0: 20 ms); synchronises.  The roofline is priced against
 * the datasheet peak (2.5 PFLOP/s dense f16); under matrix load the chip lowers its clock to its power budget and holds about
 * 1.3 - 1.7 PFLOP/s -- bench.py prints this number next to the kernels' own matrix rate.  No handle: nothing of the path calls it.
 * Returns 0 and TFLOP/s in *tflops_out (the launch's milliseconds in *ms_out if not NULL). */
int genie_probe_mfma(genie_stream_t stream, double ms_target, double* tflops_out, double* ms_out);

/* Workspace bytes currently held (HBM layout report for DESIGN.md). */
size_t genie_workspace_bytes(genie_handle_t h);

#ifdef __cplusplus
}
#endif
#endif
