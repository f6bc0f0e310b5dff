/*
 * gank.h -- C ABI of libgank.so: the MI355X (gfx950 / CDNA4) kernels behind the SNGAN-ResNet
 * CIFAR-10 training hot path of watsonyanghx/GAN_Lib_Tensorflow.
 *
 * The reference has no FFI: every FLOP runs inside TensorFlow-1.5 ops called from Python
 * (SURVEY.md section 8b).  Each entry point below therefore cites the reference Python call site
 * whose TensorFlow op(s) it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch); the caller allocates outputs
 *     and workspaces; nothing here allocates, frees or synchronises (hipGraph-capturable);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it asynchronously;
 *   - activations (and every buffer documented as "bf16") are 16-bit floats, NHWC: bfloat16 in libgank.so, IEEE half in
 *     libgank_f16.so -- the same sources and the same entry points built twice (gank_act_dtype()), MFMA bf16 / f16 tiles with
 *     fp32 accumulation; master weights, gradients of weights, statistics, losses and optimiser state are fp32; conv filters
 *     are HWIO ([k,k,Cin,Cout]); linear weights [in,out]; labels are int32;
 *   - return value 0 = launched, non-zero = argument/launch error, message in gank_last_error();
 *   - thread-compatible: no global mutable state besides the per-thread error string and the
 *     opt-in profiling event pool (gank_prof_*).
 */
#ifndef GANK_H
#define GANK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GANK_VERSION 100

/* conv flags */
#define GANK_IN_UPSAMPLE2X 1   /* input is [N,H/2,W/2,C]; nearest-neighbour 2x on the fly            */
#define GANK_IN_RELU 2         /* relu applied to the input operand while staging                     */
#define GANK_OUT_TANH 4        /* tanh applied last in the epilogue                                   */
#define GANK_DY_UPSAMPLE2X 8   /* wgrad only: dy is [N,H/2,W/2,Cout] (gradient of a 2x2 mean pool)    */
#define GANK_STAT_SLOTS 16     /* copies of each tower's statistics sums the conv epilogues spread their atomics over */
#define GANK_STATS_PREZEROED 256 /* *_fprop_stats: stat_sums was cleared by the caller (one fill for all layers of a pass) */
#define GANK_RES_UPSAMPLE2X 64 /* fprop: residual is [N,H/2,W/2,Cout] and is added nearest-neighbour upsampled: the
                                  shortcut of an 'up' residual block (gan_cifar_resnet.py:179-182,209) without
                                  materialising the upsampled tensor */
#define GANK_OUT_POOLSUM2X 512 /* gank_res8_conv3x3: the result leaves as its 2x2 sums [N,H/2,W/2,Cout] (the gradient of an
                                  NN-upsample in front of the conv whose input gradient this launch computes) */

int gank_version(void);
/* element type of every `bf16` buffer of THIS library: 0 = bfloat16 (libgank.so), 1 = IEEE half (libgank_f16.so: the same sources
 * built with -DGANK_ACT_F16 -- identical entry points, v_mfma_f32_32x32x16_f16 instead of ..._bf16, fp32 accumulation either way) */
int gank_act_dtype(void);
const char* gank_last_error(void);

/* ---- weight preparation: fp32 HWIO master/normalised filter -> bf16 MFMA operand layouts --------
 * wf [CoutPad][Kpad]  : wf[co][tap*Cin+ci]        = w[tap][ci][co]                (fprop operand)
 * wd [CinPad ][Kpad'] : wd[ci][tap'*Cout+co]      = w[taps-1-tap'][ci][co]        (dgrad operand)
 * CoutPad = roundup(Cout,32), Kpad = roundup(taps*Cin,64); CinPad/Kpad' likewise.  Either output may
 * be NULL.  Replaces the implicit filter transforms inside tf.nn.conv2d / its gradient ops
 * (common/ops/conv2d.py:180-187). */
int gank_conv2d_prep_weights(const float* w, void* wf, void* wd, int ksize, int Cin, int Cout, void* stream);
/* the same for up to any number of weights in one launch per 16 (host table, copied into kernel arguments) */
typedef struct gank_prep_desc {
  const float* w; /* fp32 [k,k,Cin,Cout] */
  void* wf;       /* kind 0: bf16 [CoutPad][Kpad] or NULL;  kind 1: wph;  kind 2: wp4   (the fprop operand) */
  void* wd;       /* kind 0: bf16 [CinPad][Kpad'] or NULL;  kind 1: wd4;  kind 2: wphd  (the dgrad operand) */
  int ksize, Cin, Cout;
  int kind;       /* 0 plain conv/linear; 1 UpsampleConv 3x3 (gank_upconv3x3_prep_weights layouts);
                     2 ConvMeanPool 3x3 (gank_convpool3x3_prep_weights layouts);
                     (3 is retired: the register-weight patch kernel it fed measured equal to the LDS-weight one and was removed);
                     4 "rfrag" operands of the resident kernels (gank_res8_chain_*): bf16 [rows/32][taps][k/16][64 lanes][8],
                       lane = 32*h + r holding k = 16*kk + 8*h .. +7 of row 32*tile + r; wf: rows = co, k = ci;
                       wd: rows = ci, k = co, taps flipped.  Cin % 32 == 0 and Cout % 32 == 0; taps*Cin*Cout elements each;
                     5 ConvMeanPool 3x3 operands of the resident kernels (gank_cpool_res_*), 16*Cin*Cout elements each, same
                       tap algebra as kind 2: wf bf16 [Cout/32][Cin/64][16 taps][4][64 lanes][8] (the 4x4 stride-2 kernel),
                       wd bf16 [4 phases][Cin/32][4 taps][Cout/16][64][8] (its transposed conv).  Cin % 64 == 0, Cout % 32 == 0 */
  int cin_pitch;  /* kind 4 only: w is [k,k,cin_pitch,Cout] and the operands are built from its FIRST Cin input channels
                     (0 = Cin: the whole filter) -- the feature half of a conv whose other input channels are factored out
                     (gank_label_conv3x3_table) */
} gank_prep_desc;
int gank_conv2d_prep_weights_batched(const gank_prep_desc* table, int count, void* stream);

/* ---- conv2d forward: y = epilogue(conv_SAME_stride1(in(x), w) * scale + bias) --------------------
 * Replaces tf.nn.conv2d + tf.nn.bias_add (common/ops/conv2d.py:180-187,212-216), with the
 * surrounding graph ops of the block library fused behind flags: NN-upsample of UpsampleConv
 * (SNGAN/gan_cifar_resnet.py:143-145), pre-activation relu (:186,:198), shortcut add (:209),
 * tanh (:261).  Epilogue order: v=acc*scale+bias; if relu_ref: v=(relu_ref>0)?v:0; if residual:
 * v+=residual; if OUT_TANH: v=tanh(v).  x [N,Hin,Win,Cin] bf16, wf from gank_conv2d_prep_weights,
 * y/residual/relu_ref [N,H,W,Cout] bf16 where (H,W) is the OUTPUT size. Implicit-GEMM on
 * v_mfma_f32_32x32x16_bf16, fp32 accumulate. */
int gank_conv2d_fprop(const void* x, const void* wf, const float* bias, const void* residual,
                      const void* relu_ref, void* y, int N, int H, int W, int Cin, int Cout, int ksize,
                      int flags, float scale, void* stream);

/* The same with the conditional-batch-norm statistics of y accumulated by the epilogue (normalization.py:47: the
 * tf.nn.moments of the layer that consumes y): stat_sums [groups][GANK_STAT_SLOTS][2][Cout] fp32 <- per tower (N/groups
 * consecutive samples) the sum and the sum of squares of (y - bias), spread over GANK_STAT_SLOTS partial copies (add them up).  Only the two-group LDS-DMA kernel does it: *produced = 1 if the
 * sums were written (feed them to gank_cbn_fwd_from_sums), 0 if another kernel ran (use gank_cbn_fwd).  The buffer is
 * zeroed here.  gank_upconv3x3_fprop_stats: likewise for the phase-decomposed UpsampleConv. */
int gank_conv2d_fprop_stats(const void* x, const void* wf, const float* bias, const void* residual,
                            const void* relu_ref, void* y, int N, int H, int W, int Cin, int Cout, int ksize,
                            int flags, float scale, float* stat_sums, int groups, int* produced, void* stream);

/* ---- conv2d input gradient: dx = epilogue(conv_SAME(in(dy), flip(w)^T) * scale) ------------------
 * Replaces the Conv2DBackpropInput op TensorFlow autodiff emits for conv2d.py:180-187.  Same engine
 * as fprop with the roles of Cin/Cout swapped and the wd layout; flags/epilogue as above
 * (relu_ref implements the relu backward mask of the preceding nonlinearity; residual the gradient
 * fan-in of the shortcut).  dy [N,Hin,Win,Cout], dx [N,H,W,Cin]. */
int gank_conv2d_dgrad(const void* dy, const void* wd, const void* residual, const void* relu_ref, void* dx,
                      int N, int H, int W, int Cin, int Cout, int ksize, int flags, float scale, void* stream);

/* Both gradients of a 1x1 conv (stride 1, no flags) in ONE launch: the filter / bias gradient exactly as gank_conv2d_wgrad (ACCUMULATED
 * into dw [Cin][Cout] and dbias) and the input gradient dx [N,H,W,Cin] = dy wd^T (gank_conv2d_dgrad with ksize 1) computed by extra
 * workgroups of the filter-gradient launch (32 pixels x all input channels each, operands straight from memory).  wd: the plain-conv
 * dgrad operand, bf16 [Cin][wd_pitch] (rows = input channels).  Cin % 64 == 0, Cout % 16 == 0, Cout <= 128, N*H*W % 32 == 0.  A layer
 * whose filter gradient runs on a kernel without the rider gets its input gradient from gank_conv2d_dgrad inside this call. */
int gank_conv1x1_wgrad_dgrad(const void* x, const void* dy, float* dw, float* dbias, const void* wd, int wd_pitch, void* dx, int N, int H, int W,
                             int Cin, int Cout, void* stream);

/* ---- conv2d filter gradient: dw[tap][ci][co] += scale * sum_pixels in(x)[p+tap][ci] * dy[p][co] --
 * Replaces Conv2DBackpropFilter for conv2d.py:180-187.  ACCUMULATES (fp32 atomics, split over pixel
 * ranges) into dw [k,k,Cin,Cout] -- zero it first for a plain gradient.  (H,W) is the conv OUTPUT
 * size; x is [N,H,W,Cin] (or half-size with IN_UPSAMPLE2X), dy [N,H,W,Cout] (or half-size with
 * DY_UPSAMPLE2X).  dbias (optional, [Cout]) += scale * column sums of dy: the tf.nn.bias_add gradient
 * (conv2d.py:216) fused into the pass that already streams dy.
 * ws (optional): fp32 scratch of gank_conv2d_wgrad_ws_elems(...) elements; when given, the split partial tiles
 * are written as slabs and summed deterministically instead of with atomics. */
long gank_conv2d_wgrad_ws_elems(int N, int H, int W, int Cin, int Cout, int ksize, int flags);
int gank_conv2d_wgrad(const void* x, const void* dy, float* dw, float* dbias, float* ws, long ws_elems, int N, int H,
                      int W, int Cin, int Cout, int ksize, int flags, float scale, void* stream);

/* `count` filter gradients of identical geometry (N,H,W,Cin,Cout,ksize; flags: GANK_IN_RELU only) in as few launches
 * as possible (up to 4 layers per launch); same arithmetic as `count` calls of gank_conv2d_wgrad without workspace. */
typedef struct gank_wgrad_item {
  const void* x;   /* bf16 [N,H,W,Cin]  */
  const void* dy;  /* bf16 [N,H,W,Cout] */
  float* dw;       /* fp32 [k,k,Cin,Cout], accumulated */
  float* dbias;    /* fp32 [Cout] or NULL, accumulated  */
} gank_wgrad_item;
int gank_conv2d_wgrad_batched(const gank_wgrad_item* items, int count, int N, int H, int W, int Cin, int Cout,
                              int ksize, int flags, float scale, void* stream);
/* Split-K partial results as plain SLABS instead of fp32 atomics (round 5).  A job: out[i] += scale * sum_{s < nslabs}
 * slabs[s * stride + i], i < n, the slabs added in ascending order (deterministic); gank_sum_slabs runs up to 8 jobs per launch.
 * gank_conv2d_wgrad_batched_slabs = gank_conv2d_wgrad_batched whose partial tiles go to `ws` (gank_conv2d_wgrad_batched_ws_elems
 * floats; 0 = this geometry is not on the slab-capable filter-row kernel: use gank_conv2d_wgrad_batched) and which fills
 * jobs[count] for a LATER gank_sum_slabs call of the caller (several producers' jobs in one launch); bias gradients are
 * accumulated directly as before. */
typedef struct gank_slab_job {
  const float* slabs;
  float* out;
  long n;          /* outputs                                  */
  long stride;     /* floats between consecutive slabs (>= n)  */
  int nslabs;
  float scale;
  int fold;        /* 0: plain sum.  1: each slab holds the SIXTEEN taps of a 4x4 stride-2 filter gradient [16][n / 9] (ConvMeanPool as one
                      conv, gank_convpool3x3_wgrad) and out the nine of the 3x3 filter: out[3i+j] += scale * sum_{s,t in {0,1}} slab[4(i+s)+j+t] */
  long out_run;    /* 0: out is linear.  > 0 (plain sums of <= 32 slabs): output i goes to out[(i / out_run) * out_pitch + i % out_run] -- runs  */
  long out_pitch;  /* of out_run elements out_pitch apart: the first Cin input channels of every tap of a wider filter (both % 4 == 0)   */
} gank_slab_job;
int gank_sum_slabs(const gank_slab_job* jobs, int count, void* stream);    /* up to 12 jobs per launch */
/* gank_conv2d_wgrad whose split-K kernels (per-tap, 1x1 / narrow-channel forms) store their partial tiles into per-split copies
 * of the filter inside `slab_ws` (gank_conv2d_wgrad_slab_elems floats; 0 = not worth it / not applicable) instead of adding
 * them to dw with fp32 atomics; *job then describes the sum (nslabs = 0: the launch accumulated into dw directly, nothing to do). */
long gank_conv2d_wgrad_slab_elems(int N, int H, int W, int Cin, int Cout, int ksize, int flags);
/* ... of the FIRST Cin input channels of a [k,k,Cin_total,Cout] filter (x holds those channels only), accumulated into their rows
 * of dw_full by the job (out_run / out_pitch); needs the all-taps kernel with at most 32 pixel splits -- gank_conv2d_wgrad_slab_splits
 * says how many (0: this entry fails and launches nothing). */
int gank_conv2d_wgrad_slab_splits(int N, int H, int W, int Cin, int Cout, int ksize, int flags);
int gank_conv2d_wgrad_slabs_rows(const void* x, const void* dy, float* dw_full, float* dbias, int N, int H, int W, int Cin, int Cin_total, int Cout,
                                 int ksize, int flags, float scale, float* slab_ws, long slab_elems, gank_slab_job* job, void* stream);
/* ... and, as extra workgroups of the same launch, the per-label tap sums of dy that gank_label_conv3x3_bwd (below) needs: lists from
 * gank_label_conv3x3_table, V labels, tap_sums_ws of gank_label_conv3x3_bwd_ws_floats floats; call gank_label_conv3x3_bwd with
 * dy = NULL and that workspace afterwards (its own first launch is then skipped). */
int gank_conv2d_wgrad_slabs_rows_tap_sums(const void* x, const void* dy, float* dw_full, float* dbias, int N, int H, int W, int Cin, int Cin_total,
                                          int Cout, int ksize, int flags, float scale, float* slab_ws, long slab_elems, gank_slab_job* job,
                                          const int32_t* lists, int V, float* tap_sums_ws, void* stream);
int gank_conv2d_wgrad_slabs(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int Cin, int Cout, int ksize,
                            int flags, float scale, float* slab_ws, long slab_elems, gank_slab_job* job, void* stream);
/* (layers on the all-taps kernel: gank_conv2d_wgrad_slab_elems = gank_conv2d_wgrad_ws_elems, and the slab reduction that
 *  gank_conv2d_wgrad launches itself becomes the job.)  The same for the ConvMeanPool filter gradient: the filter-row kernel's
 *  slabs (ws16) are left for the caller's gank_sum_slabs (job->fold = 1); job->nslabs = 0 when the call did everything itself. */
int gank_convpool3x3_wgrad_job(const void* x, const void* dy, float* dw, float* dbias, float* ws16, long ws_elems, int N, int Hp, int Wp,
                               int Cin, int Cout, int flags, gank_slab_job* job, void* stream);
long gank_conv2d_wgrad_batched_ws_elems(int count, int N, int H, int W, int Cin, int Cout, int ksize, int flags);
int gank_conv2d_wgrad_batched_slabs(const gank_wgrad_item* items, int count, int N, int H, int W, int Cin, int Cout, int ksize,
                                    int flags, float scale, float* ws, long ws_elems, gank_slab_job* jobs, void* stream);
/* Two filter gradients of 3-channel-input layers (ksize 1 or 3, SAME, stride 1; Cout % 128 == 0) with DIFFERENT geometry in
 * one launch: the first critic block's Conv1 (3x3, 32x32) and Shortcut (1x1 on the pooled 16x16 image),
 * gan_cifar_resnet.py:212-234.  Same arithmetic as two gank_conv2d_wgrad calls (which it falls back to). */
int gank_conv2d_wgrad_narrow_pair(const void* x0, const void* dy0, float* dw0, float* db0, int N0, int H0, int W0, int Cout0, int ks0,
                                  const void* x1, const void* dy1, float* dw1, float* db1, int N1, int H1, int W1, int Cout1, int ks1,
                                  float scale, void* stream);

/* ---- identity-shortcut residual blocks on 8x8 images, fused (SNGAN/gan_cifar_resnet.py:156-209, resample=None,
 * no normalisation: the critic's D.Block.3 / D.Block.4, :291-297) ---------------------------------------------------
 * One workgroup per sample keeps  y = x + conv_2(relu(conv_1(relu(x)) + b1)) + b2  of up to two consecutive blocks on
 * chip (and, optionally, the relu + spatial mean that follows the last block, :299-301).  C must be 128.
 *   fwd: x [N,8,8,C]; w_rfrag[2*nblocks]: prep kind 4 `wf` operands (conv_1, conv_2 per block); bias[2*nblocks] (entries
 *        may be NULL); h1[b] (optional) <- conv_1 output of block b before its relu, y[b] (optional) <- block output,
 *        both [N,8,8,C]: what the backward pass needs; pooled (optional) [N,C].
 *   bwd: dy [N,8,8,C] gradient of the last block's output -- or dy = NULL and dpool [N,C] + ylast: the gradient of
 *        `pooled`, expanded on the fly (kept in dy_out when given: the last conv_2's filter gradient reads it).
 *        wd_rfrag / h1 / xin / g1 / dx are listed in the order the backward pass visits them: LAST block first;
 *        wd_rfrag[2*i], [2*i+1] = kind 4 `wd` operands of conv_2, conv_1 of that block; xin = the block's input.
 *        g1[i] (optional) <- gradient of conv_1's output, dx[i] (optional except the last) <- gradient of the block input.
 *        Filter gradients: gank_conv2d_wgrad(_batched) on (relu(xin), g1) and (relu(h1), dy of the block). */
int gank_res8_chain_fwd(const void* x, const void* const* w_rfrag, const float* const* bias, void* const* h1,
                        void* const* y, void* pooled, int N, int C, int nblocks, void* stream);
int gank_res8_chain_bwd(const void* dy, const void* dpool, const void* ylast, void* dy_out, const void* const* wd_rfrag,
                        const void* const* h1, const void* const* xin, void* const* g1, void* const* dx, int N, int C,
                        int nblocks, void* stream);
/* The same pair with the critic's head inside: D.Output (a dense layer to ONE logit on the pooled features,
 * gan_cifar_resnet.py:303-304) and the hinge loss on it (:379-381 mode 0 with the first n_real samples real, :492 mode 1) --
 * what gank_critic_head_hinge does as a launch of its own between the two chains (a single latency-bound workgroup).
 *   fwd_head: additionally logits[n] = bf16(pooled[n] . head_w + head_b[0]) (head_w fp32 [C], head_b fp32 [1] or NULL,
 *        logits bf16 [N]); `pooled` is required.  logits == NULL: exactly gank_res8_chain_fwd.
 *   bwd_head: instead of dy / dpool the gradient of the pooled features is built from the logits: d loss / d pooled[n][c] =
 *        bf16(dl[n] * head_w[c]), dl[n] = bf16(loss_scale * d hinge / d logit) -- bit for bit what gank_critic_head_hinge_scaled
 *        hands to gank_res8_chain_bwd -- and one extra workgroup writes loss[0] and ACCUMULATES w_grad [C] / b_grad [1] (either
 *        may be NULL) = the layer's weight / bias gradients (fixed summation order: deterministic). */
typedef struct gank_res8_head {
  const void* logits;   /* bf16 [N], written by gank_res8_chain_fwd_head */
  const float* head_w;  /* fp32 [C] */
  const void* pooled;   /* bf16 [N,C], written by gank_res8_chain_fwd_head */
  float* loss;          /* fp32 [1] <- the hinge loss */
  float* w_grad;        /* fp32 [C] += , or NULL */
  float* b_grad;        /* fp32 [1] += , or NULL */
  int n_real, mode;     /* mode 0: hinge_d (n_real real samples first); mode 1: hinge_g */
  float loss_scale;     /* power of two: multiplies the gradients, not the loss */
} gank_res8_head;
int gank_res8_chain_fwd_head(const void* x, const void* const* w_rfrag, const float* const* bias, void* const* h1,
                             void* const* y, void* pooled, const float* head_w, const float* head_b, void* logits, int N, int C,
                             int nblocks, void* stream);
int gank_res8_chain_bwd_head(const gank_res8_head* head, const void* ylast, void* dy_out, const void* const* wd_rfrag,
                             const void* const* h1, const void* const* xin, void* const* g1, void* const* dx, int N, int C,
                             int nblocks, void* stream);

/* ---- 3x3 SAME stride-1 conv on 16x16 images, one image x 128 output channels per workgroup, the image resident in LDS per
 * 64-channel chunk and the weights streamed in fragment order (operand: prep kind 4, rows = output channels): the critic's
 * D.Block.2.Conv1 (tf.nn.conv2d of conv2d.py:180-187 behind the pre-activation relu of gan_cifar_resnet.py:186) and, with the
 * kind 4 `wd` operand and Cin / Cout swapped, its input gradient (Conv2DBackpropInput).  x [N,16,16,Cin], y [N,16,16,Cout];
 * y = conv(in(x)) + bias, zeroed where relu_ref <= 0 (optional, [N,16,16,Cout]), + residual (optional, same shape).
 * flags: GANK_IN_RELU.  Cin % 64 == 0, Cout % 128 == 0. */
int gank_img16_conv3x3(const void* x, const void* w_rfrag, const float* bias, const void* relu_ref, const void* residual, void* y,
                       int N, int Cin, int Cout, int flags, void* stream);
/* conv3x3_SAME(relu(cond_batchnorm(x))) + bias (+ residual) on 16x16 images with the normalisation (common/ops/normalization.py:47-57)
 * and the nonlinearity (SNGAN/gan_cifar_resnet.py:186) applied while the image-resident kernel stages its operand: for passes
 * that keep nothing for a backward pass (the 320-sample generator pass behind the critic updates, :326-332; sampling, :530-555) the
 * normalised tensor is never written or read.  stats [groups][2][Cin] = (mean, invstd) per tower; flags: GANK_RES_UPSAMPLE2X,
 * GANK_STATS_PREZEROED; stat_sums as gank_img16_conv3x3_stats.  Bit-identical to gank_cbn_fwd* followed by gank_img16_conv3x3_stats. */
int gank_cbn_relu_img16_conv3x3(const void* x, const int32_t* labels, const float* gamma, const float* beta, const float* stats,
                                const void* w_rfrag, const float* bias, const void* residual, void* y, int N, int Cin, int Cout,
                                int groups, int n_labels, int flags, float* stat_sums, int stat_groups, void* stream);
/* the same with the conditional-batch-norm statistics of y accumulated by the epilogue (stat_sums as gank_conv2d_fprop_stats: the
 * generator's G.Block.2.Conv2, whose output feeds G.Block.3's first normalisation, gan_cifar_resnet.py:176-209) and, with
 * GANK_RES_UPSAMPLE2X, a half-resolution residual [N,8,8,Cout] added nearest-neighbour upsampled (the 'up' block's shortcut, :179-182).
 * flags: GANK_IN_RELU | GANK_RES_UPSAMPLE2X | GANK_STATS_PREZEROED; stat_sums NULL: no statistics. */
int gank_img16_conv3x3_stats(const void* x, const void* w_rfrag, const float* bias, const void* relu_ref, const void* residual, void* y,
                             int N, int Cin, int Cout, int flags, float* stat_sums, int stat_groups, void* stream);

/* ---- 3x3 SAME conv on 8x8 images, one LDS-resident image per workgroup (operand: prep kind 4, rows = output channels):
 * the generator's first residual block (gan_cifar_resnet.py:179-207, resample='up' at 4x4 -> 8x8) -- tf.nn.conv2d of
 * conv2d.py:180-187 behind the depth_to_space upsample of :139-146 -- and, with the dgrad operand, the input gradients.
 *   x [N,8,8,Cin] (GANK_IN_UPSAMPLE2X: [N,4,4,Cin], NN-upsampled by the loader); Cin 128 | 256; Cout % 128 == 0;
 *   y [N,8,8,Cout] = conv + residual ([N,8,8,Cout], or [N,4,4,Cout] added upsampled with GANK_RES_UPSAMPLE2X) + bias;
 *   stat_sums (optional) as in gank_conv2d_fprop_stats (GANK_STATS_PREZEROED honoured);
 *   GANK_OUT_POOLSUM2X: y [N,4,4,Cout] = 2x2 sums of the conv (no bias / residual / statistics). */
int gank_res8_conv3x3(const void* x, const void* w_rfrag, const float* bias, const void* residual, void* y, int N, int Cin,
                      int Cout, int flags, float* stat_sums, int stat_groups, void* stream);

/* ---- ConvMeanPool 3x3 on LDS-resident images (same arithmetic as gank_convpool3x3_fprop / _dgrad; operands: prep kind 5)
 * A workgroup owns an 8x16 (or 8x8) patch of POOLED pixels and 128 output channels, stages its input region once per
 * 64-channel chunk as four parity planes (stride-2 taps become unit-stride LDS reads) and streams the weights from L2 in
 * MFMA-fragment order: no per-tap re-gather of the input (16 taps x 33.5 MB for D.Block.1.Conv2), no barrier per K-step.
 *   fprop: x [N,2Hp,2Wp,Cin] -> y [N,Hp,Wp,Cout]; Cin % 64 == 0, Cout % 128 == 0, Hp % 8 == 0, Wp % 16 == 0 or Wp == 8;
 *          flags: GANK_IN_RELU; bias / residual (pooled resolution) optional
 *   dgrad: dy [N,Hp,Wp,Cout] -> dx [N,2Hp,2Wp,Cin]; Cout == 128, Cin % 128 == 0; relu_ref optional (masks with relu_ref > 0) */
int gank_cpool_res_fprop(const void* x, const void* w_rfrag, const float* bias, const void* residual, void* y, int N, int Hp,
                         int Wp, int Cin, int Cout, int flags, void* stream);
int gank_cpool_res_dgrad(const void* dy, const void* w_rfrag, const void* relu_ref, void* dx, int N, int Hp, int Wp, int Cin,
                         int Cout, void* stream);
/* The same input gradient when its ONLY consumer is the filter gradient of a 3x3 conv on a 3-channel image in front of it
 * (OptimizedResBlockDisc1 in a critic update, SNGAN/gan_cifar_resnet.py:212-234; tf.gradients of disc_cost w.r.t.
 * D.Block.1.Conv1 / D.Block.1.Shortcut, :523-526): dx is never stored -- each finished tile (masked by relu_ref > 0, rounded to the
 * activation dtype as the stored tensor was) feeds  dw1 [3,3,3,Cin] += x_image (*) dx  and  db1 [Cin] += sum dx  inside the launch;
 * with x_pooled [N,Hp,Wp,3] also the 1x1 shortcut conv whose output gradient dy is:  dws [1,1,3,Cout] += x_pooled^T dy,
 * dbs [Cout] += sum dy.  x_image [N,2Hp,2Wp,3]; Wp == 16, Hp % 8 == 0, Cout == 128, Cin % 128 == 0; db1 / x_pooled / dws / dbs
 * optional.  One [32][Cin] tile per workgroup: fp32 atomics into the four targets, or, with `slabs` (Cin == 128 only;
 * N * Hp / 8 slabs of 32 x 128 floats), plain stores -- rows 0..26 of a slab belong to dw1, row 27 to db1, rows 28..30 to dws,
 * row 31 to dbs -- that the caller sums later (gank_sum_slabs, stride 4096): the targets are then not touched by this call. */
int gank_cpool_res_dgrad_image_wgrad(const void* dy, const void* w_rfrag, const void* relu_ref, const void* x_image, float* dw1,
                                     float* db1, const void* x_pooled, float* dws, float* dbs, int N, int Hp, int Wp, int Cin,
                                     int Cout, float* slabs, void* stream);

/* ---- UpsampleConv 3x3 (SNGAN/gan_cifar_resnet.py:140-153) as a stride-2 transposed conv -------------
 * nearest-neighbour 2x followed by a 3x3 SAME conv equals a 4x4 stride-2 transposed conv whose taps are sums
 * of the 3x3 taps.  fprop runs its 4 output phases (2x2 taps each over the LOW-RES input): 4 instead of 9
 * MACs per output and weight; dgrad is the 4x4 stride-2 conv of dy.  Exact in real arithmetic; the summed
 * taps are rounded to bf16 once.  prep: w fp32 [3,3,Cin,Cout] -> wph bf16 [4][roundup(Cout,32)][4*Cin] and
 * wd4 bf16 [roundup(Cin,32)][roundup(16*Cout,64)].  x [N,Hl,Wl,Cin] -> y [N,2Hl,2Wl,Cout]; residual/relu_ref
 * as in gank_conv2d_fprop/_dgrad.  (The filter gradient keeps gank_conv2d_wgrad with GANK_IN_UPSAMPLE2X.) */
int gank_upconv3x3_prep_weights(const float* w, void* wph, void* wd4, int Cin, int Cout, void* stream);
int gank_upconv3x3_fprop(const void* x, const void* wph, const float* bias, const void* residual, void* y,
                         int N, int Hl, int Wl, int Cin, int Cout, int flags, void* stream);
int gank_upconv3x3_fprop_stats(const void* x, const void* wph, const float* bias, const void* residual, void* y,
                               int N, int Hl, int Wl, int Cin, int Cout, int flags, float* stat_sums, int groups, int* produced,
                               void* stream);
int gank_upconv3x3_dgrad(const void* dy, const void* wd4, const void* relu_ref, void* dx, int N, int Hl, int Wl,
                         int Cin, int Cout, void* stream);

/* ---- ConvMeanPool 3x3 (SNGAN/gan_cifar_resnet.py:112-123; common/resnet_block.py:53-65) as a 4x4 stride-2 conv ---
 * mean_pool2x2(conv3x3_SAME(x) + b) == conv4x4_stride2_pad1(x, W4) + b with
 *   W4[a][b] = 1/4 sum_{i in I(a), j in I(b)} W3[i][j],  I(0)={0} I(1)={0,1} I(2)={1,2} I(3)={2}:
 * 16 taps per POOLED pixel (= 4 per conv output instead of 9) and no full-resolution intermediate.  Exact in real
 * arithmetic; the summed taps are rounded to bf16 once.
 *   prep : w fp32 [3,3,Cin,Cout] -> wp4 bf16 [roundup(Cout,32)][roundup(16*Cin,64)]   (fprop operand)
 *                                   wphd bf16 [4][roundup(Cin,32)][4*Cout]             (dgrad operand, 4 phases)
 *   fprop: x [N,2Hp,2Wp,Cin] -> y [N,Hp,Wp,Cout]; flags: GANK_IN_RELU; bias/residual (pooled resolution) optional
 *   dgrad: dy [N,Hp,Wp,Cout] -> dx [N,2Hp,2Wp,Cin] (stride-2 transposed conv as 4 phases of 2x2 taps);
 *          relu_ref (optional, like dx) masks the result with relu_ref > 0.  Needs Cout % 64 == 0.
 *   wgrad: dw fp32 [3,3,Cin,Cout] += fold(dW4), dbias (optional) += column sums of dy; ws16 = fp32 scratch of
 *          ws_elems >= gank_convpool3x3_wgrad_ws_elems(...) elements (the partial tiles of the pixel splits);
 *          flags: GANK_IN_RELU (relu applied to x while staging). */
int gank_convpool3x3_prep_weights(const float* w, void* wp4, void* wphd, int Cin, int Cout, void* stream);
int gank_convpool3x3_fprop(const void* x, const void* wp4, const float* bias, const void* residual, void* y,
                           int N, int Hp, int Wp, int Cin, int Cout, int flags, void* stream);
int gank_convpool3x3_dgrad(const void* dy, const void* wphd, const void* relu_ref, void* dx, int N, int Hp, int Wp,
                           int Cin, int Cout, void* stream);
/* Filter gradient of UpsampleConv 3x3 (gan_cifar_resnet.py:138-153: NN-upsample 2x, then Conv2D 3x3) in its phase form: x_low
 * bf16 [N,H,W,Cin] (the conv's input BEFORE the upsample), dy bf16 [N,2H,2W,Cout]; ACCUMULATES into dw fp32 [3,3,Cin,Cout]
 * (Conv2DBackpropFilter of conv2d.py:180-187 composed with the upsample); 4/9 of the multiply-adds of the all-taps form.
 * ws16: workspace of gank_upconv3x3_wgrad_ws_elems floats -- 0 = this shape is not served (use gank_conv2d_wgrad with
 * GANK_IN_UPSAMPLE2X).  No bias gradient (the sum of dy over pixels: gank_colsum_bf16). */
long gank_upconv3x3_wgrad_ws_elems(int N, int H, int W, int Cin, int Cout);
int gank_upconv3x3_wgrad(const void* x_low, const void* dy, float* dw, float* ws16, long ws_elems, int N, int H, int W, int Cin, int Cout,
                         void* stream);
long gank_convpool3x3_wgrad_ws_elems(int N, int Hp, int Wp, int Cin, int Cout);
int gank_convpool3x3_wgrad(const void* x, const void* dy, float* dw, float* dbias, float* ws16, long ws_elems, int N, int Hp, int Wp,
                           int Cin, int Cout, int flags, void* stream);

/* ---- general convolution (Pix2Pix/networks.py:366-536: 4x4 stride-2 SAME encoders, 4x4 stride-1 SAME decoders on
 * NN-upsampled inputs, tf.pad + VALID 4x4 convs of the PatchGAN critic) -----------------------------------------------
 * Any ksize <= 7, stride 1 | 2, `pad` = rows/columns of zeros in FRONT (TF SAME: (total pad)/2 rounded down; tf.pad 1 +
 * VALID: 1); taps past the stored image read zeros, so the trailing pad is implied by the output size.  x [N,Hin,Win,Cin]
 * as stored, y [N,Hout,Wout,Cout]; wf / wd from gank_conv2d_prep_weights(w [k,k,Cin,Cout]).  flags: GANK_IN_RELU,
 * GANK_IN_UPSAMPLE2X (stride 1 only), GANK_OUT_TANH (fprop).  dgrad covers stride 1 (dx at the gathered -- upsampled --
 * size; relu_ref optional mask); the stride-2 input gradient is the transposed conv: gank_deconv2d_prep_phases on the same
 * filter memory ([k,k,Cin,Cout] read as [k,k,Cout',Cin']) + gank_upconv3x3_fprop.  wgrad ACCUMULATES. */
int gank_conv2d_general_fprop(const void* x, const void* wf, const float* bias, void* y, int N, int Hin, int Win,
                              int Hout, int Wout, int Cin, int Cout, int ksize, int stride, int pad, int flags, void* stream);
int gank_conv2d_general_dgrad(const void* dy, const void* wd, const void* relu_ref, void* dx, int N, int Hx, int Wx,
                              int Hdy, int Wdy, int Cin, int Cout, int ksize, int pad, void* stream);
int gank_conv2d_general_wgrad(const void* x, const void* dy, float* dw, float* dbias, int N, int Hx, int Wx, int Hdy, int Wdy,
                              int Cin, int Cout, int ksize, int stride, int pad, int flags, void* stream);

/* MeanPoolConv with a 1x1 filter on a 3-channel image (D.Block.1.Shortcut: gan_cifar_resnet.py:125-137, 218-221):
 * y [N,H,W,Cout] = conv1x1(mean_pool2x2(x [N,2H,2W,3])) + bias with the pool inside the conv's gather (the pooled value is
 * rounded to the element type exactly as gank_pool2x2 + gank_conv2d_fprop round it); `pooled` [N,H,W,3] (optional) receives
 * the pooled image, which the filter gradient (gank_conv2d_wgrad on it) needs.  wf from gank_conv2d_prep_weights. */
int gank_meanpool_conv1x1_fprop(const void* x, const void* wf, const float* bias, void* y, void* pooled,
                                int N, int H, int W, int Cin, int Cout, void* stream);
/* gank_conv2d_fprop(x [N,H,W,3], wf1: 3x3 -> Cout1, bias1, no flags) AND gank_meanpool_conv1x1_fprop(x, wfs: 1x1 -> Couts at H/2 x W/2,
 * pooled side output) in ONE launch: the two image-side layers of OptimizedResBlockDisc1 (gan_cifar_resnet.py:212-234) read the same
 * image; the shortcut's workgroups run behind conv_1's.  Results bit for bit those of the two entries.  Cout1, Couts % 128 == 0. */
int gank_image_conv_pair_fprop(const void* x, const void* wf1, const float* bias1, void* y1, const void* wfs, const float* biass, void* ys,
                               void* pooled, int N, int H, int W, int Cout1, int Couts, void* stream);

/* ---- Deconv2D (common/ops/deconv2d.py:99-114): tf.nn.conv2d_transpose stride 2 SAME --------------
 * x [N,H,W,Cin] -> y [N,2H,2W,Cout]; master filter F fp32 [k,k,Cout,Cin].  The op has no caller in
 * the reference; it is provided at op level on the same two MFMA engines:
 *   fprop = zero-insertion gather + stride-1 conv with the flipped filter; its operand wz is the `wd`
 *           output of gank_conv2d_prep_weights(F, ksize, Cin:=Cout, Cout:=Cin)   (F viewed as HWIO);
 *   dgrad = the stride-2 SAME conv; its operand wfz is the `wf` output of the same prep call;
 *   wgrad accumulates into dF [k,k,Cout,Cin]. */
/* fprop by output phase for ksize 3 and 4 (4 MACs per output instead of k*k on inserted zeros): build wph
 * [4][roundup(Cout,32)][4*Cin] from the filter f fp32 [k,k,Cout,Cin] and run gank_upconv3x3_fprop(x, wph, bias, NULL, y, ...)
 * -- the same four 2x2-tap phases over the low-resolution input; needs Cin % 64 == 0. */
int gank_deconv2d_prep_phases(const float* f, void* wph, int ksize, int Cin, int Cout, void* stream);
int gank_deconv2d_fprop(const void* x, const void* wz, const float* bias, void* y, int N, int H, int W,
                        int Cin, int Cout, int ksize, void* stream);
int gank_deconv2d_dgrad(const void* dy, const void* wfz, void* dx, int N, int H, int W, int Cin, int Cout,
                        int ksize, void* stream);
int gank_deconv2d_wgrad(const void* x, const void* dy, float* df, int N, int H, int W, int Cin, int Cout,
                        int ksize, void* stream);

/* ---- column sum: out[c] += scale * sum_rows x[r][c]   (bias gradients; tf.nn.bias_add grad) ------ */
int gank_colsum_bf16(const void* x, float* out, long rows, int C, float scale, void* stream);

/* ---- spectral normalisation (common/ops/sn.py:15-69) ---------------------------------------------
 * Batched over `count` weights described by a HOST array of gank_sn_desc (copied into the kernel
 * arguments, 16 per launch group: no device table, capturable).  One power
 * iteration from u_in: a=W u, v=a/(|a|+eps); b=W^T v, u'=b/(|b|+eps); sigma=v W u'^T; W_bar=W/sigma.
 * fwd writes W_bar (fp32, same shape), v, u_out, and {sigma, |a|, |b|, ...} to `scal`.  The caller copies
 * u_out over u only under update_collection=None (sn.py:55-56); NO_OPS never writes u (sn.py:62-65).
 * bwd is the FULL gradient through the iteration (no stop_gradient in sn.py:34-61) and ACCUMULATES
 * into dW.  Two launches each way: fwd = {row dots + partial column sums + norms by the last chunk of each weight
 * (ticket), W/sigma + the backward pass's per-row terms}; bwd = {partial <G,W>, apply}.
 * Workspaces per weight: `a` [K], `v` [K], `ga` [K], `b` [C], `bpart` [gank_sn_ws_floats(K, C)] (partial column sums and
 * |a|^2 partials per row chunk, <G,W> partials per finer piece).  `rowdot` is no longer used (may be NULL). */
typedef struct gank_sn_desc {
  const float* W;      /* [K,C] master weight                                   */
  const float* u_in;   /* [C]                                                    */
  float* u_out;        /* [C]                                                    */
  float* v;            /* [K]                                                    */
  float* W_bar;        /* [K,C]                                                  */
  float* scal;         /* [8]: sigma, n=|a|, m=|b|, s, <G,W>, a.g_v, -, -        */
  float* a;            /* [K]   workspace, kept for backward                     */
  float* b;            /* [C]   kept for backward                                */
  float* bpart;        /* [gank_sn_ws_floats(K,C)] workspace, kept for backward   */
  const float* dW_bar; /* [K,C] backward input                                   */
  float* dW;           /* [K,C] backward output (accumulated)                    */
  float* rowdot;       /* unused (earlier revisions: [K] workspace)              */
  float* ga;           /* [K]   written by fwd, read by bwd                      */
  float* u_snap;       /* [C] or NULL: fwd also copies u_in here and bwd reads it instead of u_in, so that
                          u_out may alias u_in (u.assign(u_final) of sn.py:55-56 without a snapshot/copy-back pair) */
  int K, C;
  int row_offset;      /* filled by the library                                  */
  int chunk_offset;    /* filled by the library                                  */
} gank_sn_desc;
long gank_sn_ws_floats(int K, int C);
int gank_sn_power_iter_fwd(const gank_sn_desc* table, int count, void* stream);
int gank_sn_power_iter_bwd(const gank_sn_desc* table, int count, void* stream);
/* The forward pass with what always follows it in a spectrally normalised network folded into its second launch
 * (count <= 16): the bf16 MFMA operand copies of W / sigma -- `prep[i]` as for gank_conv2d_prep_weights_batched, except
 * that prep[i].w is the MASTER weight table[prep_weight[i]].W (the division by sigma happens inside the launch: same
 * arithmetic as preparing W_bar, without reading it back) -- and, optionally (`label` != NULL), the per-label rows of a
 * small dense layer on an embedding table, out[l] = bf16(bf16(emb[l]) W_bar + bias) with W_bar = entry `weight` of the
 * table: the critic's label branch embed_y -> Linear('D.Embedding_y') (gan_cifar_resnet.py:276-281) computed once per
 * label instead of once per sample (same arithmetic as gank_embedding_fwd + gank_linear_fwd on the 10 labels). */
typedef struct gank_label_dense_desc {
  const float* table;  /* [V, D] fp32 embedding table (common/ops/embedding.py:28-40)   */
  const float* bias;   /* [Cout] fp32 or NULL                                             */
  void* out;           /* [V, Cout] bf16                                                  */
  int V, D;
  int weight;          /* index into the gank_sn_desc table: its W is [D, Cout]           */
} gank_label_dense_desc;
int gank_sn_power_iter_fwd_prep(const gank_sn_desc* table, int count, const gank_prep_desc* prep, const int* prep_weight,
                                int prep_count, const gank_label_dense_desc* label, void* stream);
/* The pieces of the two passes, for a train step that folds the end of one critic update into the start of the next (round 5):
 *   gank_sn_power_iter_fwd_a     : the first launch of the forward pass alone (power iteration into the table's workspaces, u' -> u_out);
 *   gank_sn_power_iter_fwd_b_prep: the second launch alone (W / sigma, operand copies, label table) on workspaces a forward-A or the
 *                                  fused tail below already filled; with u_total > 0 it also performs u.assign(u_final) (sn.py:55-56) as a
 *                                  flat copy over the CONCATENATED u vectors of the table: u_snap_flat <- u_flat (what the backward pass
 *                                  reads; may be NULL), u_flat <- u_next_flat;
 *   gank_sn_power_iter_bwd_gw    : the first backward launch alone (partial <dW_bar, W>);
 *   gank_sn_adam_fwd_a           : ONE launch for  sn backward apply (the table's dW_bar, workspaces, u_snap as after _bwd_gw)  +
 *                                  tf.train.AdamOptimizer over the WHOLE flat buffer p / g / m / v [n] (SNGAN/gan_cifar_resnet.py:521-526;
 *                                  arguments as gank_adam_tf_health; the consumed gradients -- g and every dW_bar -- are cleared)  +  the
 *                                  NEXT forward pass's first launch on the updated weights (from the table's u_in, u' -> u_next[i]).
 *                                  Every table entry's W must be a disjoint view of p and its dW the view of g at the same offset;
 *                                  C <= 256.  flags bit 0: the caller guarantees that every entry's dW is zero on entry (nothing but
 *                                  this backward pass contributes to those weights): it is then neither read nor cleared.
 *                                  bump (optional): an int64 counter that advances by one, after every block has read `iteration`,
 *                                  when bump_when_zero[0] == 0 -- the train loop's iteration count behind the last critic update
 *                                  of an iteration (the feed-ring slot has wrapped to 0 by then) without a launch of its own.
 *                                  Bit-identical to gank_sn_power_iter_bwd + gank_adam_tf + gank_sn_power_iter_fwd_a. */
int gank_sn_power_iter_fwd_a(const gank_sn_desc* table, int count, void* stream);
int gank_sn_power_iter_fwd_b_prep(const gank_sn_desc* table, int count, const gank_prep_desc* prep, const int* prep_weight,
                                  int prep_count, const gank_label_dense_desc* label, float* u_flat, float* u_snap_flat,
                                  const float* u_next_flat, int u_total, void* stream);
/* gank_sn_power_iter_fwd_b_prep with the critic's input feed of the update (gank_critic_feed below: same arguments, same arithmetic,
 * same counters) as one more block range of the launch: the feed of a critic update and the second spectral-norm launch of its
 * forward pass are independent and both precede every layer, so the update starts with one launch instead of two. */
typedef struct gank_critic_feed_desc {
  const uint8_t* real_all;     /* as gank_critic_feed */
  const int32_t* labels_all;
  const void* fake_all;
  void* both;
  int32_t* labels2;
  int32_t* slot;
  uint64_t* rng_state;
  uint32_t* done_counter;
  int B, n_slots;
} gank_critic_feed_desc;
int gank_sn_power_iter_fwd_b_prep_feed(const gank_sn_desc* table, int count, const gank_prep_desc* prep, const int* prep_weight,
                                       int prep_count, const gank_label_dense_desc* label, float* u_flat, float* u_snap_flat,
                                       const float* u_next_flat, int u_total, const gank_critic_feed_desc* feed, void* stream);
int gank_sn_power_iter_bwd_gw(const gank_sn_desc* table, int count, void* stream);
int gank_sn_adam_fwd_a(const gank_sn_desc* table, int count, float* const* u_next, float* p, float* g, float* m, float* v, long n,
                       float* hp, int64_t* t_state, const int64_t* iteration, uint64_t* health, int flags, int64_t* bump,
                       const int32_t* bump_when_zero, void* stream);

/* ---- conditional batch norm (common/ops/normalization.py:27-59) ----------------------------------
 * Batch moments over (N/groups, H, W) per tower (biased variance, eps 1e-5), per-sample gamma/beta
 * rows gathered by label from [n_labels,C] tables.  `groups` towers of N/groups consecutive samples
 * have independent statistics (one Generator() call per tower in the reference,
 * SNGAN/gan_cifar_resnet.py:326-332,464-482).  relu!=0 fuses the following nonlinearity (:186).
 * fwd: x,y bf16 [N,HW,C]; stats fp32 [groups][2][C] (mean, invstd); ws >= groups*parts*3*C floats
 * with parts = gank_cbn_parts(N/groups*HW).  bwd: dy (gradient w.r.t. y; masked by y>0 when relu),
 * writes dx and ACCUMULATES dgamma/dbeta [n_labels,C]; ws >= N*2*C + groups*2*C floats. */
int gank_cbn_parts(long rows_per_group);
int gank_cbn_fwd(const void* x, const int32_t* labels, const float* gamma, const float* beta, void* y,
                 float* stats, float* ws, int N, int HW, int C, int groups, int n_labels, int relu, void* stream);
/* the same with the variance epsilon as an argument: instance_norm (normalization.py:105-122, epsilon 1e-6) is this
 * kernel set with groups = N (one tower per sample) and a one-row gamma/beta table */
int gank_cbn_fwd_eps(const void* x, const int32_t* labels, const float* gamma, const float* beta, void* y,
                     float* stats, float* ws, int N, int HW, int C, int groups, int n_labels, int relu, float eps, void* stream);
/* forward with the statistics taken from the producing conv's epilogue (gank_conv2d_fprop_stats): sums [groups][GANK_STAT_SLOTS][2][C]
 * of (x - shift[c]) and its square per tower (partial copies, added here), shift = that conv's bias or NULL.  ONE launch (no statistics pass, no merge);
 * writes stats [groups][2][C] (mean, invstd) for gank_cbn_bwd. */
int gank_cbn_fwd_from_sums(const void* x, const int32_t* labels, const float* gamma, const float* beta, void* y, float* stats,
                           const float* sums, const float* shift, int N, int HW, int C, int groups, int n_labels, int relu,
                           float eps, void* stream);
/* statistics alone (mean, invstd per tower and channel), and the same from a conv epilogue's sums -- for a consumer that
 * normalises inside its own kernel: gank_cbn_relu_conv3x3_fprop = conv3x3_SAME(relu(cond_batchnorm(x))) + bias [tanh] with the
 * normalisation applied while the conv stages its operand (the normalised tensor is never stored; for passes that keep nothing
 * for a backward pass: gan_cifar_resnet.py:257-261 in the critic-feed and sampling passes) */
int gank_cbn_stats(const void* x, float* stats, float* ws, int N, int HW, int C, int groups, float eps, void* stream);
int gank_cbn_stats_from_sums(const float* sums, const float* shift, float* stats, int C, int groups, long rows_per_group,
                             float eps, void* stream);
int gank_cbn_relu_conv3x3_fprop(const void* x, const int32_t* labels, const float* gamma, const float* beta, const float* stats,
                                const void* wf, const float* bias, void* y, int N, int H, int W, int Cin, int Cout,
                                int groups, int n_labels, int flags, void* stream);
int gank_cbn_bwd(const void* dy, const void* x, const void* y, const int32_t* labels, const float* gamma,
                 const float* stats, void* dx, float* dgamma, float* dbeta, float* ws, int N, int HW, int C,
                 int groups, int n_labels, int relu, void* stream);
/* backward with the relu mask RECOMPUTED from x ((x - mean) * invstd * gamma + beta > 0: the forward pass's expression, bit for
 * bit) instead of read from y: one tensor read less in each of its two passes */
int gank_cbn_bwd_remask(const void* dy, const void* x, const float* beta, const int32_t* labels, const float* gamma,
                        const float* stats, void* dx, float* dgamma, float* dbeta, float* ws, int N, int HW, int C,
                        int groups, int n_labels, int relu, void* stream);
/* Both forms on a workspace whose size the caller states (y, or beta with the mask recomputed and y ignored).  With
 * gank_cbn_bwd_ws_floats(N, HW, C, groups) floats the partial sums of a sample's pixel parts go to rows of their own and the last
 * part of the sample to finish adds them in a fixed order: no fill launch, no fp32 atomics on the sums, the same bits every run.  With the smaller workspace of
 * the two entries above (N*2*C + groups*2*C floats) it behaves as they do. */
long gank_cbn_bwd_ws_floats(int N, int HW, int C, int groups);
int gank_cbn_bwd_ws(const void* dy, const void* x, const void* y, const float* beta, const int32_t* labels, const float* gamma,
                    const float* stats, void* dx, float* dgamma, float* dbeta, float* ws, long ws_floats, int N, int HW, int C,
                    int groups, int n_labels, int relu, void* stream);

/* ---- resampling / elementwise glue of the block library ------------------------------------------
 * pool2x2: y = scale * (sum of the 2x2 window) (+ residual)   -- tf.add_n(...)/4. at
 *          gan_cifar_resnet.py:120-121,129-130 with scale=.25; scale=1 is the NN-upsample gradient.
 * unpool2x2_add: y = base + scale * nn_upsample2x(g)            -- gradient of the mean pool. */
/* ---- layer_norm (normalization.py:62-102; tf.contrib.layers.layer_norm, begin_norm_axis=1, begin_params_axis=-1):
 * moments over (H,W,C) per sample (biased variance, eps as given: TF uses 1e-12), gamma/beta fp32 [C].
 * x,y,dy,dx bf16 [N,HW,C]; stats fp32 [N][2] (mean, invstd) from fwd; bwd ACCUMULATES dgamma/dbeta. C % 8 == 0. */
int gank_layer_norm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int N, int HW, int C,
                        float eps, void* stream);
int gank_layer_norm_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx, float* dgamma,
                        float* dbeta, int N, int HW, int C, void* stream);
/* ---- pixel_norm (normalization.py:125-140): y = x * rsqrt(mean_c(x^2) + eps) per pixel; bwd recomputes the scale */
int gank_pixel_norm_fwd(const void* x, void* y, long pixels, int C, float eps, void* stream);
int gank_pixel_norm_bwd(const void* dy, const void* x, void* dx, long pixels, int C, float eps, void* stream);

int gank_pool2x2(const void* x, const void* residual, void* y, int N, int Hout, int Wout, int C, float scale, void* stream);
int gank_unpool2x2_add(const void* g, const void* base, void* y, int N, int Hin, int Win, int C, float scale, void* stream);
int gank_add_bf16(const void* a, const void* b, void* y, long n, void* stream);
int gank_relu_fwd(const void* x, void* y, long n, float leak, void* stream);           /* gan_cifar_resnet.py:80-85 */
int gank_relu_bwd(const void* dy, const void* x, void* dx, long n, float leak, void* stream);
int gank_tanh_bwd(const void* dy, const void* y, void* dx, long n, void* stream);      /* tf.tanh grad, :261 */
int gank_scale_f32(const float* x, const float* s, float* y, long n, void* stream);    /* y = x * s[0] */
/* y = wa a + wb b + wc c + wd d over n floats, null terms skipped, y may alias a term: the weighted sums of scalar losses in the
 * train steps (ACGAN/train.py:108-121; Pix2Pix/model.py gen_loss = gan_weight * GAN + l1_weight * L1) and fp32 accumulation */
int gank_weighted_sum4_f32(const float* a, const float* b, const float* c, const float* d, float wa, float wb, float wc, float wd,
                           float* y, long n, void* stream);
/* dst[0:nbytes] = src[0:nbytes], device to device, as a kernel launch (the tf.assign / feed copies of the captured
 * step: sn.py:55-56 u.assign, gan_cifar_resnet.py:616-620 feeds); 16-byte vectorised when both are 16-B aligned */
int gank_copy_bytes(void* dst, const void* src, long nbytes, void* stream);
/* dst[i*nbytes_each : (i+1)*nbytes_each] = srcs[i][0:nbytes_each] for i < count (<= 16): the N_CRITIC feed batches of an
 * iteration (gan_cifar_resnet.py:616-620: five session.run feeds) into the feed ring in ONE launch.  `srcs` is a HOST array
 * of device pointers, copied into the kernel arguments */
int gank_copy_bytes_gather(void* dst, const void* const* srcs, int count, long nbytes_each, void* stream);
/* two gathers of that kind (different row sizes) in one launch: the iteration's image batches and label vectors into the two feed rings */
int gank_copy_bytes_gather2(void* dst_a, const void* const* srcs_a, int count_a, long nbytes_a, void* dst_b, const void* const* srcs_b, int count_b,
                            long nbytes_b, void* stream);
int gank_cast_f32_bf16(const float* x, void* y, long n, void* stream);
int gank_cast_bf16_f32(const void* x, float* y, long n, void* stream);

/* ---- small dense layers (common/ops/linear.py:161-180: tf.matmul + tf.nn.bias_add) on the fp32 master weights ----
 * y[M,C] = x[M,K] w[K,C] + bias;  x, y, dy, dx bf16; w, dw, dbias fp32, no operand preparation.  For the critic's
 * D.Embedding_y (300->128) and D.Output (128->1) (gan_cifar_resnet.py:296-304), which are launch-latency problems;
 * large layers (G.Input 128->16384) use gank_conv2d_* with ksize 1.  bwd: dx (optional) = dy w^T; dw (optional)
 * += x^T dy; dbias (optional) += column sums of dy. */
int gank_linear_fwd(const void* x, const float* w, const float* bias, void* y, int M, int K, int C, void* stream);
/* the same with the result left in fp32 (y fp32 [M,C]): the 2048 -> 1008 `logits:0` layer of the Inception graph
 * (common/inception/inception_score.py:44-56), whose outputs are exponentiated by the score -- rounding them to 16 bits
 * first would move every class probability by a few percent */
int gank_linear_fwd_f32out(const void* x, const float* w, const float* bias, float* y, int M, int K, int C, void* stream);
int gank_linear_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* dbias, int M, int K, int C,
                    void* stream);

/* relu + tf.reduce_mean(axis=[1,2])  (gan_cifar_resnet.py:299-301): x [N,HW,C] -> y [N,C] */
int gank_relu_meanpool_hw_fwd(const void* x, void* y, int N, int HW, int C, void* stream);
int gank_relu_meanpool_hw_bwd(const void* dy, const void* x, void* dx, int N, int HW, int C, void* stream);

/* expand_dims x2 + tf.tile + tf.concat(axis=3)  (gan_cifar_resnet.py:282-284):
 * y[n,hw,:C1]=a[n,hw,:], y[n,hw,C1:]=e[n,:].  bwd: da = dy[..,:C1], de[n,:] = sum_hw dy[n,hw,C1:]. */
int gank_concat_tile_fwd(const void* a, const void* e, void* y, int N, int HW, int C1, int C2, void* stream);
int gank_concat_tile_bwd(const void* dy, void* da, void* de, int N, int HW, int C1, int C2, void* stream);

/* The label branch of the SNGAN critic (gan_cifar_resnet.py:276-284: embed_y -> Linear -> expand_dims x2 -> tile ->
 * concat) through a per-label table T [V, C2] bf16 = bf16(bf16(emb) W + bias) (gank_label_dense_table, or the
 * `label` option of gank_sn_power_iter_fwd_prep; sigma: a device word W is divided by, or NULL):
 *   fwd: y[n,hw,:C1] = a[n,hw,:], y[n,hw,C1:] = T[labels[n]]   (a label outside [0,V): zeros)
 *   bwd: da = dy[..,:C1]; de32[n,:] = sum_hw dy[n,hw,C1:] (fp32, not rounded per sample);
 *        gank_label_dense_bwd: dT[l] = sum_{n: labels[n]=l} de32[n] (in sample order: deterministic), then
 *        dW [D,C2] += bf16(emb)^T dT, dbias [C2] += sum_l dT[l], demb [V,D] += dT W^T  (each optional).  C2 must divide 1024; (1024 + C2) * V + N floats must fit the LDS. */
int gank_label_dense_table(const float* table, const float* W, const float* sigma, const float* bias, void* out,
                           int V, int D, int Cout, void* stream);
int gank_concat_label_fwd(const void* a, const void* T, const int32_t* labels, void* y, int N, int HW, int C1, int C2, int V,
                          void* stream);
int gank_concat_label_bwd(const void* dy, void* da, float* de32, int N, int HW, int C1, int C2, void* stream);
/* ... and the same with the fan-out of the down-sampling block behind the concat (gan_cifar_resnet.py:286-289: its main path
 * reads y, its shortcut mean_pool2x2(y), resnet_block.py:53-64) folded in: fwd writes y [N,H,W,C1+C2] and y_pooled [N,H/2,W/2,.]
 * in one pass (pool2x2's arithmetic on the first C1 channels, the table row itself on the tiled half); bwd takes the two
 * branch gradients g_main [N,H,W,C] (or NULL) and g_pooled [N,H/2,W/2,C] and produces da / de32 of
 * dy = g_main + 0.25 unpool(g_pooled) without writing dy. */
int gank_concat_label_pool_fwd(const void* a, const void* T, const int32_t* labels, void* y, void* y_pooled, int N, int H, int W,
                               int C1, int C2, int V, void* stream);
int gank_concat_label_unpool_bwd(const void* g_main, const void* g_pooled, void* da, float* de32, int N, int H, int W, int C1, int C2,
                                 void* stream);
int gank_concat_label_unpool_bwd_factored(const void* g_main_c1, const void* g_pooled, void* da, float* de32, const float* de_add, int de_parts,
                                          const int32_t* labels, const int32_t* lists, int V, int N, int H, int W, int C1, int C2, void* stream);

/* ---- the spatially constant input channels of a 3x3 conv, factored out (round 5) ------------------------------------------
 * The critic tiles the label embedding over the 16x16 grid and concatenates it to its features in front of D.Block.2
 * (SNGAN/gan_cifar_resnet.py:276-284), so C2 = 128 of the 256 input channels of D.Block.2.Conv1 (:186-190; 35 % of the critic's
 * multiply-adds) hold one vector per sample, relu(T[label]) after the pre-activation.  Their contribution to the conv depends on a
 * sample only through its label and on a pixel only through its border class (3 row x 3 column classes: first / inner / last), so a
 * table [V][9][Cout] replaces half of the layer in all three passes -- exact algebra:
 *   gank_label_conv3x3_table    : bias_table[v][cls][co] = bias[co] + sum_{taps valid in cls} sum_c bf16(w[t][c0+c][co]) relu(T[v][c]);
 *                                 w is the WHOLE fp32 filter [3,3,Cin_total,Cout], the constant channels are c0 .. c0+C2-1;
 *                                 lists (optional, int32 [V][N+1], with labels [N]): row v = {count, the samples of label v ascending}
 *                                 -- what the backward entries walk;
 *   gank_img16_conv3x3_label_bias: the image-resident 16x16 conv on the OTHER channels (x [N,16,16,Cin], operands prepared with
 *                                 gank_prep_desc.cin_pitch = Cin_total) adding row (label of the sample, class of the pixel);
 *   gank_label_conv3x3_bwd      : from dy [N,H,W,Cout] and the lists: dw[t][c0+c][co] += sum_v relu(T[v][c]) Sl[v][t][co] and
 *                                 de_parts[t][v][c] = [T[v][c] > 0] sum_co bf16(w[t][c0+c][co]) Sl[v][t][co] -- the gradient of the tiled
 *                                 vector summed per LABEL (what gank_label_dense_bwd adds up anyway; gank_concat_label_unpool_bwd_factored
 *                                 adds it to the row of the label's first sample) --, Sl[v][t] = the sum of dy over the samples of label v
 *                                 and the pixels where tap t is valid (ws: gank_label_conv3x3_bwd_ws_floats; dy = NULL: ws already holds
 *                                 them -- gank_conv2d_wgrad_slabs_rows_tap_sums); at most 16 labels.  dw_feat_tmp (optional,
 *                                 contiguous [9][c0][Cout]): the other channels' filter gradient, accumulated there by an ordinary
 *                                 filter-gradient launch, is added into rows [0, c0) of dw and the buffer cleared. */
int gank_label_conv3x3_table(const float* w, int Cin_total, int c0, int C2, int Cout, const void* T, int V, const float* bias,
                             float* bias_table, const int32_t* labels, int N, int32_t* lists, void* stream);
/* gank_label_conv3x3_table + gank_concat_label_pool_fwd(y = NULL) in one launch (the pooled concatenation for the block's shortcut) */
int gank_label_conv3x3_table_pooled(const float* w, int Cin_total, int c0, int C2, int Cout, const void* T, int V, const float* bias,
                                    float* bias_table, const int32_t* labels, int N, int32_t* lists, const void* a, void* y_pooled,
                                    int H, int W, int C1, void* stream);
/* ... and the block's 1x1 shortcut conv on that pooled concatenation (gan_cifar_resnet.py:172-184 with resample='down': Conv2D 1x1 on the
 * mean-pooled input) in the same launch: one workgroup per sample pools into LDS and multiplies from there.  ws_f = the plain-conv
 * operand of the 1x1 filter (bf16 [Cs][ws_pitch], rows = output channels); shortcut [N, H/2, W/2, Cs] bf16.  (H/2) * (W/2) == 64,
 * Cs == 128, C1 + C2 <= 256 and a multiple of 16. */
int gank_label_conv3x3_table_pooled_shortcut(const float* w, int Cin_total, int c0, int C2, int Cout, const void* T, int V, const float* bias,
                                             float* bias_table, const int32_t* labels, int N, int32_t* lists, const void* a, void* y_pooled,
                                             int H, int W, int C1, const void* ws_f, int ws_pitch, const float* bias_s, int Cs,
                                             void* shortcut, void* stream);
int gank_img16_conv3x3_label_bias(const void* x, const void* w_rfrag, const float* bias_table, const int32_t* labels, int V, void* y,
                                  int N, int Cin, int Cout, int flags, void* stream);
/* the image-resident conv as the input gradient behind a fork whose other branch is a 2x2 mean pool: dx = relu_mask(conv(dy)) +
 * res_scale * unpool2x(g_pooled[..., :Cout]), g_pooled [N,8,8,res_pitch] the pooled branch's gradient -- the join of the two branch gradients
 * in the conv's epilogue (gank_concat_label_unpool_bwd_factored with da = NULL then computes the tiled vector's gradient alone) */
int gank_img16_conv3x3_dgrad_unpool(const void* x, const void* w_rfrag, const void* relu_ref, const void* g_pooled, int res_pitch,
                                    float res_scale, void* y, int N, int Cin, int Cout, void* stream);
/* the image-resident conv (here: the factored layer's input gradient, x = dy [N,16,16,Cin], w_rfrag = the feature half's dgrad operand,
 * relu_ref = the features) with gank_label_conv3x3_bwd's label-gradient launch as extra workgroups (tap_sums as left by
 * gank_conv2d_wgrad_slabs_rows_tap_sums; w [3,3,Cin_total,CoutW] and dw the whole filter and its gradient) */
int gank_img16_conv3x3_label_bwd(const void* x, const void* w_rfrag, const void* relu_ref, void* y, int N, int Cin, int Cout, int flags,
                                 const float* tap_sums, const void* T, int V, const float* w, int Cin_total, int c0, int C2, int CoutW,
                                 float* dw, float* de_parts, void* stream);
long gank_label_conv3x3_bwd_ws_floats(int N, int Cout);
int gank_label_conv3x3_bwd(const void* dy, const int32_t* lists, const void* T, int V, const float* w, int Cin_total, int c0, int C2,
                           int Cout, int N, int H, int W, float* dw, float* dw_feat_tmp, float* de_parts, float* ws, void* stream);
int gank_label_dense_bwd(const float* de32, const int32_t* labels, const float* table, const float* W, float* dW, float* dbias,
                         float* dtable, int N, int V, int D, int C2, void* stream);
/* gank_label_conv3x3_bwd's label-gradient launch (tap sums as left by gank_conv2d_wgrad_slabs_rows_tap_sums) with a TENTH part from extra
 * workgroups: de_parts [10][V][C2], part 9 = the gradient the tiled vector receives through the block's POOLED shortcut branch, summed per
 * label: sum over the samples of label v and the HWp pooled pixels of g_pooled[n][p][c0g + c] (g_pooled [N][HWp][pitch] bf16; each of the
 * four pixels under a pooled pixel receives 0.25 g, so a sample's sum is the sum of the pooled gradient).  With the feature channels joined
 * in gank_img16_conv3x3_dgrad_unpool, gank_concat_label_unpool_bwd_factored has nothing left to do: its launch is gone.
 * gank_label_dense_bwd_parts: gank_label_dense_bwd from rows that are summed per label already, dT[l] = sum_p rows[p][l] in part order. */
int gank_label_conv3x3_bwd_pooled(const float* tap_sums, const int32_t* lists, const void* T, int V, const float* w, int Cin_total, int c0,
                                  int C2, int Cout, int N, float* dw, float* dw_feat_tmp, float* de_parts, const void* g_pooled, int HWp,
                                  int pitch, int c0g, void* stream);
int gank_label_dense_bwd_parts(const float* de_parts_rows, int parts, const float* table, const float* W, float* dW, float* dbias,
                               float* dtable, int V, int D, int C2, void* stream);
/* gank_sum_slabs(jobs, count <= 12) + gank_label_conv3x3_bwd_pooled (without dw_feat_tmp) in ONE launch: the label gradients of a critic
 * update are independent of every slab and small, so they run as extra workgroups behind the summing ones. */
int gank_sum_slabs_label_bwd(const gank_slab_job* jobs, int count, const float* tap_sums, const int32_t* lists, const void* T, int V, const float* w,
                             int Cin_total, int c0, int C2, int Cout, int N, float* dw, float* de_parts, const void* g_pooled, int HWp, int pitch,
                             int c0g, void* stream);

/* tf.nn.embedding_lookup (common/ops/embedding.py:51) and its IndexedSlices gradient (dense, accumulated) */
int gank_embedding_fwd(const float* table, const int32_t* idx, void* y, int N, int D, int vocab, void* stream);
int gank_embedding_bwd(const void* dy, const int32_t* idx, float* dtable, int N, int D, int vocab, void* stream);

/* ---- losses (loss value fp32[1] and d loss/d logits in one launch) ------------------------------
 * dlogits: bf16, the gradient for an upstream gradient of 1 (loss.backward() on the loss itself);
 * dlogits_f32 (may be NULL): the same values unrounded, for gank_loss_grad_scale when the loss is a term of a
 * weighted sum (gen_cost + ACGAN_SCALE_G*xent, gan_cifar_resnet.py:476; ACGAN/train.py:119-121).
 * hinge_d: mean(relu(1-l[:n_real])) + mean(relu(1+l[n_real:]))   (gan_cifar_resnet.py:362-363,379-381)
 * hinge_g: -mean(l)                                               (gan_cifar_resnet.py:492)
 * softmax_xent: mean sparse softmax cross-entropy                 (gan_cifar_resnet.py:390-394) */
/* critic head in one launch: logits [M] = x [M,K] w [K] + b[0] (D.Output, gan_cifar_resnet.py:303-304), their hinge loss
 * (mode 0: hinge_d with the first n_real rows real, :379-381; mode 1: hinge_g, :492), dx [M,K] = d loss / d x (NULL: skip),
 * and the layer's weight / bias gradients ACCUMULATED into w_grad [K] / b_grad [1] (NULL: skip).  Same arithmetic as
 * gank_linear_fwd + gank_hinge_*_loss + gank_linear_bwd (bf16 logits, bf16 d loss / d logits). */
int gank_critic_head_hinge(const void* x, const float* w, const float* b, void* logits, float* loss, void* dx, float* w_grad,
                           float* b_grad, int M, int K, int n_real, int mode, void* stream);
/* ... with a static loss scale (a power of two): d loss / d logit is multiplied by loss_scale before it is rounded to 16 bits, so
 * dx, w_grad and b_grad are loss_scale x the true gradients -- the fp16 build's protection against underflow of the
 * activation gradients behind this layer; the optimiser divides it out again (hp.grad_scale of gank_adam_tf).  `loss` is unscaled. */
int gank_critic_head_hinge_scaled(const void* x, const float* w, const float* b, void* logits, float* loss, void* dx, float* w_grad,
                                  float* b_grad, int M, int K, int n_real, int mode, float loss_scale, void* stream);
/* the other branches of get_loss (common/misc.py:353-394), same conventions (critic kinds: the first n_real logits are real):
 * kind 0 LSGAN critic, 1 LSGAN generator, 2 sigmoid-cross-entropy critic (CGAN / Modified_MiniMax / MiniMax), 3 its
 * non-saturating generator (-log sigmoid(fake): CGAN, Modified_MiniMax), 4 the MiniMax generator (log(1 - sigmoid(fake))).
 * The SOFT_PLUS = True branches of SNGAN/gan_cifar_resnet.py:364-386,483-497: kind 5 'Goodfellow' critic (-softplus(log sigmoid(real)) -
 * softplus(log(1 - sigmoid(fake)))), 6 its generator (softplus(-log sigmoid(fake))), 7 'HINGE' critic (softplus(-min(0, -1 + real)) +
 * softplus(-min(0, -1 - fake)); tf.minimum passes the gradient where the second argument is strictly smaller); the 'WGAN' critic
 * with SOFT_PLUS is kind 2 and both remaining generators (softplus(-fake)) are kind 3. */
int gank_gan_pointwise_loss(const void* logits, float* loss, void* dlogits, float* dlogits_f32, int n, int n_real, int kind, void* stream);
int gank_hinge_d_loss(const void* logits, float* loss, void* dlogits, float* dlogits_f32, int n, int n_real, void* stream);
int gank_hinge_g_loss(const void* logits, float* loss, void* dlogits, float* dlogits_f32, int n, void* stream);
/* wgan_d: -mean(l[:n_real]) + mean(l[n_real:])  (common/misc.py:328-331 'WGAN', :337-352 'WGAN-GP' before its penalty) */
int gank_wgan_d_loss(const void* logits, float* loss, void* dlogits, float* dlogits_f32, int n, int n_real, void* stream);
int gank_softmax_xent(const void* logits, const int32_t* labels, float* loss, void* dlogits, float* dlogits_f32, int n, int classes, void* stream);
/* dlogits (bf16) = g[0] * dlogits_f32: the chain rule through `total = ... + g * loss` (tf.gradients of a scaled loss) */
int gank_loss_grad_scale(const float* dlogits_f32, const float* g, void* dlogits, long n, void* stream);

/* ---- ACGAN configuration (ACGAN/train.py:89-121; ACGAN/model.py:49-90) ---------------------------------------------
 * The WGAN-GP term (train.py:99-107) differentiates the critic's INPUT gradient with respect to the critic's weights.
 * Conv / dense / pooling / leaky-relu second derivatives compose from the first-order entry points above (the host layer
 * makes fprop / dgrad / wgrad mutually differentiable); train-mode batch norm needs its own second-order kernel:
 *   gank_bn_bwd_bwd: given ggI = dL/d(dx) of the first backward pass dx = BN'(dy; x, gamma, mean, invstd) over `rows` =
 *   N*H*W rows of C channels (stats = [mean[C], invstd[C]] as gank_cbn_fwd leaves them for one tower, gamma [C]):
 *   gI <- dL/dx, ggO <- dL/d(dy) (bf16 [rows,C]), gG (fp32 [C], may be NULL) += dL/dgamma; ws: fp32 scratch of 5*C.
 *   gank_bn_moving_update: tf.contrib.layers.batch_norm's moving statistics (decay, zero-debiased mean; normalization.py:
 *   10-22) for `groups` towers in one launch; stats [groups][2][C], count = rows per tower (for the unbiased variance).
 *   gank_gp_loss: loss <- lambda * mean_n (sqrt(sum_d g[n,d]^2 + 1e-10) - 1)^2, dgrad <- its derivative (fp32 [N,D]: scaled by
 *   the upstream gradient and rounded once by gank_loss_grad_scale); ws fp32 [N].
 *   gank_lerp_rows: out[n,:] = real[n,:] + alpha[n] * (fake[n,:] - real[n,:]).
 *   gank_sum_hw / gank_bcast_hw: y[n,c] = scale * sum_hw x[n,hw,c] (tf.reduce_mean(axis=[1,2]), model.py:72) and its adjoint.
 *   gank_rng_uniform_f32: U[0,1) from the device RNG state (advances it). */
int gank_bn_bwd_bwd(const void* ggI, const void* dy, const void* x, const float* gamma, const float* stats, void* gI, void* ggO,
                    float* gG, float* ws, long rows, int C, void* stream);
int gank_bn_moving_update(const float* stats, float* moving_mean, float* moving_var, float* biased, float* local_step, int C,
                          int groups, long count, float decay, float eps, void* stream);
int gank_gp_loss(const void* grad, float* loss, void* dgrad, float* ws, int N, long D, float lambda, void* stream);
int gank_lerp_rows(const void* real, const void* fake, const float* alpha, void* out, int N, long D, void* stream);
int gank_sum_hw(const void* x, void* y, int N, int HW, int C, float scale, void* stream);
int gank_bcast_hw(const void* g, void* y, int N, int HW, int C, float scale, void* stream);
int gank_rng_uniform_f32(float* y, long n, uint64_t* rng_state, void* stream);

/* ---- small operators of the PGGAN (config 4) and Pix2Pix (config 5) paths --------------------------------------------
 * axpby: y = alpha*a + beta*b (b may be NULL): the fade-in blend (1-alpha)*toRGB2 + alpha*toRGB1 (PGGAN/model_nvidia.py:116,206).
 * minibatch_std (model_nvidia.py:20-29): y [B,HW,C+1] = concat(x, mean_{hwc} sqrt(var_batch(x)+1e-8)); ws fp32 [HW*C + 2] is
 *   kept from fwd to bwd; bwd: dx from dy [B,HW,C+1].
 * resize_bilinear: tf.image.resize_images (BILINEAR, align_corners=False; PGGAN/train.py:88-92).
 * concat_channels / split_channels: tf.concat(axis=3) of two NHWC tensors and its gradient (Pix2Pix/networks.py U-Net skips).
 * l1_loss: mean |a-b| (Pix2Pix/train.py:510-512); dl32 fp32 = d loss / d a (gank_loss_grad_scale rounds it once); ws fp32 [1024].
 * dropout: tf.nn.dropout(x, keep) with a byte mask from the device RNG (networks.py decoder), and its gradient. */
int gank_axpby_bf16(const void* a, const void* b, float alpha, float beta, void* y, long n, void* stream);
/* fade-in blend with the weight in device memory (graph replay): mode 0: y = (1-alpha[0]) a + alpha[0] b; 1: (1-alpha[0]) a; 2: alpha[0] a */
int gank_blend_dev(const void* a, const void* b, const float* alpha, void* y, long n, int mode, void* stream);
int gank_minibatch_std_fwd(const void* x, void* y, float* ws, int B, int HW, int C, void* stream);
int gank_minibatch_std_bwd(const void* dy, const void* x, float* ws, void* dx, int B, int HW, int C, void* stream);
int gank_resize_bilinear(const void* x, void* y, int N, int Hi, int Wi, int Ho, int Wo, int C, void* stream);
int gank_concat_channels(const void* a, const void* b, void* y, long pixels, int Ca, int Cb, void* stream);
/* ---- Inception-v3 classifier of the Inception-score harness (common/inception/inception_score.py:29-47): its pooling layers
 * and the branch concat.  pool2d: x [N,H,W,C] -> y[..., c_off : c_off + C] of [N,Ho,Wo,Cy]; mode 0 max, 1 average over the
 * in-image elements of a k x k window (tf.nn.avg_pool SAME excludes the padding), stride 1 | 2, `pad` leading rows / columns.
 * relu_to_channels: y[p, c_off : c_off + C] = relu(x[p, :]) -- a branch's last ReLU written into the block's concat output. */
int gank_pool2d(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int mode, int Cy, int c_off,
                void* stream);
int gank_relu_to_channels(const void* x, void* y, long pixels, int C, int Cy, int c_off, void* stream);
/* im2col of a narrow-channel image: y [N,Ho,Wo,Kpad] bf16, column tap*Cin + c = x[n, oy*stride - pad + ky, ox*stride - pad + kx, c]
 * (zeros outside the image and for columns >= k*k*Cin; Kpad % 8 == 0).  Turns the filter gradient of Pix2Pix's 4x4 stride-2 layers on
 * 3- / 6-channel inputs (networks.py:335-342, :474-486) into the filter gradient of a 1x1 conv (gank_conv2d_wgrad, ksize 1, Cin = Kpad). */
int gank_im2col_narrow(const void* x, void* y, int N, int Hin, int Win, int Cin, int Ho, int Wo, int ksize, int stride, int pad,
                       int Kpad, void* stream);
/* A conv with <= 4 output channels behind a 2x NN-upsample (Pix2Pix decoder_1: relu, upsample, 4x4 SAME, 3 channels, tanh;
 * networks.py:424-452) = a 1x1 conv at low resolution to Z [N,h,w,Zc] (column t*Cout + co = tap t's partial output; any MFMA 1x1
 * kernel) followed by gank_tap_gather_up2: y [N,2h,2w,Cout] = tanh(bias + the taps' partials gathered from Z).  Backward:
 * gank_tap_scatter_up2 turns g [N,2h,2w,Cout] into col [N,h,w,Zc] (the gradient of Z); input and filter gradients are the 1x1
 * conv's (gank_conv2d_dgrad / gank_conv2d_wgrad on col).  pad = leading zero rows / columns of the k x k window on the upsampled grid. */
int gank_tap_gather_up2(const void* Z, const float* bias, void* y, int N, int h, int w, int ksize, int pad, int Cout, int Zc,
                        int tanh_out, void* stream);
int gank_tap_scatter_up2(const void* g, void* col, int N, int h, int w, int ksize, int pad, int Cout, int Zc, void* stream);
/* depth_to_space / space_to_depth, block 2, channel order (a, b, c): y [N,2h,2w,C] <-> x [N,h,w,4C], y[n,2i+a,2j+b,c] = x[n,i,j,(2a+b)C+c]
 * (tf.depth_to_space of gan_cifar_resnet.py:145 in NHWC; C % 8 == 0).  Behind a phase-stacked conv: NN-upsample + 4x4 SAME (Pix2Pix
 * decoders, networks.py:424-439) = one 3x3 conv at low resolution to 4 C channels (36 instead of 64 taps per 2x2 outputs) + this. */
int gank_depth_to_space2(const void* x, void* y, int N, int h, int w, int C, void* stream);
int gank_space_to_depth2(const void* y, void* x, int N, int h, int w, int C, void* stream);
int gank_split_channels(const void* y, void* a, void* b, long pixels, int Ca, int Cb, void* stream);
int gank_l1_loss(const void* a, const void* b, float* loss, float* dl32, float* ws, long n, void* stream);
int gank_dropout_fwd(const void* x, void* y, uint8_t* mask, long n, float keep, uint64_t* rng_state, void* stream);
int gank_dropout_bwd(const void* dy, const uint8_t* mask, void* dx, long n, float keep, void* stream);

/* ---- tf.train.AdamOptimizer (gan_cifar_resnet.py:521-526), one launch over a flat buffer ---------
 * All step state lives on the device so a captured update replays without host traffic:
 * hp (float[8]) = {lr, beta1, beta2, eps, grad_scale, decay_on, <ticket word, keep 0>, -}; t_state[0] = updates applied
 * so far (incremented by this call, inside the same launch); iteration[0] = the `_iteration` feed (:320) driving the LR
 * decay (:454-459), may be NULL.  lr_t = lr*decay*sqrt(1-b2^t)/(1-b1^t); g is multiplied by grad_scale first (1/world_size
 * after a sum all-reduce).  zero_n > 0: the first zero_n floats of g (>= n: a scratch tail may follow the gradients) are
 * cleared once consumed, so the next backward pass needs no fill launch.  gank_counter_add advances a device counter. */
int gank_adam_tf(float* p, float* g, float* m, float* v, float* hp, int64_t* t_state,
                 const int64_t* iteration, long n, long zero_n, void* stream);
/* ... and counters for a loss-scaled run: health[0] += gradients that are not finite (the scale overflowed the 16-bit range
 * somewhere behind them), health[1] += gradients that are exactly zero (compare with an unscaled bf16 run: the excess is
 * underflow).  Cumulative; the caller clears them. */
int gank_adam_tf_health(float* p, float* g, float* m, float* v, float* hp, int64_t* t_state, const int64_t* iteration, long n,
                        long zero_n, uint64_t* health, void* stream);
int gank_counter_add(int64_t* counter, int64_t inc, void* stream);

/* ---- input pipeline (gan_cifar_resnet.py:334-337) and graph-safe RNG -----------------------------
 * preprocess: uint8 CHW-planar rows [B,3072] -> bf16 HWC rows: 2*(x/256-.5) + U[0,1/128).
 * RNG is counter based (Philox4x32-10): `state` (device, uint64[2] = {seed, offset}); each call
 * consumes and advances the offset on the device, so a captured graph draws fresh numbers per replay
 * (tf.random_normal :240, tf.random_uniform :335,:467). */
int gank_preprocess_real(const uint8_t* data, void* y, uint64_t* rng_state, int B, void* stream);
/* One launch that lays out a critic update's inputs from slot *slot of the iteration's feed ring (the per-update
 * feed of gan_cifar_resnet.py:616-620 + :334-338 + the concat of :361): both [2B,3072] bf16 = {preprocessed
 * real_all[slot] (same arithmetic and random numbers as gank_preprocess_real), fake_all[slot]}, labels2 [2B] =
 * labels_all[slot] twice; then *slot = (*slot + 1) % n_slots and the RNG offset advances by one.  real_all
 * uint8 [n_slots,B,3072], labels_all int32 [n_slots,B], fake_all bf16 [n_slots,B,3072]; done_counter: one zeroed
 * uint32 of scratch that the call leaves zero. */
int gank_critic_feed(const uint8_t* real_all, const int32_t* labels_all, const void* fake_all, void* both,
                     int32_t* labels2, int32_t* slot, uint64_t* rng_state, uint32_t* done_counter, int B, int n_slots,
                     void* stream);
int gank_rng_normal_bf16(void* y, long n, uint64_t* rng_state, void* stream);
int gank_rng_labels(int32_t* y, long n, int n_labels, uint64_t* rng_state, void* stream);
/* The three launches in front of a generator pass as one: labels (int32 [n_lab]; NULL with n_lab = 0: no draw) as gank_rng_labels
 * (gan_cifar_resnet.py:467), noise (bf16 [n]) as gank_rng_normal_bf16 (:240) drawn at the next stream offset, zero_buf (fp32 [zero_n],
 * 16-byte aligned; NULL with 0) cleared (the pass's statistics arena).  Outputs and the stream offset are those of the separate calls. */
int gank_generator_feed(int32_t* labels, long n_lab, int n_labels, void* noise, long n, float* zero_buf, long zero_n, uint64_t* rng_state,
                        void* stream);

/* ---- opt-in per-kernel timing with HIP events on the launch stream (bench.py roofline leg) ------- */
int gank_prof_enable(int on);
int gank_prof_reset(void);
/* fills up to `cap` records {launches, total_ms, total_flops} for kernel family `family`
 * (0 = conv_fprop/dgrad igemm, 1 = conv_wgrad); synchronises.  Returns number of launches. */
int gank_prof_collect(int family, double* total_ms, double* total_flops);
/* per-kernel breakdown of a family: entry `index` (longest total time first) -> kernel symbol name (as rocprofv3
 * prints it, without the argument list), launches, total ms, FLOPs, algorithmic bytes; returns 0 past the end */
int gank_prof_kernel_stats(int family, int index, char* name, int name_cap, int* launches, double* total_ms,
                           double* total_flops, double* total_bytes);
/* sum of the algorithmic bytes (operands read once + result written once) of the recorded launches of `family` */
double gank_prof_bytes(int family);
/* average milliseconds an event pair around an EMPTY kernel reads (n launches): the fixed cost inside every record */
double gank_prof_calibrate(int n, void* stream);

/* debug: what ds_read_b64_tr_b16 delivers for a known LDS image (layout self-check) */
int gank_debug_tr_probe(int32_t* out, void* stream);

/* ---- weight-side transforms of the Pix2Pix / PGGAN routes (fp32; no framework arithmetic on the product path) ----
 * gank_phase_stack4: the stacked 3x3 filter [3,3,Cin,4*Cout] (output channels (a, b, co)) of "NN-upsample + 4x4 SAME conv"
 *   (Pix2Pix/networks.py:420-445 decoders) from the 4x4 filter [4,4,Cin,Cout]; adjoint != 0: the 4x4 filter's gradient
 *   ACCUMULATED from the stacked filter's (w4 is then written, w3 read).
 * gank_pad_rows: `rows` rows of w_in elements -> rows of w_out elements with zeros behind (2- or 4-byte elements); adjoint:
 *   the first w_in elements of every wide row back (fp32 accumulates, 16-bit overwrites) -- zero channels behind a filter's Cin
 *   or an activation's C (PGGAN's 513-channel conv, model_nvidia.py:128-129).
 * gank_tile_rows: b [n] -> [reps][n]; adjoint: b[c] += sum_j g[j][c].
 * gank_fewout_pack: filter [k,k,Cin,Cout<=4] -> [Cin][Zc] with column t*Cout+co (zeros behind); adjoint: accumulated back. */
int gank_phase_stack4(const float* w4, float* w3, int Cin, int Cout, int adjoint, void* stream);
int gank_pad_rows(const void* src, void* dst, long rows, int w_in, int w_out, int elem_bytes, int adjoint, void* stream);
int gank_tile_rows(const float* src, float* dst, int reps, int n, int adjoint, void* stream);
int gank_fewout_pack(const float* src, float* dst, int ksize, int Cin, int Cout, int Zc, int adjoint, void* stream);
/* fp32 zero fill (scratch gradients of the routes above) */
int gank_zero_f32(float* p, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GANK_H */
