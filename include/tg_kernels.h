/*
 * tg_kernels.h — C ABI of libtg_hip.so: the hand-written gfx950 (MI355X / CDNA4) kernels behind the
 * Triple-GAN three-player training step.
 *
 * The reference (Wenyuan-Vincent-Li/Tensorflow-Implementation-of-Triple-GAN) has no FFI of its own: its
 * layer primitives call TensorFlow-1.x ops directly.  Each entry point below therefore replaces the TF
 * op(s) cited beside it (paths relative to the reference root) — the "lower side" of the drop-in boundary
 * of SURVEY.md §8b.  The "upper side" (Model/nn.py, Model/model_base.py, Training/ of the package)
 * keeps the reference's Python names and binds these symbols with ctypes.
 *
 * Contract
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked host.
 *  - the caller owns every buffer (inputs, outputs, scratch); the library never allocates device memory.
 *  - every call is asynchronous on the hipStream_t passed as `stream` (void*, may be NULL = default
 *    stream) and is legal inside hipStream capture (no allocation, no synchronisation).
 *  - returns 0 (TG_OK) or a negative tg_status; never throws.  tg_last_error_string() is thread-local.
 *  - activations NHWC fp32, conv filters HWIO, transposed-conv filters [kh,kw,Cout,Cin], dense [in,out]
 *    (the reference's variable layouts: Model/nn.py:477,530, Model/modle_base.py:96,141-142).
 *  - "channel padding": activation tensors that feed the MFMA kernels carry a channel STRIDE that is a
 *    multiple of 32 (ld); channels >= the logical count hold zeros.
 */
#ifndef TG_KERNELS_H
#define TG_KERNELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum tg_status {
  TG_OK = 0,
  TG_ERR_INVALID = -1,   /* bad argument / shape the kernel does not support */
  TG_ERR_HIP = -2,       /* a HIP runtime call failed; see tg_last_error_string() */
  TG_ERR_STATE = -3      /* call not legal in the current state (e.g. graph not captured) */
} tg_status;

enum { TG_ACT_NONE = 0, TG_ACT_LRELU = 1, TG_ACT_RELU = 2, TG_ACT_TANH = 3, TG_ACT_SIGMOID = 4, TG_ACT_SOFTPLUS = 5 };

/* ---- runtime ------------------------------------------------------------------------------------ */
int tg_version(void);
const char* tg_last_error_string(void);
int tg_device_count(void);                          /* hipGetDeviceCount; <0 on error */

/* hipGraph capture of a launch sequence on `stream` (replaces the TF session's cached sub-graph
 * execution, Training/Train_goodGAN.py:266-276: three sess.run calls per iteration). */
int tg_graph_begin_capture(void* stream);
int tg_graph_end_capture(void* stream, void** graph_exec_out);
int tg_graph_launch(void* graph_exec, void* stream);
int tg_graph_destroy(void* graph_exec);

/* per-kernel-class timing with HIP events on the launch stream (eager mode only). */
int tg_prof_enable(int on);
int tg_prof_reset(void);
int tg_prof_num_classes(void);
const char* tg_prof_class_name(int cls);
/* host outputs: total milliseconds, launch count, algorithmic flops and bytes of class `cls`
 * since the last reset; synchronises the recorded events. */
int tg_prof_collect(int cls, double* ms, int64_t* launches, double* flops, double* bytes);
/* one CSV row per recorded launch (class, ms, executed GFLOP, GB, geometry) — host path. */
int tg_prof_dump(const char* path);

/* ---- implicit-GEMM convolution family (MFMA v_mfma_f32_32x32x2_f32) ----------------------------- */
#define TG_MAX_TAPS 25
typedef struct tg_igemm_desc {
  int32_t n_img;                 /* images in the batch */
  int32_t h_in, w_in, ld_in;     /* gathered tensor [n_img,h_in,w_in,ld_in]; ld_in % 32 == 0 = reduction channels */
  int32_t h_v, w_v;              /* virtual output grid per image; GEMM rows M = n_img*h_v*w_v */
  int32_t s_y, s_x;              /* gathered pixel of virtual pixel v, tap t: (v_y*s_y + dy[t], v_x*s_x + dx[t]) */
  int32_t h_out, w_out, ld_out;  /* output tensor [n_img,h_out,w_out,ld_out] */
  int32_t os_y, os_x, oo_y, oo_x;/* output pixel of virtual pixel: (v_y*os_y + oo_y, v_x*os_x + oo_x) */
  int32_t c_out;                 /* GEMM columns N (multiple of 32) */
  int32_t n_store;               /* columns actually stored (<= c_out, <= ld_out) */
  int32_t n_taps;
  int8_t dy[TG_MAX_TAPS], dx[TG_MAX_TAPS];
  int16_t tapw[TG_MAX_TAPS];     /* weight tap index used for tap t */
  int64_t w_sn, w_st;            /* weight element (n,t,c) at n*w_sn + tapw[t]*w_st + c (c contiguous) */
  int32_t act;                   /* TG_ACT_* applied after +bias */
  float alpha;                   /* leaky slope */
  int32_t n_group;               /* 0: off.  > 0 (tg_igemm_f32 / _bf16 only): GEMM column n is channel n % n_group of output pixel
                                  * (v_y*os_y + oo_y + g / os_x, v_x*os_x + oo_x + g % os_x), g = n / n_group; stored if channel < n_store;
                                  * bias indexed by channel.  A stride-2 transposed conv then is ONE balanced 3x3 problem with
                                  * 4 * n_group columns (zero weights for the taps a parity does not have) instead of four unequal ones. */
} tg_igemm_desc;

/* Scratch of the MFMA launches (every tg_igemm_* entry point takes `scratch, scratch_bytes` in front of `stream`): device memory the
 * launch may overwrite, OWNED BY THE CALLER — the library never allocates device memory — 16-byte aligned, free for reuse once the launch
 * has completed on `stream` (launches on concurrent streams need their own).  tg_igemm_workspace_bytes answers, on the host, how much a
 * launch of these descriptors can use (seg_rows / nseg as passed to tg_igemm_colsum_* / _actsum_*, nseg = 0 otherwise; bf16: the
 * *_bf16 entry point; < 0 on a bad descriptor).  Two users:
 *   - the generic kernel cuts the tiles of the last partial round of resident workgroups (1 128 tiles on 512 slots), of an under-filled
 *     launch (225 tiles) or of the long sub-problems of a stride-2 launch (9 / 6 / 6 / 4 taps) along K; the partial sums pass through the
 *     scratch and a second launch adds them up in a fixed order (deterministic) and runs the epilogue.  OPTIONAL: with scratch == NULL or
 *     fewer bytes than the answer the launch runs one workgroup per tile (the round-2 schedule) — same result to fp32 summation order;
 *   - the halo-tiled 3x3 kernel packs the bf16 filter there and reads it back by LDS-DMA (tg_igemm_*_bf16 on a layer of that kernel's
 *     shape).  REQUIRED: such a launch with less scratch than the answer fails with TG_ERR_INVALID — it does not take another kernel. */
int64_t tg_igemm_workspace_bytes(const tg_igemm_desc* descs, int n_desc, const int32_t* seg_rows, int nseg, int bf16);

/* out[p,n] = act( sum_t sum_c in[pix(p,t),c] * w[n,t,c] + bias[n] ).
 * Replaces tf.nn.conv2d / tf.layers.conv2d (Model/nn.py:504, Model/modle_base.py:102,161), their
 * input-gradient, tf.layers.conv2d_transpose (Model/modle_base.py:250; one launch per output parity),
 * tf.matmul / tf.layers.dense (Model/nn.py:553, Model/modle_base.py:40) and the ZCA matmul
 * (Model/Good_GAN_cifar10.py:296).  bias may be NULL. */
int tg_igemm_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, void* scratch, int64_t scratch_bytes,
                 void* stream);
/* tg_igemm_f32 for a layer whose output goes straight into _conv_cond_concat (Model/modle_base.py:239-244, the discriminators' conv -> leaky
 * relu -> concat(labels) pairs, Model/Good_GAN_cifar10.py:66-91): `out` IS the concatenated tensor [n_img,h_out,w_out,ld_out] — the launch stores
 * its n_store channels as usual and ALSO writes channels [n_store, n_store + n_labels) of every output pixel = labels[image][0..n_labels) and zeros
 * from there up to ld_out, so that no separate concat launch (tg_cond_concat_f32) reads and re-writes the activation.  One ungrouped sub-problem
 * with os = 1, 4 | n_store, 4 | ld_out >= n_store + n_labels; labels: [n_img][n_labels] device floats.  Always the generic kernel. */
int tg_igemm_labels_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, const float* labels, int n_labels, float* out,
                        void* scratch, int64_t scratch_bytes, void* stream);
int tg_igemm_labels_bf16(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, const float* labels, int n_labels, float* out,
                         void* scratch, int64_t scratch_bytes, void* stream);
/* up to 4 sub-problems in ONE launch (the output parities of a stride-2 transposed conv / strided-conv input-gradient):
 * same buffers, M, N and gathered tensor; each descriptor brings its own taps and output offsets. */
int tg_igemm_multi_f32(const tg_igemm_desc* descs, int n_desc, const float* in, const float* w, const float* bias, float* out,
                       void* scratch, int64_t scratch_bytes, void* stream);

/* tg_igemm_f32 (no bias, no activation) that also accumulates the per-(application segment, channel) sums of its output
 * into colsum[nseg][c_out] (fp64; zeroed by the call unless colsum_zeroed != 0 — the caller then guarantees zeros, e.g. one
 * memset over an arena holding the accumulators of a whole solver run instead of one memset per layer): the tf.nn.moments pass of mean_only_batch_norm_impl
 * (Model/nn.py:171-175) fused into the convolution.  seg_rows: HOST array, at most 8 entries; a tile may straddle one
 * application boundary, so every entry must be at least the row count of some tile that divides c_out (32 ... 128). */
int tg_igemm_colsum_f32(const tg_igemm_desc* d, const float* in, const float* w, float* out, const int32_t* seg_rows, int nseg,
                        double* colsum, int colsum_zeroed, void* scratch, int64_t scratch_bytes, void* stream);

/* Input gradient of a convolution whose INPUT was produced by a mean-only-BN layer with nonlinearity `act`: the same launch as
 * tg_igemm_colsum_f32 (d = the input-gradient geometry, in = dpre of this conv, w = padded HWIO filter), but every output element is
 * multiplied by act'(yact) (yact = that producing layer's activated output, same shape and channel stride as `out`) before it is
 * stored and summed:  out = dx * act'(yact),  colsum[seg][k] = sum over the segment's rows of out[:,k].
 * Together with tg_mobn_center_f32 this replaces tg_mobn_bwd_f32 for that producing layer (its gradient never makes a separate
 * statistics pass). */
int tg_igemm_actsum_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* yact, int act, float alpha, float* out,
                        const int32_t* seg_rows, int nseg, double* colsum, int colsum_zeroed, void* scratch, int64_t scratch_bytes, void* stream);
int tg_igemm_actsum_bf16(const tg_igemm_desc* d, const float* in, const float* w, const float* yact, int act, float alpha, float* out,
                         const int32_t* seg_rows, int nseg, double* colsum, int colsum_zeroed, void* scratch, int64_t scratch_bytes, void* stream);

/* Convolution + bias + activation (none / relu / leaky relu) whose output feeds a training-mode batch norm (the SVHN / MNIST classifier's
 * conv -> leaky relu -> BN, Model/Good_GAN.py:249-350): the launch of tg_igemm_f32 / _bf16 that ALSO adds the per-(application segment,
 * channel) sum and sum of squares of its stored output into replica 0 of the batch norm's statistics buffer — `sums` = the
 * [8][nseg][2][c_out] fp64 buffer of tg_bn_train_f32, zeroed by the call (all replicas) unless sums_zeroed — so that the batch norm
 * needs no statistics pass of its own: follow with tg_bn_train_apply_f32.  seg_rows as for tg_igemm_colsum_f32. */
int tg_igemm_bnstat_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, const int32_t* seg_rows, int nseg,
                        double* sums, int sums_zeroed, void* scratch, int64_t scratch_bytes, void* stream);
int tg_igemm_bnstat_bf16(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, const int32_t* seg_rows, int nseg,
                         double* sums, int sums_zeroed, void* scratch, int64_t scratch_bytes, void* stream);

/* A convolution-shaped launch (in practice: the input gradient of the NEXT convolution) whose output dy is the gradient a training-mode batch
 * norm's backward pass consumes: the launch of tg_igemm_f32 / _bf16 (no bias, no activation) that ALSO adds that pass's two statistics —
 * S0 = sum dy and S1 = sum dy * x per (application segment, channel), x = the batch norm's INPUT, same shape and channel stride as `out` — into
 * replica 0 of `sums`, the [8][nseg][2][c_out] fp64 buffer of tg_bn_train_bwd_f32 (zeroed by the call unless sums_zeroed).  Follow with
 * tg_bn_train_bwd_f32 / tg_bn_train_bwd_act_f32 and sums_zeroed = 2 ("the sums are given"): its statistics launch over dy and x disappears.
 * Only valid when this launch is the ONLY contribution to dy.  seg_rows as for tg_igemm_colsum_f32. */
int tg_igemm_bnbwdstat_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* x, float* out, const int32_t* seg_rows, int nseg,
                           double* sums, int sums_zeroed, void* scratch, int64_t scratch_bytes, void* stream);
int tg_igemm_bnbwdstat_bf16(const tg_igemm_desc* d, const float* in, const float* w, const float* x, float* out, const int32_t* seg_rows, int nseg,
                            double* sums, int sums_zeroed, void* scratch, int64_t scratch_bytes, void* stream);

/* filter gradient, split over `n_split` pixel ranges:
 * slab[s][t][c][n] = sum_{p in split s} in[pix(p,t),c] * dout[p,n]   (c < ld_in, n < c_out).
 * `dout` is read through (h_out,w_out,ld_out,os,oo) exactly as tg_igemm_f32 writes `out`.
 * Replaces Conv2DBackpropFilter / MatMul-grad emitted by optimizer.minimize (Training/train_base.py:65).
 * slab holds n_split*n_taps*ld_in*c_out floats; deterministic (no atomics). */
int tg_wgrad_f32(const tg_igemm_desc* d, const float* in, const float* dout, float* slab, int n_split, void* stream);

/* bf16 MFMA variants of the four launches above (BASELINE.json configs[3], "bf16 MFMA conv path"): identical arguments; tensors stay
 * fp32 in HBM; every MFMA operand (gathered activation, filter, output gradient) is rounded to bf16 (round-to-nearest-even) inside
 * the kernel and the products accumulate in fp32 (v_mfma_f32_32x32x16_bf16).  Bias, activation, statistics and the stored result are
 * fp32. */
int tg_igemm_bf16(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, void* scratch, int64_t scratch_bytes,
                  void* stream);
int tg_igemm_multi_bf16(const tg_igemm_desc* descs, int n_desc, const float* in, const float* w, const float* bias, float* out,
                        void* scratch, int64_t scratch_bytes, void* stream);
int tg_igemm_colsum_bf16(const tg_igemm_desc* d, const float* in, const float* w, float* out, const int32_t* seg_rows, int nseg,
                         double* colsum, int colsum_zeroed, void* scratch, int64_t scratch_bytes, void* stream);
int tg_wgrad_bf16(const tg_igemm_desc* d, const float* in, const float* dout, float* slab, int n_split, void* stream);

/* 3x3 / stride 1 / SAME convolution of FEW input channels (c_in <= 16; the discriminators' first layer: 3 image + 10 label channels -> 32,
 * Model/Good_GAN_cifar10.py:63-66, Model/Good_GAN.py:129-132; tf.layers.conv2d, Model/modle_base.py:157-168) as K-PACKED fp32 MFMA products:
 * the contraction index is the (tap, channel) pair, K = 9 c_in, instead of taps x 32 padded channels (csrc/packed_conv.hip).
 * x: [n, h, w, ld_x] (c_in channels used); kernel: the [3,3,Cin,Cout] variable itself (no preparation launch); c_out = 32 or 64; 16 | w,
 * w <= 32, 4 | h.
 *   fwd:   y[n,i,j,co] = act(sum x[n,i+ky-1,j+kx-1,ci] W[ky,kx,ci,co] + bias[co]), act in {none, relu, leaky relu}; channels [c_out, ld_y) of
 *          y are written too: labels[n][0 .. lab_n) (labels != NULL: the conv -> cond_concat pair in one launch, as tg_igemm_labels_*), zeros behind.
 *   wgrad: dw[3,3,Cin,Cout] = sum_{n,i,j} x[n,i+ky-1,j+kx-1,ci] dy[n,i,j,co] — the variable's layout; dy: [n, h, w, ld_dy] (4 | ld_dy);
 *          workspace: tg_conv3x3_packed_wgrad_workspace_bytes bytes of caller-owned scratch (per-block partial sums, reduced in a fixed order).
 * tg_conv3x3_packed_supported: 1 when the shape is served (else use tg_igemm_* / tg_wgrad_*). */
int tg_conv3x3_packed_supported(int n, int h, int w, int c_in, int c_out);
int64_t tg_conv3x3_packed_wgrad_workspace_bytes(int n, int h, int w, int c_in, int c_out);
int tg_conv3x3_packed_fwd_f32(const float* x, int ld_x, int c_in, const float* kernel, const float* bias, int act, float alpha, const float* labels,
                              int lab_n, float* y, int ld_y, int n, int h, int w, int c_out, void* stream);
int tg_conv3x3_packed_wgrad_f32(const float* x, int ld_x, int c_in, const float* dy, int ld_dy, int n, int h, int w, int c_out, float* workspace,
                                float* dw, void* stream);

/* Backward pass of a 5x5 / stride-2 / 'same' transposed convolution with c_out <= 4 output channels (the generator's image layer,
 * Model/Good_GAN_cifar10.py:55-57, Model/modle_base.py:246-259) as K-PACKED fp32 MFMA products (contraction index = the (tap, channel)
 * pair: 75 for three channels) — the generic tiles pad 3 channels to 32 and do ten times the layer's arithmetic.  dy: gradient at the layer's pre-activation output [n, 2h, 2w, ld_dy] (c_out channels used); x: the layer input
 * [n, h, w, ld_x] (ci_p = channel-padded width, a multiple of 32, <= 256; 16 | w, w <= 32, 4 | h).
 *   dgrad: dx[n,i,j,ci] = sum_{ky,kx,co} dy[n, 2i+ky-1, 2j+kx-1, co] * W[ky,kx,co,ci], W = the [5,5,Cout,Cin] variable itself, times
 *          scale_a[co] when not NULL (weight norm: tg_wn_scale_tab_f32); all ci_p channels of dx are written (zeros beyond c_in).
 *   wgrad: dw[25][c_out][c_in] = sum_{n,i,j} dy[n, 2i+ky-1, 2j+kx-1, co] * x[n,i,j,ci] — the layout of the [5,5,Cout,Cin] variable;
 *          workspace: tg_deconv5x5s2_narrow_wgrad_workspace_bytes bytes of caller-owned scratch (per-block partial sums, reduced in a
 *          fixed order).  tg_deconv5x5s2_narrow_supported: 1 when the shape is served (else use the tg_igemm_* / tg_wgrad_* path). */
int tg_deconv5x5s2_narrow_supported(int n, int h, int w, int c_out, int ci_p);
int64_t tg_deconv5x5s2_narrow_wgrad_workspace_bytes(int n, int h, int w, int c_out, int ci_p);
int tg_deconv5x5s2_narrow_dgrad_f32(const float* dy, int ld_dy, const float* kernel, const float* scale_a, int n, int h, int w, int c_out, int c_in,
                                    int ci_p, float* dx, int ld_dx, void* stream);
int tg_deconv5x5s2_narrow_wgrad_f32(const float* dy, int ld_dy, const float* x, int ld_x, int n, int h, int w, int c_out, int c_in, int ci_p,
                                    float* workspace, float* dw, void* stream);

/* ---- descriptor builders and workspace sizes (host code, no device work) -------------------------------------------------------
 * Every conv-like op of the hot path as tg_igemm_desc(s), with TensorFlow's padding arithmetic (SAME: out = ceil(in/s), total =
 * max((out-1)*s + k - in, 0), before = total/2, the extra pixel after; VALID: none) — so that a host binding carries no geometry code
 * of its own.  Conventions: ld_* = channel strides (multiples of 32 for gathered tensors); ld_out <= 0 means "c_out", n_store < 0 means
 * "c_out"; pad_same: 1 = 'SAME', 0 = 'VALID'; act / alpha: TG_ACT_* fused after +bias.  Filter layouts the descriptors expect:
 *   conv2d fwd      OTI   [c_out][k*k][ld_in]        (tg_filter_prep_f32: dst_tr with tr_sb = k*k*ld_in, tr_st = ld_in)
 *   conv2d dgrad    HWIO  [k*k][c_in_pad][ld_dy]     (dst_same)                       — up to 4 descriptors (input parities of a stride-2 conv)
 *   deconv fwd      [25][c_out_pad][ld_in] (dst_same of the [kh,kw,Cout,Cin] filter)  — 4 descriptors (output parities), or the merged form
 *   deconv dgrad    [25][c_in_pad][ld_dy] (dst_tr, per-tap transpose)
 * The *_wgrad descriptors are what tg_wgrad_f32 takes: `in` = the layer input (conv) or dy (transposed conv), `dout` = dy (conv) or the
 * layer input (transposed conv). */
int tg_conv2d_desc_fwd(int n, int h, int w, int ld_in, int c_out, int k, int stride, int pad_same, int ld_out, int n_store, int act, float alpha,
                       tg_igemm_desc* d);
int tg_conv2d_desc_dgrad(int n, int h, int w, int c_in_pad, int ld_dy, int k, int stride, int pad_same, int ld_out, int n_store, tg_igemm_desc* descs,
                         int32_t* n_desc_out);
int tg_conv2d_desc_wgrad(int n, int h, int w, int ld_in, int c_out_pad, int k, int stride, int pad_same, int ld_dy, tg_igemm_desc* d);
/* tf.layers.conv2d_transpose 5x5, stride 2, 'same' on [n,h,w,ld_in] -> [n,2h,2w,.] (Model/modle_base.py:246-259) */
int tg_deconv5x5s2_desc_fwd(int n, int h, int w, int ld_in, int c_out_pad, int ld_out, int n_store, int act, tg_igemm_desc* descs, int32_t* n_desc_out);
/* the same transposed conv as ONE 3x3 problem whose GEMM columns are (output parity, channel) (tg_igemm_desc.n_group; filter from
 * tg_deconv_merge_prep_f32 with the returned tapmap[36]): pays for narrow outputs (the 3-channel image layer). */
int tg_deconv5x5s2_desc_fwd_merged(int n, int h, int w, int ld_in, int c_out, int ld_out, int n_store, int act, tg_igemm_desc* d, int32_t* n_group_out,
                                   int32_t* tapmap);
int tg_deconv5x5s2_desc_dgrad(int n, int h, int w, int c_in_pad, int ld_dy, int ld_out, int n_store, tg_igemm_desc* d);
int tg_deconv5x5s2_desc_wgrad(int n, int h, int w, int ld_dy, int c_in_pad, int ld_x, tg_igemm_desc* d);
/* y[m, c_out] = x[m, ld_in] @ Wt[c_out][w_sn]^T (w_sn <= 0: ld_in); and the same product with the reduction cut into `splits` (<= 4)
 * sub-problems of one tg_igemm_multi_f32 launch writing part[m][s][n_out] (finish with tg_splitk_reduce_f32). */
int tg_dense_desc(int m, int ld_in, int c_out, int ld_out, int n_store, int act, int64_t w_sn, tg_igemm_desc* d);
int tg_dense_splitk_desc(int m, int k_dim, int n_out, int splits, tg_igemm_desc* descs);
/* pixel splits tg_wgrad_f32 should be given for descriptor d (fills one round of the resident workgroups; >= 1, < 0 on error) and the
 * bytes of its slab for n_split; bytes of one prepared filter layout [n_taps][c_in_pad][c_out_pad]. */
int tg_wgrad_splits(const tg_igemm_desc* d);
int tg_wgrad_splits_bf16(const tg_igemm_desc* d);    /* the same for tg_wgrad_bf16 (its 3x3 / stride-1 layers run on a kernel of their own: csrc/wgrad3x3.hip) */
int64_t tg_wgrad_workspace_bytes(const tg_igemm_desc* d, int n_split);
int64_t tg_filter_workspace_bytes(int n_taps, int c_in_pad, int c_out_pad);
/* the workgroup tile (rows x columns) the launcher picks for these sub-problems (nseg > 0: the tg_igemm_colsum_* constraint that a
 * tile straddles at most one segment boundary), and whether tg_igemm_colsum_f32 / tg_igemm_actsum_f32 accept (d, seg_rows): 1 / 0. */
int tg_igemm_tile(const tg_igemm_desc* descs, int n_desc, const int32_t* seg_rows, int nseg, int bf16, int32_t* bm_out, int32_t* bn_out);
int tg_igemm_colsum_supported(const tg_igemm_desc* d, const int32_t* seg_rows, int nseg);

/* The halo-tiled 3x3 / stride-1 / SAME kernel (csrc/conv3x3_bf16.hip) behind tg_igemm_{f32,bf16} and tg_igemm_colsum_{f32,bf16}:
 * one 256-pixel tile of whole image rows per workgroup, the halo loaded once for the nine taps.  policy 0 (default): taken where the
 * layer applies AND the launch fills the chip's rounds of one workgroup per CU well enough to beat the generic implicit GEMM;
 * 1: wherever the layer applies (kernel tests); 2: never (A/B).  Returns the previous policy; any other argument only queries.
 * tg_conv3x3_launches: launches of that kernel since the library was loaded. */
int tg_conv3x3_policy(int policy);
int64_t tg_conv3x3_launches(void);

/* ---- parameter-side kernels ---------------------------------------------------------------------- */
/* scale[c] = g[c] * rsqrt(max(sum_r V[r][c]^2, 1e-12)), V row-major [rows][c].
 * tf.nn.l2_normalize(V,[0,1,2])*g of conv2d_WN (Model/nn.py:502) and g/sqrt(sum V^2) of dense_WN (nn.py:554). */
int tg_wn_scale_f32(const float* v, const float* g, int rows, int c, float* scale, void* stream);

/* Re-layout of a filter src[t][a][b] (b contiguous; optional per-b `scale`) with zero channel padding:
 *   dst_same[t][a][b]            -> [t][a_pad][b_pad]         (may be NULL)
 *   dst_tr[b*tr_sb + t*tr_st + a] for b < b_pad, a < a_pad   (may be NULL)
 * `scale` multiplies per b, `scale_a` per a (either may be NULL).
 * HWIO conv filter -> OTI for tg_igemm_f32 forward (tr_sb = t*a_pad, tr_st = a_pad) and padded HWIO for the
 * input-gradient; [kh,kw,Cout,Cin] transposed-conv filter -> padded copy and per-tap transpose. */
int tg_filter_prep_f32(const float* src, const float* scale, const float* scale_a, int t, int a, int b, int a_pad, int b_pad, float* dst_same,
                       float* dst_tr, int64_t tr_sb, int64_t tr_st, void* stream);

/* weight norm of a transposed-conv filter V[t][a][b] = [kh*kw][Cout][Cin] over axes (0,1,3) (NN_Base._WN_deconv2d,
 * Model/modle_base.py:148): scale_a[a] = g[a]*rsqrt(max(sum_{t,b} V^2, 1e-12)) (feed it to tg_filter_prep_f32), and the
 * gradients of W = g V/||V|| given dW in the same layout. */
int tg_wn_scale_tab_f32(const float* v, const float* g, int t, int a, int b, float* scale_a, void* stream);
int tg_wn_bwd_tab_f32(const float* dw, const float* v, const float* g, int t, int a, int b, float* dv, float* dg, void* stream);

/* Merged filter of a stride-2 transposed conv for a tg_igemm_desc with n_group: dst[g*n_group + co][t9][c] (zero padded to
 * [n_pad][9][c_pad]) = w[tapmap[g*9 + t9]][co][c] * scale_a[co] (scale_a may be NULL), 0 where tapmap is negative.
 * w: [25][c_out][c_in] (tf conv2d_transpose filter [kh,kw,Cout,Cin]); tapmap: HOST array of 36 entries. */
int tg_deconv_merge_prep_f32(const float* w, const float* scale_a, int c_out, int c_in, int n_group, int n_pad, int c_pad,
                             const int32_t* tapmap, float* dst, void* stream);
/* tg_wn_scale_f32 + tg_filter_prep_f32 of up to 24 layers in two launches (one when no layer is weight-normalised): for every job
 * the arguments of those two calls (g == NULL: no weight norm; scale: scratch of b floats for the weight-normalised ones). */
typedef struct tg_prep_job {
  const float* src;
  const float* g;
  float* scale;
  float* dst_same;
  float* dst_tr;
  int64_t tr_sb, tr_st;
  int32_t t, a, b, a_pad, b_pad;
} tg_prep_job;
int tg_filter_prep_multi_f32(const tg_prep_job* jobs, int n_jobs, void* stream);
/* The tails of up to 16 filter-gradient launches in three launches instead of up to three each: for every job
 *   dw[t][c_in][c_out] = sum_s slab[s][t][c_pad][n_pad]                                  (as tg_slab_reduce_f32)
 *   and, when v != NULL (weight-normalised layer), dv / dg from dw as tg_wn_bwd_f32 (rows = t*c_in).
 * Jobs without v write dw as the final gradient.  coef: scratch of 2*c_out floats per weight-normalised job. */
typedef struct tg_wn_job {
  const float* slab;
  float* dw;
  const float* v;
  const float* g;
  float* dv;
  float* dg;
  float* coef;
  int32_t n_split, t, c_pad, n_pad, c_in, c_out;
} tg_wn_job;
int tg_filter_grad_tail_multi_f32(const tg_wn_job* jobs, int n_jobs, void* stream);
/* dst[t][c][n] = sum_s slab[s][t][c][n] (c < c_dim, n < n_dim): finishes tg_wgrad_f32, drops channel padding. */
int tg_slab_reduce_f32(const float* slab, int n_split, int t, int c_pad, int n_pad, int c_dim, int n_dim, float* dst, void* stream);

/* gradients of W = g V/||V|| given dW (all row-major [rows][c]); coef = scratch of 2*c floats.
 * Autodiff of Model/nn.py:502,554 emitted by optimizer.minimize (Training/train_base.py:65). */
int tg_wn_bwd_f32(const float* dw, const float* v, const float* g, int rows, int c, float* dv, float* dg, float* coef, void* stream);

/* ---- statistics / normalisation (HBM-bound) ------------------------------------------------------ */
/* Rows of a batched activation belong to up to 8 consecutive "application segments" (one per classifier /
 * discriminator application batched into the launch); seg_rows is a HOST array of nseg row counts. */
int64_t tg_colstats_workspace_floats(int rows, int nseg, int c);
/* Per-(segment, channel) sums, deterministic two-stage reduction.  mode 0: s1 = sum a; 1: s1 = sum a, s2 = sum a^2;
 * 2: s1 = sum a*act'(b) (b = activation output); 3: s1 = sum a, s2 = sum a*b; 4: s1 = sum (a - b[c]*alpha)^2 with b the
 * [c] sums of a mode-0 pass and alpha = 1/rows (centred second moment, one segment).  s1/s2: [nseg][c].
 * tf.nn.moments of mean_only_batch_norm_impl (Model/nn.py:171-175), tf.contrib.layers.batch_norm
 * (Model/modle_base.py:229-237), bias-gradient reductions. */
int tg_colstats_f32(int mode, const float* a, int ld_a, const float* b, int ld_b, int rows, int c, const int32_t* seg_rows, int nseg, int act,
                    float alpha, float* workspace, float* s1, float* s2, void* stream);
/* y[r][k] = act(x[r][k]*scale[k] + shift[seg(r)][k]) for k < c, 0 for c <= k < c_zero_to (scale may be NULL). */
int tg_seg_scale_shift_act_f32(const float* x, int ld_x, float* y, int ld_y, int rows, int c, int c_zero_to, const int32_t* seg_rows, int nseg,
                               const float* scale, const float* shift, int act, float alpha, void* stream);
/* dx[r][k] = dy[r][k]*act'(yact[r][k]) + shift[seg(r)][k]: backward of mean-only BN + nonlinearity. */
int tg_seg_actgrad_shift_f32(const float* dy, int ld_dy, const float* yact, int ld_y, float* dx, int ld_dx, int rows, int c,
                             const int32_t* seg_rows, int nseg, const float* shift, int act, float alpha, void* stream);
/* mean-only BN (Model/nn.py:147-187): train: shift[s][k] = b[k] - sums[s][k]/rows_s and pop_mean <- decay*pop_mean +
 * (1-decay)*mean_s sequentially over s; eval (train = 0): shift[s][k] = b[k] - pop_mean[k]. */
int tg_mobn_finalize_f32(const float* sums, const int32_t* seg_rows, int nseg, int rows, int c, const float* b, float* pop_mean, float decay,
                         int train, float* shift, void* stream);
/* fused tail of mean-only BN on x (in place): y = act(x - mean_seg + b) with mean_seg = sums[seg]/rows_seg (sums from
 * tg_igemm_colsum_f32; training: pop_mean <- decay*pop_mean + (1-decay)*mean_seg sequentially over the segments), or
 * y = act(x - pop_mean + b) when sums is NULL (evaluation).  c <= 512, c % 4 == 0. */
int tg_mobn_apply_f32(float* x, int ld, int rows, int c, const int32_t* seg_rows, int nseg, const double* sums, const float* b, float* pop_mean,
                      float decay, int act, float alpha, void* stream);
/* tg_mobn_apply_f32 AND the max-pool 2x2 + dropout behind the layer (Model/Good_GAN_cifar10.py:121-124,140-143: conv1_3 / conv2_3 -> tf.nn.max_pool ->
 * tf.layers.dropout) in one pass over the convolution's raw output x [n,h,w,c]: x = act(x - mean + b) in place (the backward pass reads it) and
 * out [n,h/2,w/2,c] = max over the 2x2 window of that * mask * mscale (mask NULL: no dropout — evaluation).  Segments must be whole images. */
int tg_mobn_apply_pool_f32(float* x, int ld, int n, int h, int w, int c, const int32_t* seg_rows, int nseg, const double* sums, const float* b,
                           float* pop_mean, float decay, int act, float alpha, float* out, int ld_out, const float* mask, int ld_mask, float mscale,
                           void* stream);
/* fused backward of mean-only BN + nonlinearity: dx = dy*act'(yact) - mean_seg(dy*act'(yact)), db[k] = sum over all rows
 * (db may be NULL).  sums: scratch of 8*nseg*c doubles (8 replicas of the accumulators; sums_zeroed as colsum_zeroed above).  Two launches (sums with fp64 atomics, apply).  c <= 512, c % 4 == 0;
 * segments of any size. */
int tg_mobn_bwd_f32(const float* dy, int ld_dy, const float* yact, int ld_y, float* dx, int ld_dx, int rows, int c, const int32_t* seg_rows, int nseg,
                    int act, float alpha, double* sums, int sums_zeroed, float* db, void* stream);
/* second half of the mean-only-BN backward when t = dy*act'(y) and its per-segment column sums already exist (tg_igemm_actsum_f32):
 * dx = t - sums[seg]/rows_seg (dx may alias t), db[k] = sum_s sums[s][k] (db may be NULL).  sums: n_repl copies of [nseg][c] doubles that are
 * added up (1 after tg_igemm_actsum_f32, 8 after tg_maxpool2_bwd_actsum_f32).  c <= 512, c % 4 == 0. */
int tg_mobn_center_f32(const float* t, int ld_t, float* dx, int ld_dx, int rows, int c, const int32_t* seg_rows, int nseg, const double* sums,
                       int n_repl, float* db, void* stream);
/* 2x2 max-pool (+ dropout) backward in front of a mean-only-BN layer (Model/Good_GAN_cifar10.py:123-124,142-143): t = routed pooled
 * gradient * act'(y) written to the layer's gradient buffer and its per-application column sums accumulated in the same pass
 * (sums: 8 replicas of [nseg][c] doubles -> tg_mobn_center_f32 with n_repl = 8).  seg_rows: HOST array in pre-pool pixel rows
 * (whole images per application).  c <= 512, c % 4 == 0. */
int tg_maxpool2_bwd_actsum_f32(const float* dout, int ld_do, const float* mask, int ld_mask, float mscale, const float* y, int ld_y, float* t, int ld_t,
                               int n, int h, int w, int c, const int32_t* seg_rows, int nseg, int act, float alpha, double* sums, int sums_zeroed,
                               void* stream);
/* shift[s][k] = -sums[s][k]/rows_s; db[k] = sum_s sums[s][k]. */
int tg_mobn_bwd_finalize_f32(const float* sums, const int32_t* seg_rows, int nseg, int rows, int c, float* shift, float* db, void* stream);
/* Fused training-mode batch norm over application segments (two launches): per segment s and column k
 *   mean = sum x / n_s, var = sum x^2 / n_s - mean^2 (fp64), y = gamma*(x-mean)/sqrt(var+eps) + beta;
 * mean_inv[s][0][k] = mean, mean_inv[s][1][k] = 1/sqrt(var+eps) (for the backward pass); the moving statistics (both NULL: none)
 * are updated sequentially over the segments with the unbiased variance.  sums: scratch of 16*nseg*c doubles (8 replicas).  c need not be a
 * multiple of 4 (columns up to the next multiple of 4 are read and written; they must lie inside ld). */
int tg_bn_train_f32(const float* x, int ld_x, float* y, int ld_y, int rows, int c, const int32_t* seg_rows, int nseg, const float* gamma,
                    const float* beta, float eps, float decay, float* moving_mean, float* moving_var, double* sums, int sums_zeroed, float* mean_inv,
                    void* stream);
/* tg_bn_train_f32 without its statistics launch: `sums` already holds S0 = sum x, S1 = sum x^2 per (segment, column) — left there by
 * tg_igemm_bnstat_* (the producing convolution) — in the layout above.  One launch: normalise, mean / inv-std, moving statistics. */
int tg_bn_train_apply_f32(const float* x, int ld_x, float* y, int ld_y, int rows, int c, const int32_t* seg_rows, int nseg, const float* gamma,
                          const float* beta, float eps, float decay, float* moving_mean, float* moving_var, const double* sums, float* mean_inv,
                          void* stream);
/* One more moving-statistics update from the batch sums a tg_bn_train_f32 launch left in `sums` (same seg_rows / nseg / c / decay; the
 * buffer must not have been cleared since): what TensorFlow does when a later sess.run re-executes the same training-mode batch norm on
 * the same feed and weights (tf.contrib.layers.batch_norm, updates_collections=None, Model/modle_base.py:229-237) while this build
 * re-uses the kept forward pass — bit-identical to re-running tg_bn_train_f32, without touching x / y. */
int tg_bn_moving_update_f32(const double* sums, int rows, int c, const int32_t* seg_rows, int nseg, float decay, float* moving_mean,
                            float* moving_var, void* stream);
/* its backward: dx = gamma*inv*(dy - mean_s(dy) - xhat*mean_s(dy*xhat)) per segment (masked by x > 0 when relu_input: the gradient is
 * then with respect to the pre-ReLU value), dgamma = sum_s sum dy*xhat, dbeta = sum_s sum dy (both NULL: not wanted).
 * sums_zeroed: 0 = the call clears `sums`, 1 = the caller did, 2 = `sums` already HOLDS S0 = sum dy, S1 = sum dy*x (tg_igemm_bnbwdstat_*):
 * no statistics launch. */
int tg_bn_train_bwd_f32(const float* dy, int ld_dy, const float* x, int ld_x, float* dx, int ld_dx, int rows, int c, const int32_t* seg_rows, int nseg,
                        const float* gamma, const float* mean_inv, int relu_input, double* sums, int sums_zeroed, float* dgamma, float* dbeta,
                        void* stream);
/* The same pass with the derivative of the activation that PRODUCED x folded in (x = act(conv + bias) feeding a batch norm: the SVHN /
 * MNIST classifier's conv -> leaky relu -> BN, the generator's deconv -> relu -> BN; Model/Good_GAN.py:249-350, Model/Good_GAN_cifar10.py:44-53):
 *   dx = [ gamma*inv*(dy - mean_s(dy) - xhat*mean_s(dy*xhat)) ] * act'(x)      act: TG_ACT_NONE / TG_ACT_RELU / TG_ACT_LRELU (alpha)
 * is the gradient at the producing layer's PRE-activation output, and — dsum / dbias not NULL — dbias[k] = sum over all rows of dx[:,k] is
 * that layer's bias gradient: tg_actgrad_bias_f32's read-modify-write pass over the activation disappears.  dsum: scratch of 8*c doubles. */
int tg_bn_train_bwd_act_f32(const float* dy, int ld_dy, const float* x, int ld_x, float* dx, int ld_dx, int rows, int c, const int32_t* seg_rows,
                            int nseg, const float* gamma, const float* mean_inv, int act, float alpha, double* sums, int sums_zeroed, float* dgamma,
                            float* dbeta, double* dsum, int dsum_zeroed, float* dbias, void* stream);
/* batch norm (training mode, biased variance) from s1 = sum x and s2 = sum (x-mean)^2 (modes 0 and 4 above):
 * scale = gamma*inv, shift = beta - mean*scale, mean_inv = [mean | inv];
 * moving statistics updated in place when non-NULL (bessel = use the unbiased variance, the fused 4-D kernel). */
int tg_bn_finalize_f32(const float* s1, const float* s2, int rows, int c, const float* gamma, const float* beta, float eps, float* scale,
                       float* shift, float* mean_inv, float* moving_mean, float* moving_var, float decay, int bessel, void* stream);
/* inference-mode batch norm (is_training=False): scale = gamma*rsqrt(moving_var+eps), shift = beta - moving_mean*scale. */
int tg_bn_eval_finalize_f32(int c, const float* gamma, const float* beta, const float* moving_mean, const float* moving_var, float eps,
                            float* scale, float* shift, void* stream);
/* from s_dy = sum dy, s_dyx = sum dy*x: dgamma, dbeta and abc = [A | B | C] with dx = A*dy + B*x + C. */
int tg_bn_bwd_finalize_f32(const float* s_dy, const float* s_dyx, int rows, int c, const float* gamma, const float* mean_inv, float* abc,
                           float* dgamma, float* dbeta, void* stream);
/* dx = (A*dy + B*x + C) * (relu_mask ? x > 0 : 1): BN backward fused with the ReLU that precedes it in the generator
 * (ReLU -> BN order, Model/Good_GAN_cifar10.py:41-42). */
int tg_bn_bwd_apply_f32(const float* dy, int ld_dy, const float* x, int ld_x, float* dx, int ld_dx, int rows, int c, const float* abc,
                        int relu_mask, void* stream);

/* ---- pointwise / pooling / concat ---------------------------------------------------------------- */
/* out[r][:c] = x[r][:c] + add[r][:c] (add may be NULL), zero up to ld_out.  Gaussian input noise of the classifier
 * (Model/modle_base.py:193-202 via Good_GAN_cifar10.py:104) + channel padding. */
int tg_pad_add_f32(const float* x, int ld_x, int c, const float* add, int ld_add, float* out, int ld_out, int rows, void* stream);
/* 3x3 SAME stride-1 patches of (x + add), x dense [n,h,w,c]: out[(n,y,x)][t*c+k], zero outside the image and up to ld_out.
 * Lets the classifier's first convolution (Cin = 3, Model/Good_GAN_cifar10.py:106) run as a K = 27 -> 32 dense product. */
int tg_im2col3x3_add_f32(const float* x, const float* add, int n, int h, int w, int c, float* out, int ld_out, void* stream);
/* out[n,p,:] = [x[n,p,:c]*mask*mscale, y[n,:ncls], 0...]: dropout (modle_base.py:190-191) + _conv_cond_concat
 * (modle_base.py:239-244).  mask may be NULL. */
int tg_cond_concat_f32(const float* x, int ld_x, int c, const float* mask, int ld_mask, float mscale, const float* y, int ncls, float* out,
                       int ld_out, int n_img, int hw, void* stream);
/* y[r][:c] = act(x[r][:c]), zero up to ld_y: an activation CALLED on a tensor (tf.nn.relu / leaky_relu / softplus / tanh / sigmoid,
 * Model/modle_base.py:176-188, Good_GAN_cifar10.py:19-27; tf.nn.sigmoid of the discriminator's logit, :97).  The models fuse their
 * activations into the producing kernel instead (tg_igemm_desc.act). */
int tg_act_f32(const float* x, int ld_x, float* y, int ld_y, int rows, int c, int act, float alpha, void* stream);
/* out[r][:c] = dy[r][:c]*mask*mscale*act'(yact[r][:c]), zero up to ld_out (mask, yact may be NULL). */
int tg_actgrad_f32(const float* dy, int ld_dy, const float* yact, int ld_y, const float* mask, int ld_mask, float mscale, float* out, int ld_out,
                   int rows, int c, int act, float alpha, void* stream);
/* the same for a layer with a bias (no mask): out = dy*act'(yact) (yact NULL: out = dy), padding zeroed up to ld_out, AND
 * bias_grad[k] = sum over rows of out[:,k] in the same pass (tf.nn.bias_add gradient).  sums: scratch of 8*c doubles
 * (sums_zeroed as in tg_igemm_colsum_f32).  ld_out % 4 == 0. */
int tg_actgrad_bias_f32(const float* dy, int ld_dy, const float* yact, int ld_y, float* out, int ld_out, int rows, int c, int act, float alpha,
                        double* sums, int sums_zeroed, float* bias_grad, void* stream);
/* tf.nn.max_pool 2x2 s2 (Good_GAN_cifar10.py:123,142) fused with the dropout that follows (:124,143). */
int tg_maxpool2_fwd_f32(const float* y, int ld_y, float* out, int ld_out, const float* mask, int ld_mask, float mscale, int n, int h, int w, int c,
                        void* stream);
int tg_maxpool2_bwd_f32(const float* dout, int ld_do, const float* mask, int ld_mask, float mscale, const float* y, int ld_y, float* dy, int ld_dy,
                        int n, int h, int w, int c, void* stream);
/* global MAX pool (the layer named avg_pool_0, Good_GAN_cifar10.py:163); pads of out are zeroed. */
int tg_gmaxpool_fwd_f32(const float* x, int ld_x, float* out, int ld_out, int n, int hw, int c, void* stream);
int tg_gmaxpool_bwd_f32(const float* dfeat, int ld_d, const float* x, int ld_x, float* dx, int ld_dx, int n, int hw, int c, void* stream);
/* out[n,:] = [mean_p x[n,p,:c], y[n,:ncls], 0...]: average_pooling2d(8) + squeeze + concat (Good_GAN_cifar10.py:94-96). */
int tg_gavgpool_concat_f32(const float* x, int ld_x, int c, const float* y, int ncls, float* out, int ld_out, int n, int hw, void* stream);
int tg_gavgpool_bwd_f32(const float* dfeat, int ld_d, const float* yact, int ld_y, float* out, int ld_out, int n, int hw, int c, int act,
                        float alpha, void* stream);
int tg_copy2d_f32(const float* src, int64_t ld_s, float* dst, int64_t ld_d, int64_t rows, int64_t c, void* stream);
/* up to 16 contiguous copies dst[0:n) = src[0:n) in ONE launch: tf.concat along the batch axis (Model/Good_GAN_cifar10.py:258-259, the
 * batched network applications of this package) and the feed of one iteration's placeholders (Training/Train_goodGAN.py:249-263). */
typedef struct tg_copy_job {
  const float* src;
  float* dst;
  int64_t n;
} tg_copy_job;
int tg_copy_multi_f32(const tg_copy_job* jobs, int n_jobs, void* stream);
int tg_fill_f32(float* dst, float value, int64_t n, void* stream);
/* out[m][n] = bias[n] + sum_s part[m][s][n] (bias may be NULL): finishes a dense product whose reduction dimension was split
 * into s_dim sub-problems of one tg_igemm_multi_f32 launch (skinny GEMMs such as the ZCA product: few rows, long K). */
int tg_splitk_reduce_f32(const float* part, const float* bias, float* out, int ld_out, int64_t m, int s_dim, int n, void* stream);
/* input-pipeline tail on the device (Input_Pipeline/cifar10Dataset.py:52-62): dst = float(src)/255 * scale + shift
 * (scale 2, shift -1 for SVHN / CIFAR-10; scale 1, shift 0 for MNIST, mnistDataset.py:65), and tf.one_hot(label, k). */
int tg_u8_affine_f32(const uint8_t* src, float* dst, int64_t n, float scale, float shift, void* stream);
int tg_onehot_i32_f32(const int32_t* labels, float* out, int64_t n, int k, void* stream);
/* dst = a + b (dst may alias a or b): sums the per-application gradient buffers of a network applied several times. */
int tg_add_f32(float* dst, const float* a, const float* b, int64_t n, void* stream);
/* tf.one_hot(tf.argmax(logits,1)) (Good_GAN_cifar10.py:232,237,259,270); out [n][k]. */
int tg_argmax_onehot_f32(const float* logits, int ld, int n, int k, float* out, void* stream);

/* ---- loss heads: value + d/dlogits in one launch (Training/train_base.py:113-154) ---------------- */
/* rows [real | fake | unl]: d_loss = BCE(real,1) + .5 BCE(fake,0) + .5 BCE(unl,0). */
int tg_d_loss_f32(const float* logits, int ld, int n_real, int n_fake, int n_unl, float* dlogits, int ld_d, float* loss, void* stream);
/* g_loss = .5 BCE(D_fake,1). */
int tg_g_loss_f32(const float* logits, int ld, int n, float* dlogits, int ld_d, float* loss, void* stream);
/* rows [real | unl | unl_rep | fake] (n_rep = n_unl, or 0 when there is no consistency term); lambdas = device {l1,l2}:
 * c_loss = .005 c_unl + CE(real) + 1e-6 H(unl) + 1e-3 Bal(unl) + l1 CE(fake) + l2 MSE(unl,rep). */
int tg_c_loss_f32(const float* c_logits, int ld, int n_real, int n_unl, int n_rep, int n_fake, const float* y_real, const float* y_fake,
                  const float* d_unl_logits, int ld_dunl, const float* lambdas, float* dlogits, int ld_d, float* loss, void* stream);
/* feature matching (train_base.py:172) and pull-away (masked :175-181, unmasked :204-207) terms; dense [n][c] features. */
int tg_feature_match_f32(const float* f_fake, int n_fake, const float* f_unl, int n_unl, int c, float* df_fake, float* df_unl, float* loss,
                         void* stream);
int tg_pull_away_f32(const float* f, int n, int c, int masked, float* scratch /* n*c+n*n+n floats */, float* df, float* loss, void* stream);
/* ---- loss variants of Training/train_base.py:156-574 (_loss_BGAN, _loss_GoodBadGAN, _loss_GoodRegGAN[_cifar10|_BS|_BS_cifar10],
 * _loss_GoodRegBadGAN; SURVEY §8f N4 — no trainer of the reference repository calls them).  They are weighted sums of the terms
 * above plus the two below; Training/train_base.py of the package composes them. */
/* tg_d_loss_f32 that also reports terms[3] = {BCE(real,1), .5 BCE(fake,0), .5 BCE(unl,0)} (the lists train_base.py:304 returns). */
int tg_d_loss_terms_f32(const float* logits, int ld, int n_real, int n_fake, int n_unl, float* dlogits, int ld_d, float* loss, float* terms,
                        void* stream);
/* tg_c_loss_f32 with explicit HOST weights[6] = {CE(real), c_unl, H(unl), Bal(unl), CE(fake), MSE(unl,rep)} (_loss_GAN: {1, .005,
 * 1e-6, 1e-3, l1, l2}); n_fake may be 0 (then y_fake may be NULL), d_unl_logits may be NULL when weights[1] == 0.
 * terms (optional, device [6]): the unweighted term values in the same order. */
int tg_c_loss_terms_f32(const float* c_logits, int ld, int n_real, int n_unl, int n_rep, int n_fake, const float* y_real, const float* y_fake,
                        const float* d_unl_logits, int ld_dunl, const float* weights, float* dlogits, int ld_d, float* loss, float* terms,
                        void* stream);
/* bad-GAN "true-fake" terms (train_base.py:162-166,226-229,296-298), lse = logsumexp_k(logits):
 *   T_unl = mean(-.5 lse + .5 softplus(lse)) over the unlabelled rows, T_fake = .5 mean softplus(lse) over the bad generator's rows;
 * loss[3] = {w_unl T_unl + w_fake T_fake, T_unl, T_fake}; the gradients are written to d_unl / d_fake or, accumulate_* != 0, added. */
int tg_true_fake_loss_f32(const float* unl_logits, int ld_u, int n_unl, const float* fake_logits, int ld_f, int n_fake, float w_unl, float w_fake,
                          float* d_unl, int ld_du, int accumulate_unl, float* d_fake, int ld_df, int accumulate_fake, float* loss, void* stream);
/* T = mean_n sum_k (a - b)^2 (train_base.py:299); loss[2] = {w T, T}; da / db (either may be NULL) written or added. */
int tg_sqdiff_rows_loss_f32(const float* a, int ld_a, const float* b, int ld_b, int n, int k, float w, float* da, int ld_da, int accumulate_a,
                            float* db, int ld_db, int accumulate_b, float* loss, void* stream);
/* minibatch discrimination (Model/modle_base.py:110-128): act = x @ W viewed [n][kernels][dim] (dim <= 8);
 * out[i] = [x[i,:c], f[i,:], 0...] with f[i,k] = sum_j exp(-sum_d |act[i,k,d] - act[j,k,d]|) + b[k].
 * bwd: df = the gradient's columns c.. ([n][kernels], stride ld_df) -> dact (pads zeroed), db[k] = sum_i df[i,k] (db may be NULL). */
int tg_minibatch_disc_fwd_f32(const float* act, int ld_a, const float* x, int ld_x, int c, const float* b, float* out, int ld_out, int n, int kernels,
                              int dim, void* stream);
int tg_minibatch_disc_bwd_f32(const float* act, int ld_a, const float* df, int ld_df, float* dact, int ld_da, float* db, int n, int kernels, int dim,
                              void* stream);
/* Stand-alone heads behind the reference's Train_base helper methods (value + d/dlogits; dlogits may be NULL; accumulate != 0 adds to
 * dlogits instead of writing, so a loss assembled from several helper calls sums into one gradient buffer):
 *   tg_softmax_ce_f32     T = mean_n softmax-CE(labels, logits)                (_softmax_cross_entropy_loss_w_logits, train_base.py:75-79);
 *                         labels dense [n][k]; loss[2] = {w T, T}
 *   tg_bce_logits_f32     T = mean over n*c elements of sigmoid-CE(labels, logits) (_sigmoid_cross_entopy_w_logits, :81-84); labels NULL:
 *                         the constant `label` everywhere (tf.ones_like / tf.zeros_like, :123-128); loss[2] = {w T, T}
 *   tg_entropy_terms_f32  H = mean_n(lse - sum_k p_k l_k) (_entropy, :43-48), Bal = -sum_k (1/K) log(mean_n p_k + 1e-12) (_balance_entropy,
 *                         :50-57); loss[3] = {w_h H + w_bal Bal, H, Bal}
 * k must be 10 (NUM_CLASSES of every config of the reference). */
int tg_softmax_ce_f32(const float* logits, int ld, const float* labels, int n, int k, float w, float* dlogits, int ld_d, int accumulate, float* loss,
                      void* stream);
int tg_bce_logits_f32(const float* logits, int ld, const float* labels, int ld_y, float label, int n, int c, float w, float* dlogits, int ld_d,
                      int accumulate, float* loss, void* stream);
int tg_entropy_terms_f32(const float* logits, int ld, int n, int k, float w_h, float w_bal, float* dlogits, int ld_d, int accumulate, float* loss,
                         void* stream);
/* counters[0] += #correct, counters[1] += n (tf.metrics.accuracy, Training/Train_goodGAN.py:428-447). */
int tg_accuracy_count_f32(const float* logits, int ld, const float* labels, int n, int k, float* counters, void* stream);

/* ---- optimiser ------------------------------------------------------------------------------------ */
/* TF-form Adam over a flat buffer (Training/train_base.py:91-97); *step_dev is incremented first, lr read from device;
 * g is multiplied by grad_scale (1/world_size after a sum all-reduce). */
int tg_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, float beta1, float beta2, float eps, int* step_dev,
                float grad_scale, void* stream);
/* shadow -= (1-decay)*(shadow - p) (tf.train.ExponentialMovingAverage, Training/Train_goodGAN.py:101-103). */
int tg_ema_f32(float* shadow, const float* p, int64_t n, float decay, void* stream);

/* ---- RNG (Philox4x32-10; state = device {seed, step}) -------------------------------------------- */
int tg_rng_uniform_f32(float* out, int64_t n, float lo, float hi, const uint64_t* state, uint32_t stream_id, void* stream);
int tg_rng_keep_mask_f32(float* out, int64_t n, float keep_prob, const uint64_t* state, uint32_t stream_id, void* stream);
int tg_rng_normal_f32(float* out, int64_t n, float stddev, const uint64_t* state, uint32_t stream_id, void* stream);
int tg_rng_onehot_f32(float* out, int rows, int k, const uint64_t* state, uint32_t stream_id, void* stream);
int tg_rng_advance(uint64_t* state, void* stream);
/* Up to 16 of the draws above in ONE launch (all random inputs of a solver run): mode 0 uniform [a,b), 1 keep-mask with probability a,
 * 2 normal(0,a), 3 one-hot of a classes per row (n = rows).  Bit-identical to the single calls (same counters). */
typedef struct tg_rng_job {
  float* out;
  int64_t n;
  int32_t mode;
  float a, b;
  uint32_t stream_id;
} tg_rng_job;
int tg_rng_multi_f32(const tg_rng_job* jobs, int n_jobs, const uint64_t* state, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TG_KERNELS_H */
