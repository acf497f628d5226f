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
} tg_igemm_desc;

/* out[p,n] = act( sum_t sum_c in[pix(p,t),c] * w[n,t,c] + bias[n] ).
 * Replaces tf.nn.conv2d / tf.layers.conv2d (Model/nn.py:504, Model/modle_base.py:102,161), their
 * input-gradient, tf.layers.conv2d_transpose (Model/modle_base.py:250; one launch per output parity),
 * tf.matmul / tf.layers.dense (Model/nn.py:553, Model/modle_base.py:40) and the ZCA matmul
 * (Model/Good_GAN_cifar10.py:296).  bias may be NULL. */
int tg_igemm_f32(const tg_igemm_desc* d, const float* in, const float* w, const float* bias, float* out, void* stream);

/* filter gradient, split over `n_split` pixel ranges:
 * slab[s][t][c][n] = sum_{p in split s} in[pix(p,t),c] * dout[p,n]   (c < ld_in, n < c_out).
 * `dout` is read through (h_out,w_out,ld_out,os,oo) exactly as tg_igemm_f32 writes `out`.
 * Replaces Conv2DBackpropFilter / MatMul-grad emitted by optimizer.minimize (Training/train_base.py:65).
 * slab holds n_split*n_taps*ld_in*c_out floats; deterministic (no atomics). */
int tg_wgrad_f32(const tg_igemm_desc* d, const float* in, const float* dout, float* slab, int n_split, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TG_KERNELS_H */
