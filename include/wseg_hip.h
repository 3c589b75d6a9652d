/* wseg_hip.h — C ABI of libwseg_hip.so (MI355X / gfx950).
 *
 * The reference (obeychoi0120/wseg) has no FFI: its hot path is PyTorch ATen calls made from
 * network/resnet38d.py, network/resnet38_contrast.py and the loop body of contrast_train.py.
 * Each entry point below replaces one group of those ATen call sites (file:line cited per
 * function, paths relative to the reference root).  The Python host (wseg_amd/) binds them
 * with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocates); the library never
 *     allocates, frees or synchronises, keeps no mutable global state, and is re-entrant
 *     (contrast_infer.py:69-73 calls forward from 8 threads);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it;
 *   - activations are NHWC ("pixel rows"): element (n,y,x,c) at ((n*H+y)*W+x)*ld + c;
 *   - dtype: 0 = f32 (exact-f32 MFMA, parity mode), 1 = bf16 (bf16 MFMA, f32 accumulate);
 *   - return 0 on success, negative on error; wseg_last_error() gives the thread-local message.
 */
#ifndef WSEG_HIP_H
#define WSEG_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define WSEG_F32 0
#define WSEG_BF16 1

int wseg_version(void);
const char* wseg_last_error(void);

/* ---- convolution as implicit GEMM ------------------------------------------------------
 * Replaces nn.Conv2d forward / input-gradient for every 3x3 (stride 1|2, dilation 1|2|4) and
 * 1x1 (stride 1|2) conv of the backbone and heads: network/resnet38d.py:17,22,25,61,65,69,72,124;
 * network/resnet38_contrast.py:15-20,36-38,50-51,67.  The frozen BatchNorm(eval)+ReLU
 * (+Dropout2d scale) that follows a conv (resnet38d.py:29-30,40-41,76-77,84-91,187) is the
 * fused epilogue, the residual add (resnet38d.py:44,94) too.
 *
 *   mode 0 (forward):   out[n,oy,ox,:] = sum_{ky,kx} in[n, oy*stride+ky*dil-pad, ox*stride+kx*dil-pad, :] . w[:,ky,kx,:]
 *   mode 1 (data grad): out[n,y,x,:]   = sum_{ky,kx} in[n, (y+pad-ky*dil)/stride, (x+pad-kx*dil)/stride, :] . w[:,ky,kx,:]
 *                       (terms whose division is inexact or out of range are zero)
 *   weights: w[OC][KH*KW][IC], IC contiguous (for mode 1 the caller passes the transposed pack).
 *
 *   epilogue, v = acc:
 *     if r_pre   : v += r_pre
 *     epi 0      : out  = v (+ r_post)                                   raw sum
 *                  out2 = relu(v*scale[c]+shift[c]) * drop[n,c]          (when out2 != NULL)
 *     epi 1      : out  = v*scale[c]*drop[n,c]*(mask>0) + r_post          (BN-ReLU backward)
 *     epi 2      : out  = relu(v)                                         (f8_3/f8_4/fc_proj)
 *   NULL scale/shift/drop/mask/r_post mean 1/0/1/all-pass/0.
 */
typedef struct wseg_conv_desc {
  const void* in;  const void* w;  void* out;  void* out2;
  const void* r_pre;  const void* r_post;  const void* mask;
  const float* scale;  const float* shift;  const float* drop;
  int32_t N, IH, IW, IC, ld_in;
  int32_t OH, OW, OC, ld_out, ld_out2;
  int32_t ld_rpre, ld_rpost, ld_mask;
  int32_t KH, KW, stride, dil, pad;
  int32_t mode, epi, dtype;
  int32_t relu_out2;   /* 1: out2 gets the ReLU (default); 0: affine only */
  int32_t relu_lt;     /* epi 0: ReLU on `out` channels < relu_lt (fused head: f_proj | cam); 0 = none */
} wseg_conv_desc;
int wseg_conv_igemm(const wseg_conv_desc* d, void* stream);

/* ---- weight gradient -------------------------------------------------------------------
 * Replaces the weight-gradient half of Conv2d backward (autograd of the same call sites).
 *   dw[oc][ky*KW+kx][ic] (+)= sum_{n,oy,ox} dy[n,oy,ox,oc] * x[n, oy*stride+ky*dil-pad, ox*stride+kx*dil-pad, ic]
 * dw is f32, accumulated with atomics (split over pixel ranges); the caller zeroes it once per step
 * so that both views accumulate (contrast_train.py:398 sums both forward graphs).
 */
typedef struct wseg_wgrad_desc {
  const void* x;  const void* dy;  float* dw;
  int32_t N, IH, IW, IC, ld_x;
  int32_t OH, OW, OC, ld_dy;
  int32_t KH, KW, stride, dil, pad;
  int32_t dtype, split_k;      /* split_k <= 0: library heuristic */
  int32_t IC_dw, OC_dw;        /* real extents of dw ([OC_dw][KH*KW][IC_dw]); IC/OC may be padded */
} wseg_wgrad_desc;
int wseg_conv_wgrad(const wseg_wgrad_desc* d, void* stream);

/* ---- weight packing ----------------------------------------------------------------------
 * master f32 [OC][T][IC] -> fwd pack [OCp][T][ICp] and transposed pack [ICp][T][OCp] in `dtype`
 * (zero padded).  Either destination may be NULL. */
int wseg_pack_weights(const float* master, void* fwd, void* tr, int OC, int T, int IC,
                      int OCp, int ICp, int dtype, void* stream);

/* ---- stem: conv1a (3->64, 3x3, pad 1) + the next block's frozen BN-ReLU -------------------
 * network/resnet38d.py:124,162 (+ :29-30 of b2).  x is the reference's NCHW f32 input;
 * raw = conv1a(x) NHWC, act = relu(raw*scale+shift) NHWC (either may be NULL). */
int wseg_stem_conv(const float* x_nchw, const float* w /*[64][3][3][3] = [oc][ky][kx][ic]*/,
                   const float* scale, const float* shift, void* raw, void* act,
                   int N, int H, int W, int dtype, void* stream);

/* ---- CAM head (network/resnet38_contrast.py:34-59) -------------------------------------------
 * Fused head GEMM rows are [f_proj(128) | cam logits(21) | zero pad] (ld = 192).
 * head_split : rows -> planar cam_low [N][21][hw] f32 and cmax[n][c] = max_hw relu(cam)   (:41-43)
 * cam_gate   : the no_grad normalise / bg = 1-max fg / keep-arg-max gate (:44-48) -> G [N*hw][32],
 *              G[:,21] = 1 (PCM column-sum channel), G[:,22:] = 0
 * pcm_xs     : x_s = bilinear(x, (h,w), align_corners=True) (:52) into cols [c_xs, c_xs+3) of the
 *              feature rows `feat` (row stride ld), zeroing the padding cols [c_xs+3, c_end)
 * head_grad_rows : assemble d(head rows) from d_f_proj (planar [N][128][hw], ReLU-masked by the stored
 *              rows) and d_cam_low (planar [N][21][hw]).
 */
int wseg_head_split(const void* head, int ld, int c0, float* cam_low, float* cmax, int N, int hw, int dtype, void* stream);
int wseg_cam_gate(const float* cam_low, const float* cmax, float* G, int N, int hw, void* stream);
int wseg_pcm_xs(const float* x_nchw, void* feat, int ld, int c_xs, int c_end, int N, int H, int W, int h, int w, int dtype, void* stream);
int wseg_head_grad_rows(const float* d_fproj, const float* d_cam_low, const void* head, void* d_head, int ld, int N, int hw, int dtype, void* stream);
int wseg_planar_to_rows(const float* planar, void* rows, int ld, int c0, int C, int N, int hw, int dtype, void* stream);

/* planar bilinear resize of [planes][ih][iw] f32 (F.interpolate(mode='bilinear'), align_corners
 * 0/1 — resnet38_contrast.py:57-59, contrast_train.py:131-134,145-152,180; contrast_infer.py:62).
 * bwd is the exact adjoint computed by gather (deterministic). plane_mul (nullable) scales plane p. */
int wseg_resize_planar_fwd(const float* in, float* out, const float* plane_mul, long planes, int ih, int iw, int oh, int ow, int align, void* stream);
int wseg_resize_planar_bwd(const float* d_out, float* d_in, const float* plane_mul, long planes, int ih, int iw, int oh, int ow, int align, int accumulate, void* stream);

/* ---- PCM (network/resnet38_contrast.py:63-75), flash-style, exact-f32 MFMA ---------------------
 * l2norm  : Fh = F/(||F||_2 + 1e-5) over the 192 f9 channels of each pixel row (:70)
 * forward : cam_rv[n][c][j] = sum_i G[i][c] relu(Fh_i.Fh_j) / (sum_i relu(Fh_i.Fh_j) + 1e-5)  (:71-73)
 * backward: gradient w.r.t. Fh only (the CAM input is no_grad in the reference, :41-48). */
int wseg_l2norm_forward(const void* F, int ldf, float* Fh, float* nrm, long rows, int dtype, void* stream);
int wseg_l2norm_backward(const void* F, int ldf, const float* dFh, const float* nrm, void* dF, int lddf, long rows, int dtype, void* stream);
int wseg_pcm_forward(const float* Fh, const float* G, float* cam_rv, float* den, int N, int hw, void* stream);
int wseg_pcm_backward(const float* Fh, const float* G, const float* d_cam_rv, const float* cam_rv, const float* den,
                      float* DN, float* dFh, int N, int hw, void* stream);

/* ---- fused SGD step (tool/torchutils.py:23-33 -> torch.optim.SGD.step) ------------------------
 * One pass over the flat buffers: d = g*grad_scale + wd*p; buf = first ? d : momentum*buf + d;
 * p -= lr*buf.  Segments [begin,end) carry the per-group lr / weight_decay (contrast_train.py:91-96). */
int wseg_sgd_step(float* params, const float* grads, float* momentum_buf, long numel,
                  const long* seg_begin, const long* seg_end, const float* seg_lr, const float* seg_wd, int nseg,
                  float momentum, float grad_scale, int first_step, void* stream);

#ifdef __cplusplus
}
#endif
#endif
