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
 *   - dtype: 0 = f32 (exact-f32 MFMA, parity mode), 1 = bf16 (bf16 MFMA, f32 accumulate), 2 = f32 storage with split-bf16 products
 *     (conv / wgrad only: every f32 operand x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the product is hi.hi + lo.hi + hi.lo on
 *     the bf16 MFMA with f32 accumulate — 16-17 operand bits at 3 bf16 MFMAs per product instead of the 8x slower f32 MFMA; activations
 *     are split in the kernel, weights arrive pre-split: wseg_pack_x3);
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
#define WSEG_F32X3 2

int wseg_version(void);
/* sizeof(wseg_conv_desc) / sizeof(wseg_wgrad_desc) as this build sees them: a binding checks its mirror of the descriptors against
 * these before the first call (a stale, shorter mirror would hand the kernels uninitialised trailing fields). */
size_t wseg_sizeof_conv_desc(void);
size_t wseg_sizeof_wgrad_desc(void);
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
  int32_t bm_hint;     /* 0 = library chooses the tile (64 / 128 pixel rows x 128 channels, or the 256 x 256 phase-pipelined
                          bf16 kernel for large layers with OC % 256 == 0, or the 512 x 128 one for OC = 128 layers with many pixels);
                          64 / 128 / 224 / 256 = force; 259 = test hook (512 x 128 tile); negative values: development
                          probes, refused unless the library was built with -DWSEG_PROBES */
  /* optional SECOND row segment (the 128x128 view batched behind the 448x448 view in one launch): rows
   * [0, N*OH*OW) use (IH,IW,OH,OW); rows beyond use (IH2,IW2,OH2,OW2), same N, their input pixels follow the
   * first segment's N*IH*IW rows; drop then has 2N rows.  OH2 == 0: single segment. */
  int32_t IH2, IW2, OH2, OW2;
  /* optional SECOND INPUT (K-concatenation): out = conv(in; W[:, :KH*KW*IC]) + in2 . W[:, KH*KW*IC:] with w = [OC][KH*KW*IC + IC2] —
   * a residual block's last conv and its 1x1 skip conv (network/resnet38d.py:35-47, 83-97: branch1 + branch2) as ONE product: the
   * second source is one extra K segment read at the output pixel itself, so the skip output is neither written nor re-read.  With
   * the transposed packs and mode 1 the same form is the sum of the two data gradients into the block's input.  Same-size stride-1
   * convolution (pad = dil*(KH/2)), bf16, OC % 256 == 0, IC2 % 64 == 0 (0: IC2 = IC), in2 on the OUTPUT pixel grid; NULL: none. */
  const void* in2;
  int32_t ld_in2, IC2;
  int32_t w_rows;      /* rows the weight pack really holds (>= OC; 0 = OC).  A pack zero-padded to a multiple of 256 rows lets a launch whose OC is
                          not one (the fused head: 149 of 192 columns, resnet38_contrast.py:34-38) run on the 256-tile kernel: it reads whole 256-row
                          weight tiles and masks the columns >= OC in its epilogue */
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
  int32_t tile_hint;           /* 0 = library chooses (256x256 tiles for bf16 with OC,IC >= 256), 128 = force 128x128 */
  int32_t IH2, IW2, OH2, OW2;  /* optional second row segment, as in wseg_conv_desc (OH2 == 0: none) */
  int32_t dw_rot;              /* column rotation of dw: input channel ic accumulates into column (ic + dw_rot) % IC_dw — the PCM feature rows are
                                  [f8_3 | f8_4 | x_s] where f9.weight's columns are [x_s | f8_3 | f8_4] (resnet38_contrast.py:53: torch.cat([x_s, f8_3, f8_4])):
                                  dw_rot = 3 writes f9's gradient in the parameter's own column order (128-tile kernel only; 0 = none) */
} wseg_wgrad_desc;
int wseg_conv_wgrad(const wseg_wgrad_desc* d, void* stream);
/* A layer's data gradient (`dg`: mode 1) and a weight gradient (`wg`) as ONE launch: both depend only on dY (autograd runs them back to back:
 * network/resnet38d.py:17-44 backward), and as one grid the weight-gradient tiles back-fill the data gradient's partly filled last round and the two
 * kinds of tile do not end — and store — at the same time.  Qualifies: bf16, stride-1 single-source data gradient on the 256-tile kernel + a
 * weight gradient on the 256 x 256 phase-pipelined kernel; anything else runs as the two ordinary launches, in that order.  Same results. */
int wseg_conv_bwd_pair(const wseg_conv_desc* dg, const wseg_wgrad_desc* wg, void* stream);
int wseg_conv_bwd_pair_fuses(const wseg_conv_desc* dg, const wseg_wgrad_desc* wg);   /* 1: one grid, 0: two launches, < 0: bad arguments (launches nothing) */

/* ---- weight packing ----------------------------------------------------------------------
 * master f32 [OC][T][IC] -> fwd pack [OCp][T][ICp] and transposed pack [ICp][T][OCp] in `dtype`
 * (zero padded).  Either destination may be NULL.  ic_rot: packed input channel ic is the master's column (ic + ic_rot) % IC
 * (f9.weight's columns [x_s | f8_3 | f8_4] against the feature rows [f8_3 | f8_4 | x_s]: ic_rot = 3, resnet38_contrast.py:53); 0 = none. */
int wseg_pack_weights(const float* master, void* fwd, void* tr, int OC, int T, int IC,
                      int OCp, int ICp, int ic_rot, int dtype, void* stream);
/* the K-concatenated packs of the two-source launches ([W_a[r] | W_b[r]] rows, see wseg_conv_desc.in2) from the per-layer packs, every piece of a
 * step in ONE launch: piece p copies `rows` x `cols16` 16-byte chunks from src + src_off (row stride ld_src) to dst + dst_off (row stride ld_dst), all in
 * 16-byte units; `table` (device, int64) holds per piece {first chunk in the launch, src_off, dst_off, rows, cols16, ld_src, ld_dst}.
 * Replaces torch.cat of the packs (the reference has no counterpart: it runs the skip conv and the last conv of a block as two convolutions,
 * network/resnet38d.py:35-47, 83-97). */
int wseg_copy2d_batch(const void* src, void* dst, const long* table, int npieces, long total_chunks, void* stream);
/* every transposed pack of a training step in ONE launch: layer l is the f32 master [OC][T][IC] at master + off_in[l],
 * written as [IC][T][OC] in `dtype` at out + off_out[l] (element offsets).  `table` (device, int64) holds per layer
 * {index of its first 32x32 tile, off_in, off_out, OC, T, IC}; total_tiles = sum of ceil(OC/32)*ceil(IC/32)*T. */
int wseg_pack_transposed_batch(const float* master, void* out, const long* table, int nlayers, long total_tiles, int dtype, void* stream);
/* the same from the bf16 weight mirror (bf16 -> bf16), `table` counted in 64x64 tiles */
int wseg_pack_transposed_batch_bf16(const void* mirror, void* out, const long* table, int nlayers, long total_tiles, void* stream);
/* Dropout2d scales (resnet38d.py:64,68,86,91; resnet38_contrast.py:14,34) from uniforms u in [0,1):
 * out[i] = u[i] >= p ? 1/(1-p) : 0 with p = p0 for i < split_at, p1 after (one launch for all five masks). */
int wseg_dropout_scale(const float* u, float* out, long total, long split_at, float p0, float p1, void* stream);

/* ---- stem: conv1a (3->64, 3x3, pad 1) + the next block's frozen BN-ReLU -------------------
 * network/resnet38d.py:124,162 (+ :29-30 of b2).  x is the reference's NCHW f32 input;
 * raw = conv1a(x) NHWC, act = relu(raw*scale+shift) NHWC (either may be NULL). */
int wseg_stem_conv(const float* x_nchw, const float* w /*[64][3][3][3] = [oc][ky][kx][ic]*/,
                   const float* scale, const float* shift, void* raw, void* act,
                   int N, int H, int W, int dtype, void* stream);
/* the same with the weights transposed to [27 = (ky*3+kx)*3+ic][64 oc] (f32): the channel pairs become packed FMAs — same products,
 * same order, same results, half the vector instructions (conv1a is frozen: the caller transposes once). */
int wseg_stem_conv_kc(const float* x, const float* w_kc, const float* scale, const float* shift,
                      void* raw, void* act, int N, int H, int W, int dtype, void* stream);

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
 * bwd is the exact adjoint computed by gather (deterministic). plane_mul (nullable) scales plane p;
 * plane_add (nullable, bwd) is a constant added to every d_out element of plane p (the GAP gradient). */
int wseg_resize_planar_fwd(const float* in, float* out, const float* plane_mul, long planes, int ih, int iw, int oh, int ow, int align,
                           int flip_x, int accumulate, void* stream);   /* flip_x: out[..,x] = resized[..,ow-1-x]; accumulate: out += */
int wseg_resize_planar_bwd(const float* d_out, float* d_in, const float* plane_mul, const float* plane_add, long planes, int ih, int iw, int oh, int ow, int align, int accumulate, void* stream);

/* inference post-process, contrast_infer.py:75-98: sum of 8 label-gated CAMs -> clamp, per-class
 * (x-min-1e-5)/(max-min+1e-5) with entries < min+1e-5 zeroed first, argmax against the constant bg score alpha.
 * stats = wseg_plane_stats(sum_cam, 20 planes). */
int wseg_infer_finish(const float* sum_cam, const float* stats, float alpha, float* norm_cam, unsigned char* pred, int npix, void* stream);

/* ---- PCM (network/resnet38_contrast.py:63-75), flash-style, exact-f32 MFMA ---------------------
 * l2norm  : Fh = F/(||F||_2 + 1e-5) over the 192 f9 channels of each pixel row (:70)
 * forward : cam_rv[n][c][j] = sum_i G[i][c] relu(Fh_i.Fh_j) / (sum_i relu(Fh_i.Fh_j) + 1e-5)  (:71-73)
 * backward: gradient w.r.t. Fh only (the CAM input is no_grad in the reference, :41-48). */
int wseg_l2norm_forward(const void* F, int ldf, float* Fh, float* nrm, long rows, int dtype, void* stream);
int wseg_l2norm_backward(const void* F, int ldf, const float* dFh, const float* nrm, void* dF, int lddf, long rows, int dtype, void* stream);
int wseg_pcm_forward(const float* Fh, const float* G, float* cam_rv, float* den, int N, int hw, void* stream);
int wseg_pcm_backward(const float* Fh, const float* G, const float* d_cam_rv, const float* cam_rv, const float* den,
                      float* DN, float* dFh, int N, int hw, void* stream);
/* split-bf16 weight pack for dtype WSEG_F32X3: f32 [.. x K] (K % 32 == 0 per row, numel % 32 == 0) -> per 32-element group
 * [32 hi bf16 | 32 lo bf16] in the same 128 bytes; `src` = a forward pack [OC][T][IC] or a transposed pack [IC][T][OC] in f32. */
int wseg_pack_x3(const float* src, void* dst, long numel, void* stream);
/* f32 [total] -> bf16 planes hi, lo with in = hi + lo to 16-17 bits (16-B aligned, contiguous) */
int wseg_split_bf16(const float* in, void* hi, void* lo, long total, void* stream);
/* bf16-MFMA variants for the bf16 throughput mode (Fb/Gb/DNb = bf16 copies made with wseg_to_bf16) */
int wseg_to_bf16(const float* in, void* out, long total, void* stream);
int wseg_pcm_forward_bf16(const void* Fb, const void* Gb, float* cam_rv, float* den, int N, int hw, void* stream);
/* backward: Gl / DNl = the LOW parts of the split G = Gb + Gl and DN = DNb + DNl (wseg_split_bf16; DNb and DNl are written here).  The gate-channel
 * product of the backward pass is a difference of two nearly cancelling sums and runs in split precision (3 bf16 MFMAs), everything else in bf16. */
int wseg_pcm_backward_bf16(const void* Fb, const void* Gb, const void* Gl, const float* d_cam_rv, const float* cam_rv, const float* den,
                           float* DN, void* DNb, void* DNl, float* dFh, int N, int hw, void* stream);

/* ---- loss step, contrast_train.py:138-395 (forward values + hand-written gradients) -----------
 * All maps planar f32 [N][21][npix]; label20 = float [N][20] multi-hot; loss outputs are device
 * scalars accumulated with atomicAdd (caller zeroes them).
 *  plane_stats        per (n,c): {max relu(U), min relu(U), sum U, argmax, argmin, 0}      (:142, visualization.py:62-66)
 *  cls_loss           mean BCE-with-logits of the GAP logits (:159-160) + its per-pixel gradient as a plane bias
 *  rvmin_values       q = max_{c>=1} U*L and its arg channel (:19-21)
 *  select_kth         k-th order statistic per row by 4-pass radix select + strict sums (torch.topk(...)[0].sum(), :22, :170-171)
 *  rvmin_backward     gradient of adaptive_min_pooling_loss into dU
 *  norm_resize_*      L * bilinear_{S->OS}(max_norm(U)) and its backward incl. the max/min routes (:145-158)
 *  er_ecr_prep        ER sum + gradients, bg = 1-max fg, max_onehot, signed ECR differences (:163-169)
 *  ecr_backward       gradient of the top-K mean (:170-171)
 *  rows_resize_forward  f_proj rows -> [N*oh*ow][128] f32, bilinear align_corners=True (:179)
 *  head_grad_fused    d(head rows) = [relu-masked adjoint of rows_resize(dF) | d_cam_low | 0]
 *  pseudo_label       Q6 normalisation, bg threshold, argmax softmax(.*label) (:186-197)
 *  proto_candidates / proto_merge   per-class top-K (K = npix/8) over the batch, weighted mean, L2 norm (:199-209);
 *                     candidates are what ranks all-gather for global-batch prototypes
 *  nce_sims           L2-normalised features and their similarities to both prototype sets (:245-246, :262, :289)
 *  intra_weights      hard-pixel sampling: random half + similarity rank band per class (:302-331)
 *  nce_loss_grad      cross-prototype, cross-pseudo-label and intra-view InfoNCE + gradient w.r.t. the features (:261-334)
 */
size_t wseg_plane_stats_workspace_bytes(long planes);
int wseg_plane_stats(const float* U, float* stats, long planes, int npix, void* workspace, void* stream);
/* The same map losses evaluated ON THE FLY from the stride-8 maps `low` [planes][h][w] (csrc/maps.hip): the
 * reference's upsampled [N,21,S,S] tensors (resnet38_contrast.py:57-59, 540 MB per view) and their gradients are
 * never materialised; U(y,x) = bilinear(low, align_corners=True) is recomputed by one pinned expression.
 *  up_plane_stats          = plane_stats(U)                      (contrast_train.py:142,155; visualization.py:62-67)
 *  up_rvmin_values         = rvmin_values(U_rv)                  (:16-22)
 *  up_norm_resize_forward  = norm_resize_forward(U)              (:145-158)
 *  resize_adjoint_ones     wvec[y] = sum over the S upsampled rows of their weight on low-res row y (the GAP gradient)
 *  up_maps_backward        d_low[pl] = all gradients of plane pl: max_norm + resize backward of G (NULL: none) with the
 *                          max/min routes, plane_bias[pl]*wvec_y*wvec_x (NULL: none), and the min-pool selection
 *                          (q/argc/res of select_kth, NULL: none; k, coef as in rvmin_backward). */
int wseg_up_plane_stats(const float* low, float* stats, long planes, int h, int w, int S, const float* label20 /* nullable: all planes; else only bg + labelled classes of [N][20] */, void* workspace, void* stream);
int wseg_up_rvmin_values(const float* low, const float* label20, float* q, unsigned char* argc, int N, int h, int w, int S, void* stream);
int wseg_up_norm_resize_forward(const float* low, const float* stats, const float* label20, float* out, int N, int h, int w, int S, int OS, void* stream);
int wseg_resize_adjoint_ones(float* wvec, int h, int S, void* stream);
int wseg_up_maps_backward(const float* G, const float* low, const float* stats, const float* label20, const float* plane_bias,
                          const float* wvec_y, const float* wvec_x, const float* q, const unsigned char* argc, const float* res,
                          int k, float coef, float* d_low, int N, int h, int w, int S, int OS, void* stream);
int wseg_cls_loss(const float* stats, const float* label20, float* loss_out, float* plane_bias, int N, int npix, float coef, void* stream);
int wseg_rvmin_values(const float* U, const float* label20, float* q, unsigned char* argc, int N, int npix, void* stream);
size_t wseg_select_workspace_bytes(int rows);
int wseg_select_kth(const float* vals, int rows, int n, int k, int largest, int use_abs, int relu_vals, float* res, void* workspace, void* stream);
int wseg_select_finish(const float* res, int rows, int k, int relu_vals, float scale, float* loss_out, void* stream);
/* the 8 logged scalars (contrast_train.py:174, 389-395, 401-408) from the step's accumulators acc = [cls1+cls2, (rvmin1+rvmin2)/2, er_sum, ecr, cross,
 * cross2, intra, -]:  out8 = [loss, loss_cls, loss_er, loss_ecr, loss_nce, loss_intra_nce, loss_cross_nce, loss_cross_nce2] */
int wseg_loss_finish(const float* acc, float er_coef, float* out8, void* stream);
int wseg_rvmin_backward(const float* q, const unsigned char* argc, const float* res, const float* label20, float* dU, int N, int npix, int k, float coef, void* stream);
int wseg_norm_resize_forward(const float* U, const float* stats, const float* label20, float* out, int N, int S, int OS, void* stream);
int wseg_norm_resize_backward(const float* G, const float* U, const float* stats, const float* label20, float* dU, int N, int S, int OS, void* stream);
int wseg_er_ecr_prep(const float* c1, const float* c2, const float* r1, const float* r2, float* Gc1, float* Gc2, float* dlt1, float* dlt2,
                     float* er_sum, int N, int npix, float er_coef, void* stream);
int wseg_ecr_backward(const float* dlt, const float* res, float* Gr, int N, int per_row, int k, float coef, void* stream);
int wseg_rows_resize_forward(const void* head, int ld, float* F, int N, int ih, int iw, int oh, int ow, int dtype, void* stream);
int wseg_head_grad_fused(const float* dF, const float* d_cam_low, const void* head, void* d_head, int ld, int N, int ih, int iw, int oh, int ow, int dtype, void* stream);
int wseg_pseudo_label(const float* R, const float* label20, float bg_thr, int* y, float* ncam, int N, int npix, void* stream);
int wseg_proto_candidates(const float* ncam, const float* F, const int* tie_idx, float* cand_val, float* cand_feat, int* cand_const, int N, int npix, int K, void* stream);
int wseg_proto_merge(const float* cand_val, const float* cand_feat, const int* cand_const, float* protos, int world, int K,
                     long rank_stride /* 0: contiguous [world][...] arrays; else elements between consecutive ranks' blocks (one gathered buffer) */, void* stream);
int wseg_nce_sims(const float* F, const float* p_own, const float* p_oth, float* fn, float* nrm, float* S_own, float* S_oth, int P, void* stream);
/* S_own: ld_s == 21: the [P,21] similarity table (the own-class entry is S_own[p*21 + y[p]]); ld_s == 1: one own-class similarity per pixel
 * (row 1 of the records below) */
int wseg_intra_weights(const int* y, const float* S_own, int ld_s, const float* rkey, const unsigned char* rand_flag, float* w, int P, void* stream);
/* the same sampling over the GLOBAL batch under data parallelism (the reference runs :302-334 on the gathered batch):
 * intra_pack writes this rank's records rec[3][P] = {label (int bits), own-class similarity, random key}; after an
 * all-gather rank r's block lies at rec + r*rank_stride and intra_weights_global returns this rank's weights, multiplied by `scale`
 * (= ranks when the gradient all-reduce averages). */
int wseg_intra_pack(const int* y, const float* S_own, const float* rkey, float* rec, int P, void* stream);
int wseg_intra_weights_global(const float* rec, float* w, int P, int ranks, int own_rank, float scale, long rank_stride, void* stream);
int wseg_nce_loss_grad(const float* fn, const float* nrm, const float* S_own, const float* S_oth, const int* y_own, const int* y_oth,
                       const float* w_intra, const float* p_own, const float* p_oth, float* dF, float* sums, int P,
                       float coef_cross, float coef_intra, void* stream);

/* ---- fused pixel-to-prototype contrast: the product path of contrast_train.py:245-334 (nce_sims / nce_loss_grad above are its
 * unfused reference formulation).  Two launches per step serve BOTH views; the normalised features and the [P,21] similarity rows
 * never reach HBM: every launch reads the raw features once (P*128*4 B) and writes only records (12 B / pixel) or dF (P*128*4 B).
 *   nce_records : rec[3][P] = {label (int bits), similarity of the pixel to its OWN class's prototype of p_own, rkey (when given)} — what the
 *                 hard-pixel sampling (intra_weights / the all-gather + intra_weights_global) needs; fields used: F, p_own, y_own, rkey, rec
 *   nce_fused   : similarities (exact-f32 MFMA) -> cross-prototype, cross-pseudo-label and intra-view InfoNCE (tau = 0.1) -> dF, and
 *                 sums[0..2] += {cross, cross2, intra} * coef; fields used: F, p_own, p_oth, y_own, y_oth, w_intra, dF
 * Per view v the "other" set is the other view's prototypes / labels (views[0] <-> views[1]); P rows each, D = 128, C = 21. */
typedef struct {
  const float* F;        /* [P][128] un-normalised projected features (f_proj resized to 16x16, rows n,i,j) */
  const float* p_own;    /* [21][128] this view's prototypes */
  const float* p_oth;    /* [21][128] the other view's prototypes */
  const int32_t* y_own;  /* [P] this view's pseudo-labels */
  const int32_t* y_oth;  /* [P] the other view's pseudo-labels */
  const float* w_intra;  /* [P] hard-pixel weights (intra_weights*) */
  const float* rkey;     /* [P] random keys copied into the records (nullable) */
  float* rec;            /* [3][P] out (nce_records) */
  float* dF;             /* [P][128] out (nce_fused) */
} wseg_nce_view;
/* nce_records, split_bf16 = 0: exact-f32 MFMA (the fp32 parity mode: the record's similarity is bit-identical to the one nce_fused uses);
 * 1: split-bf16 products (hi.hi + lo.hi + hi.lo on the bf16 MFMA, 16-17 operand bits; the bf16 and bf16x3 modes — the similarity only RANKS pixels) */
int wseg_nce_records(const wseg_nce_view* views, int nviews, int P, int split_bf16, void* stream);
int wseg_nce_fused(const wseg_nce_view* views, int nviews, int P, float coef_cross, float coef_intra, float* sums /* [3], accumulated */, void* stream);

/* ---- training augmentation on the device (contrast_train.py:64-75, tool/imutils.py:6-67, network/resnet38d.py:104-118): a batch of
 * DECODED uint8 HWC images -> RandomResizeLong (Pillow's two-pass 8-bit bicubic, coefficient tables made by the host) -> horizontal flip ->
 * ColorJitter ops in the drawn order (Pillow's ImageEnhance / HSV arithmetic) -> normalise (3 x 256 table) -> RandomCrop placement -> CHW f32.
 * The host draws the random parameters (wseg_amd/augment.py, same draws in the same order as the host pipeline) and fills one
 * descriptor per image; all pointers are device pointers; `tmp` / `img` are scratch of H*rw*3 and rh*rw*3 bytes. */
typedef struct {
  const uint8_t* src; int32_t H, W;                   /* decoded image [H][W][3] */
  int32_t rh, rw;                                     /* size after RandomResizeLong */
  const int32_t* xb; const int32_t* xk; int32_t xks;  /* horizontal pass: bounds [rw][2] = (first source column, count), coefficients [rw][xks] (22 fractional bits) */
  const int32_t* yb; const int32_t* yk; int32_t yks;  /* vertical pass, over rh */
  uint8_t* tmp; uint8_t* img;
  int32_t flip;
  int32_t op[4];                                      /* colour ops in execution order: 0 brightness, 1 contrast, 2 saturation, 3 hue, -1 none */
  float factor[4];                                    /* enhancement factor of ops 0..2 */
  int32_t hue_shift;                                  /* added to the uint8 hue (mod 256) */
  int32_t cont_top, cont_left, img_top, img_left, ch, cw;   /* RandomCrop: [ch][cw] pixels from (img_top, img_left) land at (cont_top, cont_left) */
  float* out;                                         /* [3][crop][crop] */
} wseg_aug_desc;
size_t wseg_sizeof_aug_desc(void);
int wseg_augment_batch(const wseg_aug_desc* descs_dev, int n, int max_pixels /* largest H*rw or rh*rw of the batch */, const float* lut /* [3][256] */,
                       int crop, unsigned long long* lum_sums /* scratch [n][4] */, void* stream);

/* ---- fused SGD step (tool/torchutils.py:23-33 -> torch.optim.SGD.step) ------------------------
 * One pass over the flat buffers: d = g*grad_scale + wd*p; buf = first ? d : momentum*buf + d;
 * p -= lr*buf.  Segments [begin,end) carry the per-group lr / weight_decay (contrast_train.py:91-96).
 * bf16_mirror (nullable): also receives the updated weights in bf16 (the next step's forward packs). */
int wseg_sgd_step(float* params, const float* grads, float* momentum_buf, long numel,
                  const long* seg_begin, const long* seg_end, const float* seg_lr, const float* seg_wd, int nseg,
                  float momentum, float grad_scale, int first_step, void* bf16_mirror, void* stream);

#ifdef __cplusplus
}
#endif
#endif
