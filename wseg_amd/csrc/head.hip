// head.hip — small HBM-bound kernels around the CAM head (network/resnet38_contrast.py:34-59):
// split the fused head GEMM output into planar CAM logits, the no_grad CAM gate that feeds PCM,
// the 3 image channels of the PCM feature, and planar bilinear resize (forward + exact gather
// backward).  All maps with 21 channels are kept planar f32 ([N][21][h*w]): every consumer reduces
// per (n, class).
#include "common.h"

namespace {

// ---- head GEMM rows [pixels][ld] (cols 0..20 = fc8 logits) -> cam_low [N][21][hw] + per-(n,c) max of relu
template <int DT>
__global__ void head_split_kernel(const void* __restrict__ head, int ld, int c0, float* __restrict__ cam, float* __restrict__ cmax, int hw, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over N*hw; a block never straddles...
  const long n = idx / hw;                                          // (it may: handled by per-thread atomics below)
  const int p = (int)(idx - n * hw);
  const bool ok = idx < total;
  for (int c = 0; c < 21; ++c) {
    float v = 0.f;
    if (ok) {
      v = elem<DT>::ld(head, (size_t)idx * ld + c0 + c);
      cam[((size_t)n * 21 + c) * hw + p] = v;
    }
    float m = ok ? fmaxf(v, 0.f) : 0.f;
    // wave max, then one atomic per wave when the wave lies inside one image
    const int n_first = __shfl((int)n, 0, 64), n_last = __shfl((int)n, 63, 64);
    if (n_first == n_last) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      if ((threadIdx.x & 63) == 0 && ok) atomicMax(reinterpret_cast<int*>(&cmax[n * 21 + c]), __float_as_int(m));
    } else if (ok) {
      atomicMax(reinterpret_cast<int*>(&cmax[n * 21 + c]), __float_as_int(m));
    }
  }
}

// ---- no_grad CAM gate, resnet38_contrast.py:41-48 -> G [N*hw][32] (col 21 = 1, cols 22..31 = 0)
__global__ void cam_gate_kernel(const float* __restrict__ cam, const float* __restrict__ cmax, float* __restrict__ G, int hw, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long n = idx / hw; const int p = (int)(idx - n * hw);
  float v[21];
  float fgmax = -INFINITY;
  for (int c = 0; c < 21; ++c) {
    const float d = fmaxf(cam[((size_t)n * 21 + c) * hw + p], 0.f);
    const float m = cmax[n * 21 + c] + 1e-5f;
    v[c] = fmaxf(d - 1e-5f, 0.f) / m;
    if (c >= 1) fgmax = fmaxf(fgmax, v[c]);
  }
  v[0] = 1.f - fgmax;
  float* o = G + idx * 32;
  o[0] = v[0];
  for (int c = 1; c < 21; ++c) o[c] = v[c] < fgmax ? 0.f : v[c];
  o[21] = 1.f;
  for (int c = 22; c < 32; ++c) o[c] = 0.f;
}

// ---- torch-compatible bilinear source index (area_pixel_compute_source_index)
__device__ __forceinline__ void src_index(int o, float scale, bool align, int in_size, int& i0, int& i1, float& f) {
  float s = align ? scale * o : fmaxf(scale * (o + 0.5f) - 0.5f, 0.f);
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  f = s - i0;
}
__host__ __device__ inline float resize_scale(int in_size, int out_size, bool align) {
  if (align) return out_size > 1 ? (float)(in_size - 1) / (out_size - 1) : 0.f;
  return (float)in_size / out_size;
}

// ---- x_s = bilinear(x -> h x w, align_corners=True) into feature rows [N*hw][ld] cols 0..2; zero cols [c_zero0, ld)
template <int DT>
__global__ void pcm_xs_kernel(const float* __restrict__ x, void* __restrict__ feat, int ld, int c_xs, int c_end,
                              int H, int W, int h, int w, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int hw = h * w;
  const long n = idx / hw; const int p = (int)(idx - n * hw);
  const int oy = p / w, ox = p - oy * w;
  int y0, y1, x0, x1; float fy, fx;
  src_index(oy, resize_scale(H, h, true), true, H, y0, y1, fy);
  src_index(ox, resize_scale(W, w, true), true, W, x0, x1, fx);
  for (int c = 0; c < 3; ++c) {
    const float* pl = x + ((size_t)n * 3 + c) * H * W;
    const float v = (1.f - fy) * ((1.f - fx) * pl[(size_t)y0 * W + x0] + fx * pl[(size_t)y0 * W + x1]) +
                    fy * ((1.f - fx) * pl[(size_t)y1 * W + x0] + fx * pl[(size_t)y1 * W + x1]);
    elem<DT>::st(feat, (size_t)idx * ld + c_xs + c, v);
  }
  for (int c = c_xs + 3; c < c_end; ++c) elem<DT>::st(feat, (size_t)idx * ld + c, 0.f);
}

// ---- planar bilinear resize [planes][ih][iw] -> [planes][oh][ow], optional per-plane multiplier
__global__ void resize_planar_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ plane_mul,
                                         int ih, int iw, int oh, int ow, int align, int flip_x, int accumulate, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int ox = (int)(idx % ow); const long r = idx / ow;
  if (flip_x) ox = ow - 1 - ox;                            // out[.., x] = resized[.., ow-1-x]  (np.flip(cam, -1))
  const int oy = (int)(r % oh); const long pl = r / oh;
  int y0, y1, x0, x1; float fy, fx;
  src_index(oy, resize_scale(ih, oh, align), align, ih, y0, y1, fy);
  src_index(ox, resize_scale(iw, ow, align), align, iw, x0, x1, fx);
  const float* p = in + (size_t)pl * ih * iw;
  float v = (1.f - fy) * ((1.f - fx) * p[(size_t)y0 * iw + x0] + fx * p[(size_t)y0 * iw + x1]) +
            fy * ((1.f - fx) * p[(size_t)y1 * iw + x0] + fx * p[(size_t)y1 * iw + x1]);
  if (plane_mul) v *= plane_mul[pl];
  if (accumulate) out[idx] += v; else out[idx] = v;
}

// ---- inference post-process (contrast_infer.py:75-98): clamp, per-class min/max normalise, argmax vs alpha
//      stats = plane_stats of sum_cam (max/min of relu == clamp-then-max/min)
__global__ void infer_finish_kernel(const float* __restrict__ sum_cam, const float* __restrict__ stats, float alpha,
                                    float* __restrict__ norm_cam, unsigned char* __restrict__ pred, int npix) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  float best = alpha; int bc = 0;
  for (int c = 0; c < 20; ++c) {
    float v = fmaxf(sum_cam[(size_t)c * npix + p], 0.f);
    const float mx = stats[c * 6 + 0], mn = stats[c * 6 + 1];
    if (v < mn + 1e-5f) v = 0.f;
    v = (v - mn - 1e-5f) / (mx - mn + 1e-5f);
    norm_cam[(size_t)c * npix + p] = v;
    if (v > best) { best = v; bc = c + 1; }
  }
  pred[p] = (unsigned char)bc;
}

// exact adjoint by GATHER (deterministic, no atomics): d_in[y][x] = sum over outputs that touch it
__global__ void resize_planar_bwd_kernel(const float* __restrict__ d_out, float* __restrict__ d_in, const float* __restrict__ plane_mul,
                                         const float* __restrict__ plane_add, int ih, int iw, int oh, int ow, int align, int accumulate, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int x = (int)(idx % iw); const long r = idx / iw;
  const int y = (int)(r % ih); const long pl = r / ih;
  const float sy = resize_scale(ih, oh, align), sx = resize_scale(iw, ow, align);
  // candidate output range: source coordinate within (y-1, y+1)
  int oy_lo, oy_hi, ox_lo, ox_hi;
  if (sy > 0.f) { oy_lo = max(0, (int)floorf((y - 1) / sy) - 1); oy_hi = min(oh - 1, (int)ceilf((y + 1) / sy) + 1); } else { oy_lo = 0; oy_hi = oh - 1; }
  if (sx > 0.f) { ox_lo = max(0, (int)floorf((x - 1) / sx) - 1); ox_hi = min(ow - 1, (int)ceilf((x + 1) / sx) + 1); } else { ox_lo = 0; ox_hi = ow - 1; }
  const float* g = d_out + (size_t)pl * oh * ow;
  const float gadd = plane_add ? plane_add[pl] : 0.f;      // a constant added to every d_out element of the plane
  float acc = 0.f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    int y0, y1; float fy;
    src_index(oy, sy, align, ih, y0, y1, fy);
    const float wy = (y == y0 ? 1.f - fy : 0.f) + (y == y1 ? fy : 0.f);
    if (wy == 0.f) continue;
    float row = 0.f;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      int x0, x1; float fx;
      src_index(ox, sx, align, iw, x0, x1, fx);
      const float wx = (x == x0 ? 1.f - fx : 0.f) + (x == x1 ? fx : 0.f);
      if (wx != 0.f) row += wx * (g[(size_t)oy * ow + ox] + gadd);
    }
    acc += wy * row;
  }
  if (plane_mul) acc *= plane_mul[pl];
  if (accumulate) d_in[idx] += acc; else d_in[idx] = acc;
}

// ---- planar f32 [N][C][hw] <-> pixel rows [N*hw][ld] (cols c0..c0+C-1) in the mode dtype
template <int DT>
__global__ void planar_to_rows_kernel(const float* __restrict__ pl, void* __restrict__ rows, int ld, int c0, int C, int hw, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long n = idx / hw; const int p = (int)(idx - n * hw);
  for (int c = 0; c < C; ++c) elem<DT>::st(rows, (size_t)idx * ld + c0 + c, pl[((size_t)n * C + c) * hw + p]);
}

// ---- d_head rows [N*hw][ld]: cols [0,128) = d_f_proj * (f_proj > 0), cols [128,149) = d_cam_low, rest 0
template <int DT>
__global__ void head_grad_rows_kernel(const float* __restrict__ d_fproj, const float* __restrict__ d_cam, const void* __restrict__ head,
                                      void* __restrict__ d_head, int ld, int hw, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over N*hw*(ld/8)
  if (idx >= total) return;
  const int v8 = ld / 8;
  const long pix = idx / v8; const int c8 = (int)(idx - pix * v8) * 8;
  const long n = pix / hw; const int p = (int)(pix - n * hw);
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 + e;
    float g = 0.f;
    if (c < 128) {
      if (d_fproj) g = elem<DT>::ld(head, (size_t)pix * ld + c) > 0.f ? d_fproj[((size_t)n * 128 + c) * hw + p] : 0.f;
    } else if (c < 149) {
      if (d_cam) g = d_cam[((size_t)n * 21 + (c - 128)) * hw + p];
    }
    o[e] = g;
  }
  store8<DT>(d_head, (size_t)pix * ld + c8, o);
}

}  // namespace

#define GRID1(total) dim3((unsigned)(((total) + 255) / 256)), dim3(256)

extern "C" int wseg_head_split(const void* head, int ld, int c0, float* cam_low, float* cmax, int N, int hw, int dtype, void* stream) {
  WSEG_CHECK(head && cam_low && cmax && N > 0 && hw > 0, "head_split: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)N * hw;
  (void)hipMemsetAsync(cmax, 0, sizeof(float) * N * 21, s);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (dtype == WSEG_BF16) hipLaunchKernelGGL(head_split_kernel<WSEG_BF16>, dim3(blocks), dim3(256), 0, s, head, ld, c0, cam_low, cmax, hw, total);
  else hipLaunchKernelGGL(head_split_kernel<WSEG_F32>, dim3(blocks), dim3(256), 0, s, head, ld, c0, cam_low, cmax, hw, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_cam_gate(const float* cam_low, const float* cmax, float* G, int N, int hw, void* stream) {
  WSEG_CHECK(cam_low && cmax && G && N > 0 && hw > 0, "cam_gate: bad arguments");
  const long total = (long)N * hw;
  hipLaunchKernelGGL(cam_gate_kernel, GRID1(total), 0, (hipStream_t)stream, cam_low, cmax, G, hw, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_pcm_xs(const float* x_nchw, void* feat, int ld, int c_xs, int c_end, int N, int H, int W, int h, int w, int dtype, void* stream) {
  WSEG_CHECK(x_nchw && feat && c_xs >= 0 && c_xs + 3 <= c_end && c_end <= ld, "pcm_xs: bad arguments");
  const long total = (long)N * h * w;
  if (dtype == WSEG_BF16) hipLaunchKernelGGL(pcm_xs_kernel<WSEG_BF16>, GRID1(total), 0, (hipStream_t)stream, x_nchw, feat, ld, c_xs, c_end, H, W, h, w, total);
  else hipLaunchKernelGGL(pcm_xs_kernel<WSEG_F32>, GRID1(total), 0, (hipStream_t)stream, x_nchw, feat, ld, c_xs, c_end, H, W, h, w, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_resize_planar_fwd(const float* in, float* out, const float* plane_mul, long planes, int ih, int iw, int oh, int ow, int align,
                                      int flip_x, int accumulate, void* stream) {
  WSEG_CHECK(in && out && planes > 0 && ih > 0 && iw > 0 && oh > 0 && ow > 0, "resize_planar_fwd: bad arguments");
  const long total = planes * oh * ow;
  hipLaunchKernelGGL(resize_planar_fwd_kernel, GRID1(total), 0, (hipStream_t)stream, in, out, plane_mul, ih, iw, oh, ow, align, flip_x, accumulate, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_resize_planar_bwd(const float* d_out, float* d_in, const float* plane_mul, const float* plane_add, long planes, int ih, int iw, int oh, int ow, int align, int accumulate, void* stream) {
  WSEG_CHECK(d_out && d_in && planes > 0 && ih > 0 && iw > 0 && oh > 0 && ow > 0, "resize_planar_bwd: bad arguments");
  const long total = planes * ih * iw;
  hipLaunchKernelGGL(resize_planar_bwd_kernel, GRID1(total), 0, (hipStream_t)stream, d_out, d_in, plane_mul, plane_add, ih, iw, oh, ow, align, accumulate, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_planar_to_rows(const float* planar, void* rows, int ld, int c0, int C, int N, int hw, int dtype, void* stream) {
  WSEG_CHECK(planar && rows && C > 0 && c0 >= 0 && c0 + C <= ld, "planar_to_rows: bad arguments");
  const long total = (long)N * hw;
  if (dtype == WSEG_BF16) hipLaunchKernelGGL(planar_to_rows_kernel<WSEG_BF16>, GRID1(total), 0, (hipStream_t)stream, planar, rows, ld, c0, C, hw, total);
  else hipLaunchKernelGGL(planar_to_rows_kernel<WSEG_F32>, GRID1(total), 0, (hipStream_t)stream, planar, rows, ld, c0, C, hw, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_head_grad_rows(const float* d_fproj, const float* d_cam_low, const void* head, void* d_head, int ld, int N, int hw, int dtype, void* stream) {
  WSEG_CHECK(head && d_head && ld % 8 == 0 && ld >= 152, "head_grad_rows: bad arguments");
  const long total = (long)N * hw * (ld / 8);
  if (dtype == WSEG_BF16) hipLaunchKernelGGL(head_grad_rows_kernel<WSEG_BF16>, GRID1(total), 0, (hipStream_t)stream, d_fproj, d_cam_low, head, d_head, ld, hw, total);
  else hipLaunchKernelGGL(head_grad_rows_kernel<WSEG_F32>, GRID1(total), 0, (hipStream_t)stream, d_fproj, d_cam_low, head, d_head, ld, hw, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_infer_finish(const float* sum_cam, const float* stats, float alpha, float* norm_cam, unsigned char* pred, int npix, void* stream) {
  WSEG_CHECK(sum_cam && stats && norm_cam && pred && npix > 0, "infer_finish: bad arguments");
  hipLaunchKernelGGL(infer_finish_kernel, GRID1((long)npix), 0, (hipStream_t)stream, sum_cam, stats, alpha, norm_cam, pred, npix);
  WSEG_LAUNCH_CHECK();
  return 0;
}
