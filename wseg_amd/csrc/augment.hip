// augment.hip — the training augmentation of contrast_train.py:64-75 on the device.
//
// The reference augments on the host, per image, inside DataLoader workers:
//   RandomResizeLong(448, 768) [PIL bicubic] -> RandomHorizontalFlip -> ColorJitter(0.3, 0.3, 0.3, 0.1) -> np.asarray ->
//   Normalize -> RandomCrop(448) -> HWC_to_CHW                       (tool/imutils.py:6-67, network/resnet38d.py:104-118)
// which costs 45-60 ms of a host core per image — the limit of an end-to-end run beside a 437 images/s training step.
// Here the host only decodes the JPEG and draws the random parameters (same draws, same order: wseg_amd/augment.py); a batch of
// decoded uint8 images goes through 11 launches (blockIdx.y = image):
//   resize_h, resize_v   Pillow's two-pass 8-bit resampler as it computes it: integer coefficients of 22 fractional bits (made on
//                        the host in float64 exactly as Resample.c makes them), rounding constant, clip, uint8 intermediate
//   4 x (lum_sum, op)    ImageEnhance.Brightness / Contrast / Color = Blend.c's float arithmetic incl. its truncation and its
//                        extrapolation clip; contrast needs the image's mean luminance (Convert.c rgb2l fixed point);
//                        hue = Convert.c rgb2hsv / hsv2rgb with the float / double mix of that file, uint8 hue wrap
//   finish               flip, 256-entry normalisation table per channel, crop / zero-pad placement, CHW f32
// Every stage reproduces the host pipeline of wseg_amd/data.py BIT FOR BIT (tests/test_gpu_augment.py); the reference's own
// torchvision transforms are not importable offline, so parity with them stays unpinned.
#include <algorithm>
#include "common.h"

namespace {

__device__ __forceinline__ unsigned char clip8i(long long v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// one pass of Pillow's ImagingResampleHorizontal_8bpc / Vertical_8bpc: out = clip8((2^21 + sum_j in[min + j] * k[j]) >> 22)
template <int VERTICAL>
__global__ void aug_resize_kernel(const wseg_aug_desc* __restrict__ descs) {
  const wseg_aug_desc d = descs[blockIdx.y];
  const int ow = d.rw, oh = VERTICAL ? d.rh : d.H;           // this pass's output size
  const long total = (long)ow * oh;
  const unsigned char* in = VERTICAL ? d.tmp : d.src;
  unsigned char* out = VERTICAL ? d.img : d.tmp;
  const int in_w = VERTICAL ? d.rw : d.W;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int y = (int)(idx / ow), x = (int)(idx - (long)y * ow);
    const int o = VERTICAL ? y : x;
    const int* b = (VERTICAL ? d.yb : d.xb) + 2 * o;
    const int ks = VERTICAL ? d.yks : d.xks;
    const int* k = (VERTICAL ? d.yk : d.xk) + (long)o * ks;
    const int lo = b[0], cnt = b[1];
    long long s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int j = 0; j < cnt; ++j) {
      const unsigned char* p = VERTICAL ? in + ((long)(lo + j) * in_w + x) * 3 : in + ((long)y * in_w + lo + j) * 3;
      const long long kk = k[j];
      s0 += p[0] * kk; s1 += p[1] * kk; s2 += p[2] * kk;
    }
    unsigned char* q = out + idx * 3;
    q[0] = clip8i(s0 >> 22); q[1] = clip8i(s1 >> 22); q[2] = clip8i(s2 >> 22);
  }
}

__device__ __forceinline__ int lum_of(const unsigned char* p) {      // Convert.c rgb2l
  return (int)(((unsigned)p[0] * 19595u + (unsigned)p[1] * 38470u + (unsigned)p[2] * 7471u + 0x8000u) >> 16);
}

// sum of the luminance of the current image, for the images whose colour op of this stage is the contrast
__global__ void aug_lum_sum_kernel(const wseg_aug_desc* __restrict__ descs, int stage, unsigned long long* __restrict__ sums) {
  const wseg_aug_desc d = descs[blockIdx.y];
  if (d.op[stage] != 1) return;
  const long total = (long)d.rw * d.rh;
  unsigned long long s = 0;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) s += lum_of(d.img + idx * 3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(&sums[blockIdx.y * 4 + stage], s);
}

// Blend.c: out = degenerate + alpha * (image - degenerate) in C float arithmetic (separate multiply and add), truncated;
// outside [0, 1] the result is clipped to [0, 255] first
__device__ __forceinline__ unsigned char blend_u8(int deg, int img, float alpha) {
  const float t = __fadd_rn((float)deg, __fmul_rn(alpha, (float)(img - deg)));
  if (alpha >= 0.f && alpha <= 1.f) return (unsigned char)(int)t;
  if (t <= 0.f) return 0;
  if (t >= 255.f) return 255;
  return (unsigned char)t;
}

// Convert.c rgb2hsv_row / hsv2rgb (float variables, double constants)
__device__ __forceinline__ void rgb2hsv_u8(const unsigned char* in, unsigned char* out) {
  const int r = in[0], g = in[1], b = in[2];
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  unsigned char uh = 0, us = 0;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
    float h;
    if (r == maxc) h = __fsub_rn(bc, gc);
    else if (g == maxc) h = (float)__dsub_rn(__dadd_rn(2.0, (double)rc), (double)bc);
    else h = (float)__dsub_rn(__dadd_rn(4.0, (double)gc), (double)rc);
    const double t = __dadd_rn(__ddiv_rn((double)h, 6.0), 1.0);      // in [1/3, 2): fmod(t, 1.0) = t - floor(t), exact
    h = (float)(t - floor(t));
    int ih = (int)__dmul_rn((double)h, 255.0), is = (int)__dmul_rn((double)s, 255.0);
    uh = (unsigned char)min(255, max(0, ih)); us = (unsigned char)min(255, max(0, is));
  }
  out[0] = uh; out[1] = us; out[2] = (unsigned char)maxc;
}
__device__ __forceinline__ void hsv2rgb_u8(const unsigned char* in, unsigned char* out) {
  const int h = in[0], s = in[1], v = in[2];
  if (s == 0) { out[0] = out[1] = out[2] = (unsigned char)v; return; }
  const double t6 = __ddiv_rn(__dmul_rn((double)(float)h, 6.0), 255.0);
  const int i = (int)floor(t6);
  const float f = (float)__dsub_rn(t6, (double)(float)i);
  const float fs = (float)__ddiv_rn((double)(float)s, 255.0);
  const double vf = (double)(float)v;
  auto cround = [](double a) { return floor(a + 0.5); };             // C round() on non-negative values
  const int p = (int)cround(__dmul_rn(vf, __dsub_rn(1.0, (double)fs)));
  const int q = (int)cround(__dmul_rn(vf, __dsub_rn(1.0, __dmul_rn((double)fs, (double)f))));
  const int t = (int)cround(__dmul_rn(vf, __dsub_rn(1.0, __dmul_rn((double)fs, __dsub_rn(1.0, (double)f)))));
  const unsigned char up = (unsigned char)min(255, max(0, p)), uq = (unsigned char)min(255, max(0, q)), ut = (unsigned char)min(255, max(0, t));
  const unsigned char uv = (unsigned char)v;
  switch (i % 6) {
    case 0: out[0] = uv; out[1] = ut; out[2] = up; break;
    case 1: out[0] = uq; out[1] = uv; out[2] = up; break;
    case 2: out[0] = up; out[1] = uv; out[2] = ut; break;
    case 3: out[0] = up; out[1] = uq; out[2] = uv; break;
    case 4: out[0] = ut; out[1] = up; out[2] = uv; break;
    default: out[0] = uv; out[1] = up; out[2] = uq; break;
  }
}

// colour op of one stage, in place: 0 brightness, 1 contrast, 2 saturation (ImageEnhance.Color), 3 hue shift, -1 none
__global__ void aug_color_kernel(const wseg_aug_desc* __restrict__ descs, int stage, const unsigned long long* __restrict__ sums) {
  const wseg_aug_desc d = descs[blockIdx.y];
  const int op = d.op[stage];
  if (op < 0) return;
  const long total = (long)d.rw * d.rh;
  const float alpha = d.factor[stage];
  int mean = 0;
  if (op == 1) mean = (int)(__ddiv_rn((double)sums[blockIdx.y * 4 + stage], (double)total) + 0.5);   // int(Stat(L).mean[0] + 0.5)
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    unsigned char* p = d.img + idx * 3;
    if (op == 3) {
      unsigned char hsv[3];
      rgb2hsv_u8(p, hsv);
      hsv[0] = (unsigned char)((((int)hsv[0] + d.hue_shift) % 256 + 256) % 256);
      hsv2rgb_u8(hsv, p);
    } else {
      const int deg = op == 0 ? 0 : (op == 1 ? mean : lum_of(p));
      const unsigned char r = blend_u8(deg, p[0], alpha), g = blend_u8(deg, p[1], alpha), b = blend_u8(deg, p[2], alpha);
      p[0] = r; p[1] = g; p[2] = b;
    }
  }
}

// flip, normalise (256-entry table per channel = float32((v / 255. - mean) / std) as numpy computes it), crop / pad, CHW
__global__ void aug_finish_kernel(const wseg_aug_desc* __restrict__ descs, const float* __restrict__ lut, int crop) {
  const wseg_aug_desc d = descs[blockIdx.y];
  const long total = (long)crop * crop;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int y = (int)(idx / crop), x = (int)(idx - (long)y * crop);
    const int cy = y - d.cont_top, cx = x - d.cont_left;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (cy >= 0 && cy < d.ch && cx >= 0 && cx < d.cw) {
      const int ry = cy + d.img_top;
      int rx = cx + d.img_left;
      if (d.flip) rx = d.rw - 1 - rx;
      const unsigned char* p = d.img + ((long)ry * d.rw + rx) * 3;
      v0 = lut[p[0]]; v1 = lut[256 + p[1]]; v2 = lut[512 + p[2]];
    }
    d.out[idx] = v0; d.out[total + idx] = v1; d.out[2 * total + idx] = v2;
  }
}

}  // namespace

extern "C" size_t wseg_sizeof_aug_desc(void) { return sizeof(wseg_aug_desc); }

extern "C" int wseg_augment_batch(const wseg_aug_desc* descs_dev, int n, int max_pixels, const float* lut, int crop,
                                  unsigned long long* lum_sums, void* stream) {
  WSEG_CHECK(descs_dev && n > 0 && max_pixels > 0 && lut && crop > 0 && lum_sums, "augment_batch: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)std::min(1024, (max_pixels + 255) / 256), n), blk(256);
  (void)hipMemsetAsync(lum_sums, 0, sizeof(unsigned long long) * 4 * n, s);
  hipLaunchKernelGGL(aug_resize_kernel<0>, grid, blk, 0, s, descs_dev);
  hipLaunchKernelGGL(aug_resize_kernel<1>, grid, blk, 0, s, descs_dev);
  for (int stage = 0; stage < 4; ++stage) {
    hipLaunchKernelGGL(aug_lum_sum_kernel, grid, blk, 0, s, descs_dev, stage, lum_sums);
    hipLaunchKernelGGL(aug_color_kernel, grid, blk, 0, s, descs_dev, stage, (const unsigned long long*)lum_sums);
  }
  hipLaunchKernelGGL(aug_finish_kernel, dim3((unsigned)((crop * crop + 255) / 256), n), blk, 0, s, descs_dev, lut, crop);
  WSEG_LAUNCH_CHECK();
  return 0;
}
