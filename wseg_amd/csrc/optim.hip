// optim.hip — fused SGD step of PolyOptimizer (tool/torchutils.py:23-33 -> torch.optim.SGD.step) on the
// flat parameter / gradient / momentum buffers: one HBM-bound pass, 16 B per lane.
//   d = g * gscale + wd * p ;  buf = first ? d : mom * buf + d ;  p -= lr * buf
// (dampening 0, nesterov off; `mom` is 5e-4 because of the reference's positional-argument quirk.)
// Segments carry the per-group lr / weight_decay (contrast_train.py:91-96).
#include <algorithm>
#include "common.h"

namespace {

struct Seg { long begin, end; float lr, wd; };
constexpr int MAX_SEG = 8;
struct SegTable { Seg s[MAX_SEG]; int n; };

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  const SegTable tab, float mom, float gscale, int first, long total4,
                                                  bf16_t* __restrict__ mirror) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long e0 = i * 4;
    float lr = 0.f, wd = 0.f;
#pragma unroll
    for (int k = 0; k < MAX_SEG; ++k)
      if (k < tab.n && e0 >= tab.s[k].begin && e0 < tab.s[k].end) { lr = tab.s[k].lr; wd = tab.s[k].wd; }
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 bv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(buf)[i];
    float pe[4] = {pv.x, pv.y, pv.z, pv.w};
    const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
    float be[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d = ge[k] * gscale;
      if (wd != 0.f) d = fmaf(wd, pe[k], d);
      be[k] = first ? d : fmaf(mom, be[k], d);
      pe[k] = fmaf(-lr, be[k], pe[k]);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    reinterpret_cast<float4*>(buf)[i] = make_float4(be[0], be[1], be[2], be[3]);
    if (mirror) {                                  // bf16 copy of the updated weights (next step's forward packs)
      const unsigned lo = (unsigned)f32_to_bf16(pe[0]) | ((unsigned)f32_to_bf16(pe[1]) << 16);
      const unsigned hi = (unsigned)f32_to_bf16(pe[2]) | ((unsigned)f32_to_bf16(pe[3]) << 16);
      reinterpret_cast<uint2*>(mirror)[i] = make_uint2(lo, hi);
    }
  }
}

}  // namespace

extern "C" int wseg_sgd_step(float* params, const float* grads, float* momentum_buf, long numel,
                             const long* seg_begin, const long* seg_end, const float* seg_lr, const float* seg_wd, int nseg,
                             float momentum, float grad_scale, int first_step, void* bf16_mirror, void* stream) {
  WSEG_CHECK(params && grads && momentum_buf && numel > 0, "sgd_step: null pointer");
  WSEG_CHECK(nseg >= 1 && nseg <= MAX_SEG, "sgd_step: 1..%d segments", MAX_SEG);
  WSEG_CHECK(numel % 4 == 0, "sgd_step: numel must be a multiple of 4");
  SegTable tab;
  tab.n = nseg;
  for (int k = 0; k < nseg; ++k) {
    WSEG_CHECK(seg_begin[k] % 4 == 0 && seg_end[k] % 4 == 0, "sgd_step: segment bounds must be multiples of 4");
    tab.s[k] = Seg{seg_begin[k], seg_end[k], seg_lr[k], seg_wd[k]};
  }
  const long total4 = numel / 4;
  const int blocks = (int)std::min<long>((total4 + 255) / 256, 8192);
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, momentum_buf, tab, momentum, grad_scale, first_step, total4, (bf16_t*)bf16_mirror);
  WSEG_LAUNCH_CHECK();
  return 0;
}
