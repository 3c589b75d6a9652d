// maps.hip — the SEAM map losses of contrast_train.py:142-158 computed ON THE FLY from the stride-8 maps.
//
// The reference upsamples cam / cam_rv to the input size (resnet38_contrast.py:57-59: 2 x [N,21,448,448] = 540 MB
// per view) and then makes ~6 elementwise passes over them (GAP, min-pool, max_norm, 448->128 resize, and the same
// again backwards).  Every one of those consumers only needs U(y,x) = bilinear(low, align_corners=True) at points
// it can compute itself from the [h][w] low-resolution plane (12.5 KB for 56x56: L1/L2 resident), so these kernels
// never materialise U or dU:
//   up_plane_stats        max / min / arg / sum of relu(U) per plane          (GAP :142,155; max_norm :145-158)
//   up_rvmin_values       q = max_c U_rv*label and its arg channel per pixel  (adaptive_min_pooling_loss :16-22)
//   up_norm_resize_fwd    label * resize_{S->128}(max_norm(U))                (:145-158, visualization.py:62-67)
//   up_maps_backward      ALL gradients of one plane w.r.t. the low-res map: max_norm+resize backward with the
//                         max/min routes, the GAP constant, the min-pool selection — accumulated in an LDS image
//                         of the low-res plane (LDS float atomics), written once.
// U is evaluated by ONE pinned expression (explicit fma/mul, no compiler contraction) so every kernel sees
// bit-identical values (arg-max positions, `u > 0` gates and thresholds stay consistent between kernels).
#include <algorithm>
#include "common.h"

namespace {

__device__ __forceinline__ void src_index(int o, float scale, int in_size, int& i0, int& i1, float& f) {
  const float s = scale * o;                       // align_corners=True
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  f = s - i0;
}
__device__ __forceinline__ float ac_scale(int in_size, int out_size) {
  return out_size > 1 ? (float)(in_size - 1) / (out_size - 1) : 0.f;
}
__device__ __forceinline__ float up_value(const float* __restrict__ p, int w, int y0, int y1, int x0, int x1, float fy, float fx) {
  const float a = __fmaf_rn(fx, p[y0 * w + x1], __fmul_rn(1.f - fx, p[y0 * w + x0]));
  const float b = __fmaf_rn(fx, p[y1 * w + x1], __fmul_rn(1.f - fx, p[y1 * w + x0]));
  return __fmaf_rn(fy, b, __fmul_rn(1.f - fy, a));
}
__device__ __forceinline__ float up_at(const float* __restrict__ p, int h, int w, float sy, float sx, int oy, int ox) {
  int y0, y1, x0, x1; float fy, fx;
  src_index(oy, sy, h, y0, y1, fy); src_index(ox, sx, w, x0, x1, fx);
  return up_value(p, w, y0, y1, x0, x1, fy, fx);
}
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
// 256 threads sweep rows [r0, r1) of an S-wide image: `tpr` (power of two) threads per row
__device__ __forceinline__ int threads_per_row(int S) { return S > 128 ? 256 : (S > 64 ? 128 : 64); }

// ---- per-plane statistics of relu(U): stats[pl] = {mx, mn, sum(U), argmax, argmin, 0}; index order = hi-res row major.
// A plane is split over `chunks` row ranges; partials meet in packed 64-bit atomics (see loss.hip plane_stats).
__global__ __launch_bounds__(256) void up_stats_partial_kernel(const float* __restrict__ low, unsigned long long* __restrict__ kmax,
                                                              unsigned long long* __restrict__ kmin, float* __restrict__ ksum,
                                                              int h, int w, int S, int chunks, const float* __restrict__ label20) {
  __shared__ unsigned long long s_mx[256], s_mn[256];
  __shared__ float s_sum[256];
  const int pl = blockIdx.x / chunks, ck = blockIdx.x - pl * chunks, tid = threadIdx.x;
  if (label20) {                                  // only labelled planes (+ bg) are needed: the others' statistics are never read
    const int c = pl % 21;
    if (c >= 1 && label20[(pl / 21) * 20 + c - 1] == 0.f) return;
  }
  const float* p = low + (size_t)pl * h * w;
  const float sy = ac_scale(h, S), sx = ac_scale(w, S);
  const int per = (S + chunks - 1) / chunks;
  const int r0 = ck * per, r1 = min(S, r0 + per);
  const int tpr = threads_per_row(S), rpi = 256 / tpr;
  const int tr = tid / tpr, tx = tid & (tpr - 1);
  // A thread owns COLUMNS (ox = tx, tx + tpr, ...: at most MAXC) and walks its rows downwards, so the x-interpolated values of the two low-res rows
  // in use — the `a` and `b` of up_value, the SAME expressions — are recomputed only when the low-res row changes (every ~S/h rows; moving down one
  // low-res row turns b into a), and a value costs one multiply + one fma.  Indices grow along the walk, so "first occurrence wins a tie" is a strict
  // compare on the value; the 64-bit keys are packed once at the end.
  constexpr int MAXC = 4;
  int x0[MAXC], x1[MAXC]; float fx[MAXC], ra[MAXC], rb[MAXC];
  int nc = 0;
  for (int ox = tx; ox < S && nc < MAXC; ox += tpr, ++nc) src_index(ox, sx, w, x0[nc], x1[nc], fx[nc]);
  unsigned mxv = 0u, mxi = 0u, mnv = 0xFFFFFFFFu, mni = 0xFFFFFFFFu;
  bool any = false;
  float sum = 0.f;
  int cy0 = -1, cy1 = -1;
  for (int oy = r0 + tr; oy < r1; oy += rpi) {
    int y0, y1; float fy;
    src_index(oy, sy, h, y0, y1, fy);
    if (y0 != cy0 || y1 != cy1) {
#pragma unroll
      for (int j = 0; j < MAXC; ++j)
        if (j < nc) {
          ra[j] = (y0 == cy1) ? rb[j] : __fmaf_rn(fx[j], p[y0 * w + x1[j]], __fmul_rn(1.f - fx[j], p[y0 * w + x0[j]]));
          rb[j] = (y1 == y0) ? ra[j] : __fmaf_rn(fx[j], p[y1 * w + x1[j]], __fmul_rn(1.f - fx[j], p[y1 * w + x0[j]]));
        }
      cy0 = y0; cy1 = y1;
    }
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      if (j < nc) {
        const float u = __fmaf_rn(fy, rb[j], __fmul_rn(1.f - fy, ra[j]));
        const unsigned i = (unsigned)(oy * S + tx + j * tpr);
        const unsigned vb = __float_as_uint(fmaxf(u, 0.f));
        sum += u;
        if (!any || vb > mxv) { mxv = vb; mxi = i; }
        if (!any || vb < mnv) { mnv = vb; mni = i; }
        any = true;
      }
  }
  // columns beyond MAXC * tpr (S > 1024): the plain form
  for (int oy = r0 + tr; oy < r1; oy += rpi) {
    int y0, y1; float fy;
    src_index(oy, sy, h, y0, y1, fy);
    for (int ox = tx + MAXC * tpr; ox < S; ox += tpr) {
      int xa, xb; float f;
      src_index(ox, sx, w, xa, xb, f);
      const float u = up_value(p, w, y0, y1, xa, xb, fy, f);
      const unsigned i = (unsigned)(oy * S + ox);
      const unsigned vb = __float_as_uint(fmaxf(u, 0.f));
      sum += u;
      if (!any || vb > mxv || (vb == mxv && i < mxi)) { mxv = vb; mxi = i; }
      if (!any || vb < mnv || (vb == mnv && i < mni)) { mnv = vb; mni = i; }
      any = true;
    }
  }
  s_mx[tid] = any ? (((unsigned long long)mxv << 32) | (unsigned)(~mxi)) : 0ull;
  s_mn[tid] = any ? (((unsigned long long)mnv << 32) | mni) : ~0ull;
  s_sum[tid] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      if (s_mx[tid + o] > s_mx[tid]) s_mx[tid] = s_mx[tid + o];
      if (s_mn[tid + o] < s_mn[tid]) s_mn[tid] = s_mn[tid + o];
      s_sum[tid] += s_sum[tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) { atomicMax(&kmax[pl], s_mx[0]); atomicMin(&kmin[pl], s_mn[0]); atomicAdd(&ksum[pl], s_sum[0]); }
}
__global__ void up_stats_init_kernel(unsigned long long* __restrict__ kmax, unsigned long long* __restrict__ kmin, float* __restrict__ ksum, long planes) {
  const long pl = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pl < planes) { kmax[pl] = 0ull; kmin[pl] = ~0ull; ksum[pl] = 0.f; }
}
__global__ void up_stats_final_kernel(const unsigned long long* __restrict__ kmax, const unsigned long long* __restrict__ kmin,
                                      const float* __restrict__ ksum, float* __restrict__ stats, long planes) {
  const long pl = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pl >= planes) return;
  float* o = stats + pl * 6;
  o[0] = __uint_as_float((unsigned)(kmax[pl] >> 32)); o[1] = __uint_as_float((unsigned)(kmin[pl] >> 32)); o[2] = ksum[pl];
  o[3] = __int_as_float((int)(~(unsigned)(kmax[pl] & 0xFFFFFFFFull))); o[4] = __int_as_float((int)(unsigned)(kmin[pl] & 0xFFFFFFFFull)); o[5] = 0.f;
}

// ---- adaptive min-pooling values: q[n][p] = max_{c>=1} U_rv[n,c,p]*L[n,c] (first arg channel).  Absent classes
//      contribute U*0 = +0 (U_rv >= 0), so only labelled planes are evaluated.
__global__ __launch_bounds__(256) void up_rvmin_values_kernel(const float* __restrict__ low, const float* __restrict__ label20,
                                                             float* __restrict__ q, unsigned char* __restrict__ argc,
                                                             int h, int w, int S, int rows_per_block) {
  const int n = blockIdx.y, tid = threadIdx.x;
  const float sy = ac_scale(h, S), sx = ac_scale(w, S);
  const int r0 = blockIdx.x * rows_per_block, r1 = min(S, r0 + rows_per_block);
  const int tpr = threads_per_row(S), rpi = 256 / tpr;
  const int tr = tid / tpr, tx = tid & (tpr - 1);
  const float* lab = label20 + n * 20;
  for (int oy = r0 + tr; oy < r1; oy += rpi) {
    int y0, y1; float fy;
    src_index(oy, sy, h, y0, y1, fy);
    for (int ox = tx; ox < S; ox += tpr) {
      int x0, x1; float fx;
      src_index(ox, sx, w, x0, x1, fx);
      float best = -INFINITY; int bc = 1;
      for (int c = 1; c < 21; ++c) {
        const float L = lab[c - 1];
        float v = 0.f;
        if (L != 0.f) v = up_value(low + ((size_t)n * 21 + c) * h * w, w, y0, y1, x0, x1, fy, fx) * L;
        if (v > best) { best = v; bc = c; }
      }
      const size_t idx = (size_t)n * S * S + (size_t)oy * S + ox;
      q[idx] = best; argc[idx] = (unsigned char)bc;
    }
  }
}

// ---- out[n,c,P] = L * resize_{S->OS}( relu(relu(U) - mn - e) / (mx - mn + e) )      (visualization.py:62-67 + :145-158)
__global__ void up_norm_resize_fwd_kernel(const float* __restrict__ low, const float* __restrict__ stats, const float* __restrict__ label20,
                                          float* __restrict__ out, int h, int w, int S, int OS, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int ox = (int)(idx % OS); const long r = idx / OS;
  const int oy = (int)(r % OS); const long pl = r / OS;
  const int c = (int)(pl % 21); const long n = pl / 21;
  const float L = c == 0 ? 1.f : label20[n * 20 + c - 1];
  float v = 0.f;
  if (L != 0.f) {
    const float mx = stats[pl * 6 + 0], mn = stats[pl * 6 + 1];
    const float invD = 1.f / (mx - mn + 1e-5f);
    int y0, y1, x0, x1; float fy, fx;
    const float sc = ac_scale(S, OS);
    src_index(oy, sc, S, y0, y1, fy); src_index(ox, sc, S, x0, x1, fx);
    const float* p = low + (size_t)pl * h * w;
    const float sy = ac_scale(h, S), sx = ac_scale(w, S);
    auto f = [&](int y, int x) { return fmaxf(fmaxf(up_at(p, h, w, sy, sx, y, x), 0.f) - mn - 1e-5f, 0.f) * invD; };
    v = (1.f - fy) * ((1.f - fx) * f(y0, x0) + fx * f(y0, x1)) + fy * ((1.f - fx) * f(y1, x0) + fx * f(y1, x1));
    v *= L;
  }
  out[idx] = v;
}

// ---- adjoint of the align_corners upsample applied to all-ones: wvec[y] = sum_oy weight(oy -> y)   (the GAP gradient)
__global__ void resize_adjoint_ones_kernel(float* __restrict__ wvec, int h, int S) {
  const int y = blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= h) return;
  const float sc = ac_scale(h, S);
  float acc = 0.f;
  for (int o = 0; o < S; ++o) {
    int y0, y1; float f;
    src_index(o, sc, h, y0, y1, f);
    acc += (y == y0 ? 1.f - f : 0.f) + (y == y1 ? f : 0.f);
  }
  wvec[y] = acc;
}

// ---- backward of one plane into d_low[pl] ([h][w], ACCUMULATED with float atomics: the host zeroes it).  A plane is
// split over `chunks` workgroups (ranges of the OS x OS outputs and of the S x S pixels); every term is linear, so
// each workgroup scatters its own share — including its share of the max / min routes:
//   (1) max_norm + S->OS resize backward of G (label-gated), with the max / min gradient routes;
//   (2) plane_bias[pl] * wy[y] * wx[x]: the GAP (classification) gradient, a constant over the upsampled plane (chunk 0);
//   (3) min-pool selection (rv map only, q != nullptr): pixels among the k smallest with q > 0 whose arg channel is
//       this plane send coef * L.
// Hi-res gradients are scattered to their 4 low-res taps in an LDS image of the plane (LDS float atomics), which is
// added to d_low once at the end.
__global__ __launch_bounds__(256) void up_maps_bwd_kernel(const float* __restrict__ G, const float* __restrict__ low, const float* __restrict__ stats,
                                                         const float* __restrict__ label20, const float* __restrict__ plane_bias,
                                                         const float* __restrict__ wvec_y, const float* __restrict__ wvec_x,
                                                         const float* __restrict__ q, const unsigned char* __restrict__ argc,
                                                         const float* __restrict__ res, int k, float coef,
                                                         float* __restrict__ d_low, int h, int w, int S, int OS, int chunks) {
  extern __shared__ float dl[];                    // [h*w] gradient image
  __shared__ float red[4];
  const int pl = blockIdx.x / chunks, ck = blockIdx.x - pl * chunks, tid = threadIdx.x;
  const int c = pl % 21; const int n = pl / 21;
  const float L = c == 0 ? 1.f : label20[n * 20 + c - 1];
  const float bias = (plane_bias && ck == 0) ? plane_bias[pl] : 0.f;
  const bool do_norm = L != 0.f && G != nullptr, do_sel = q != nullptr && c >= 1 && L != 0.f;
  if (!do_norm && !do_sel && bias == 0.f) return;  // (workgroup-uniform)
  const float* p = low + (size_t)pl * h * w;
  const float sy = ac_scale(h, S), sx = ac_scale(w, S);
  for (int i = tid; i < h * w; i += 256) dl[i] = 0.f;
  __syncthreads();
  auto scatter = [&](int oy, int ox, float t) {    // hi-res gradient t at (oy, ox) -> its 4 low-res taps
    int y0, y1, x0, x1; float fy, fx;
    src_index(oy, sy, h, y0, y1, fy); src_index(ox, sx, w, x0, x1, fx);
    atomicAdd(&dl[y0 * w + x0], t * (1.f - fy) * (1.f - fx));
    atomicAdd(&dl[y0 * w + x1], t * (1.f - fy) * fx);
    atomicAdd(&dl[y1 * w + x0], t * fy * (1.f - fx));
    atomicAdd(&dl[y1 * w + x1], t * fy * fx);
  };
  if (do_norm) {
    const float mx = stats[(size_t)pl * 6 + 0], mn = stats[(size_t)pl * 6 + 1];
    const float invD = 1.f / (mx - mn + 1e-5f);
    const float sc = ac_scale(S, OS);
    const float* g = G + (size_t)pl * OS * OS;
    const int per = (OS * OS + chunks - 1) / chunks;
    const int o_end = min(OS * OS, (ck + 1) * per);
    float A = 0.f, B = 0.f;
    for (int o = ck * per + tid; o < o_end; o += 256) {
      const float go = g[o] * L;
      if (go == 0.f) continue;
      const int oy = o / OS, ox = o - oy * OS;
      int y0, y1, x0, x1; float fy, fx;
      src_index(oy, sc, S, y0, y1, fy); src_index(ox, sc, S, x0, x1, fx);
      const int ys[2] = {y0, y1}, xs[2] = {x0, x1};
      const float wy[2] = {1.f - fy, fy}, wx[2] = {1.f - fx, fx};
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float wgt = wy[a] * wx[b];
          if (wgt == 0.f) continue;
          const float u = up_at(p, h, w, sy, sx, ys[a], xs[b]);
          const float av = fmaxf(fmaxf(u, 0.f) - mn - 1e-5f, 0.f);
          if (av > 0.f) {                                   // implies u > 0
            const float t = go * wgt * invD;
            scatter(ys[a], xs[b], t);
            B += t;
            A += t * av * invD;
          }
        }
    }
    const float At = block_sum(A, red), Bt = block_sum(B, red);
    if (tid == 0) {
      const int imx = __float_as_int(stats[(size_t)pl * 6 + 3]), imn = __float_as_int(stats[(size_t)pl * 6 + 4]);
      const int ymx = imx / S, xmx = imx - ymx * S, ymn = imn / S, xmn = imn - ymn * S;
      if (up_at(p, h, w, sy, sx, ymx, xmx) > 0.f) scatter(ymx, xmx, -At);          // d/d mx (this chunk's share)
      if (up_at(p, h, w, sy, sx, ymn, xmn) > 0.f) scatter(ymn, xmn, At - Bt);      // d/d mn
    }
  }
  if (do_sel) {
    const float thr = res[n * 4 + 0];
    const float ce = res[n * 4 + 3];
    const float wtie = ce > 0.f ? ((float)k - res[n * 4 + 2]) / ce : 0.f;
    const float* qn = q + (size_t)n * S * S;
    const unsigned char* an = argc + (size_t)n * S * S;
    const float val = coef * L;
    const int npix = S * S;
    const int per = ((npix + chunks - 1) / chunks + 3) & ~3;
    const int i_end = min(npix, (ck + 1) * per);
    auto one = [&](int i, unsigned char ac, float v) {
      if (ac != (unsigned char)c) return;
      float wsel = 0.f;
      if (v < thr) wsel = 1.f; else if (v == thr) wsel = wtie;
      if (wsel > 0.f && v > 0.f) { const int oy = i / S; scatter(oy, i - oy * S, wsel * val); }
    };
    if ((npix & 3) == 0) {                          // 4 pixels per lane per iteration: one 4-B and one 16-B load
      for (int i = ck * per + tid * 4; i < i_end; i += 1024) {
        const unsigned a4 = *reinterpret_cast<const unsigned*>(an + i);
        const unsigned cc = (unsigned)c;
        if (((a4 & 255u) != cc) && (((a4 >> 8) & 255u) != cc) && (((a4 >> 16) & 255u) != cc) && ((a4 >> 24) != cc)) continue;
        const float4 v4 = *reinterpret_cast<const float4*>(qn + i);
        one(i, (unsigned char)(a4 & 255u), v4.x); one(i + 1, (unsigned char)((a4 >> 8) & 255u), v4.y);
        one(i + 2, (unsigned char)((a4 >> 16) & 255u), v4.z); one(i + 3, (unsigned char)(a4 >> 24), v4.w);
      }
    } else {
      for (int i = ck * per + tid; i < i_end; i += 256) one(i, an[i], qn[i]);
    }
  }
  __syncthreads();
  float* o = d_low + (size_t)pl * h * w;
  for (int i = tid; i < h * w; i += 256) {
    const int y = i / w;
    const float v = dl[i] + (bias != 0.f ? bias * wvec_y[y] * wvec_x[i - y * w] : 0.f);
    if (v != 0.f) atomicAdd(&o[i], v);
  }
}

}  // namespace

#define GRID1(total) dim3((unsigned)(((total) + 255) / 256)), dim3(256)
#define ST ((hipStream_t)stream)

// workspace: planes * 24 bytes (wseg_plane_stats_workspace_bytes)
extern "C" int wseg_up_plane_stats(const float* low, float* stats, long planes, int h, int w, int S, const float* label20, void* workspace, void* stream) {
  WSEG_CHECK(low && stats && workspace && planes > 0 && h > 0 && w > 0 && S > 0 && S <= 32768, "up_plane_stats: bad arguments");
  unsigned long long* kmax = (unsigned long long*)workspace;
  unsigned long long* kmin = kmax + planes;
  float* ksum = (float*)(kmin + planes);
  hipLaunchKernelGGL(up_stats_init_kernel, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, ST, kmax, kmin, ksum, planes);   // (one launch, not three memsets: these sit on the loss phase's critical chain)
  const int chunks = std::max(1, std::min(std::min(16, S / 16), (int)std::max(1L, 4096 / planes)));
  hipLaunchKernelGGL(up_stats_partial_kernel, dim3((unsigned)(planes * chunks)), dim3(256), 0, ST, low, kmax, kmin, ksum, h, w, S, chunks, label20);
  hipLaunchKernelGGL(up_stats_final_kernel, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, ST, kmax, kmin, ksum, stats, planes);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_up_rvmin_values(const float* low, const float* label20, float* q, unsigned char* argc, int N, int h, int w, int S, void* stream) {
  WSEG_CHECK(low && label20 && q && argc && N > 0 && h > 0 && w > 0 && S > 0, "up_rvmin_values: bad arguments");
  const int rpb = 8;
  hipLaunchKernelGGL(up_rvmin_values_kernel, dim3((S + rpb - 1) / rpb, N), dim3(256), 0, ST, low, label20, q, argc, h, w, S, rpb);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_up_norm_resize_forward(const float* low, const float* stats, const float* label20, float* out, int N, int h, int w, int S, int OS, void* stream) {
  WSEG_CHECK(low && stats && label20 && out && N > 0 && h > 0 && w > 0 && S > 0 && OS > 0, "up_norm_resize_forward: bad arguments");
  const long total = (long)N * 21 * OS * OS;
  hipLaunchKernelGGL(up_norm_resize_fwd_kernel, GRID1(total), 0, ST, low, stats, label20, out, h, w, S, OS, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_resize_adjoint_ones(float* wvec, int h, int S, void* stream) {
  WSEG_CHECK(wvec && h > 0 && S > 0, "resize_adjoint_ones: bad arguments");
  hipLaunchKernelGGL(resize_adjoint_ones_kernel, dim3((h + 63) / 64), dim3(64), 0, ST, wvec, h, S);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_up_maps_backward(const float* G, const float* low, const float* stats, const float* label20, const float* plane_bias,
                                     const float* wvec_y, const float* wvec_x, const float* q, const unsigned char* argc, const float* res,
                                     int k, float coef, float* d_low, int N, int h, int w, int S, int OS, void* stream) {
  WSEG_CHECK(low && stats && label20 && d_low && N > 0 && h > 0 && w > 0 && S > 0 && OS > 0, "up_maps_backward: bad arguments");
  WSEG_CHECK((size_t)h * w * 4 <= 60000, "up_maps_backward: low-res plane %dx%d does not fit the LDS image", h, w);
  WSEG_CHECK(plane_bias == nullptr || (wvec_y && wvec_x), "up_maps_backward: plane_bias needs the adjoint-of-ones vectors");
  WSEG_CHECK(q == nullptr || (argc && res), "up_maps_backward: q needs argc and res");
  (void)hipMemsetAsync(d_low, 0, sizeof(float) * (size_t)N * 21 * h * w, ST);
  // (many short chunks: the selection loop of a workgroup is a latency chain of divergent LDS scatters — measured 180 us
  //  per launch with 8-12 iterations per workgroup, independent of the map size)
  const int chunks = std::max(1, std::min(64, (S * S) / 2048));
  hipLaunchKernelGGL(up_maps_bwd_kernel, dim3((unsigned)(N * 21 * chunks)), dim3(256), (size_t)h * w * 4, ST, G, low, stats, label20, plane_bias,
                     wvec_y, wvec_x, q, argc, res, k, coef, d_low, h, w, S, OS, chunks);
  WSEG_LAUNCH_CHECK();
  return 0;
}
