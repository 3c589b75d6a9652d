// loss.hip — the SEAM + pixel-to-prototype contrast losses of contrast_train.py:138-395 as HIP
// kernels (forward values and hand-written gradients).  All 21-class maps are planar f32.
// The kernels are HBM/latency-bound integer+float work: coalesced planar access, wavefront
// shuffles for per-pixel 21-wide reductions / ranks, LDS for per-plane and per-class reductions.
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

// ---- per-plane statistics of U [planes][npix]: max/min of relu(U) with their first index, sum of U
//      stats[pl] = {mx, mn, sum, (float)argmax, (float)argmin, 0}
// A plane is split over `chunks` workgroups; partials meet in packed 64-bit atomics:
//   max key = (bits(relu) << 32) | ~index   (atomicMax: largest value, then lowest index)
//   min key = (bits(relu) << 32) |  index   (atomicMin: smallest value, then lowest index)   (relu >= 0: bits are ordered)
__global__ __launch_bounds__(256) void plane_stats_partial_kernel(const float* __restrict__ U, unsigned long long* __restrict__ kmax,
                                                                  unsigned long long* __restrict__ kmin, float* __restrict__ ksum,
                                                                  int npix, int chunks) {
  __shared__ unsigned long long s_mx[256], s_mn[256];
  __shared__ float s_sum[256];
  const int pl = blockIdx.x / chunks, ck = blockIdx.x - pl * chunks, tid = threadIdx.x;
  const float* p = U + (size_t)pl * npix;
  const int per = (npix + chunks - 1) / chunks;
  const int i0 = ck * per, i1 = min(npix, i0 + per);
  unsigned long long mx = 0ull, mn = ~0ull;
  float sum = 0.f;
  for (int i = i0 + tid; i < i1; i += 256) {
    const float u = p[i];
    const unsigned rb = __float_as_uint(fmaxf(u, 0.f));
    sum += u;
    const unsigned long long a = ((unsigned long long)rb << 32) | (unsigned)(~i), b = ((unsigned long long)rb << 32) | (unsigned)i;
    mx = a > mx ? a : mx; mn = b < mn ? b : mn;
  }
  s_mx[tid] = mx; s_mn[tid] = mn; s_sum[tid] = sum;
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
__global__ void plane_stats_final_kernel(const unsigned long long* __restrict__ kmax, const unsigned long long* __restrict__ kmin,
                                         const float* __restrict__ ksum, float* __restrict__ stats, long planes) {
  const long pl = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pl >= planes) return;
  float* o = stats + pl * 6;
  o[0] = __uint_as_float((unsigned)(kmax[pl] >> 32)); o[1] = __uint_as_float((unsigned)(kmin[pl] >> 32)); o[2] = ksum[pl];
  o[3] = __int_as_float((int)(~(unsigned)(kmax[pl] & 0xFFFFFFFFull))); o[4] = __int_as_float((int)(unsigned)(kmin[pl] & 0xFFFFFFFFull)); o[5] = 0.f;
}

// ---- classification loss (contrast_train.py:142,155,159-160): z = GAP, mean BCE-with-logits over N*20
//      out[0] += loss ; dz[n][c] = coef * (sigmoid(z) - y) / (20 N)   (c >= 1; dz[n][0] = 0), as a per-pixel bias /npix
__global__ void cls_loss_kernel(const float* __restrict__ stats, const float* __restrict__ label20, float* __restrict__ loss_out,
                                float* __restrict__ plane_bias, int N, int npix, float coef) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < N * 21; i += blockDim.x) {
    const int n = i / 21, c = i - n * 21;
    float bias = 0.f;
    if (c >= 1) {
      const float z = stats[(size_t)i * 6 + 2] / npix;
      const float y = label20[n * 20 + c - 1];
      // -(y*logsigmoid(z) + (1-y)*logsigmoid(-z)); logsigmoid(z) = min(z,0) - log1p(exp(-|z|))
      const float l1p = log1pf(expf(-fabsf(z)));
      const float ls_p = fminf(z, 0.f) - l1p, ls_n = fminf(-z, 0.f) - l1p;
      acc += -(y * ls_p + (1.f - y) * ls_n);
      const float sg = 1.f / (1.f + expf(-z));
      bias = coef * (sg - y) / (20.f * N) / npix;           // d loss / d U[n,c,pixel]
    }
    plane_bias[i] = bias;
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) atomicAdd(loss_out, tot / (20.f * N));
}

// ---- adaptive min-pooling values (contrast_train.py:16-22): q = max_{c>=1} U[n,c,p]*L[n,c], arg channel
__global__ void rvmin_values_kernel(const float* __restrict__ U, const float* __restrict__ label20, float* __restrict__ q,
                                    unsigned char* __restrict__ argc, int npix, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long n = idx / npix; const int p = (int)(idx - n * npix);
  float best = -INFINITY; int bc = 1;
  for (int c = 1; c < 21; ++c) {
    const float v = U[((size_t)n * 21 + c) * npix + p] * label20[n * 20 + c - 1];
    if (v > best) { best = v; bc = c; }
  }
  q[idx] = best; argc[idx] = (unsigned char)bc;
}

// ---- radix select of the k-th order statistic per row (values as order-preserving uint keys)
__device__ __forceinline__ unsigned f2key(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float key2f(unsigned k) { unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; return __uint_as_float(u); }

// state[row] = {prefix, mask, k_remaining}; one pass handles 8 bits (shift = 24,16,8,0)
__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ vals, int n, int use_abs, const unsigned* __restrict__ state,
                                                          unsigned* __restrict__ hist, int shift) {
  __shared__ unsigned h[4][256];                            // one sub-histogram per wave: values cluster, LDS atomics on one
  const int row = blockIdx.y, wv = threadIdx.x >> 6;        // bucket serialise
#pragma unroll
  for (int w = 0; w < 4; ++w) h[w][threadIdx.x] = 0;
  __syncthreads();
  const unsigned prefix = state[row * 4 + 0], mask = state[row * 4 + 1];
  const float* v = vals + (size_t)row * n;
  auto count = [&](float f) {
    if (use_abs) f = fabsf(f);
    const unsigned k = f2key(f);
    if ((k & mask) == prefix) atomicAdd(&h[wv][(k >> shift) & 255u], 1u);
  };
  if ((n & 3) == 0) {                                       // 16-B loads (the rows of both callers: 448^2 and 21 * 128^2 values)
    const float4* v4 = reinterpret_cast<const float4*>(v);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += gridDim.x * 256) {
      const float4 q = v4[i];
      count(q.x); count(q.y); count(q.z); count(q.w);
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) count(v[i]);
  }
  __syncthreads();
  const unsigned t = h[0][threadIdx.x] + h[1][threadIdx.x] + h[2][threadIdx.x] + h[3][threadIdx.x];
  if (t) atomicAdd(&hist[row * 256 + threadIdx.x], t);
}
// choose the bucket holding the k-th (k counted from the small end: 1-based rank) and narrow the prefix.
// One wave per row: lane l owns buckets 4l..4l+3, a shuffle scan finds the lane whose running count crosses k.
__global__ __launch_bounds__(64) void select_scan_kernel(unsigned* __restrict__ state, unsigned* __restrict__ hist, int shift, int rows) {
  const int row = blockIdx.x, lane = threadIdx.x;
  if (row >= rows) return;
  const unsigned k = state[row * 4 + 2];
  uint4* hrow = reinterpret_cast<uint4*>(hist + (size_t)row * 256);
  const uint4 c = hrow[lane];
  const unsigned s = c.x + c.y + c.z + c.w;
  unsigned inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  const unsigned exc = inc - s;
  const unsigned total = __shfl(inc, 63, 64);
  hrow[lane] = make_uint4(0u, 0u, 0u, 0u);
  int b = -1; unsigned cum = exc;
  if (exc < k && inc >= k) {                       // exactly one lane (k >= 1); first bucket with running count >= k
    const unsigned cs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (b < 0) { if (cum + cs[j] >= k) b = lane * 4 + j; else cum += cs[j]; }
    }
  } else if (lane == 63 && total < k) {            // (cannot happen for finite inputs: keep the serial loop's result)
    b = 255; cum = total;
  }
  if (b >= 0) {
    state[row * 4 + 0] |= ((unsigned)b << shift);
    state[row * 4 + 1] |= (255u << shift);
    state[row * 4 + 2] = k - cum;
  }
}
__global__ void select_init_kernel(unsigned* __restrict__ state, unsigned* __restrict__ hist, int rows, unsigned rank_small, float* __restrict__ res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) { state[i * 4 + 0] = 0; state[i * 4 + 1] = 0; state[i * 4 + 2] = rank_small; state[i * 4 + 3] = 0; }
  if (i < rows * 4) res[i] = 0.f;                  // (the sums the last pass accumulates into: cleared here, not by a memset launch in front of it)
  if (i < rows * 256) hist[i] = 0;
}
// after 4 passes state.prefix is the key of the threshold.  Accumulate per row:
//   res[row] = {thr, sum of relu?(v) strictly beyond thr, count strictly beyond, count equal}
__global__ __launch_bounds__(256) void select_sum_kernel(const float* __restrict__ vals, int n, int use_abs, int largest, int relu_vals,
                                                         const unsigned* __restrict__ state, float* __restrict__ res) {
  __shared__ float red[4];
  const int row = blockIdx.y;
  const float thr = key2f(state[row * 4 + 0]);
  const float* v = vals + (size_t)row * n;
  float s = 0.f, cs = 0.f, ce = 0.f;
  auto take = [&](float f) {
    if (use_abs) f = fabsf(f);
    const bool beyond = largest ? (f > thr) : (f < thr);
    if (beyond) { s += relu_vals ? fmaxf(f, 0.f) : f; cs += 1.f; }
    else if (f == thr) ce += 1.f;
  };
  if ((n & 3) == 0) {
    const float4* v4 = reinterpret_cast<const float4*>(v);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < (n >> 2); i += gridDim.x * 256) {
      const float4 q = v4[i];
      take(q.x); take(q.y); take(q.z); take(q.w);
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) take(v[i]);
  }
  const float S = block_sum(s, red), CS = block_sum(cs, red), CE = block_sum(ce, red);
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) res[row * 4 + 0] = thr;
    atomicAdd(&res[row * 4 + 1], S); atomicAdd(&res[row * 4 + 2], CS); atomicAdd(&res[row * 4 + 3], CE);
  }
}
// loss += scale * sum_rows ( sum_strict + (k - cnt_strict) * f(thr) )
__global__ void select_finish_kernel(const float* __restrict__ res, int rows, int k, int relu_vals, float scale, float* __restrict__ loss_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = 0.f;
    for (int r = 0; r < rows; ++r) {
      const float thr = relu_vals ? fmaxf(res[r * 4 + 0], 0.f) : res[r * 4 + 0];
      t += res[r * 4 + 1] + ((float)k - res[r * 4 + 2]) * thr;
    }
    atomicAdd(loss_out, t * scale);
  }
}

// ---- rvmin backward: pixels among the k smallest with q > 0 send coef*L to their arg channel
__global__ void rvmin_bwd_kernel(const float* __restrict__ q, const unsigned char* __restrict__ argc, const float* __restrict__ res,
                                 const float* __restrict__ label20, float* __restrict__ dU, int npix, int k, float coef, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long n = idx / npix; const int p = (int)(idx - n * npix);
  const float v = q[idx], thr = res[n * 4 + 0];
  float w = 0.f;
  if (v < thr) w = 1.f;
  else if (v == thr) { const float ce = res[n * 4 + 3]; w = ce > 0.f ? ((float)k - res[n * 4 + 2]) / ce : 0.f; }
  if (w > 0.f && v > 0.f) {
    const int c = argc[idx];
    dU[((size_t)n * 21 + c) * npix + p] += w * coef * label20[n * 20 + c - 1];
  }
}

// ---- out[n,c,P] = L * resize_{S->OS}( relu(relu(U) - mn - e) / (mx - mn + e) )      (visualization.py:62-67 + :145-158)
__global__ void norm_resize_fwd_kernel(const float* __restrict__ U, const float* __restrict__ stats, const float* __restrict__ label20,
                                       float* __restrict__ out, int S, int OS, long total) {
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
    const float* p = U + (size_t)pl * S * S;
    auto f = [&](int y, int x) { return fmaxf(fmaxf(p[(size_t)y * S + x], 0.f) - mn - 1e-5f, 0.f) * invD; };
    v = (1.f - fy) * ((1.f - fx) * f(y0, x0) + fx * f(y0, x1)) + fy * ((1.f - fx) * f(y1, x0) + fx * f(y1, x1));
    v *= L;
  }
  out[idx] = v;
}

// backward of the above; one workgroup per plane: scatters into dU and routes the max/min gradients
__global__ __launch_bounds__(256) void norm_resize_bwd_kernel(const float* __restrict__ G, const float* __restrict__ U, const float* __restrict__ stats,
                                                              const float* __restrict__ label20, float* __restrict__ dU, int S, int OS) {
  __shared__ float red[4];
  const int pl = blockIdx.x;
  const int c = pl % 21; const int n = pl / 21;
  const float L = c == 0 ? 1.f : label20[n * 20 + c - 1];
  if (L == 0.f) return;
  const float mx = stats[(size_t)pl * 6 + 0], mn = stats[(size_t)pl * 6 + 1];
  const float invD = 1.f / (mx - mn + 1e-5f);
  const float sc = ac_scale(S, OS);
  const float* p = U + (size_t)pl * S * S;
  float* dp = dU + (size_t)pl * S * S;
  const float* g = G + (size_t)pl * OS * OS;
  float A = 0.f, B = 0.f;
  for (int o = threadIdx.x; o < OS * OS; o += 256) {
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
        const float w = wy[a] * wx[b];
        if (w == 0.f) continue;
        const size_t qi = (size_t)ys[a] * S + xs[b];
        const float u = p[qi];
        const float av = fmaxf(fmaxf(u, 0.f) - mn - 1e-5f, 0.f);
        if (av > 0.f) {                                   // implies u > 0
          const float t = go * w * invD;
          atomicAdd(&dp[qi], t);
          B += t;
          A += t * av * invD;
        }
      }
  }
  const float At = block_sum(A, red), Bt = block_sum(B, red);
  if (threadIdx.x == 0) {
    const int imx = __float_as_int(stats[(size_t)pl * 6 + 3]), imn = __float_as_int(stats[(size_t)pl * 6 + 4]);
    if (p[imx] > 0.f) atomicAdd(&dp[imx], -At);          // d/d mx
    if (p[imn] > 0.f) atomicAdd(&dp[imn], At - Bt);      // d/d mn
  }
}

// ---- ER + ECR preparation on the 128x128 maps (contrast_train.py:163-169)
// per pixel: ER sum, G_c1/G_c2 (ER gradients, fg only), dlt1 = r1 - oh(c2), dlt2 = r2 - oh(c1) (all 21 channels)
__global__ void er_ecr_prep_kernel(const float* __restrict__ c1, const float* __restrict__ c2, const float* __restrict__ r1,
                                   const float* __restrict__ r2, float* __restrict__ Gc1, float* __restrict__ Gc2,
                                   float* __restrict__ dlt1, float* __restrict__ dlt2, float* __restrict__ er_out,
                                   int npix, float er_coef, long total) {
  __shared__ float red[4];
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float er = 0.f;
  if (idx < total) {
    const long n = idx / npix; const int p = (int)(idx - n * npix);
    const size_t base = (size_t)n * 21 * npix + p;
    float a[21], b[21];
    float m1 = -INFINITY, m2 = -INFINITY;
    for (int c = 1; c < 21; ++c) {
      a[c] = c1[base + (size_t)c * npix]; b[c] = c2[base + (size_t)c * npix];
      m1 = fmaxf(m1, a[c]); m2 = fmaxf(m2, b[c]);
      const float df = a[c] - b[c];
      er += fabsf(df);
      const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
      Gc1[base + (size_t)c * npix] = sg * er_coef;
      Gc2[base + (size_t)c * npix] = -sg * er_coef;
    }
    Gc1[base] = 0.f; Gc2[base] = 0.f;
    a[0] = 1.f - m1; b[0] = 1.f - m2;                    // cam[:,0] = 1 - max fg
    for (int c = 0; c < 21; ++c) {
      const float oh2 = (c == 0 || b[c] == m2) ? b[c] : 0.f;
      const float oh1 = (c == 0 || a[c] == m1) ? a[c] : 0.f;
      dlt1[base + (size_t)c * npix] = r1[base + (size_t)c * npix] - oh2;
      dlt2[base + (size_t)c * npix] = r2[base + (size_t)c * npix] - oh1;
    }
  }
  const float t = block_sum(er, red);
  if (threadIdx.x == 0 && t != 0.f) atomicAdd(er_out, t);
}

// ECR backward: G_r = coef * sign(dlt) for the K largest |dlt| of each sample (ties at thr share the remainder)
__global__ void ecr_bwd_kernel(const float* __restrict__ dlt, const float* __restrict__ res, float* __restrict__ Gr, int per_row, int k, float coef, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long row = idx / per_row;
  const float d = dlt[idx], v = fabsf(d), thr = res[row * 4 + 0];
  float w = 0.f;
  if (v > thr) w = 1.f;
  else if (v == thr) { const float ce = res[row * 4 + 3]; w = ce > 0.f ? ((float)k - res[row * 4 + 2]) / ce : 0.f; }
  const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  Gr[idx] = w * sg * coef;
}

// ---- f_proj rows (head rows cols [0,128), any dtype) -> F [N*oh*ow][128] f32, bilinear align_corners=True
template <int DT>
__global__ void rows_resize_fwd_kernel(const void* __restrict__ head, int ld, float* __restrict__ F, int ih, int iw, int oh, int ow, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over N*oh*ow*128
  if (idx >= total) return;
  const int ch = (int)(idx & 127); const long pix = idx >> 7;
  const int ox = (int)(pix % ow); const long r = pix / ow;
  const int oy = (int)(r % oh); const long n = r / oh;
  int y0, y1, x0, x1; float fy, fx;
  src_index(oy, ac_scale(ih, oh), ih, y0, y1, fy); src_index(ox, ac_scale(iw, ow), iw, x0, x1, fx);
  auto f = [&](int y, int x) { return elem<DT>::ld(head, (((size_t)n * ih + y) * iw + x) * ld + ch); };
  F[idx] = (1.f - fy) * ((1.f - fx) * f(y0, x0) + fx * f(y0, x1)) + fy * ((1.f - fx) * f(y1, x0) + fx * f(y1, x1));
}

// ---- d(head rows): cols [0,128) = relu-masked adjoint of the row resize applied to dF, cols [128,149) = d_cam_low, rest 0
template <int DT>
__global__ void head_grad_fused_kernel(const float* __restrict__ dF, const float* __restrict__ d_cam, const void* __restrict__ head,
                                       void* __restrict__ d_head, int ld, int ih, int iw, int oh, int ow, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over N*ih*iw*(ld/8)
  if (idx >= total) return;
  const int v8 = ld / 8;
  const long pix = idx / v8; const int c8 = (int)(idx - pix * v8) * 8;
  const int x = (int)(pix % iw); const long r = pix / iw;
  const int y = (int)(r % ih); const long n = r / ih;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;
  if (c8 < 128) {
    const float sy = ac_scale(ih, oh), sx = ac_scale(iw, ow);
    int oy_lo, oy_hi, ox_lo, ox_hi;
    if (sy > 0.f) { oy_lo = max(0, (int)floorf((y - 1) / sy) - 1); oy_hi = min(oh - 1, (int)ceilf((y + 1) / sy) + 1); } else { oy_lo = 0; oy_hi = oh - 1; }
    if (sx > 0.f) { ox_lo = max(0, (int)floorf((x - 1) / sx) - 1); ox_hi = min(ow - 1, (int)ceilf((x + 1) / sx) + 1); } else { ox_lo = 0; ox_hi = ow - 1; }
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1; float fy;
      src_index(oy, sy, ih, y0, y1, fy);
      const float wy = (y == y0 ? 1.f - fy : 0.f) + (y == y1 ? fy : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1; float fx;
        src_index(ox, sx, iw, x0, x1, fx);
        const float w = wy * ((x == x0 ? 1.f - fx : 0.f) + (x == x1 ? fx : 0.f));
        if (w == 0.f) continue;
        const float4* g = reinterpret_cast<const float4*>(dF + ((((size_t)n * oh + oy) * ow + ox) << 7) + c8);
        const float4 g0 = g[0], g1 = g[1];
        o[0] += w * g0.x; o[1] += w * g0.y; o[2] += w * g0.z; o[3] += w * g0.w;
        o[4] += w * g1.x; o[5] += w * g1.y; o[6] += w * g1.z; o[7] += w * g1.w;
      }
    }
    float hv[8];
    load8<DT>(head, (size_t)pix * ld + c8, hv);
#pragma unroll
    for (int e = 0; e < 8; ++e) if (!(hv[e] > 0.f)) o[e] = 0.f;
  } else if (c8 < 152 && d_cam) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c8 + e - 128;
      if (c < 21) o[e] = d_cam[(((size_t)n * 21 + c) * ih + y) * iw + x];
    }
  }
  store8<DT>(d_head, (size_t)pix * ld + c8, o);
}

// ---- pseudo labels (contrast_train.py:186-197): one workgroup per image over npix (=256) pixels
__global__ __launch_bounds__(256) void pseudo_label_kernel(const float* __restrict__ R, const float* __restrict__ label20, float bg_thr,
                                                           int* __restrict__ y, float* __restrict__ ncam, int npix) {
  __shared__ float s_mx[21], s_mn[21];
  __shared__ float red_a[256], red_b[256];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* Rn = R + (size_t)n * 21 * npix;
  for (int c = 0; c < 21; ++c) {
    float mx = 0.f, mn = INFINITY;
    for (int p = tid; p < npix; p += 256) { const float v = fmaxf(Rn[(size_t)c * npix + p], 0.f); mx = fmaxf(mx, v); mn = fminf(mn, v); }
    red_a[tid] = mx; red_b[tid] = mn;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) { red_a[tid] = fmaxf(red_a[tid], red_a[tid + o]); red_b[tid] = fminf(red_b[tid], red_b[tid + o]); } __syncthreads(); }
    if (tid == 0) { s_mx[c] = red_a[0]; s_mn[c] = red_b[0]; }
    __syncthreads();
  }
  for (int p = tid; p < npix; p += 256) {
    float best = -INFINITY; int bc = 0;
    for (int c = 0; c < 21; ++c) {
      float v = fmaxf(Rn[(size_t)c * npix + p], 0.f);
      if (v < s_mn[c] + 1e-5f) v = 0.f;
      v = (v - s_mn[c] - 1e-5f) / (s_mx[c] - s_mn[c] + 1e-5f);
      if (c == 0) v = bg_thr;
      ncam[((size_t)n * 21 + c) * npix + p] = v;
      const float L = c == 0 ? 1.f : label20[n * 20 + c - 1];
      const float s = v * L;                               // softmax is monotone: argmax of the logits
      if (s > best) { best = s; bc = c; }
    }
    y[(size_t)n * npix + p] = bc;
  }
}

// ---- batch prototypes (contrast_train.py:199-209): one workgroup per class; candidates = top-K of T[c][:]
//      T[c][p] = ncam[n][c][pix], p = n*npix + pix.  Constant rows use the supplied tie index set (Q5).
__global__ __launch_bounds__(256) void proto_candidates_kernel(const float* __restrict__ ncam, const float* __restrict__ F, const int* __restrict__ tie_idx,
                                                               float* __restrict__ cand_val, float* __restrict__ cand_feat, int* __restrict__ cand_const,
                                                               int N, int npix, int K) {
  extern __shared__ float sv[];                            // [P] values
  __shared__ float r_v[256]; __shared__ int r_i[256];
  __shared__ int sel[64];
  __shared__ float s_gmin;
  const int c = blockIdx.x, tid = threadIdx.x;
  const int P = N * npix;
  float lmx = -INFINITY, lmn = INFINITY;
  for (int p = tid; p < P; p += 256) {
    const int n = p / npix, pix = p - n * npix;
    const float v = ncam[((size_t)n * 21 + c) * npix + pix];
    sv[p] = v; lmx = fmaxf(lmx, v); lmn = fminf(lmn, v);
  }
  r_v[tid] = lmx; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) r_v[tid] = fmaxf(r_v[tid], r_v[tid + o]); __syncthreads(); }
  const float gmx = r_v[0]; __syncthreads();
  r_v[tid] = lmn; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) r_v[tid] = fminf(r_v[tid], r_v[tid + o]); __syncthreads(); }
  if (tid == 0) s_gmin = r_v[0];
  __syncthreads();
  const bool is_const = (gmx == s_gmin);
  if (is_const) {
    if (tid < K) sel[tid] = tie_idx[tid];
    __syncthreads();
  } else {
    for (int k = 0; k < K; ++k) {                          // K rounds of (max value, lowest index): wave shuffles + one LDS hop
      float bv = -INFINITY; int bi = 0x7fffffff;
      for (int p = tid; p < P; p += 256) { const float v = sv[p]; if (v > bv || (v == bv && p < bi)) { bv = v; bi = p; } }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if ((tid & 63) == 0) { r_v[tid >> 6] = bv; r_i[tid >> 6] = bi; }
      __syncthreads();
      if (tid == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) if (r_v[w] > bv || (r_v[w] == bv && r_i[w] < bi)) { bv = r_v[w]; bi = r_i[w]; }
        sel[k] = bi; sv[bi] = -INFINITY;
      }
      __syncthreads();
    }
  }
  if (tid == 0) cand_const[c] = is_const ? 1 : 0;
  for (int k = 0; k < K; ++k) {
    const int p = sel[k];
    const int n = p / npix, pix = p - n * npix;
    if (tid == 0) cand_val[c * K + k] = ncam[((size_t)n * 21 + c) * npix + pix];
    if (tid < 128) cand_feat[((size_t)c * K + k) * 128 + tid] = F[(size_t)p * 128 + tid];
  }
}
// merge world*K candidates per class -> top K -> weighted mean -> L2 normalise (F.normalize eps 1e-12)
__global__ __launch_bounds__(128) void proto_merge_kernel(const float* __restrict__ cand_val, const float* __restrict__ cand_feat, const int* __restrict__ cand_const,
                                                          float* __restrict__ protos, int world, int K, long rs_val, long rs_feat, long rs_const) {
  __shared__ float vals[512]; __shared__ int order[64]; __shared__ float red[2]; __shared__ int cst_s;
  const int c = blockIdx.x, tid = threadIdx.x;
  const int M = world * K;                                  // rank w's [21][K] values / [21][K][128] features / [21] flags at w * rs_*
  for (int i = tid; i < M; i += 128) { const int w = i / K, k = i - w * K; vals[i] = cand_val[(size_t)w * rs_val + c * K + k]; }
  if (tid == 0) {
    bool cst = true;
    for (int w = 0; w < world; ++w) cst = cst && cand_const[(size_t)w * rs_const + c];
    cst_s = cst ? 1 : 0;
  }
  __syncthreads();
  if (cst_s) { if (tid < K) order[tid] = tid; }             // fully tied: rank 0's set (global pixels first)
  else {
    // rank by counting (round 2 selected serially on one thread: K x M compares, 40 us at world 8): candidate i is the
    // (#{j : v_j > v_i or (v_j == v_i and j < i)})-th largest — the order of K rounds of "first maximum"
    for (int i = tid; i < M; i += 128) {
      const float v = vals[i];
      int r = 0;
      for (int j = 0; j < M; ++j) { const float u = vals[j]; r += (u > v || (u == v && j < i)) ? 1 : 0; }
      if (r < K) order[r] = i;
    }
  }
  __syncthreads();
  float acc = 0.f, wsum = 0.f;
  for (int k = 0; k < K; ++k) {
    const int i = order[k]; const int w = i / K, kk = i - w * K;
    const float v = cand_val[(size_t)w * rs_val + c * K + kk];
    acc += v * cand_feat[(size_t)w * rs_feat + ((size_t)c * K + kk) * 128 + tid];
    wsum += v;
  }
  const float pr = acc / wsum;
  const float ss = block_sum(pr * pr, red);
  protos[c * 128 + tid] = pr / fmaxf(sqrtf(ss), 1e-12f);
}

// ---- fn = F/max(||F||,1e-12); S_own = fn.protos_own^T ; S_oth = fn.protos_oth^T  on exact-f32 MFMA.
// One wave = 16 pixels at a time: [16 px] x [48 = 21 own | 21 other | 6 pad classes] x K=128 as 3 x 32
// v_mfma_f32_16x16x4_f32.  Lane (row = lane&15, g = lane>>4) loads F[row][16b + 4g .. +3] (8 x 16 B), which is
// both its A fragment (k-slot g <-> channel 16b+4g+e in MFMA (b,e)) and the piece of fn it writes back; the row
// norm is two xor-shuffles across g.  The prototypes live in registers in the matching B layout.
__global__ __launch_bounds__(256) void nce_sims_kernel(const float* __restrict__ F, const float* __restrict__ p_own, const float* __restrict__ p_oth,
                                                       float* __restrict__ fn, float* __restrict__ nrm, float* __restrict__ S_own, float* __restrict__ S_oth, int P) {
  const int lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4;
  f32x4 pb[3][8];                                            // B operand: class cc = t*16+col, channels 16b+4g..
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int cc = t * 16 + col;
    const float* src = cc < 21 ? p_own + cc * 128 : (cc < 42 ? p_oth + (cc - 21) * 128 : nullptr);
#pragma unroll
    for (int b = 0; b < 8; ++b) pb[t][b] = src ? *reinterpret_cast<const f32x4*>(src + b * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int ngrp = (P + 15) >> 4;
  for (int grp = blockIdx.x * 4 + (threadIdx.x >> 6); grp < ngrp; grp += gridDim.x * 4) {
    const int row = grp * 16 + col;
    const bool ok = row < P;
    const float* fr = F + (size_t)min(row, P - 1) * 128 + 4 * g;
    f32x4 a[8];
    float ss = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b) { a[b] = *reinterpret_cast<const f32x4*>(fr + b * 16); ss += a[b][0] * a[b][0] + a[b][1] * a[b][1] + a[b][2] * a[b][2] + a[b][3] * a[b][3]; }
    ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
    const float nr = sqrtf(ss);
    const float inv = 1.f / fmaxf(nr, 1e-12f);
    if (ok) {
      float* fo = fn + (size_t)row * 128 + 4 * g;
#pragma unroll
      for (int b = 0; b < 8; ++b) *reinterpret_cast<f32x4*>(fo + b * 16) = a[b] * inv;
      if (g == 0) nrm[row] = nr;
    }
    f32x4 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][e], pb[t][b][e], acc[t], 0, 0, 0);
    // acc[t][r] = F[row 4g+r] . P[class t*16+col]; scale by that row's 1/norm (held by lane 4g+r)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ir = __shfl(inv, 4 * g + r, 64);
      const int prow = grp * 16 + 4 * g + r;
      if (prow < P) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int cc = t * 16 + col;
          if (cc < 21) S_own[(size_t)prow * 21 + cc] = acc[t][r] * ir;
          else if (cc < 42) S_oth[(size_t)prow * 21 + cc - 21] = acc[t][r] * ir;
        }
      }
    }
  }
}

// ---- hard pixel sampling weights (contrast_train.py:302-331), single workgroup, P <= 8192.
//   key1 = S_own[p][y_p] (similarity order), key2 = random key or host flag.  w[p] = (#selections)/(2*half*C).
__global__ __launch_bounds__(1024) void intra_weights_kernel(const int* __restrict__ y, const float* __restrict__ S_own, int ld_s, const float* __restrict__ rkey,
                                                             const unsigned char* __restrict__ rand_flag, float* __restrict__ w, int P) {
  extern __shared__ unsigned long long keys[];              // [P2] sort buffer
  __shared__ int cnt[21], start[21], nclass;
  const int tid = threadIdx.x;
  int P2 = 1; while (P2 < P) P2 <<= 1;
  if (tid < 21) cnt[tid] = 0;
  __syncthreads();
  for (int p = tid; p < P; p += 1024) atomicAdd(&cnt[y[p]], 1);
  __syncthreads();
  if (tid == 0) { int s = 0, C = 0; for (int c = 0; c < 21; ++c) { start[c] = s; s += cnt[c]; if (cnt[c] > 0) ++C; } nclass = C; }
  for (int p = tid; p < P; p += 1024) w[p] = 0.f;
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && rand_flag) {                           // host-provided random half (RNG-parity mode)
      for (int p = tid; p < P; p += 1024) if (rand_flag[p]) { const int c = y[p]; const int half = cnt[c] / 2; if (cnt[c] >= 2) w[p] += 1.f / (2.f * half * nclass); }
      break;
    }
    for (int i = tid; i < P2; i += 1024) {
      unsigned long long k = ~0ull;
      if (i < P) {
        const float f = pass == 0 ? (ld_s == 1 ? S_own[i] : S_own[(size_t)i * ld_s + y[i]]) : rkey[i];
        k = ((unsigned long long)y[i] << 56) | ((unsigned long long)f2key(f) << 24) | (unsigned long long)i;   // i < 2^24
      }
      keys[i] = k;
    }
    __syncthreads();
    for (int k2 = 2; k2 <= P2; k2 <<= 1)
      for (int j = k2 >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < P2; i += 1024) {
          const int l = i ^ j;
          if (l > i) {
            const bool up = (i & k2) == 0;
            const unsigned long long a = keys[i], b = keys[l];
            if ((a > b) == up) { keys[i] = b; keys[l] = a; }
          }
        }
        __syncthreads();
      }
    for (int i = tid; i < P; i += 1024) {
      const unsigned long long k = keys[i];
      const int c = (int)(k >> 56); const int p = (int)(k & 0xFFFFFFull);
      const int len = cnt[c], r = i - start[c];
      if (len < 2) continue;
      const int half = len / 2;
      bool selected;
      if (pass == 0) { const int kk = (int)((double)len * 0.6); selected = r >= kk - half && r < kk; }
      else selected = r < half;
      if (selected) w[p] += 1.f / (2.f * half * nclass);
    }
    __syncthreads();
  }
}

// ---- hard-pixel sampling over the GLOBAL batch (data parallel; contrast_train.py:302-334 runs on the gathered batch).
// intra_pack: this rank's per-pixel record {label, own-class similarity, random key} for the all-gather.
__global__ void intra_pack_kernel(const int* __restrict__ y, const float* __restrict__ S_own, const float* __restrict__ rkey,
                                  float* __restrict__ rec, int P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int c = y[p];
  rec[p] = __int_as_float(c);
  rec[P + p] = S_own[(size_t)p * 21 + c];
  rec[2 * P + p] = rkey[p];
}
// intra_weights_global: `rec` = gathered records, rank r's [3][P] block at rec + r*rank_stride (global pixel g = r*P + p).  Per class c with
// len >= 2 pixels the reference keeps (a) a random half and (b) the pixels whose similarity rank lies in
// [int(0.6 len) - len/2, int(0.6 len)); both are order statistics of unique 56-bit keys (value << 24 | g), found by a 7-pass radix
// select — no global sort.  ONE WORKGROUP PER CLASS (grid 21; round 2 ran all 63 selections in a single workgroup, which at world 8 —
// 32 768 records per view — sat serially between the all-gather and the fused NCE launch): workgroup c counts its members (and which classes
// occur at all), runs the three selections of its class over the gathered records (per-wave private histograms: the first digit of a
// similarity or a uniform key is the same for most of a class, so a shared histogram serialises on two or three buckets), and writes the
// weights of ITS class's pixels of this rank: w[p] = scale * (#selections of p) / (2 * (len/2) * classes present).
__global__ __launch_bounds__(1024) void intra_weights_global_kernel(const float* __restrict__ rec, float* __restrict__ w, int P, int ranks,
                                                                    int own_rank, float scale, long rank_stride) {
  __shared__ unsigned hist[16][3][256];                       // [wave][selection][bucket]
  __shared__ unsigned long long prefix[3];
  __shared__ unsigned remaining[3];
  __shared__ int present[21], cnt_s, nclass_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = blockIdx.x;
  // Sweep over the records of ranks [r0, r1): every lane takes FOUR consecutive pixels per step as 16-byte loads of the label / similarity / key
  // rows (all three requested before any is used; no per-record division), fn(g, label, similarity bits, key bits) per record.  vec: rows 16-byte aligned.
  const bool vec = (P & 3) == 0 && (rank_stride & 3) == 0 && (reinterpret_cast<size_t>(rec) & 15) == 0;
  auto sweep = [&](int r0, int r1, bool need_keys, auto&& fn) {
    for (int r = r0; r < r1; ++r) {
      const float* base = rec + (size_t)r * rank_stride;
      for (int p0 = tid * 4; p0 < P; p0 += 4096) {
        float lb[4], sv[4] = {0.f, 0.f, 0.f, 0.f}, kv[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(base + p0);
          f32x4 s4 = (f32x4){0.f, 0.f, 0.f, 0.f}, k4 = s4;
          if (need_keys) { s4 = *reinterpret_cast<const f32x4*>(base + P + p0); k4 = *reinterpret_cast<const f32x4*>(base + 2 * (size_t)P + p0); }
#pragma unroll
          for (int e = 0; e < 4; ++e) { lb[e] = l4[e]; sv[e] = s4[e]; kv[e] = k4[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int p = min(p0 + e, P - 1);
            lb[e] = base[p];
            if (need_keys) { sv[e] = base[P + p]; kv[e] = base[2 * (size_t)P + p]; }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (p0 + e < P) fn(r * P + p0 + e, __float_as_int(lb[e]), f2key(sv[e]), f2key(kv[e]));
      }
    }
  };
  if (tid < 21) present[tid] = 0;
  if (tid == 0) cnt_s = 0;
  __syncthreads();
  int mine = 0;
  sweep(0, ranks, false, [&](int, int l, unsigned, unsigned) {
    present[l] = 1;                                          // (plain store: every writer writes the same value)
    mine += l == c ? 1 : 0;
  });
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if (lane == 0 && mine) atomicAdd(&cnt_s, mine);
  __syncthreads();
  if (tid == 0) { int C = 0; for (int k = 0; k < 21; ++k) C += present[k]; nclass_s = C; }
  const int len = cnt_s, half = len / 2, kk = (int)((double)len * 0.6);
  const int g0 = own_rank * P;
  if (len < 2) {                                             // (uniform) absent or single-pixel class: counted in C, no term (contrast_train.py:312-313)
    sweep(own_rank, own_rank + 1, false, [&](int g, int l, unsigned, unsigned) { if (l == c) w[g - g0] = 0.f; });
    return;
  }
  if (tid < 3) {
    prefix[tid] = 0ull;
    remaining[tid] = (unsigned)(tid == 0 ? kk - half : (tid == 1 ? kk : half));     // 0-based target rank within the class
  }
  for (int shift = 48; shift >= 0; shift -= 8) {
    for (int i = tid; i < 16 * 3 * 256; i += 1024) (&hist[0][0][0])[i] = 0u;
    __syncthreads();
    const unsigned long long pf0 = prefix[0], pf1 = prefix[1], pf2 = prefix[2];
    sweep(0, ranks, true, [&](int g, int l, unsigned sk, unsigned rk) {
      if (l != c) return;
      const unsigned long long ks = ((unsigned long long)sk << 24) | (unsigned long long)g, kr = ((unsigned long long)rk << 24) | (unsigned long long)g;
      const unsigned bs = (unsigned)(ks >> shift) & 255u, br = (unsigned)(kr >> shift) & 255u;
      if (shift == 48 || (ks >> (shift + 8)) == (pf0 >> (shift + 8))) atomicAdd(&hist[wv][0][bs], 1u);
      if (shift == 48 || (ks >> (shift + 8)) == (pf1 >> (shift + 8))) atomicAdd(&hist[wv][1][bs], 1u);
      if (shift == 48 || (kr >> (shift + 8)) == (pf2 >> (shift + 8))) atomicAdd(&hist[wv][2][br], 1u);
    });
    __syncthreads();
    if (wv < 3) {                                  // one wave per selection: lane l owns buckets 4l..4l+3 (summed over the 16 private copies)
      const int s_ = wv;
      unsigned cs[4] = {0u, 0u, 0u, 0u};
      for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[j] += hist[k][s_][lane * 4 + j];
      }
      const unsigned rem = remaining[s_];
      const unsigned sum = cs[0] + cs[1] + cs[2] + cs[3];
      unsigned inc = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
      unsigned cum = inc - sum;
      if (cum <= rem && rem < inc) {               // exactly one lane when the class holds more than `rem` pixels
        int b = -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (b < 0) { if (rem < cum + cs[j]) b = lane * 4 + j; else cum += cs[j]; }
        prefix[s_] |= (unsigned long long)b << shift;
        remaining[s_] = rem - cum;
      }
    }
    __syncthreads();
  }
  const float unit = scale / (2.f * (float)half * (float)nclass_s);
  const unsigned long long t0 = prefix[0], t1 = prefix[1], t2 = prefix[2];
  sweep(own_rank, own_rank + 1, true, [&](int g, int l, unsigned sk, unsigned rk) {
    if (l != c) return;
    const unsigned long long ks = ((unsigned long long)sk << 24) | (unsigned long long)g, kr = ((unsigned long long)rk << 24) | (unsigned long long)g;
    const int n_sel = (ks >= t0 && ks < t1 ? 1 : 0) + (kr < t2 ? 1 : 0);
    w[g - g0] = unit * (float)n_sel;
  });
}

// ---- per-pixel NCE losses + gradient w.r.t. the un-normalised features F (contrast_train.py:261-334).
// One wave = 64 pixels.  Phase A, one lane per pixel: the 21+21 similarities in registers -> the three InfoNCE
// terms and d(loss)/d(similarity) (44 values, written to the wave's LDS slab).  Phase B, exact-f32 MFMA:
//   d fn[ch][px] = sum_class [P_own|P_oth]^T[ch][class] * dS[class][px]      (8 channel tiles x 11 x 16x16x4 per 16 px)
// whose C layout gives every lane 4 consecutive channels of one pixel: fn is loaded and dF stored 16 B per lane;
// the F.normalize backward needs one dot product per pixel = two xor-shuffles across the lane groups.
__global__ __launch_bounds__(256) void nce_loss_grad_kernel(const float* __restrict__ fn, const float* __restrict__ nrm, const float* __restrict__ S_own,
                                                            const float* __restrict__ S_oth, const int* __restrict__ y_own, const int* __restrict__ y_oth,
                                                            const float* __restrict__ w_intra, const float* __restrict__ p_own, const float* __restrict__ p_oth,
                                                            float* __restrict__ dF, float* __restrict__ sums, int P, float coef_cross, float coef_intra) {
  __shared__ float dS[4][64][45];                            // [wave][pixel][class 0..43] (+1 pad)
  __shared__ float red[3][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, col = lane & 15, g = lane >> 4;
  const float itau = 10.f;                                   // 1 / 0.1
  // A operand of phase B: [P_own|P_oth]^T[ch = mt*16+col][class 4kk+g], kk = 0..10
  float pa[8][11];
#pragma unroll
  for (int kk = 0; kk < 11; ++kk) {
    const int cc = 4 * kk + g;
    const float* src = cc < 21 ? p_own + cc * 128 : (cc < 42 ? p_oth + (cc - 21) * 128 : nullptr);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) pa[mt][kk] = src ? src[mt * 16 + col] : 0.f;
  }
  float l_cross = 0.f, l_cross2 = 0.f, l_intra = 0.f;
  const int ngrp = (P + 63) >> 6;
  for (int grp = blockIdx.x * 4 + wv; grp < ngrp; grp += gridDim.x * 4) {
    // ---------------- phase A: lane = pixel
    {
      const int p = grp * 64 + lane;
      const bool ok = p < P;
      const int pc = min(p, P - 1);
      float so[21], st[21];
#pragma unroll
      for (int c = 0; c < 21; ++c) { so[c] = S_own[(size_t)pc * 21 + c]; st[c] = S_oth[(size_t)pc * 21 + c]; }
      const int yo = y_own[pc], yt = y_oth[pc];
      const float wi = ok ? w_intra[pc] : 0.f;
      float eo[21], et[21], sum_o = 0.f, sum_t = 0.f, s_yo = 0.f, e_yo_t = 0.f, e_yt_o = 0.f, e_yo_o = 0.f;
#pragma unroll
      for (int c = 0; c < 21; ++c) {
        eo[c] = expf(so[c] * itau); et[c] = expf(st[c] * itau);
        sum_o += eo[c]; sum_t += et[c];
        if (c == yo) { s_yo = so[c]; e_yo_t = et[c]; e_yo_o = eo[c]; }
        if (c == yt) e_yt_o = eo[c];
      }
      // semi-hard negatives: similarity ranks 3..12 (descending, lower index first on ties)
      float a2 = e_yo_o;
      unsigned negmask = 0;
#pragma unroll
      for (int c = 0; c < 21; ++c) {
        int rank = 0;
#pragma unroll
        for (int c2 = 0; c2 < 21; ++c2) rank += (so[c2] > so[c] || (so[c2] == so[c] && c2 < c)) ? 1 : 0;
        if (rank >= 3 && rank <= 12) { negmask |= 1u << c; a2 += eo[c]; }
      }
      if (ok) {
        l_cross += -logf(e_yo_t / sum_t) * coef_cross;       // 1.1 cross-prototype: other view's prototypes, own label
        l_cross2 += -logf(e_yt_o / sum_o) * coef_cross;      // 1.2 cross-pseudo-label: own prototypes, other label
        if (wi != 0.f) l_intra += -logf(e_yo_o / a2) * wi * coef_intra;
      }
      (void)s_yo;
      const float kc = ok ? coef_cross * itau : 0.f, ki = coef_intra * wi * itau;
#pragma unroll
      for (int c = 0; c < 21; ++c) {
        float go = kc * (eo[c] / sum_o - (c == yt ? 1.f : 0.f));
        if (wi != 0.f) go += ki * (((c == yo ? 1.f : 0.f) + ((negmask >> c) & 1u)) * eo[c] / a2 - (c == yo ? 1.f : 0.f));
        dS[wv][lane][c] = go;
        dS[wv][lane][21 + c] = kc * (et[c] / sum_t - (c == yo ? 1.f : 0.f));
      }
      dS[wv][lane][42] = 0.f; dS[wv][lane][43] = 0.f;
    }
    __builtin_amdgcn_wave_barrier();
    // ---------------- phase B: 4 sub-groups of 16 pixels
#pragma unroll 1
    for (int sub = 0; sub < 4; ++sub) {
      const int p = grp * 64 + sub * 16 + col;               // B/C column = pixel
      f32x4 acc[8];
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 11; ++kk) {
        const float bv = dS[wv][sub * 16 + col][4 * kk + g];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[mt][kk], bv, acc[mt], 0, 0, 0);
      }
      // acc[mt][r] = d fn[p][ch = mt*16 + 4g + r]
      const int pc = min(p, P - 1);
      f32x4 f[8];
      float dot = 0.f;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        f[mt] = *reinterpret_cast<const f32x4*>(fn + (size_t)pc * 128 + mt * 16 + 4 * g);
        dot += acc[mt][0] * f[mt][0] + acc[mt][1] * f[mt][1] + acc[mt][2] * f[mt][2] + acc[mt][3] * f[mt][3];
      }
      dot += __shfl_xor(dot, 16, 64); dot += __shfl_xor(dot, 32, 64);
      const float nr = nrm[pc];
      const float inv = nr > 1e-12f ? 1.f / nr : 0.f;        // below eps F.normalize divides by a constant: treat as dead
      if (p < P) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
          *reinterpret_cast<f32x4*>(dF + (size_t)p * 128 + mt * 16 + 4 * g) = (acc[mt] - f[mt] * dot) * inv;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { l_cross += __shfl_xor(l_cross, o, 64); l_cross2 += __shfl_xor(l_cross2, o, 64); l_intra += __shfl_xor(l_intra, o, 64); }
  if (lane == 0) { red[0][wv] = l_cross; red[1][wv] = l_cross2; red[2][wv] = l_intra; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (t != 0.f) atomicAdd(&sums[threadIdx.x], t);
  }
}


// ============================================================================================================================
// Fused pixel-to-prototype contrast (contrast_train.py:245-334): the product path.  Two launches per step for BOTH views:
//   nce_records_kernel  F, prototypes, labels            -> per-pixel record {label, similarity to its own class, random key}
//                                                           (the inputs of the hard-pixel sampling)            520 B / pixel
//   nce_fused_kernel    F, both prototype sets, labels,
//                       hard-pixel weights                -> dF and the three loss sums                         1.03 KB / pixel
// The features are read once per launch and nothing but dF / 12-byte records is written: the normalised features and the
// [P,21] similarity rows of nce_sims / nce_loss_grad (which stay as the reference formulation for the tests) never reach HBM.
// Similarities: one wave = 16 pixels x [21 own | 21 other | 6 pad] classes x 128 channels as 96 v_mfma_f32_16x16x4_f32 (exact
// f32), identical arithmetic in both kernels, so the record's similarity is bit-identical to the one the loss uses.
struct NceView { const float* F; const float* p_own; const float* p_oth; const int* y_own; const int* y_oth; const float* w_intra;
                 const float* rkey; float* rec; float* dF; };
struct NceArgs { NceView v[2]; int nviews, P; float coef_cross, coef_intra; float* sums; };

// similarities of 16 pixels (rows grp16*16 ..) to the 42 prototypes held in pb; returns acc[t][r] = S[pixel 4g+r][class t*16+col]
// already divided by the pixel's norm, and the lane's own row norm in `nr` (row = lane & 15)
// Feature tile of 16 pixels (8 KB) through LDS (the record pass): lane (col, g) needs 16 B at channel 16b + 4g of pixel `col` — 64-B pieces of 16
// different rows per load instruction when fetched straight into registers.  Here 8 LDS-DMA instructions fetch two WHOLE rows each (1 KB contiguous
// per wave instruction) and cost no registers, so a wave keeps TWO tiles (16 KB) in flight; the 16-B chunks are XOR-swizzled on the source side
// (chunk ^ row in the low 4 bits) so that the fragment reads of one chunk index from 16 rows fall into 16 different bank groups.  The tile belongs
// to ONE wave: no barrier, only that wave's vmcnt.  Measured at P = 2^22 per view: 2.92 -> 3.25 TB/s (exact f32), 2.83 -> 3.84 TB/s (split-bf16);
// with one tile in flight the staging alone changed nothing — these kernels are bound by bytes in flight per CU (latency), not by the load shape.
__device__ __forceinline__ void nce_tile_dma(const float* __restrict__ F, int P, int grp16, int lane, char* tile) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = 2 * i + (lane >> 5), pc = lane & 31;           // row inside the tile, physical 16-B chunk inside the row
    const int lc = (pc & 16) | ((pc ^ r) & 15);                  // logical chunk stored there
    glds16(F + (size_t)min(grp16 * 16 + r, P - 1) * 128 + lc * 4, tile + i * 1024);
  }
}
__device__ __forceinline__ void nce_frags_global(const float* __restrict__ F, int P, int grp16, int col, int g, f32x4 (&a)[8]) {
  const float* fr = F + (size_t)min(grp16 * 16 + col, P - 1) * 128 + 4 * g;
#pragma unroll
  for (int b = 0; b < 8; ++b) a[b] = *reinterpret_cast<const f32x4*>(fr + b * 16);
}
__device__ __forceinline__ void nce_frags_lds(const char* tile, int col, int g, f32x4 (&a)[8]) {
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int lc = 4 * b + g;
    a[b] = *reinterpret_cast<const f32x4*>(tile + col * 512 + (((lc & 16) | ((lc ^ col) & 15)) << 4));
  }
}

template <int NT = 3>   // NT = 2: only the first 32 classes (the record pass needs the own-view prototypes only)
__device__ __forceinline__ void nce_sims16(const f32x4 (&a)[8], int col, int g, const f32x4 (&pb)[3][8], f32x4 (&acc)[3], float& nr) {
  float ss = 0.f;
#pragma unroll
  for (int b = 0; b < 8; ++b) ss += a[b][0] * a[b][0] + a[b][1] * a[b][1] + a[b][2] * a[b][2] + a[b][3] * a[b][3];
  ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
  nr = sqrtf(ss);
  const float inv = 1.f / fmaxf(nr, 1e-12f);
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][e], pb[t][b][e], acc[t], 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ir = __shfl(inv, 4 * g + r, 64);
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t][r] *= ir;
  }
}

__device__ __forceinline__ void nce_load_protos(const float* __restrict__ p_own, const float* __restrict__ p_oth, int col, int g, f32x4 (&pb)[3][8]) {
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int cc = t * 16 + col;
    const float* src = cc < 21 ? p_own + cc * 128 : (cc < 42 && p_oth ? p_oth + (cc - 21) * 128 : nullptr);
#pragma unroll
    for (int b = 0; b < 8; ++b) pb[t][b] = src ? *reinterpret_cast<const f32x4*>(src + b * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}

// ---- split-bf16 form of the same contraction (the record pass in the bf16 and bf16x3 precision modes; the fp32 mode and the fused loss kernel
// keep the exact-f32 MFMA above — a split-bf16 form of the fused kernel was measured: 2.97 vs 2.79 TB/s at P = 2^22, equal at the real shape, at the
// price of register spills; not kept).
// Operands x = hi + lo (bf16 each), product lo.hi + hi.lo + hi.hi on v_mfma_f32_16x16x32_bf16: 36 MFMAs of 16 cycles per 16 pixels instead of
// 96 of 32 — the exact-f32 MFMA (1/16 of the bf16 rate) is what bounds the f32 kernels, not HBM.  Lane (col, g) holds channels
// 32c + 4g + e and 32c + 16 + 4g + e (e = 0..3) of chunk c: exactly its two 16-B feature loads 2c and 2c + 1, same map for both operands.
struct NceProtosX3 { bf16x8 hi[3][4], lo[3][4]; };
__device__ __forceinline__ void nce_load_protos_x3(const float* __restrict__ p_own, const float* __restrict__ p_oth, int col, int g, NceProtosX3& pp) {
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int cc = t * 16 + col;
    const float* src = cc < 21 ? p_own + cc * 128 : (cc < 42 && p_oth ? p_oth + (cc - 21) * 128 : nullptr);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      f32x4 q0 = (f32x4){0.f, 0.f, 0.f, 0.f}, q1 = q0;
      if (src) { q0 = *reinterpret_cast<const f32x4*>(src + 32 * c + 4 * g); q1 = *reinterpret_cast<const f32x4*>(src + 32 * c + 16 + 4 * g); }
      split_bf16x8(q0, q1, pp.hi[t][c], pp.lo[t][c]);
    }
  }
}
template <int NT = 3>
__device__ __forceinline__ void nce_sims16_x3(const f32x4 (&a)[8], int col, int g, const NceProtosX3& pp, f32x4 (&acc)[3], float& nr) {
  float ss = 0.f;
#pragma unroll
  for (int b = 0; b < 8; ++b) ss += a[b][0] * a[b][0] + a[b][1] * a[b][1] + a[b][2] * a[b][2] + a[b][3] * a[b][3];
  ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
  nr = sqrtf(ss);
  const float inv = 1.f / fmaxf(nr, 1e-12f);
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bf16x8 ah, al;
    split_bf16x8(a[2 * c], a[2 * c + 1], ah, al);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, pp.hi[t][c], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, pp.lo[t][c], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, pp.hi[t][c], acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ir = __shfl(inv, 4 * g + r, 64);
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t][r] *= ir;
  }
}

// One wave walks the 16-pixel tiles id = first + k * (waves in the grid) of BOTH views (view 0's tiles, then view 1's) with TWO tiles always in
// flight: tile k + 2 is requested into the buffer tile k was read from as soon as its fragments are in registers, so the wave never drains its
// queue (the unpipelined form — request two tiles, wait for both, compute both — had nothing in flight while it computed).  The record of pixel j
// of a tile is gathered into lane j (8 ds_bpermute from the lane that holds the pixel's own class), so every tile issues the SAME number of memory
// operations (8 LDS-DMA + label + key loads, 2-3 stores) and the only wait is one counted vmcnt.
template <bool X3>
__global__ __launch_bounds__(256) void nce_records_kernel(const NceArgs a) {
  // per wave: two 8-KB feature tiles, then per buffer 64 labels | 64 keys (lanes 0..15 are the tile's pixels).  ONE __shared__ object: beside a
  // second one the compiler puts a vmcnt(0) in front of every ds_read while LDS-DMA is in flight
  __shared__ __attribute__((aligned(16))) char lds[4][2 * 8192 + 2 * 512];
  const int lane = threadIdx.x & 63, col = lane & 15, g = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* tile = lds[wv];
  char* side = lds[wv] + 2 * 8192;
  const int P = a.P, ngrp = (P + 15) >> 4, total = a.nviews * ngrp, W = gridDim.x * 4;
  const int first = blockIdx.x * 4 + wv;
  if (first >= total) return;
  const bool has_key = a.v[0].rkey != nullptr;               // (both views or none: checked on the host)
  f32x4 pb[X3 ? 1 : 3][8];
  NceProtosX3 pp;
  // (labels and keys travel through LDS as well: a register that is the destination of a load in flight across the loop edge makes the compiler
  //  copy it at the latch behind a vmcnt(0))
  auto request = [&](int id, int buf) {                       // tile `id` (clamped: a dummy request keeps the operation count fixed) -> buffer buf
    id = min(id, total - 1);
    const int vi = id >= ngrp ? 1 : 0;
    const NceView& v = a.v[vi];
    const int grp = id - vi * ngrp;
    nce_tile_dma(v.F, P, grp, lane, tile + buf * 8192);
    const int p = min(grp * 16 + col, P - 1);
    glds4(v.y_own + p, side + buf * 512);
    if (has_key) glds4(v.rkey + p, side + buf * 512 + 256);
  };
  int cur_v = -1;
  auto body = [&](int id, int buf) {                          // (buf is a literal at both call sites)
    const int vi = id >= ngrp ? 1 : 0, grp = id - vi * ngrp;
    const NceView& v = a.v[vi];
    if (vi != cur_v) {                                        // prototypes of the view (wave-uniform branch; at most twice per wave)
      cur_v = vi;
      if constexpr (X3) nce_load_protos_x3(v.p_own, nullptr, col, g, pp);
      else {
        nce_load_protos(v.p_own, nullptr, col, g, pb);
#pragma unroll
        for (int t = 0; t < 2; ++t)                           // a USE inside the branch: the compiler waits for these loads here, not (with a
#pragma unroll                                                //  vmcnt(0), which would drain the tiles in flight) at their first use in the loop
          for (int b = 0; b < 8; ++b) asm volatile("" : "+v"(pb[t][b]));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (has_key) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");   // younger than tile k: stores(k-2) 3, tile k+1 10, stores(k-1) 3
    else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");                  // (2, 9, 2)
    f32x4 acc[3], fa[8]; float nr;
    nce_frags_lds(tile + buf * 8192, col, g, fa);
    const int yc = *reinterpret_cast<const int*>(side + buf * 512 + col * 4);
    const float kc = *reinterpret_cast<const float*>(side + buf * 512 + 256 + col * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    request(id + 2 * W, buf);
    if constexpr (X3) nce_sims16_x3<2>(fa, col, g, pp, acc, nr); else nce_sims16<2>(fa, col, g, pb, acc, nr);
    // lane j < 16 <- similarity of pixel j to its class c: held by lane (c & 15) + 16 * (j >> 2) in acc[c >> 4][j & 3]
    const int src = (yc & 15) + 16 * (col >> 2);
    float sim = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = __shfl(acc[t][r], src, 64);
        if ((yc >> 4) == t && (col & 3) == r) sim = x;
      }
    const int p = grp * 16 + col;
    if (g == 0 && p < P) {
      v.rec[p] = __int_as_float(yc); v.rec[P + p] = sim;
      if (has_key) v.rec[2 * P + p] = kc;
    }
  };
  request(first, 0);
  request(first + W, 1);
  for (int id = first; id < total; id += 2 * W) {
    body(id, 0);
    if (id + W < total) body(id + W, 1);
  }
}

__global__ __launch_bounds__(256, 2) void nce_fused_kernel(const NceArgs a) {
  __shared__ float slab[4][64][45];                          // per wave: [pixel][42 similarities -> 44 dS values | norm]
  __shared__ float red[3][4];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), col = lane & 15, g = lane >> 4;
  const float itau = 10.f;                                   // 1 / 0.1
  const int P = a.P, ngrp = (P + 63) >> 6;
  float l_cross = 0.f, l_cross2 = 0.f, l_intra = 0.f;
  {
    for (int gid = blockIdx.x * 4 + wv; gid < a.nviews * ngrp; gid += gridDim.x * 4) {   // both views share the grid
      const NceView& v = a.v[gid >= ngrp ? 1 : 0];
      const int grp = gid >= ngrp ? gid - ngrp : gid;
      // ---------------- similarities of 4 x 16 pixels -> the wave's slab.  The features of sub-tile s + 1 are requested before the 96 MFMAs of
      // sub-tile s (two register sets), the labels / weights of phase A before everything: no load of this phase is waited for right after its issue
      const int pA = min(grp * 64 + lane, P - 1);
      const int yo = v.y_own[pA], yt = v.y_oth[pA];
      const float wA = v.w_intra[pA];
      {
        const float *po = v.p_own, *pt = v.p_oth;            // (opaque copies: keeps the compiler from hoisting the 96 prototype registers
        asm volatile("" : "+s"(po), "+s"(pt));               //  of this phase and the 88 of phase B over the whole loop — they never overlap)
        f32x4 pb[3][8], f0[8], f1[8];
        nce_frags_global(v.F, P, grp * 4, col, g, f0);
        nce_load_protos(po, pt, col, g, pb);
        auto sims_of = [&](int sub, const f32x4 (&fa)[8]) {
          f32x4 acc[3]; float nr;
          nce_sims16(fa, col, g, pb, acc, nr);
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
              const int cc = t * 16 + col;
              if (cc < 42) slab[wv][sub * 16 + 4 * g + r][cc] = acc[t][r];
            }
          if (g == 0) slab[wv][sub * 16 + col][44] = nr;
        };
        nce_frags_global(v.F, P, grp * 4 + 1, col, g, f1); __builtin_amdgcn_sched_barrier(0); sims_of(0, f0);
        nce_frags_global(v.F, P, grp * 4 + 2, col, g, f0); __builtin_amdgcn_sched_barrier(0); sims_of(1, f1);
        nce_frags_global(v.F, P, grp * 4 + 3, col, g, f1); __builtin_amdgcn_sched_barrier(0); sims_of(2, f0);
        sims_of(3, f1);
      }
      __builtin_amdgcn_wave_barrier();
      // ---------------- phase A, lane = pixel: the three InfoNCE terms and d(loss)/d(similarity)
      {
        const int p = grp * 64 + lane;
        const bool ok = p < P;
        const int pc = min(p, P - 1);
        float so[21], st[21];
#pragma unroll
        for (int c = 0; c < 21; ++c) { so[c] = slab[wv][lane][c]; st[c] = slab[wv][lane][21 + c]; }
        const float wi = ok ? wA : 0.f;
        float eo[21], et[21], sum_o = 0.f, sum_t = 0.f, e_yo_t = 0.f, e_yt_o = 0.f, e_yo_o = 0.f;
#pragma unroll
        for (int c = 0; c < 21; ++c) {
          eo[c] = expf(so[c] * itau); et[c] = expf(st[c] * itau);
          sum_o += eo[c]; sum_t += et[c];
          if (c == yo) { e_yo_t = et[c]; e_yo_o = eo[c]; }
          if (c == yt) e_yt_o = eo[c];
        }
        float a2 = e_yo_o;                                   // semi-hard negatives: similarity ranks 3..12 (descending, lower index first on ties)
        unsigned negmask = 0;
#pragma unroll
        for (int c = 0; c < 21; ++c) {
          int rank = 0;
#pragma unroll
          for (int c2 = 0; c2 < 21; ++c2) rank += (so[c2] > so[c] || (so[c2] == so[c] && c2 < c)) ? 1 : 0;
          if (rank >= 3 && rank <= 12) { negmask |= 1u << c; a2 += eo[c]; }
        }
        if (ok) {
          l_cross += -logf(e_yo_t / sum_t) * a.coef_cross;   // cross-prototype: other view's prototypes, own label (:262)
          l_cross2 += -logf(e_yt_o / sum_o) * a.coef_cross;  // cross-pseudo-label: own prototypes, other label (:272)
          if (wi != 0.f) l_intra += -logf(e_yo_o / a2) * wi * a.coef_intra;
        }
        const float kc = ok ? a.coef_cross * itau : 0.f, ki = a.coef_intra * wi * itau;
#pragma unroll
        for (int c = 0; c < 21; ++c) {
          float go = kc * (eo[c] / sum_o - (c == yt ? 1.f : 0.f));
          if (wi != 0.f) go += ki * (((c == yo ? 1.f : 0.f) + ((negmask >> c) & 1u)) * eo[c] / a2 - (c == yo ? 1.f : 0.f));
          slab[wv][lane][c] = go;
          slab[wv][lane][21 + c] = kc * (et[c] / sum_t - (c == yo ? 1.f : 0.f));
        }
        slab[wv][lane][42] = 0.f; slab[wv][lane][43] = 0.f;
      }
      __builtin_amdgcn_wave_barrier();
      // ---------------- phase B: d fn[ch][px] = sum_class [P_own|P_oth]^T[ch][class] * dS[class][px], then the F.normalize backward
      {
        const float *po = v.p_own, *pt = v.p_oth;
        asm volatile("" : "+s"(po), "+s"(pt));
        f32x4 f0[8], f1[8];
        nce_frags_global(v.F, P, grp * 4, col, g, f0);       // (second touch of the wave's own 32 KB of features: cache-resident; sub-tile s + 1 is
        float pa[8][11];                                     //  requested before the 88 MFMAs of sub-tile s)    A operand: [P_own|P_oth]^T[ch = mt*16+col][class 4kk+g]
#pragma unroll
        for (int kk = 0; kk < 11; ++kk) {
          const int cc = 4 * kk + g;
          const float* src = cc < 21 ? po + cc * 128 : (cc < 42 ? pt + (cc - 21) * 128 : nullptr);
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) pa[mt][kk] = src ? src[mt * 16 + col] : 0.f;
        }
        auto grad_of = [&](int sub, f32x4 (&f)[8]) {
          const int p = grp * 64 + sub * 16 + col;           // B / C column = pixel
          f32x4 acc[8];
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < 11; ++kk) {
            const float bv = slab[wv][sub * 16 + col][4 * kk + g];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[mt][kk], bv, acc[mt], 0, 0, 0);
          }
          const float nr = slab[wv][sub * 16 + col][44];
          const float inv_s = 1.f / fmaxf(nr, 1e-12f);
          float dot = 0.f;
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) {
            f[mt] = f[mt] * inv_s;
            dot += acc[mt][0] * f[mt][0] + acc[mt][1] * f[mt][1] + acc[mt][2] * f[mt][2] + acc[mt][3] * f[mt][3];
          }
          dot += __shfl_xor(dot, 16, 64); dot += __shfl_xor(dot, 32, 64);
          const float inv = nr > 1e-12f ? 1.f / nr : 0.f;    // below eps F.normalize divides by a constant: treat as dead
          if (p < P) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
              *reinterpret_cast<f32x4*>(v.dF + (size_t)p * 128 + mt * 16 + 4 * g) = (acc[mt] - f[mt] * dot) * inv;
          }
        };
        nce_frags_global(v.F, P, grp * 4 + 1, col, g, f1); __builtin_amdgcn_sched_barrier(0); grad_of(0, f0);
        nce_frags_global(v.F, P, grp * 4 + 2, col, g, f0); __builtin_amdgcn_sched_barrier(0); grad_of(1, f1);
        nce_frags_global(v.F, P, grp * 4 + 3, col, g, f1); __builtin_amdgcn_sched_barrier(0); grad_of(2, f0);
        grad_of(3, f1);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { l_cross += __shfl_xor(l_cross, o, 64); l_cross2 += __shfl_xor(l_cross2, o, 64); l_intra += __shfl_xor(l_intra, o, 64); }
  if (lane == 0) { red[0][wv] = l_cross; red[1][wv] = l_cross2; red[2][wv] = l_intra; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float t = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    if (t != 0.f) atomicAdd(&a.sums[threadIdx.x], t);
  }
}

}  // namespace

#define GRID1(total) dim3((unsigned)(((total) + 255) / 256)), dim3(256)
#define ST ((hipStream_t)stream)

// workspace: planes * 24 bytes (wseg_plane_stats_workspace_bytes)
extern "C" size_t wseg_plane_stats_workspace_bytes(long planes) { return (size_t)planes * 24; }
extern "C" int wseg_plane_stats(const float* U, float* stats, long planes, int npix, void* workspace, void* stream) {
  WSEG_CHECK(U && stats && workspace && planes > 0 && npix > 0, "plane_stats: bad arguments");
  unsigned long long* kmax = (unsigned long long*)workspace;
  unsigned long long* kmin = kmax + planes;
  float* ksum = (float*)(kmin + planes);
  (void)hipMemsetAsync(kmax, 0x00, sizeof(unsigned long long) * planes, ST);
  (void)hipMemsetAsync(kmin, 0xFF, sizeof(unsigned long long) * planes, ST);
  (void)hipMemsetAsync(ksum, 0x00, sizeof(float) * planes, ST);
  const int chunks = std::max(1, std::min(std::min(64, (int)((long)npix / 8192)), (int)std::max(1L, 2048 / planes)));
  hipLaunchKernelGGL(plane_stats_partial_kernel, dim3((unsigned)(planes * chunks)), dim3(256), 0, ST, U, kmax, kmin, ksum, npix, chunks);
  hipLaunchKernelGGL(plane_stats_final_kernel, dim3((unsigned)((planes + 255) / 256)), dim3(256), 0, ST, kmax, kmin, ksum, stats, planes);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_cls_loss(const float* stats, const float* label20, float* loss_out, float* plane_bias, int N, int npix, float coef, void* stream) {
  WSEG_CHECK(stats && label20 && loss_out && plane_bias && N > 0, "cls_loss: bad arguments");
  hipLaunchKernelGGL(cls_loss_kernel, dim3(1), dim3(256), 0, ST, stats, label20, loss_out, plane_bias, N, npix, coef);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_rvmin_values(const float* U, const float* label20, float* q, unsigned char* argc, int N, int npix, void* stream) {
  WSEG_CHECK(U && label20 && q && argc && N > 0 && npix > 0, "rvmin_values: bad arguments");
  const long total = (long)N * npix;
  hipLaunchKernelGGL(rvmin_values_kernel, GRID1(total), 0, ST, U, label20, q, argc, npix, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

// k-th order statistic per row + partial sums.  largest=1: the k largest; res[row] = {thr, sum_strict, cnt_strict, cnt_tie}
// workspace: unsigned state[rows*4] + unsigned hist[rows*256]  (wseg_select_workspace_bytes)
extern "C" size_t wseg_select_workspace_bytes(int rows) { return (size_t)rows * (4 + 256) * sizeof(unsigned); }
extern "C" int wseg_select_kth(const float* vals, int rows, int n, int k, int largest, int use_abs, int relu_vals,
                               float* res, void* workspace, void* stream) {
  WSEG_CHECK(vals && res && workspace && rows > 0 && n > 0 && k >= 1 && k <= n, "select_kth: bad arguments (rows=%d n=%d k=%d)", rows, n, k);
  unsigned* state = (unsigned*)workspace;
  unsigned* hist = state + (size_t)rows * 4;
  const unsigned rank_small = largest ? (unsigned)(n - k + 1) : (unsigned)k;
  hipLaunchKernelGGL(select_init_kernel, dim3((rows * 256 + 255) / 256), dim3(256), 0, ST, state, hist, rows, rank_small, res);
  const int gx = std::max(1, std::min(128, (n + 4095) / 4096));  // histogram passes: enough workgroups to stream the rows (4 floats per thread and trip;
                                                                 // 32 / 64 / 128 per row measured equal, 16: +10 %, 8: +40 %)
  const int gs = std::max(1, std::min(32, (n + 8191) / 8192));   // final sums: few workgroups per row (their partials meet in same-address atomics)
  for (int shift = 24; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL(select_hist_kernel, dim3(gx, rows), dim3(256), 0, ST, vals, n, use_abs, state, hist, shift);
    hipLaunchKernelGGL(select_scan_kernel, dim3(rows), dim3(64), 0, ST, state, hist, shift, rows);
  }
  hipLaunchKernelGGL(select_sum_kernel, dim3(gs, rows), dim3(256), 0, ST, vals, n, use_abs, largest, relu_vals, state, res);
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_select_finish(const float* res, int rows, int k, int relu_vals, float scale, float* loss_out, void* stream) {
  WSEG_CHECK(res && loss_out && rows > 0, "select_finish: bad arguments");
  hipLaunchKernelGGL(select_finish_kernel, dim3(1), dim3(64), 0, ST, res, rows, k, relu_vals, scale, loss_out);
  WSEG_LAUNCH_CHECK();
  return 0;
}

// the 8 logged scalars of contrast_train.py:174, 389-395, 401-408 from the step's accumulators (one launch instead of a handful of torch scalar ops):
// acc = [cls1+cls2, (rvmin1+rvmin2)/2, er_sum, ecr, cross, cross2, intra, -]; out = [loss, cls, er, ecr, nce, intra, cross, cross2]
static __global__ void loss_finish_kernel(const float* __restrict__ acc, float er_coef, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  const float cls = acc[0] * 0.5f + acc[1], er = acc[2] * er_coef, ecr = acc[3], nce = acc[4] + acc[5] + acc[6];
  out[0] = cls + er + ecr + nce; out[1] = cls; out[2] = er; out[3] = ecr; out[4] = nce; out[5] = acc[6]; out[6] = acc[4]; out[7] = acc[5];
}
extern "C" int wseg_loss_finish(const float* acc, float er_coef, float* out8, void* stream) {
  WSEG_CHECK(acc && out8, "loss_finish: bad arguments");
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, ST, acc, er_coef, out8);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_rvmin_backward(const float* q, const unsigned char* argc, const float* res, const float* label20, float* dU,
                                   int N, int npix, int k, float coef, void* stream) {
  WSEG_CHECK(q && argc && res && label20 && dU, "rvmin_backward: bad arguments");
  const long total = (long)N * npix;
  hipLaunchKernelGGL(rvmin_bwd_kernel, GRID1(total), 0, ST, q, argc, res, label20, dU, npix, k, coef, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_norm_resize_forward(const float* U, const float* stats, const float* label20, float* out, int N, int S, int OS, void* stream) {
  WSEG_CHECK(U && stats && label20 && out && N > 0 && S > 0 && OS > 0, "norm_resize_forward: bad arguments");
  const long total = (long)N * 21 * OS * OS;
  hipLaunchKernelGGL(norm_resize_fwd_kernel, GRID1(total), 0, ST, U, stats, label20, out, S, OS, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_norm_resize_backward(const float* G, const float* U, const float* stats, const float* label20, float* dU, int N, int S, int OS, void* stream) {
  WSEG_CHECK(G && U && stats && label20 && dU && N > 0, "norm_resize_backward: bad arguments");
  hipLaunchKernelGGL(norm_resize_bwd_kernel, dim3(N * 21), dim3(256), 0, ST, G, U, stats, label20, dU, S, OS);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_er_ecr_prep(const float* c1, const float* c2, const float* r1, const float* r2, float* Gc1, float* Gc2,
                                float* dlt1, float* dlt2, float* er_sum, int N, int npix, float er_coef, void* stream) {
  WSEG_CHECK(c1 && c2 && r1 && r2 && Gc1 && Gc2 && dlt1 && dlt2 && er_sum, "er_ecr_prep: bad arguments");
  const long total = (long)N * npix;
  hipLaunchKernelGGL(er_ecr_prep_kernel, GRID1(total), 0, ST, c1, c2, r1, r2, Gc1, Gc2, dlt1, dlt2, er_sum, npix, er_coef, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_ecr_backward(const float* dlt, const float* res, float* Gr, int N, int per_row, int k, float coef, void* stream) {
  WSEG_CHECK(dlt && res && Gr, "ecr_backward: bad arguments");
  const long total = (long)N * per_row;
  hipLaunchKernelGGL(ecr_bwd_kernel, GRID1(total), 0, ST, dlt, res, Gr, per_row, k, coef, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_rows_resize_forward(const void* head, int ld, float* F, int N, int ih, int iw, int oh, int ow, int dtype, void* stream) {
  WSEG_CHECK(head && F && ld >= 128, "rows_resize_forward: bad arguments");
  const long total = (long)N * oh * ow * 128;
  if (dtype == WSEG_BF16) hipLaunchKernelGGL(rows_resize_fwd_kernel<WSEG_BF16>, GRID1(total), 0, ST, head, ld, F, ih, iw, oh, ow, total);
  else hipLaunchKernelGGL(rows_resize_fwd_kernel<WSEG_F32>, GRID1(total), 0, ST, head, ld, F, ih, iw, oh, ow, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_head_grad_fused(const float* dF, const float* d_cam_low, const void* head, void* d_head, int ld,
                                    int N, int ih, int iw, int oh, int ow, int dtype, void* stream) {
  WSEG_CHECK(dF && head && d_head && ld % 8 == 0 && ld >= 152, "head_grad_fused: bad arguments");
  const long total = (long)N * ih * iw * (ld / 8);
  if (dtype == WSEG_BF16) hipLaunchKernelGGL(head_grad_fused_kernel<WSEG_BF16>, GRID1(total), 0, ST, dF, d_cam_low, head, d_head, ld, ih, iw, oh, ow, total);
  else hipLaunchKernelGGL(head_grad_fused_kernel<WSEG_F32>, GRID1(total), 0, ST, dF, d_cam_low, head, d_head, ld, ih, iw, oh, ow, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_pseudo_label(const float* R, const float* label20, float bg_thr, int* y, float* ncam, int N, int npix, void* stream) {
  WSEG_CHECK(R && label20 && y && ncam && N > 0 && npix > 0, "pseudo_label: bad arguments");
  hipLaunchKernelGGL(pseudo_label_kernel, dim3(N), dim3(256), 0, ST, R, label20, bg_thr, y, ncam, npix);
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_proto_candidates(const float* ncam, const float* F, const int* tie_idx, float* cand_val, float* cand_feat, int* cand_const,
                                     int N, int npix, int K, void* stream) {
  WSEG_CHECK(ncam && F && tie_idx && cand_val && cand_feat && cand_const && K >= 1 && K <= 64, "proto_candidates: bad arguments");
  const size_t P = (size_t)N * npix;
  WSEG_CHECK(P * 4 <= 128 * 1024 && (size_t)K <= P, "proto_candidates: P=%zu too large for one workgroup's LDS", P);
  hipLaunchKernelGGL(proto_candidates_kernel, dim3(21), dim3(256), P * 4, ST, ncam, F, tie_idx, cand_val, cand_feat, cand_const, N, npix, K);
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_proto_merge(const float* cand_val, const float* cand_feat, const int* cand_const, float* protos, int world, int K,
                                long rank_stride, void* stream) {
  WSEG_CHECK(cand_val && cand_feat && cand_const && protos && world >= 1 && world * K <= 512 && K <= 64, "proto_merge: bad arguments");
  // rank_stride 0: contiguous [world][21][K] / [world][21][K][128] / [world][21]; else all three advance by rank_stride elements per rank
  const long rv = rank_stride ? rank_stride : 21L * K, rf = rank_stride ? rank_stride : 21L * K * 128, rc = rank_stride ? rank_stride : 21L;
  hipLaunchKernelGGL(proto_merge_kernel, dim3(21), dim3(128), 0, ST, cand_val, cand_feat, cand_const, protos, world, K, rv, rf, rc);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_nce_sims(const float* F, const float* p_own, const float* p_oth, float* fn, float* nrm, float* S_own, float* S_oth, int P, void* stream) {
  WSEG_CHECK(F && p_own && p_oth && fn && nrm && S_own && S_oth && P > 0, "nce_sims: bad arguments");
  hipLaunchKernelGGL(nce_sims_kernel, dim3(std::min(2048, (P + 63) / 64)), dim3(256), 0, ST, F, p_own, p_oth, fn, nrm, S_own, S_oth, P);
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_intra_weights(const int* y, const float* S_own, int ld_s, const float* rkey, const unsigned char* rand_flag, float* w, int P, void* stream) {
  WSEG_CHECK(y && S_own && w && (rkey || rand_flag) && P > 0 && P <= 8192 && (ld_s == 1 || ld_s == 21), "intra_weights: needs 0 < P <= 8192 (got %d), ld_s 1 or 21", P);
  int P2 = 1; while (P2 < P) P2 <<= 1;
  hipLaunchKernelGGL(intra_weights_kernel, dim3(1), dim3(1024), (size_t)P2 * 8, ST, y, S_own, ld_s, rkey, rand_flag, w, P);
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_intra_pack(const int* y, const float* S_own, const float* rkey, float* rec, int P, void* stream) {
  WSEG_CHECK(y && S_own && rkey && rec && P > 0, "intra_pack: bad arguments");
  hipLaunchKernelGGL(intra_pack_kernel, GRID1(P), 0, ST, y, S_own, rkey, rec, P);
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_intra_weights_global(const float* rec, float* w, int P, int ranks, int own_rank, float scale, long rank_stride, void* stream) {
  WSEG_CHECK(rec && w && P > 0 && ranks > 0 && own_rank >= 0 && own_rank < ranks && (long)P * ranks < (1L << 24) && rank_stride >= 3L * P,
             "intra_weights_global: bad arguments (P=%d ranks=%d own=%d)", P, ranks, own_rank);
  hipLaunchKernelGGL(intra_weights_global_kernel, dim3(21), dim3(1024), 0, ST, rec, w, P, ranks, own_rank, scale, rank_stride);   // one workgroup per class
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_nce_loss_grad(const float* fn, const float* nrm, const float* S_own, const float* S_oth, const int* y_own, const int* y_oth,
                                  const float* w_intra, const float* p_own, const float* p_oth, float* dF, float* sums, int P,
                                  float coef_cross, float coef_intra, void* stream) {
  WSEG_CHECK(fn && nrm && S_own && S_oth && y_own && y_oth && w_intra && p_own && p_oth && dF && sums && P > 0, "nce_loss_grad: bad arguments");
  hipLaunchKernelGGL(nce_loss_grad_kernel, dim3(std::min(2048, (P + 255) / 256)), dim3(256), 0, ST, fn, nrm, S_own, S_oth, y_own, y_oth, w_intra, p_own, p_oth,
                     dF, sums, P, coef_cross, coef_intra);
  WSEG_LAUNCH_CHECK();
  return 0;
}

static int nce_args(const wseg_nce_view* views, int nviews, int P, NceArgs& a, bool need_grad) {
  WSEG_CHECK(views && (nviews == 1 || nviews == 2) && P > 0, "nce: needs 1 or 2 views and P > 0");
  a.nviews = nviews; a.P = P;
  for (int i = 0; i < nviews; ++i) {
    const wseg_nce_view& w = views[i];
    WSEG_CHECK(w.F && w.p_own && w.y_own, "nce: view %d: F, p_own, y_own are required", i);
    if (need_grad) WSEG_CHECK(w.p_oth && w.y_oth && w.w_intra && w.dF, "nce_fused: view %d: p_oth, y_oth, w_intra, dF are required", i);
    else WSEG_CHECK(w.rec && (w.rkey != nullptr) == (views[0].rkey != nullptr), "nce_records: view %d: rec is required; rkey for every view or none", i);
    a.v[i] = NceView{w.F, w.p_own, w.p_oth, w.y_own, w.y_oth, w.w_intra, w.rkey, w.rec, w.dF};
  }
  return 0;
}
extern "C" int wseg_nce_records(const wseg_nce_view* views, int nviews, int P, int split_bf16, void* stream) {
  NceArgs a{};
  if (int rc = nce_args(views, nviews, P, a, false)) return rc;
  const dim3 grid(std::min(2048, (nviews * ((P + 15) / 16) + 3) / 4));
  if (split_bf16) hipLaunchKernelGGL(nce_records_kernel<true>, grid, dim3(256), 0, ST, a);
  else hipLaunchKernelGGL(nce_records_kernel<false>, grid, dim3(256), 0, ST, a);
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_nce_fused(const wseg_nce_view* views, int nviews, int P, float coef_cross, float coef_intra, float* sums, void* stream) {
  NceArgs a{};
  WSEG_CHECK(sums, "nce_fused: sums is null");
  if (int rc = nce_args(views, nviews, P, a, true)) return rc;
  a.coef_cross = coef_cross; a.coef_intra = coef_intra; a.sums = sums;
  hipLaunchKernelGGL(nce_fused_kernel, dim3(std::min(2048, (nviews * ((P + 63) / 64) + 3) / 4)), dim3(256), 0, ST, a);
  WSEG_LAUNCH_CHECK();
  return 0;
}
