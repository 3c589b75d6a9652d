// api.hip — library-level pieces of libwseg_hip.so: error state, zero page, weight packing, stem.
#include <stdarg.h>
#include <algorithm>
#include "common.h"


static thread_local char g_err[512] = "";
void wseg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* wseg_last_error(void) { return g_err; }
extern "C" int wseg_version(void) { return 100; }

namespace {

// ---- weight packing: cast + zero-pad (fwd) and cast + transpose (tr) ----------------------
template <int DT>
__global__ void pack_fwd_kernel(const float* __restrict__ w, void* __restrict__ out, int OC, int T, int IC, int OCp, int ICp) {
  const size_t total = (size_t)OCp * T * ICp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ic = (int)(i % ICp);
    const size_t r = i / ICp;
    const int t = (int)(r % T), oc = (int)(r / T);
    const float v = (oc < OC && ic < IC) ? w[((size_t)oc * T + t) * IC + ic] : 0.f;
    elem<DT>::st(out, i, v);
  }
}
template <int DT>
__global__ void pack_tr_kernel(const float* __restrict__ w, void* __restrict__ out, int OC, int T, int IC, int OCp, int ICp) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int oc0 = blockIdx.y * 32, ic0 = blockIdx.x * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int oc = oc0 + r, ic = ic0 + threadIdx.x;
    tile[r][threadIdx.x] = (oc < OC && ic < IC) ? w[((size_t)oc * T + t) * IC + ic] : 0.f;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int ic = ic0 + r, oc = oc0 + threadIdx.x;
    if (ic < ICp && oc < OCp) elem<DT>::st(out, ((size_t)ic * T + t) * OCp + oc, tile[threadIdx.x][r]);
  }
}

// ---- stem: conv1a 3->64 3x3 pad 1 from NCHW f32, fused BN-ReLU, NHWC out -------------------
template <int DT>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   void* raw, void* act, int N, int H, int W) {
  __shared__ float ws[64 * 27];
  __shared__ __attribute__((aligned(16))) float tile[64 * 64];      // [px][oc] staging for coalesced stores
  const int tid = threadIdx.x;
  for (int i = tid; i < 64 * 27; i += 256) ws[i] = w[i];
  const int nxb = (W + 63) / 64;
  const int bx = blockIdx.x % nxb;
  const int y = (blockIdx.x / nxb) % H;
  const int n = blockIdx.x / (nxb * H);
  const int px = tid & 63, cg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xg = bx * 64 + px;
  __syncthreads();
  float acc[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) acc[o] = 0.f;
  // same (ic, ky, kx) nesting order as a direct NCHW convolution
  for (int ic = 0; ic < 3; ++ic)
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + ky - 1;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = xg + kx - 1;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)n * 3 + ic) * H + iy) * W + ix];
        const float* wp = &ws[(cg * 16) * 27 + (ky * 3 + kx) * 3 + ic];
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = fmaf(v, wp[o * 27], acc[o]);
      }
    }
  const size_t pix0 = ((size_t)n * H + y) * W + bx * 64;
  const int npx = min(64, W - bx * 64);
  for (int pass = 0; pass < 2; ++pass) {
    void* dst = pass == 0 ? raw : act;
    if (!dst) continue;
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      const int oc = cg * 16 + o;
      float v = acc[o];
      if (pass == 1) v = fmaxf(v * scale[oc] + shift[oc], 0.f);
      tile[px * 64 + oc] = v;
    }
    __syncthreads();
    for (int i = tid; i < 64 * 8; i += 256) {                       // 8-channel vectors
      const int p = i >> 3, c8 = (i & 7) * 8;
      if (p >= npx) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[p * 64 + c8 + e];
      store8<DT>(dst, (pix0 + p) * 64 + c8, v);
    }
  }
}

}  // namespace

extern "C" int wseg_pack_weights(const float* master, void* fwd, void* tr, int OC, int T, int IC,
                                 int OCp, int ICp, int dtype, void* stream) {
  WSEG_CHECK(master && (fwd || tr), "pack_weights: null pointer");
  WSEG_CHECK(OCp >= OC && ICp >= IC && T >= 1, "pack_weights: bad padded shape");
  hipStream_t s = (hipStream_t)stream;
  if (fwd) {
    const size_t total = (size_t)OCp * T * ICp;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    if (dtype == WSEG_BF16) hipLaunchKernelGGL(pack_fwd_kernel<WSEG_BF16>, dim3(blocks), dim3(256), 0, s, master, fwd, OC, T, IC, OCp, ICp);
    else hipLaunchKernelGGL(pack_fwd_kernel<WSEG_F32>, dim3(blocks), dim3(256), 0, s, master, fwd, OC, T, IC, OCp, ICp);
  }
  if (tr) {
    dim3 grid((ICp + 31) / 32, (OCp + 31) / 32, T);
    if (dtype == WSEG_BF16) hipLaunchKernelGGL(pack_tr_kernel<WSEG_BF16>, grid, dim3(32, 8), 0, s, master, tr, OC, T, IC, OCp, ICp);
    else hipLaunchKernelGGL(pack_tr_kernel<WSEG_F32>, grid, dim3(32, 8), 0, s, master, tr, OC, T, IC, OCp, ICp);
  }
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_stem_conv(const float* x, const float* w, const float* scale, const float* shift,
                              void* raw, void* act, int N, int H, int W, int dtype, void* stream) {
  WSEG_CHECK(x && w && (raw || act), "stem_conv: null pointer");
  WSEG_CHECK(!act || (scale && shift), "stem_conv: act needs scale/shift");
  const long blocks = (long)N * H * ((W + 63) / 64);
  WSEG_CHECK(blocks > 0 && blocks < (1L << 31), "stem_conv: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == WSEG_BF16) hipLaunchKernelGGL(stem_kernel<WSEG_BF16>, dim3((unsigned)blocks), dim3(256), 0, s, x, w, scale, shift, raw, act, N, H, W);
  else hipLaunchKernelGGL(stem_kernel<WSEG_F32>, dim3((unsigned)blocks), dim3(256), 0, s, x, w, scale, shift, raw, act, N, H, W);
  WSEG_LAUNCH_CHECK();
  return 0;
}
