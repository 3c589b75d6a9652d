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
extern "C" int wseg_version(void) { return 101; }
extern "C" size_t wseg_sizeof_conv_desc(void) { return sizeof(wseg_conv_desc); }
extern "C" size_t wseg_sizeof_wgrad_desc(void) { return sizeof(wseg_wgrad_desc); }

namespace {

// ---- weight packing: cast + zero-pad (fwd) and cast + transpose (tr) ----------------------
template <int DT>
__global__ void pack_fwd_kernel(const float* __restrict__ w, void* __restrict__ out, int OC, int T, int IC, int OCp, int ICp, int ic_rot) {
  const size_t total = (size_t)OCp * T * ICp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ic = (int)(i % ICp);
    const size_t r = i / ICp;
    const int t = (int)(r % T), oc = (int)(r / T);
    const int src = ic + ic_rot >= IC ? ic + ic_rot - IC : ic + ic_rot;        // packed column ic <- master column (ic + ic_rot) % IC
    const float v = (oc < OC && ic < IC) ? w[((size_t)oc * T + t) * IC + src] : 0.f;
    elem<DT>::st(out, i, v);
  }
}
template <int DT>
__global__ void pack_tr_kernel(const float* __restrict__ w, void* __restrict__ out, int OC, int T, int IC, int OCp, int ICp, int ic_rot) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int oc0 = blockIdx.y * 32, ic0 = blockIdx.x * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int oc = oc0 + r, ic = ic0 + threadIdx.x;
    const int src = ic + ic_rot >= IC ? ic + ic_rot - IC : ic + ic_rot;
    tile[r][threadIdx.x] = (oc < OC && ic < IC) ? w[((size_t)oc * T + t) * IC + src] : 0.f;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int ic = ic0 + r, oc = oc0 + threadIdx.x;
    if (ic < ICp && oc < OCp) elem<DT>::st(out, ((size_t)ic * T + t) * OCp + oc, tile[threadIdx.x][r]);
  }
}

// ---- all transposed packs of one step in ONE launch: layer l = master f32 [OC][T][IC] at master + off_in[l]
//      -> [IC][T][OC] (dtype) at out + off_out[l] (elements).  table[l] = {first 32x32 tile, off_in, off_out, OC, T, IC}.
template <int DT>
__global__ void pack_tr_batch_kernel(const float* __restrict__ master, void* __restrict__ out, const long* __restrict__ table, int nlayers) {
  __shared__ float tile[32][33];
  const long bid = blockIdx.x;
  int lo = 0, hi = nlayers - 1;                    // last layer whose first tile <= bid
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (table[mid * 6] <= bid) lo = mid; else hi = mid - 1; }
  const long* e = table + lo * 6;
  const int OC = (int)e[3], T = (int)e[4], IC = (int)e[5];
  const int nti = (IC + 31) / 32, nto = (OC + 31) / 32;
  long r = bid - e[0];
  const int ti = (int)(r % nti); r /= nti;
  const int to = (int)(r % nto); const int t = (int)(r / nto);
  const float* w = master + e[1];
  const int oc0 = to * 32, ic0 = ti * 32;
  for (int rr = threadIdx.y; rr < 32; rr += 8) {
    const int oc = oc0 + rr, ic = ic0 + threadIdx.x;
    tile[rr][threadIdx.x] = (oc < OC && ic < IC) ? w[((size_t)oc * T + t) * IC + ic] : 0.f;
  }
  __syncthreads();
  for (int rr = threadIdx.y; rr < 32; rr += 8) {
    const int ic = ic0 + rr, oc = oc0 + threadIdx.x;
    if (ic < IC && oc < OC) elem<DT>::st(out, (size_t)e[2] + ((size_t)ic * T + t) * OC + oc, tile[threadIdx.x][rr]);
  }
}

// bf16 -> bf16 variant on 64x64 tiles (source = the bf16 weight mirror the fused SGD writes): 128-B row segments on both sides,
// a third of the f32 version's traffic.  table[l] = {first 64x64 tile, off_in, off_out, OC, T, IC}.
__global__ __launch_bounds__(256) void pack_tr_batch_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ out, const long* __restrict__ table, int nlayers) {
  __shared__ bf16_t tile[64][66];
  const long bid = blockIdx.x;
  int lo = 0, hi = nlayers - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (table[mid * 6] <= bid) lo = mid; else hi = mid - 1; }
  const long* e = table + lo * 6;
  const int OC = (int)e[3], T = (int)e[4], IC = (int)e[5];
  const int nti = (IC + 63) / 64, nto = (OC + 63) / 64;
  long r = bid - e[0];
  const int ti = (int)(r % nti); r /= nti;
  const int to = (int)(r % nto); const int t = (int)(r / nto);
  const bf16_t* w = src + e[1];
  const int oc0 = to * 64, ic0 = ti * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int rr = ty; rr < 64; rr += 4) {
    const int oc = oc0 + rr, ic = ic0 + tx;
    tile[rr][tx] = (oc < OC && ic < IC) ? w[((size_t)oc * T + t) * IC + ic] : (bf16_t)0;
  }
  __syncthreads();
  for (int rr = ty; rr < 64; rr += 4) {
    const int ic = ic0 + rr, oc = oc0 + tx;
    if (ic < IC && oc < OC) out[(size_t)e[2] + ((size_t)ic * T + t) * OC + oc] = tile[tx][rr];
  }
}

// ---- Dropout2d scales for one step from uniforms: out[i] = u[i] >= p ? 1/(1-p) : 0, p = p0 for i < split_at else p1
__global__ void dropout_scale_kernel(const float* __restrict__ u, float* __restrict__ out, long total, long split_at, float p0, float p1) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float p = i < split_at ? p0 : p1;
  out[i] = u[i] >= p ? 1.f / (1.f - p) : 0.f;
}

// ---- stem: conv1a 3->64 3x3 pad 1 from NCHW f32, fused BN-ReLU, NHWC out -------------------
// KC = 1: weights transposed to [k = (ky*3+kx)*3+ic][64 oc]: two adjacent channels are then one SGPR pair and the FMA loop is
// 216 v_pk_fma_f32 (input broadcast by op_sel) instead of 432 v_fma_f32 — same products, same order, same IEEE fused results.
template <int DT, int KC = 0>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   void* raw, void* act, int N, int H, int W) {
  // 64 pixels of one image row x 64 output channels per workgroup; wave `cg` owns channels [16cg, 16cg+16):
  // its weight index is wave-uniform, so the 432 weights come through the scalar cache (s_load) and feed
  // v_fmac as SGPR operands — no LDS traffic in the FMA loop.  Output goes through an LDS tile so the NHWC
  // stores are full 16-B vectors of consecutive channels.
  __shared__ __attribute__((aligned(16))) float tile[64 * 68];      // [px][oc], row padded by 4 floats
  const int tid = threadIdx.x;
  const int nxb = (W + 63) / 64;
  const int bx = blockIdx.x % nxb;
  const int y = (blockIdx.x / nxb) % H;
  const int n = blockIdx.x / (nxb * H);
  const int px = tid & 63, cg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xg = bx * 64 + px;
  float in[27];
#pragma unroll
  for (int ic = 0; ic < 3; ++ic)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + ky - 1;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = xg + kx - 1;
        in[(ky * 3 + kx) * 3 + ic] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[(((size_t)n * 3 + ic) * H + iy) * W + ix] : 0.f;
      }
    }
  float acc[16];
  if constexpr (KC == 1) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t* wg2 = reinterpret_cast<const f2_t*>(w + cg * 16);   // [k][64 oc]: row stride 32 pairs, wave-uniform base
    f2_t a2[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) a2[o] = (f2_t){0.f, 0.f};
#pragma unroll
    for (int ic = 0; ic < 3; ++ic)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const f2_t xv = {in[t * 3 + ic], in[t * 3 + ic]};
#pragma unroll
        for (int o = 0; o < 8; ++o) a2[o] = __builtin_elementwise_fma(xv, wg2[(t * 3 + ic) * 32 + o], a2[o]);
      }
#pragma unroll
    for (int o = 0; o < 8; ++o) { acc[2 * o] = a2[o].x; acc[2 * o + 1] = a2[o].y; }
  } else {
  const float* wg = w + (size_t)cg * 16 * 27;                       // [oc][ky][kx][ic], wave-uniform base
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    float a = 0.f;
    // (ic, ky, kx) order of a direct NCHW convolution
#pragma unroll
    for (int ic = 0; ic < 3; ++ic)
#pragma unroll
      for (int t = 0; t < 9; ++t) a = fmaf(in[t * 3 + ic], wg[o * 27 + t * 3 + ic], a);
    acc[o] = a;
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
      tile[px * 68 + oc] = v;
    }
    __syncthreads();
    for (int i = tid; i < 64 * 8; i += 256) {                       // 8-channel vectors
      const int p = i >> 3, c8 = (i & 7) * 8;
      if (p >= npx) continue;
      float v[8];
      const float4 a0 = *reinterpret_cast<const float4*>(&tile[p * 68 + c8]);
      const float4 a1 = *reinterpret_cast<const float4*>(&tile[p * 68 + c8 + 4]);
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
      store8<DT>(dst, (pix0 + p) * 64 + c8, v);
    }
  }
}

}  // namespace

extern "C" int wseg_pack_weights(const float* master, void* fwd, void* tr, int OC, int T, int IC,
                                 int OCp, int ICp, int ic_rot, int dtype, void* stream) {
  WSEG_CHECK(master && (fwd || tr), "pack_weights: null pointer");
  WSEG_CHECK(OCp >= OC && ICp >= IC && T >= 1 && ic_rot >= 0 && ic_rot < IC, "pack_weights: bad padded shape / rotation");
  hipStream_t s = (hipStream_t)stream;
  if (fwd) {
    const size_t total = (size_t)OCp * T * ICp;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    if (dtype == WSEG_BF16) hipLaunchKernelGGL(pack_fwd_kernel<WSEG_BF16>, dim3(blocks), dim3(256), 0, s, master, fwd, OC, T, IC, OCp, ICp, ic_rot);
    else hipLaunchKernelGGL(pack_fwd_kernel<WSEG_F32>, dim3(blocks), dim3(256), 0, s, master, fwd, OC, T, IC, OCp, ICp, ic_rot);
  }
  if (tr) {
    dim3 grid((ICp + 31) / 32, (OCp + 31) / 32, T);
    if (dtype == WSEG_BF16) hipLaunchKernelGGL(pack_tr_kernel<WSEG_BF16>, grid, dim3(32, 8), 0, s, master, tr, OC, T, IC, OCp, ICp, ic_rot);
    else hipLaunchKernelGGL(pack_tr_kernel<WSEG_F32>, grid, dim3(32, 8), 0, s, master, tr, OC, T, IC, OCp, ICp, ic_rot);
  }
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_stem_conv_kc(const float* x, const float* w_kc, const float* scale, const float* shift,
                                 void* raw, void* act, int N, int H, int W, int dtype, void* stream) {
  WSEG_CHECK(x && w_kc && (raw || act), "stem_conv_kc: null pointer");
  WSEG_CHECK(!act || (scale && shift), "stem_conv_kc: act needs scale/shift");
  const long blocks = (long)N * H * ((W + 63) / 64);
  WSEG_CHECK(blocks > 0 && blocks < (1L << 31), "stem_conv_kc: bad shape");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == WSEG_BF16) hipLaunchKernelGGL((stem_kernel<WSEG_BF16, 1>), dim3((unsigned)blocks), dim3(256), 0, s, x, w_kc, scale, shift, raw, act, N, H, W);
  else hipLaunchKernelGGL((stem_kernel<WSEG_F32, 1>), dim3((unsigned)blocks), dim3(256), 0, s, x, w_kc, scale, shift, raw, act, N, H, W);
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

extern "C" int wseg_pack_transposed_batch(const float* master, void* out, const long* table, int nlayers, long total_tiles, int dtype, void* stream) {
  WSEG_CHECK(master && out && table && nlayers > 0 && total_tiles > 0 && total_tiles < (1L << 31), "pack_transposed_batch: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (dtype == WSEG_BF16) hipLaunchKernelGGL(pack_tr_batch_kernel<WSEG_BF16>, dim3((unsigned)total_tiles), dim3(32, 8), 0, s, master, out, table, nlayers);
  else hipLaunchKernelGGL(pack_tr_batch_kernel<WSEG_F32>, dim3((unsigned)total_tiles), dim3(32, 8), 0, s, master, out, table, nlayers);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_dropout_scale(const float* u, float* out, long total, long split_at, float p0, float p1, void* stream) {
  WSEG_CHECK(u && out && total > 0 && p0 >= 0.f && p0 < 1.f && p1 >= 0.f && p1 < 1.f, "dropout_scale: bad arguments");
  hipLaunchKernelGGL(dropout_scale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, out, total, split_at, p0, p1);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_pack_transposed_batch_bf16(const void* mirror, void* out, const long* table, int nlayers, long total_tiles, void* stream) {
  WSEG_CHECK(mirror && out && table && nlayers > 0 && total_tiles > 0 && total_tiles < (1L << 31), "pack_transposed_batch_bf16: bad arguments");
  hipLaunchKernelGGL(pack_tr_batch_bf16_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)mirror, (bf16_t*)out, table, nlayers);
  WSEG_LAUNCH_CHECK();
  return 0;
}

// ---- K-concatenated weight packs of the two-source conv launches (wseg_conv_desc.in2): rows [W_a[r] | W_b[r]] built from the per-layer packs.
// ONE launch copies every piece of a step: piece p = `rows` rows of `cols16` 16-byte chunks from src + src_off (row stride ld_src) to dst + dst_off
// (row stride ld_dst); offsets and strides in 16-byte units; table[p] = {first chunk of the piece in the launch, src_off, dst_off, rows, cols16,
// ld_src, ld_dst}.  (Round 2 made these packs with six torch.cat calls per step.)
static __global__ void copy2d_batch_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const long* __restrict__ table, int npieces, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int lo = 0, hi = npieces - 1;                    // last piece whose first chunk <= i
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (table[mid * 7] <= i) lo = mid; else hi = mid - 1; }
    const long* t = table + lo * 7;
    const long j = i - t[0];
    const long row = j / t[4], c = j - row * t[4];
    dst[t[2] + row * t[6] + c] = src[t[1] + row * t[5] + c];
  }
}
extern "C" int wseg_copy2d_batch(const void* src, void* dst, const long* table, int npieces, long total_chunks, void* stream) {
  WSEG_CHECK(src && dst && table && npieces > 0 && total_chunks > 0 && ((size_t)src & 15) == 0 && ((size_t)dst & 15) == 0, "copy2d_batch: bad arguments");
  const unsigned blocks = (unsigned)std::min<long>((total_chunks + 255) / 256, 8192);
  hipLaunchKernelGGL(copy2d_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, table, npieces, total_chunks);
  WSEG_LAUNCH_CHECK();
  return 0;
}

// ---- split-bf16 weight pack (dtype WSEG_F32X3): every group of 32 consecutive K elements of an f32 [rows][K] matrix becomes
// [32 hi bf16 | 32 lo bf16] (hi = RNE bf16(w), lo = RNE bf16(w - hi)) in the same 128 bytes, so the conv kernel stages B tiles
// with the f32 path's addressing and reads both parts as ready-made MFMA operands.  One thread per 4 elements.
static __global__ void pack_x3_kernel(const float* src, unsigned char* dst, long nquads) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nquads) return;
  const long g = idx >> 3;
  const int q = (int)(idx & 7);
  const float4 v = *reinterpret_cast<const float4*>(src + g * 32 + q * 4);
  const float x[4] = {v.x, v.y, v.z, v.w};
  unsigned short hi[4], lo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = f32_to_bf16(x[e]);
    lo[e] = f32_to_bf16(x[e] - bf16_to_f32(hi[e]));
  }
  *reinterpret_cast<uint2*>(dst + g * 128 + q * 8) = make_uint2((unsigned)hi[0] | ((unsigned)hi[1] << 16), (unsigned)hi[2] | ((unsigned)hi[3] << 16));
  *reinterpret_cast<uint2*>(dst + g * 128 + 64 + q * 8) = make_uint2((unsigned)lo[0] | ((unsigned)lo[1] << 16), (unsigned)lo[2] | ((unsigned)lo[3] << 16));
}

extern "C" int wseg_pack_x3(const float* src, void* dst, long numel, void* stream) {
  WSEG_CHECK(src && dst && numel > 0 && numel % 32 == 0 && (const void*)src != dst, "pack_x3: numel must be a positive multiple of 32, out of place");
  const long nquads = numel / 4;
  hipLaunchKernelGGL(pack_x3_kernel, dim3((unsigned)((nquads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (unsigned char*)dst, nquads);
  WSEG_LAUNCH_CHECK();
  return 0;
}

// ---- f32 -> (hi, lo) bf16 planes, x = hi + lo to 16-17 bits (hi = RNE bf16(x), lo = RNE bf16(x - hi)): the split-bf16 mode's
// weight gradients of the large layers run the bf16 pixel-reduction kernel three times on these planes (lo.hi + hi.lo + hi.hi
// accumulate into dW), which is faster than splitting inside the f32-tile kernel.  8 elements per thread.
static __global__ void split_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long total) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 7 < total) {
    float v[8];
    load8<WSEG_F32>(in, i, v);
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16_t h0 = f32_to_bf16(v[2 * j]), h1 = f32_to_bf16(v[2 * j + 1]);
      h[j] = (unsigned)h0 | ((unsigned)h1 << 16);
      l[j] = (unsigned)f32_to_bf16(v[2 * j] - bf16_to_f32(h0)) | ((unsigned)f32_to_bf16(v[2 * j + 1] - bf16_to_f32(h1)) << 16);
    }
    *reinterpret_cast<uint4*>(hi + i) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(lo + i) = make_uint4(l[0], l[1], l[2], l[3]);
  } else {
    for (long k = i; k < total; ++k) {
      const bf16_t h0 = f32_to_bf16(in[k]);
      hi[k] = h0; lo[k] = f32_to_bf16(in[k] - bf16_to_f32(h0));
    }
  }
}

extern "C" int wseg_split_bf16(const float* in, void* hi, void* lo, long total, void* stream) {
  WSEG_CHECK(in && hi && lo && total > 0, "split_bf16: bad arguments");
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((total / 8 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)hi, (bf16_t*)lo, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}
