// common.h — shared device helpers for the wseg gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/wseg_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned short bf16_t;

#define WAVE 64

// 256 B of zeros: the source of every padded / out-of-range 16-byte LDS-DMA chunk.
static __device__ uint4 g_wseg_zero_page[16];   // per-TU, zero-initialised

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  // round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}

template <int DT> struct elem;            // DT = WSEG_F32 / WSEG_BF16
template <> struct elem<WSEG_F32> {
  typedef float type;
  static constexpr int size = 4;
  __device__ static __forceinline__ float ld(const void* p, size_t i) { return ((const float*)p)[i]; }
  __device__ static __forceinline__ void st(void* p, size_t i, float v) { ((float*)p)[i] = v; }
};
template <> struct elem<WSEG_BF16> {
  typedef bf16_t type;
  static constexpr int size = 2;
  __device__ static __forceinline__ float ld(const void* p, size_t i) { return bf16_to_f32(((const bf16_t*)p)[i]); }
  __device__ static __forceinline__ void st(void* p, size_t i, float v) { ((bf16_t*)p)[i] = f32_to_bf16(v); }
};

template <> struct elem<WSEG_F32X3> : elem<WSEG_F32> {};   // f32 storage; only the conv / wgrad inner products differ

// f32 x 8 -> (hi, lo) bf16 x 8 with x = hi + lo to 16-17 bits: hi = RNE bf16(x), lo = RNE bf16(x - hi)
__device__ __forceinline__ void split_bf16x8(const f32x4& p0, const f32x4& p1, bf16x8& hi, bf16x8& lo) {
  const float x[8] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3]};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bf16_t h = f32_to_bf16(x[e]);
    hi[e] = (short)h;
    lo[e] = (short)f32_to_bf16(x[e] - bf16_to_f32(h));
  }
}

__device__ __forceinline__ void split_bf16x8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bf16_t h = f32_to_bf16(x[e]);
    hi[e] = (short)h;
    lo[e] = (short)f32_to_bf16(x[e] - bf16_to_f32(h));
  }
}

// 8 consecutive elements <-> 8 floats (16 B for bf16, 32 B for f32); p must be 16-B aligned.
template <int DT> __device__ __forceinline__ void load8(const void* p, size_t i, float (&v)[8]);
template <> __device__ __forceinline__ void load8<WSEG_F32>(const void* p, size_t i, float (&v)[8]) {
  const float4* q = reinterpret_cast<const float4*>((const float*)p + i);
  float4 a = q[0], b = q[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void load8<WSEG_BF16>(const void* p, size_t i, float (&v)[8]) {
  uint4 a = *reinterpret_cast<const uint4*>((const bf16_t*)p + i);
  unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = __uint_as_float(w[j] << 16);
    v[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
  }
}
template <> __device__ __forceinline__ void load8<WSEG_F32X3>(const void* p, size_t i, float (&v)[8]) { load8<WSEG_F32>(p, i, v); }
template <int DT> __device__ __forceinline__ void store8(void* p, size_t i, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<WSEG_F32>(void* p, size_t i, const float (&v)[8]) {
  float4* q = reinterpret_cast<float4*>((float*)p + i);
  q[0] = make_float4(v[0], v[1], v[2], v[3]);
  q[1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void store8<WSEG_BF16>(void* p, size_t i, const float (&v)[8]) {
  unsigned w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    w[j] = (unsigned)f32_to_bf16(v[2 * j]) | ((unsigned)f32_to_bf16(v[2 * j + 1]) << 16);
  *reinterpret_cast<uint4*>((bf16_t*)p + i) = make_uint4(w[0], w[1], w[2], w[3]);
}

template <> __device__ __forceinline__ void store8<WSEG_F32X3>(void* p, size_t i, const float (&v)[8]) { store8<WSEG_F32>(p, i, v); }

// async 16-byte global -> LDS copy (LDS destination = wave-uniform base + lane*16)
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_wave_base, 16, 0, 0);
}

// async 4-byte global -> LDS copy (LDS destination = wave-uniform base + lane*4)
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)gsrc,
                                   (void __attribute__((address_space(3)))*)lds_wave_base, 4, 0, 0);
}

// XCD-aware bijective remap of a 1-D block id: blocks that share an XCD (id % 8) get a
// contiguous run of logical tiles, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}

// Row (pixel) index of a one- or two-segment activation matrix -> image row base, output coordinates and the
// input geometry of its segment.  Segment 2 (OH2 > 0) follows segment 1's N*OH*OW output rows / N*IH*IW input rows.
struct wseg_rowgeo { long in_base; int n_glob, oy, ox, IH, IW; };
template <typename D>
__device__ __forceinline__ wseg_rowgeo wseg_decode_row(const D& d, int m) {
  wseg_rowgeo g;
  const int M1 = d.N * d.OH * d.OW;
  if (d.OH2 == 0 || m < M1) {
    const int hw = d.OH * d.OW;
    const int n = m / hw, rem = m - n * hw;
    g.oy = rem / d.OW; g.ox = rem - g.oy * d.OW;
    g.IH = d.IH; g.IW = d.IW; g.n_glob = n;
    g.in_base = (long)n * d.IH * d.IW;
  } else {
    const int mm = m - M1, hw = d.OH2 * d.OW2;
    const int n = mm / hw, rem = mm - n * hw;
    g.oy = rem / d.OW2; g.ox = rem - g.oy * d.OW2;
    g.IH = d.IH2; g.IW = d.IW2; g.n_glob = d.N + n;
    g.in_base = (long)d.N * d.IH * d.IW + (long)n * d.IH2 * d.IW2;
  }
  return g;
}

void wseg_set_error(const char* fmt, ...);
#define WSEG_CHECK(cond, ...)           \
  do {                                  \
    if (!(cond)) {                      \
      wseg_set_error(__VA_ARGS__);      \
      return -1;                        \
    }                                   \
  } while (0)
#define WSEG_LAUNCH_CHECK()                                                        \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ != hipSuccess) {                                                        \
      wseg_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return -2;                                                                   \
    }                                                                              \
  } while (0)
