// pcm.hip — PCM affinity refinement (network/resnet38_contrast.py:63-75) without ever
// materialising the hw x hw affinity.  Exact-f32 MFMA (v_mfma_f32_16x16x4_f32) in both modes.
//
//   S[i,j]   = Fh_i . Fh_j                       (Fh = L2-normalised f9 features, [hw][192])
//   forward : out[j][c] = sum_i relu(S[i,j]) * G[i][c]           G[:,21] == 1  -> out[j][21] = column sum
//             cam_rv[c][j] = out[j][c] / (out[j][21] + 1e-5)
//   backward: W[i,j] = (S[i,j] > 0) * sum_c P[i][c] * Q[j][c]
//             dFh[j][k] += sum_i W[i,j] * Fh[i][k]
//     called twice, (P,Q) = (G,DN) and (DN,G): the second call is the same sum over the transposed
//     dS, which yields the row-side gradient with the identical data flow.
//
// One workgroup = 64 columns j of one image (one 16-column sub-tile per wave, its Fh_j fragment
// held in 48 VGPRs); rows i stream through LDS in 32-row tiles by LDS-DMA (double buffered,
// XOR-swizzled 16-B chunks).  The S accumulator's C/D layout (col = lane&15, row = 4*(lane>>4)+reg)
// IS the A-operand layout of the second product when MFMA r takes k-index g <-> row 4g+r, so
// relu(S) / W goes from accumulator to operand without touching LDS.
#include "common.h"

namespace {

constexpr int KF = 192;                     // feature channels
constexpr int FROW = KF * 4;                // 768 B per Fh row
constexpr int IT = 32;                      // rows per i-tile
constexpr int F_TILE = IT * FROW;           // 24576
constexpr int G_TILE = IT * 128;            // 4096  ([32 rows][32 f32])
constexpr int BUF = F_TILE + G_TILE;        // 28672 per stage

template <int BWD>
__global__ __launch_bounds__(256, 2) void pcm_kernel(const float* __restrict__ Fh, const float* __restrict__ Pm,
                                                     const float* __restrict__ Qm, float* __restrict__ out0,
                                                     float* __restrict__ out1, int hw) {
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y;
  const int j0 = blockIdx.x * 64 + wid * 16;
  const int col = lane & 15, g = lane >> 4;
  const char* zero = reinterpret_cast<const char*>(g_wseg_zero_page);
  const float* Fn = Fh + (size_t)n * hw * KF;
  const float* Pn = Pm + (size_t)n * hw * 32;

  // ---- this wave's column fragment: Fh[j0+col][blk*16 + 4g + e]
  f32x4 bj[12];
  {
    const int jr = min(j0 + col, hw - 1);
    const f32x4* src = reinterpret_cast<const f32x4*>(Fn + (size_t)jr * KF);
#pragma unroll
    for (int b = 0; b < 12; ++b) bj[b] = src[b * 4 + g];
  }
  f32x4 qj[2];
  if (BWD) {
    const int jr = min(j0 + col, hw - 1);
    const f32x4* src = reinterpret_cast<const f32x4*>(Qm + ((size_t)n * hw + jr) * 32);
    qj[0] = src[g]; qj[1] = src[4 + g];
  }

  // ---- staging: Fh tile 32 rows x 48 chunks = 1536 chunks -> 6 per thread (24 wave pieces of 64
  //      chunks); P/G tile 32 rows x 8 chunks = 256 chunks -> 1 per thread.
  auto stage = [&](int buf, int i0) {
    char* lf = smem + buf * BUF;
    char* lg = lf + F_TILE;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int piece = wid * 6 + t;                       // 1 KiB pieces, linear in the tile
      const int ci = piece * 64 + lane;                    // physical chunk index in the tile
      const int row = ci / 48, pc = ci - row * 48;
      const int lc = (pc & ~15) | ((pc ^ row) & 15);       // logical chunk (low 4 bits swizzled)
      const int i = i0 + row;
      const char* src = (i < hw) ? reinterpret_cast<const char*>(Fn + (size_t)i * KF) + lc * 16 : zero + (lane & 15) * 16;
      glds16(src, lf + piece * 1024);
    }
    {
      const int row = wid * 8 + (lane >> 3), pc = lane & 7;
      const int lc = pc ^ ((row >> 1) & 7);
      const int i = i0 + row;
      const char* src = (i < hw) ? reinterpret_cast<const char*>(Pn + (size_t)i * 32) + lc * 16 : zero + (lane & 15) * 16;
      glds16(src, lg + wid * 1024);
    }
  };

  constexpr int NT = BWD ? 12 : 2;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nti = (hw + IT - 1) / IT;
  stage(0, 0);
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < nti; ++it) {
    if (it + 1 < nti) stage(cur ^ 1, (it + 1) * IT);
    const char* lf = smem + cur * BUF;
    const char* lg = lf + F_TILE;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int row = sub * 16 + col;                      // A-operand row (pixel i) for this lane
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 12; ++b) {
        const int lc = b * 4 + g;
        const int pc = (lc & ~15) | ((lc ^ row) & 15);
        const f32x4 a = *reinterpret_cast<const f32x4*>(lf + row * FROW + pc * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], bj[b][e], s, 0, 0, 0);
      }
      // s[r] = S[i = sub*16 + 4g + r][j = col]
      float wv[4];
      if (!BWD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) wv[r] = fmaxf(s[r], 0.f);
      } else {
        f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const int pc = (k2 * 4 + g) ^ ((row >> 1) & 7);
          const f32x4 a = *reinterpret_cast<const f32x4*>(lg + row * 128 + pc * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) t = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], qj[k2][e], t, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) wv[r] = s[r] > 0.f ? t[r] : 0.f;
      }
      // accumulate out[j][nn] += sum_i W[i,j] * V[i][nn]; MFMA r: k-index g <-> row i = 4g + r
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int vi = sub * 16 + 4 * g + r;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float v;
          if (!BWD) {
            const int lc = t * 4 + (col >> 2);             // 16-B chunk of G row holding column t*16+col
            const int pc = lc ^ ((vi >> 1) & 7);
            v = *reinterpret_cast<const float*>(lg + vi * 128 + pc * 16 + (col & 3) * 4);
          } else {
            const int lc = t * 4 + (col >> 2);
            const int pc = (lc & ~15) | ((lc ^ vi) & 15);
            v = *reinterpret_cast<const float*>(lf + vi * FROW + pc * 16 + (col & 3) * 4);
          }
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r], v, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // acc[t][reg] <-> (j = j0 + 4g + reg, nn = t*16 + col)
  if (!BWD) {
    // out0 = cam_rv [N][21][hw], out1 = den [N][hw]
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float den = __shfl(acc[1][reg], g * 16 + 5, 64);       // column c = 21 lives in tile 1, col 5
      const int j = j0 + 4 * g + reg;
      if (j < hw) {
        const float inv = 1.f / (den + 1e-5f);
        out0[((size_t)n * 21 + col) * hw + j] = acc[0][reg] / (den + 1e-5f);
        if (col < 5) out0[((size_t)n * 21 + 16 + col) * hw + j] = acc[1][reg] / (den + 1e-5f);
        if (col == 5) out1[(size_t)n * hw + j] = den;
        (void)inv;
      }
    }
  } else {
    // out0 = dFh [N][hw][192], accumulated (this launch is the only writer of row j)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int j = j0 + 4 * g + reg;
      if (j < hw) {
        float* dst = out0 + ((size_t)n * hw + j) * KF + col;
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[t * 16] += acc[t][reg];
      }
    }
  }
}


// ---- bf16-MFMA variant (throughput mode).  Same data flow as pcm_kernel, but every product runs on
// v_mfma_f32_16x16x32_bf16:
//   S      : Fb_i . Fb_j over 192 channels = 6 MFMAs per 16x16 sub-tile (A: ds_read_b128 rows, B: registers)
//   t      : P_i . Q_j over the 32 gate channels = 1 MFMA
//   acc    : out[j][n] += sum_{i in 32-row tile} W[i,j] V[i][n] = 1 MFMA per 16 columns n; its A operand is built
//            from BOTH sub-tiles' accumulators with k-slot (g,e) <-> row (e>>2)*16 + 4g + (e&3), so the values a
//            lane needs are its own; the matching B operand is two ds_read_b64_tr_b16 (rows 4g..4g+3 and 16+4g..).
// Inputs are bf16 copies (Fb [N][hw][192], P/Q [N][hw][32]); outputs stay f32.
// Backward (BWD): the gate-channel product t_ij = P_i . Q_j = sum_c G_ic g_jc / D_j - (sum_c g_jc rv_jc) / D_j is a DIFFERENCE of two sums that nearly
// cancel wherever G_i is close to the refined map rv_j (rv_j is the affinity-weighted mean of the G_i) — with both operands rounded to bf16 BEFORE the
// cancellation the f9 / f8_3 / f8_4 gradients came out at cosine 0.94 against the f32 kernels under identical gate decisions (round 3, the injected-
// decision test).  It is therefore computed in split precision: P = Ph + Pl, Q = Qh + Ql (bf16 each, 16-17 significant bits), t = Pl.Qh + Ph.Ql + Ph.Qh —
// 3 MFMAs instead of 1 beside the 6 of S.  S itself only decides the ReLU gate, and W = gate * t is rounded to bf16 AFTER the cancellation.
template <int BWD>
__global__ __launch_bounds__(256, 2) void pcm_bf16_kernel(const bf16_t* __restrict__ Fb, const bf16_t* __restrict__ Pm,
                                                          const bf16_t* __restrict__ Qm, const bf16_t* __restrict__ Pl_,
                                                          const bf16_t* __restrict__ Ql_, float* __restrict__ out0,
                                                          float* __restrict__ out1, int hw) {
  constexpr int FR = KF * 2;                  // 384 B per Fb row
  constexpr int F_T = IT * FR;                // 12288
  constexpr int P_T = IT * 64;                // 2048 ([32 rows][32 bf16])
  constexpr int STG = F_T + (BWD ? 2 : 1) * P_T;   // backward: P tile hi + lo
  __shared__ __attribute__((aligned(16))) char smem[2 * STG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y;
  const int j0 = blockIdx.x * 64 + wid * 16;
  const int col = lane & 15, g = lane >> 4;
  const char* zero = reinterpret_cast<const char*>(g_wseg_zero_page);
  const bf16_t* Fn = Fb + (size_t)n * hw * KF;
  const bf16_t* Pn = Pm + (size_t)n * hw * 32;

  bf16x8 bj[6];                               // Fb[j0+col][32b + 8g + e]
  {
    const int jr = min(j0 + col, hw - 1);
    const bf16x8* src = reinterpret_cast<const bf16x8*>(Fn + (size_t)jr * KF);
#pragma unroll
    for (int b = 0; b < 6; ++b) bj[b] = src[b * 4 + g];
  }
  bf16x8 qj = {0, 0, 0, 0, 0, 0, 0, 0}, ql = {0, 0, 0, 0, 0, 0, 0, 0};
  if (BWD) {
    const int jr = min(j0 + col, hw - 1);
    qj = reinterpret_cast<const bf16x8*>(Qm + ((size_t)n * hw + jr) * 32)[g];
    ql = reinterpret_cast<const bf16x8*>(Ql_ + ((size_t)n * hw + jr) * 32)[g];
  }
  const bf16_t* Pln = BWD ? Pl_ + (size_t)n * hw * 32 : nullptr;

  // staging: Fb tile 32 rows x 24 chunks = 768 chunks -> 3 per thread (12 pieces); P tile 32 rows x 4 chunks = 128 chunks
  auto stage = [&](int buf, int i0) {
    char* lf = smem + buf * STG;
    char* lp = lf + F_T;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int piece = wid * 3 + t;
      const int ci = piece * 64 + lane;
      const int row = ci / 24, pc = ci - row * 24;
      const int lc = (pc & ~7) | ((pc ^ (row >> 1)) & 7);
      const int i = i0 + row;
      const char* src = (i < hw) ? reinterpret_cast<const char*>(Fn + (size_t)i * KF) + lc * 16 : zero + (lane & 15) * 16;
      glds16(src, lf + piece * 1024);
    }
    if (wid < 2 || BWD) {                                    // 2 KiB: waves 0,1 (backward: waves 2,3 stage the lo part behind it)
      const int w2 = wid & 1;
      const int ci = w2 * 64 + lane;
      const int row = ci >> 2, pc = ci & 3;
      const int i = i0 + row;
      const bf16_t* Psrc = wid < 2 ? Pn : Pln;
      const char* src = (i < hw) ? reinterpret_cast<const char*>(Psrc + (size_t)i * 32) + pc * 16 : zero + (lane & 15) * 16;
      glds16(src, lp + (wid >> 1) * P_T + w2 * 1024);
    }
  };

  constexpr int NT = BWD ? 12 : 2;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int q = (lane & 15) >> 2, p = lane & 3;

  const int nti = (hw + IT - 1) / IT;
  stage(0, 0);
  __syncthreads();
  int cur = 0;
  for (int it = 0; it < nti; ++it) {
    if (it + 1 < nti) stage(cur ^ 1, (it + 1) * IT);
    const char* lf = smem + cur * STG;
    const char* lp = lf + F_T;
    float wv[2][4];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int row = sub * 16 + col;
      f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const int lc = b * 4 + g;
        const int pc = (lc & ~7) | ((lc ^ (row >> 1)) & 7);
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(lf + row * FR + pc * 16);
        sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bj[b], sv, 0, 0, 0);
      }
      if (!BWD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) wv[sub][r] = fmaxf(sv[r], 0.f);
      } else {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(lp + row * 64 + g * 16);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(lp + P_T + row * 64 + g * 16);
        f32x4 tv = (f32x4){0.f, 0.f, 0.f, 0.f};
        tv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, qj, tv, 0, 0, 0);       // small terms first
        tv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, ql, tv, 0, 0, 0);
        tv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qj, tv, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) wv[sub][r] = sv[r] > 0.f ? tv[r] : 0.f;
      }
    }
    // A operand of the accumulate product: k-slot (g, e) <-> tile row (e>>2)*16 + 4g + (e&3)
    bf16x8 wa;
#pragma unroll
    for (int e = 0; e < 8; ++e) wa[e] = (short)f32_to_bf16(wv[e >> 2][e & 3]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      bf16x8 vb;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = h * 16 + 4 * g + q;                  // lane 4q+p of the 16-lane group addresses row q of the 4-row block
        const char* base;
        int off;
        if (!BWD) { base = lp; off = row * 64 + t * 32 + p * 8; }
        else {
          const int lc = t * 2 + (p >> 1);
          const int pc = (lc & ~7) | ((lc ^ (row >> 1)) & 7);
          base = lf; off = row * FR + pc * 16 + (p & 1) * 8;
        }
        const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)(base + off));
#pragma unroll
        for (int e = 0; e < 4; ++e) vb[h * 4 + e] = v[e];
      }
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, vb, acc[t], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }

  if (!BWD) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const float den = __shfl(acc[1][reg], g * 16 + 5, 64);
      const int j = j0 + 4 * g + reg;
      if (j < hw) {
        out0[((size_t)n * 21 + col) * hw + j] = acc[0][reg] / (den + 1e-5f);
        if (col < 5) out0[((size_t)n * 21 + 16 + col) * hw + j] = acc[1][reg] / (den + 1e-5f);
        if (col == 5) out1[(size_t)n * hw + j] = den;
      }
    }
  } else {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int j = j0 + 4 * g + reg;
      if (j < hw) {
        float* dst = out0 + ((size_t)n * hw + j) * KF + col;
#pragma unroll
        for (int t = 0; t < NT; ++t) dst[t * 16] += acc[t][reg];
      }
    }
  }
}

// f32 [rows][cols] -> bf16 copy
__global__ void to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long total) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < total) {
    const float4 v = *reinterpret_cast<const float4*>(in + i);
    uint2 o;
    o.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
    o.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(out + i) = o;
  } else {
    for (long k = i; k < total; ++k) out[k] = f32_to_bf16(in[k]);
  }
}

// f32 -> (hi, lo) bf16 pair: hi = RNE bf16(x), lo = RNE bf16(x - hi)
__global__ void split_hi_lo_kernel(const float* __restrict__ in, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float x = in[i];
  const unsigned short h = f32_to_bf16(x);
  hi[i] = h;
  lo[i] = f32_to_bf16(x - bf16_to_f32(h));
}

// Fh = F / (||F|| + 1e-5)   (one wave per pixel row of 192 channels)
template <int DT>
__global__ void l2norm_fwd_kernel(const void* __restrict__ F, int ldf, float* __restrict__ Fh, float* __restrict__ nrm, long rows) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[3];
  float ss = 0.f;
#pragma unroll
  for (int t = 0; t < 3; ++t) { v[t] = elem<DT>::ld(F, (size_t)row * ldf + t * 64 + lane); ss += v[t] * v[t]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float nr = sqrtf(ss);
  const float r = 1.f / (nr + 1e-5f);
#pragma unroll
  for (int t = 0; t < 3; ++t) Fh[(size_t)row * KF + t * 64 + lane] = v[t] * r;
  if (lane == 0) nrm[row] = nr;
}

// dF = dFh/(n+eps) - F * (dFh.F) / (n (n+eps)^2)
template <int DT>
__global__ void l2norm_bwd_kernel(const void* __restrict__ F, int ldf, const float* __restrict__ dFh, const float* __restrict__ nrm,
                                  void* __restrict__ dF, int lddf, long rows) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float v[3], d[3];
  float dot = 0.f;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    v[t] = elem<DT>::ld(F, (size_t)row * ldf + t * 64 + lane);
    d[t] = dFh[(size_t)row * KF + t * 64 + lane];
    dot += v[t] * d[t];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  const float nr = nrm[row];
  const float r = 1.f / (nr + 1e-5f);
  const float k = nr > 0.f ? dot * r * r / nr : 0.f;
#pragma unroll
  for (int t = 0; t < 3; ++t) elem<DT>::st(dF, (size_t)row * lddf + t * 64 + lane, d[t] * r - v[t] * k);
}

// DN[j][c] = d_out[c][j]/D_j (c<21), DN[j][21] = -sum_c d_out[c][j]*out[c][j]/D_j, rest 0
__global__ void pcm_dn_kernel(const float* __restrict__ d_rv, const float* __restrict__ rv, const float* __restrict__ den,
                              float* __restrict__ DN, int hw, long total) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over N*hw
  if (idx >= total) return;
  const long n = idx / hw; const int j = (int)(idx - n * hw);
  const float inv = 1.f / (den[idx] + 1e-5f);
  float dd = 0.f;
  float* o = DN + idx * 32;
  for (int c = 0; c < 21; ++c) {
    const float g = d_rv[((size_t)n * 21 + c) * hw + j];
    dd += g * rv[((size_t)n * 21 + c) * hw + j];
    o[c] = g * inv;
  }
  o[21] = -dd * inv;
  for (int c = 22; c < 32; ++c) o[c] = 0.f;
}

}  // namespace

extern "C" int wseg_pcm_forward(const float* Fh, const float* G, float* cam_rv, float* den, int N, int hw, void* stream) {
  WSEG_CHECK(Fh && G && cam_rv && den && N > 0 && hw > 0, "pcm_forward: bad arguments");
  dim3 grid((hw + 63) / 64, N);
  hipLaunchKernelGGL(pcm_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, Fh, G, (const float*)nullptr, cam_rv, den, hw);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_pcm_backward(const float* Fh, const float* G, const float* d_cam_rv, const float* cam_rv, const float* den,
                                 float* DN /*[N][hw][32] scratch*/, float* dFh /*[N][hw][192], zeroed by caller*/, int N, int hw, void* stream) {
  WSEG_CHECK(Fh && G && d_cam_rv && cam_rv && den && DN && dFh && N > 0 && hw > 0, "pcm_backward: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)N * hw;
  hipLaunchKernelGGL(pcm_dn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_cam_rv, cam_rv, den, DN, hw, total);
  dim3 grid((hw + 63) / 64, N);
  hipLaunchKernelGGL(pcm_kernel<1>, grid, dim3(256), 0, s, Fh, G, (const float*)DN, dFh, (float*)nullptr, hw);
  hipLaunchKernelGGL(pcm_kernel<1>, grid, dim3(256), 0, s, Fh, (const float*)DN, G, dFh, (float*)nullptr, hw);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_l2norm_forward(const void* F, int ldf, float* Fh, float* nrm, long rows, int dtype, void* stream) {
  WSEG_CHECK(F && Fh && nrm && rows > 0 && ldf >= KF, "l2norm_forward: bad arguments");
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == WSEG_BF16) hipLaunchKernelGGL(l2norm_fwd_kernel<WSEG_BF16>, grid, dim3(256), 0, (hipStream_t)stream, F, ldf, Fh, nrm, rows);
  else hipLaunchKernelGGL(l2norm_fwd_kernel<WSEG_F32>, grid, dim3(256), 0, (hipStream_t)stream, F, ldf, Fh, nrm, rows);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_l2norm_backward(const void* F, int ldf, const float* dFh, const float* nrm, void* dF, int lddf, long rows, int dtype, void* stream) {
  WSEG_CHECK(F && dFh && nrm && dF && rows > 0, "l2norm_backward: bad arguments");
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == WSEG_BF16) hipLaunchKernelGGL(l2norm_bwd_kernel<WSEG_BF16>, grid, dim3(256), 0, (hipStream_t)stream, F, ldf, dFh, nrm, dF, lddf, rows);
  else hipLaunchKernelGGL(l2norm_bwd_kernel<WSEG_F32>, grid, dim3(256), 0, (hipStream_t)stream, F, ldf, dFh, nrm, dF, lddf, rows);
  WSEG_LAUNCH_CHECK();
  return 0;
}

extern "C" int wseg_to_bf16(const float* in, void* out, long total, void* stream) {
  WSEG_CHECK(in && out && total > 0, "to_bf16: bad arguments");
  hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)((total / 4 + 256) / 256)), dim3(256), 0, (hipStream_t)stream, in, (bf16_t*)out, total);
  WSEG_LAUNCH_CHECK();
  return 0;
}

// bf16-MFMA PCM (throughput mode): Fb / Gb are bf16 copies of Fh [N][hw][192] and G [N][hw][32]
extern "C" int wseg_pcm_forward_bf16(const void* Fb, const void* Gb, float* cam_rv, float* den, int N, int hw, void* stream) {
  WSEG_CHECK(Fb && Gb && cam_rv && den && N > 0 && hw > 0, "pcm_forward_bf16: bad arguments");
  dim3 grid((hw + 63) / 64, N);
  hipLaunchKernelGGL(pcm_bf16_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)Fb, (const bf16_t*)Gb, (const bf16_t*)nullptr,
                     (const bf16_t*)nullptr, (const bf16_t*)nullptr, cam_rv, den, hw);
  WSEG_LAUNCH_CHECK();
  return 0;
}
extern "C" int wseg_pcm_backward_bf16(const void* Fb, const void* Gb, const void* Gl, const float* d_cam_rv, const float* cam_rv, const float* den,
                                      float* DN, void* DNb, void* DNl, float* dFh, int N, int hw, void* stream) {
  WSEG_CHECK(Fb && Gb && Gl && d_cam_rv && cam_rv && den && DN && DNb && DNl && dFh && N > 0 && hw > 0, "pcm_backward_bf16: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)N * hw;
  hipLaunchKernelGGL(pcm_dn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_cam_rv, cam_rv, den, DN, hw, total);
  hipLaunchKernelGGL(split_hi_lo_kernel, dim3((unsigned)((total * 32 + 255) / 256)), dim3(256), 0, s, (const float*)DN, (bf16_t*)DNb, (bf16_t*)DNl, total * 32);
  dim3 grid((hw + 63) / 64, N);
  hipLaunchKernelGGL(pcm_bf16_kernel<1>, grid, dim3(256), 0, s, (const bf16_t*)Fb, (const bf16_t*)Gb, (const bf16_t*)DNb, (const bf16_t*)Gl, (const bf16_t*)DNl,
                     dFh, (float*)nullptr, hw);
  hipLaunchKernelGGL(pcm_bf16_kernel<1>, grid, dim3(256), 0, s, (const bf16_t*)Fb, (const bf16_t*)DNb, (const bf16_t*)Gb, (const bf16_t*)DNl, (const bf16_t*)Gl,
                     dFh, (float*)nullptr, hw);
  WSEG_LAUNCH_CHECK();
  return 0;
}
