// conv_wgrad.hip — convolution weight gradient as a pixel-reduction GEMM on MFMA.
//
//   dW[oc][tap][ic] += sum_m dY[m][oc] * X[pix(m,tap)][ic]
//
// The reduction index is the PIXEL, which is the row (slow) index of both NHWC operands, so both
// LDS tiles are staged exactly as they lie in HBM ([pixels][128 channels], LDS-DMA, 16 B/lane,
// zero page for padded taps) and are read TRANSPOSED:
//   bf16: ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group) feeding
//         v_mfma_f32_16x16x32_bf16;  32-B blocks of a pixel row are XOR-swizzled by
//         s(pix) = (pix&3) | ((pix>>3)&1)<<2  so a 32-lane half touches 8 distinct blocks.
//   f32:  one ds_read_b32 per operand element feeding v_mfma_f32_16x16x4_f32; 64-B blocks are
//         XOR-swizzled by (pix&7).
// One workgroup = 128(oc) x 128(ic) of one tap over a slice of the pixels (split-K over pixel
// ranges); partial tiles are added to the f32 dW with 256-B-contiguous float atomics.
#include <algorithm>
#include "common.h"

namespace {

constexpr int BO = 128, BI = 128;
constexpr int TILE_BYTES = 16384;
constexpr int EPI_LD = BI + 4;
constexpr int SMEM_BYTES = BO * EPI_LD * 4;

struct Args {
  wseg_wgrad_desc d;
  int M, taps, nto, nti, ntiles;
  int pix_per_split;
};

template <int DT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const Args a) {
  constexpr int ES = elem<DT>::size;
  constexpr int CH = 16 / ES;                 // channels per 16-B chunk
  constexpr int ROWB = 128 * ES;              // bytes per pixel row in LDS (128 channels)
  constexpr int PK = TILE_BYTES / ROWB;       // pixels per K-step: 64 (bf16) / 32 (f32)
  constexpr int CPR = ROWB / 16;              // chunks per row: 16 / 32
  constexpr int RPI = 64 / CPR;               // rows per wave DMA instruction: 4 / 2
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  const wseg_wgrad_desc& d = a.d;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x;
  const int tap = tile % a.taps;
  const int t2 = tile / a.taps;
  const int ti = t2 % a.nti, to = t2 / a.nti;
  const int oc0 = to * BO, ic0 = ti * BI;
  const int ky = tap / d.KW, kx = tap - ky * d.KW;
  const int m_begin = blockIdx.y * a.pix_per_split;
  const int m_end = min(a.M, m_begin + a.pix_per_split);
  if (m_begin >= m_end) return;

  const char* zero = reinterpret_cast<const char*>(g_wseg_zero_page);
  const char* X = reinterpret_cast<const char*>(d.x);
  const char* DY = reinterpret_cast<const char*>(d.dy);

  // staging: thread -> 4 rows per operand; piece i covers rows (wid*4+i)*RPI + lane/CPR
  const int prow = lane / CPR, pch = lane % CPR;
  int r_pix[4];                               // pixel row inside the tile
  int lch_o[4], lch_i[4];                     // logical chunk (channel group) this lane fetches
  bool ok_o[4], ok_i[4];
  int cn[4], cy[4], cx[4];                    // (n, oy, ox) of the row's pixel, advanced per step
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wid * 4 + i) * RPI + prow;
    r_pix[i] = r;
    int lc;
    if constexpr (DT == WSEG_BF16) {
      const int s = (r & 3) | (((r >> 3) & 1) << 2);
      lc = (((pch >> 1) ^ s) << 1) | (pch & 1);
    } else {
      lc = (((pch >> 2) ^ (r & 7)) << 2) | (pch & 3);
    }
    lch_o[i] = lc; lch_i[i] = lc;
    ok_o[i] = (oc0 + lc * CH) < d.OC;
    ok_i[i] = (ic0 + lc * CH) < d.IC;
    const int m = m_begin + r;
    const int hw = d.OH * d.OW;
    const int n = m / hw, rem = m - n * hw;
    cn[i] = n; cy[i] = rem / d.OW; cx[i] = rem - cy[i] * d.OW;
  }

  auto stage = [&](int buf, int mstep) {
    char* lo = smem + buf * 2 * TILE_BYTES + wid * 4 * RPI * ROWB;
    char* li = lo + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mstep + r_pix[i];
      const bool inr = m < m_end;
      const char* po = zero + (lane & 15) * 16;
      const char* pi = po;
      if (inr && ok_o[i]) po = DY + ((size_t)m * d.ld_dy + oc0 + lch_o[i] * CH) * ES;
      if (inr && ok_i[i]) {
        const int iy = cy[i] * d.stride + ky * d.dil - d.pad;
        const int ix = cx[i] * d.stride + kx * d.dil - d.pad;
        if (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW)
          pi = X + (((size_t)cn[i] * d.IH + iy) * d.IW + ix) * (size_t)d.ld_x * ES + (size_t)(ic0 + lch_i[i] * CH) * ES;
      }
      glds16(po, lo + i * RPI * ROWB);
      glds16(pi, li + i * RPI * ROWB);
      // advance this row's pixel coordinates by PK
      cx[i] += PK;
      while (cx[i] >= d.OW) { cx[i] -= d.OW; if (++cy[i] == d.OH) { cy[i] = 0; ++cn[i]; } }
    }
  };

  const int wr = wid >> 1, wc = wid & 1;
  const int fcol = lane & 15, fk = lane >> 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (m_end - m_begin + PK - 1) / PK;
  stage(0, m_begin);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1, m_begin + (kt + 1) * PK);
    const char* bo = smem + cur * 2 * TILE_BYTES;
    const char* bi = bo + TILE_BYTES;
    if constexpr (DT == WSEG_BF16) {
      // tr read: lane 4q+p of a 16-lane group addresses row q, 8 bytes at column 4p
      const int q = (lane & 15) >> 2, p = lane & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[4], bf[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = ks * 32 + fk * 8 + h * 4 + q;
          const int s = (row & 3) | (((row >> 3) & 1) << 2);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int blk_o = wr * 4 + i, blk_i = wc * 4 + i;
            const bf16x4 vo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (bf16x4 __attribute__((address_space(3)))*)(bo + row * ROWB + ((blk_o ^ s) << 5) + p * 8));
            const bf16x4 vi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (bf16x4 __attribute__((address_space(3)))*)(bi + row * ROWB + ((blk_i ^ s) << 5) + p * 8));
#pragma unroll
            for (int e = 0; e < 4; ++e) { af[i][h * 4 + e] = vo[e]; bf[i][h * 4 + e] = vi[e]; }
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < PK / 4; ++kk) {
        const int row = kk * 4 + fk;
        const int s = row & 7;
        float af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int blk_o = wr * 4 + i, blk_i = wc * 4 + i;     // 64-B (16-float) blocks
          af[i] = *reinterpret_cast<const float*>(bo + row * ROWB + ((blk_o ^ s) << 6) + fcol * 4);
          bf[i] = *reinterpret_cast<const float*>(bi + row * ROWB + ((blk_i ^ s) << 6) + fcol * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: acc -> LDS image [oc][ic] -> 256-B contiguous float atomics into dW
  float* img = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wr * 64 + i * 16 + fk * 4;
      const int col = wc * 64 + j * 16 + fcol;
#pragma unroll
      for (int e = 0; e < 4; ++e) img[(row + e) * EPI_LD + col] = acc[i][j][e];
    }
  __syncthreads();
  const size_t row_stride = (size_t)a.taps * d.IC_dw;
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
    const int idx = it * 256 + tid;
    const int row = idx >> 7, col = idx & 127;
    const int oc = oc0 + row, ic = ic0 + col;
    if (oc < d.OC_dw && ic < d.IC_dw)
      atomicAdd(&d.dw[(size_t)oc * row_stride + (size_t)tap * d.IC_dw + ic], img[row * EPI_LD + col]);
  }
}

}  // namespace

extern "C" int wseg_conv_wgrad(const wseg_wgrad_desc* d, void* stream) {
  WSEG_CHECK(d && d->x && d->dy && d->dw, "conv_wgrad: null pointer");
  WSEG_CHECK(d->dtype == WSEG_F32 || d->dtype == WSEG_BF16, "conv_wgrad: bad dtype");
  WSEG_CHECK(d->IC % 8 == 0 && d->OC % 8 == 0 && d->ld_x % 8 == 0 && d->ld_dy % 8 == 0,
             "conv_wgrad: IC/OC/ld must be multiples of 8 (IC=%d OC=%d ld_x=%d ld_dy=%d)", d->IC, d->OC, d->ld_x, d->ld_dy);
  WSEG_CHECK(d->ld_x >= d->IC && d->ld_dy >= d->OC, "conv_wgrad: leading dims too small");
  WSEG_CHECK(d->IC_dw > 0 && d->IC_dw <= d->IC && d->OC_dw > 0 && d->OC_dw <= d->OC, "conv_wgrad: bad dw extents");
  WSEG_CHECK(d->N > 0 && d->OH > 0 && d->OW > 0 && d->IH > 0 && d->IW > 0 && d->stride >= 1 && d->dil >= 1, "conv_wgrad: bad shape");
  const long M = (long)d->N * d->OH * d->OW;
  WSEG_CHECK(M < (1L << 31), "conv_wgrad: too many pixels");
  Args a;
  a.d = *d;
  a.M = (int)M;
  a.taps = d->KH * d->KW;
  a.nto = (d->OC + BO - 1) / BO;
  a.nti = (d->IC + BI - 1) / BI;
  a.ntiles = a.nto * a.nti * a.taps;
  const int pk = d->dtype == WSEG_BF16 ? 64 : 32;
  int split = d->split_k;
  if (split <= 0) {                            // heuristic: >= 2 workgroups per CU, >= 8 K-steps each
    split = (512 + a.ntiles - 1) / a.ntiles;
    const int max_split = (int)std::max(1L, M / (pk * 8));
    split = std::max(1, std::min(split, max_split));
  }
  long pps = (M + split - 1) / split;
  pps = (pps + pk - 1) / pk * pk;
  a.pix_per_split = (int)pps;
  split = (int)((M + pps - 1) / pps);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(a.ntiles, split);
  if (d->dtype == WSEG_BF16)
    hipLaunchKernelGGL(conv_wgrad_kernel<WSEG_BF16>, grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<WSEG_F32>, grid, dim3(256), 0, s, a);
  WSEG_LAUNCH_CHECK();
  return 0;
}
