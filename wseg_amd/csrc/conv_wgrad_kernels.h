// conv_wgrad.hip — convolution weight gradient as a pixel-reduction GEMM on MFMA.
//
//   dW[oc][tap][ic] += sum_m dY[m][oc] * X[pix(m,tap)][ic]
//
// The reduction index is the PIXEL, which is the row (slow) index of both NHWC operands, so both
// LDS tiles are staged exactly as they lie in HBM ([pixels][BO|BI channels], LDS-DMA, 16 B/lane,
// zero page for padded taps) and are read TRANSPOSED:
//   bf16: ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group) feeding
//         v_mfma_f32_16x16x32_bf16;  32-B blocks of a pixel row are XOR-swizzled (low 3 bits) by
//         s(pix) = (pix&3) | ((pix>>3)&1)<<2  so a 32-lane half touches 8 distinct blocks.
//   f32:  one ds_read_b32 per operand element feeding v_mfma_f32_16x16x4_f32; 64-B blocks are
//         XOR-swizzled by (pix&7).
// One workgroup = BO(oc) x BI(ic) of one tap over a slice of the pixels (split-K over pixel ranges).
// Two geometries: 128x128 / 4 waves (any dtype, small layers) and 256x256 / 8 waves (bf16): each
// K-step moves (BO+BI)*128 B for BO*BI*64 MACs — 64 vs 128 FLOP per byte of L2->LDS fill, which is
// what bounds this kernel (measured 3.7 GB of fills at 6.4 TB/s for one 512->512 3x3 layer at 128^2).
// Partial tiles are added to the f32 dW with 256-B-contiguous float atomics.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#pragma once
#include "common.h"

#ifdef WSEG_PROBES   // WSEG_WGRAD_DIAG timing diagnostics (results wrong by design) exist in probe builds only
#define WG_DIAG(a) ((a).diag)
#else
#define WG_DIAG(a) 0
#endif

namespace wseg_wg {
namespace {

// transposed LDS read with a compile-time immediate offset (ds_read_b64_tr_b16: 4 pixels x 16 channels per 16-lane group)
template <int OFF>
__device__ __forceinline__ void tr_read(bf16x4& dst, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

struct Args {
  wseg_wgrad_desc d;
  int M, taps, nto, nti, ntiles;
  int pix_per_split;
  int nsplit, nwg;
  int simple_adv, q64_1, r64_1, q64_2, r64_2;   // pipe kernel: 64 pixels = q*OW + r per row segment (simple_adv: one image wrap at most)
  int wave_epi;      // pipe kernel: 1 = wave-local atomic epilogue (WSEG_WGRAD_EPI)
  int diag;          // 0 = normal; timing diagnostics (WSEG_WGRAD_DIAG): 1 = no epilogue stores, 2 = plain stores, 4 / 5 = X / X and dY from the zero page
};

template <int DT, int BO, int BI, int WR, int WC>
__global__ __launch_bounds__(WR * WC * 64, 2) void conv_wgrad_kernel(const Args a) {
  constexpr int NT = WR * WC * 64;            // threads
  constexpr int NW = WR * WC;                 // waves
  constexpr int ES = elem<DT>::size;
  constexpr int CH = 16 / ES;                 // channels per 16-B chunk
  constexpr int PK = DT == WSEG_BF16 ? 64 : 32;   // pixels per K-step
  constexpr int ROWB_O = BO * ES, ROWB_I = BI * ES;
  constexpr int TILE_O = PK * ROWB_O, TILE_I = PK * ROWB_I;
  constexpr int STAGE = TILE_O + TILE_I;
  constexpr int MI = BO / WR / 16, NJ = BI / WC / 16;   // 16x16 MFMA tiles per wave
  constexpr int EPI_ROWS = 64;
  constexpr int EPI_LD = BI + 4;
  constexpr int SMEM = (2 * STAGE > EPI_ROWS * EPI_LD * 4) ? 2 * STAGE : EPI_ROWS * EPI_LD * 4;
  constexpr int PIECES_O = TILE_O / 1024, PIECES_I = TILE_I / 1024;      // 1-KiB LDS-DMA pieces
  static_assert(PIECES_O % NW == 0 && PIECES_I % NW == 0, "pieces must divide over the waves");
  constexpr int PO = PIECES_O / NW, PI = PIECES_I / NW;
  __shared__ __attribute__((aligned(16))) char smem[SMEM];
  const wseg_wgrad_desc& d = a.d;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (tile, pixel split).  All tiles of one split read the same dY / X rows: keep them on ONE XCD
  // (blocks b and b+8 share an XCD) so the rows are fetched into that XCD's L2 once and every other tile's
  // LDS-DMA hits L2 (~70 GB/s per CU) instead of MALL/HBM (~25-33 GB/s per CU).
  int tile, split;
  if ((a.nsplit & 7) == 0) {
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3, q = a.nsplit >> 3;
    tile = j % a.ntiles;
    split = xcd * q + j / a.ntiles;
  } else {
    const int l = xcd_remap(blockIdx.x, a.nwg);           // contiguous logical ids per XCD: the taps of an (oc,ic) tile pair
    tile = l % a.ntiles;
    split = l / a.ntiles;
  }
  const int tap = tile % a.taps;
  const int t2 = tile / a.taps;
  const int ti = t2 % a.nti, to = t2 / a.nti;
  const int oc0 = to * BO, ic0 = ti * BI;
  const int ky = tap / d.KW, kx = tap - ky * d.KW;
  const int m_begin = split * a.pix_per_split;
  const int m_end = min(a.M, m_begin + a.pix_per_split);
  if (m_begin >= m_end) return;

  const char* zero = reinterpret_cast<const char*>(g_wseg_zero_page);
  const char* X = reinterpret_cast<const char*>(d.x);
  const char* DY = reinterpret_cast<const char*>(d.dy);

  // logical 16-B chunk fetched into physical chunk `pch` of pixel row r (rows are ROWB bytes)
  auto logical_chunk = [](int r, int pch) {
    if constexpr (DT == WSEG_BF16) {
      const int s = (r & 3) | (((r >> 3) & 1) << 2);
      const int blk = pch >> 1;                                   // 32-B block
      return ((((blk & ~7) | ((blk ^ s) & 7))) << 1) | (pch & 1);
    } else {
      const int blk = pch >> 2;                                   // 64-B block
      return (((blk & ~7) | ((blk ^ (r & 7)) & 7)) << 2) | (pch & 3);
    }
  };

  // pixel coordinates of the rows this thread stages for X (advanced by PK per step)
  constexpr int CPR_I = ROWB_I / 16, CPR_O = ROWB_O / 16;          // chunks per row
  int xr[PI], xlc[PI];
  bool xok[PI];
#pragma unroll
  for (int i = 0; i < PI; ++i) {
    const int ci = (wid * PI + i) * 64 + lane;                    // physical chunk index in the tile
    xr[i] = ci / CPR_I;
    xlc[i] = logical_chunk(xr[i], ci % CPR_I);
    xok[i] = (ic0 + xlc[i] * CH) < d.IC;
  }
  int yr[PO], ylc[PO];
  bool yok[PO];
#pragma unroll
  for (int i = 0; i < PO; ++i) {
    const int ci = (wid * PO + i) * 64 + lane;
    yr[i] = ci / CPR_O;
    ylc[i] = logical_chunk(yr[i], ci % CPR_O);
    yok[i] = (oc0 + ylc[i] * CH) < d.OC;
  }

  // pixel coordinates of the X rows advance INCREMENTALLY by PK rows per stage (three integer divisions per piece per K-step
  // made the small layers VALU-bound); a row that crosses into the second segment is decoded afresh
  const int M1 = d.N * d.OH * d.OW;
  int xm[PI], xoy[PI], xox[PI], xbase[PI];
#pragma unroll
  for (int i = 0; i < PI; ++i) {
    xm[i] = m_begin + xr[i];
    const wseg_rowgeo rg = wseg_decode_row(d, min(xm[i], a.M - 1));
    xoy[i] = rg.oy; xox[i] = rg.ox; xbase[i] = (int)rg.in_base;
  }
  auto stage = [&](int buf, int mstep) {             // called with mstep = m_begin, m_begin + PK, ... (consecutive)
    char* lo = smem + buf * STAGE;
    char* li = lo + TILE_O;
#pragma unroll
    for (int i = 0; i < PO; ++i) {
      const int m = mstep + yr[i];
      const char* po = zero + (lane & 15) * 16;
      if (m < m_end && yok[i]) po = DY + ((size_t)m * d.ld_dy + oc0 + ylc[i] * CH) * ES;
      glds16(po, lo + (wid * PO + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < PI; ++i) {
      const bool s2 = d.OH2 != 0 && xm[i] >= M1;
      const int OHs = s2 ? d.OH2 : d.OH, OWs = s2 ? d.OW2 : d.OW, IHs = s2 ? d.IH2 : d.IH, IWs = s2 ? d.IW2 : d.IW;
      const char* pi = zero + (lane & 15) * 16;
      if (xm[i] < m_end && xok[i]) {
        const int iy = xoy[i] * d.stride + ky * d.dil - d.pad;
        const int ix = xox[i] * d.stride + kx * d.dil - d.pad;
        if (iy >= 0 && iy < IHs && ix >= 0 && ix < IWs)
          pi = X + ((size_t)(xbase[i] + iy * IWs + ix) * d.ld_x + (size_t)(ic0 + xlc[i] * CH)) * ES;
      }
      glds16(pi, li + (wid * PI + i) * 1024);
      const int mn = xm[i] + PK;
      if (d.OH2 != 0 && xm[i] < M1 && mn >= M1) {
        const wseg_rowgeo rg = wseg_decode_row(d, min(mn, a.M - 1));
        xoy[i] = rg.oy; xox[i] = rg.ox; xbase[i] = (int)rg.in_base;
      } else {
        int ox = xox[i] + PK, oy = xoy[i], bs = xbase[i];
        while (ox >= OWs) { ox -= OWs; ++oy; }
        while (oy >= OHs) { oy -= OHs; bs += IHs * IWs; }
        xox[i] = ox; xoy[i] = oy; xbase[i] = bs;
      }
      xm[i] = mn;
    }
  };

  const int wr = wid / WC, wc = wid % WC;
  const int fcol = lane & 15, fk = lane >> 4;

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (m_end - m_begin + PK - 1) / PK;
  stage(0, m_begin);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) stage(cur ^ 1, m_begin + (kt + 1) * PK);
    const char* bo = smem + cur * STAGE;
    const char* bi = bo + TILE_O;
    if constexpr (DT == WSEG_BF16) {
      // tr read: lane 4q+p of a 16-lane group addresses row q, 8 bytes at column 4p of a 16-channel block
      const int q = (lane & 15) >> 2, p = lane & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[MI], bf[NJ];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int row = ks * 32 + fk * 8 + h * 4 + q;
          const int s = (row & 3) | (((row >> 3) & 1) << 2);
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int blk = wr * MI + i;
            const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (bf16x4 __attribute__((address_space(3)))*)(bo + row * ROWB_O + (((blk & ~7) | ((blk ^ s) & 7)) << 5) + p * 8));
#pragma unroll
            for (int e = 0; e < 4; ++e) af[i][h * 4 + e] = v[e];
          }
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int blk = wc * NJ + j;
            const bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (bf16x4 __attribute__((address_space(3)))*)(bi + row * ROWB_I + (((blk & ~7) | ((blk ^ s) & 7)) << 5) + p * 8));
#pragma unroll
            for (int e = 0; e < 4; ++e) bf[j][h * 4 + e] = v[e];
          }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else if constexpr (DT == WSEG_F32X3) {
      // split-bf16 products on f32 operands: lane (fcol, fk) gathers the 8 pixels 4e + fk (e = 0..7) of its channel of both
      // operands (the scalar reads of the f32 path, same swizzle), splits them into hi + lo and issues lo.hi + hi.lo + hi.hi
      float a8[MI][8], b8[NJ][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = e * 4 + fk;
        const int s = row & 7;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int blk = wr * MI + i;
          a8[i][e] = *reinterpret_cast<const float*>(bo + row * ROWB_O + (((blk & ~7) | ((blk ^ s) & 7)) << 6) + fcol * 4);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int blk = wc * NJ + j;
          b8[j][e] = *reinterpret_cast<const float*>(bi + row * ROWB_I + (((blk & ~7) | ((blk ^ s) & 7)) << 6) + fcol * 4);
        }
      }
      bf16x8 ah[MI], al[MI], bh[NJ], bl[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) split_bf16x8(a8[i], ah[i], al[i]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) split_bf16x8(b8[j], bh[j], bl[j]);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll
      for (int kk = 0; kk < PK / 4; ++kk) {
        const int row = kk * 4 + fk;
        const int s = row & 7;
        float af[MI], bf[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int blk = wr * MI + i;                              // 64-B (16-float) block
          af[i] = *reinterpret_cast<const float*>(bo + row * ROWB_O + (((blk & ~7) | ((blk ^ s) & 7)) << 6) + fcol * 4);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int blk = wc * NJ + j;
          bf[j] = *reinterpret_cast<const float*>(bi + row * ROWB_I + (((blk & ~7) | ((blk ^ s) & 7)) << 6) + fcol * 4);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: 64 oc rows at a time: acc -> LDS image [64][BI] -> 256-B contiguous float atomics into dW
  float* img = reinterpret_cast<float*>(smem);
  const size_t row_stride = (size_t)a.taps * d.IC_dw;
  constexpr int PASSES = BO / EPI_ROWS;
  constexpr int TILES_PER_PASS = EPI_ROWS / 16;                     // row tiles (of 16) per pass
#pragma unroll 1
  for (int ps = 0; ps < PASSES; ++ps) {
    // row tile t (global, 0..BO/16) belongs to wave row wr = t / MI, local i = t % MI
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int t = wr * MI + i;
      if (t / TILES_PER_PASS == ps) {
        const int row = (t % TILES_PER_PASS) * 16 + fk * 4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = (wc * NJ + j) * 16 + fcol;
#pragma unroll
          for (int e = 0; e < 4; ++e) img[(row + e) * EPI_LD + col] = acc[i][j][e];
        }
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int idx = tid; idx < EPI_ROWS * BI; idx += NT) {
      const int row = idx / BI, col = idx - row * BI;
      const int oc = oc0 + ps * EPI_ROWS + row, ic = ic0 + col;
      if (oc < d.OC_dw && ic < d.IC_dw) {
        int icw = ic + d.dw_rot;                                     // (column rotation of dw: see wseg_wgrad_desc.dw_rot)
        icw = icw >= d.IC_dw ? icw - d.IC_dw : icw;
        atomicAdd(&d.dw[(size_t)oc * row_stride + (size_t)tap * d.IC_dw + icw], img[row * EPI_LD + col]);
      }
    }
    __syncthreads();
  }
}

// ---- 256x256 bf16 phase-pipelined variant (the schedule validated in csrc/gemm256_probe.hip: 1.1-1.2 PF on
// plain GEMM).  K-tile = 64 pixels; LDS = 2 K-tiles x 4 half-tile slots {dY ch 0-127, dY ch 128-255, X ch 0-127,
// X ch 128-255}, each [64 pixels][128 ch] = 16 KiB.  A K-tile is 4 phases of 16 MFMAs (one 64(oc) x 32(ic)
// quadrant of the wave's 128 x 64 tile); each phase refills ONE slot every wave has finished reading:
//     p1(u): dY0(u+1)   p2(u): dY1(u+1)   p3(u): X0(u+2)   p4(u): X1(u+2)
// and the only DMA wait is a counted s_waitcnt vmcnt(4) in p4.  One raw s_barrier per phase.  The transposed
// fragment reads go through inline asm (hipcc would put vmcnt(0) in front of ds_read_tr builtins while
// LDS-DMA is in flight), with an explicit lgkmcnt(0) + sched_barrier before the MFMAs.
template <int UNIT>
                                 // UNIT 1: stride 1 and IH==OH, IW==OW in both segments (the X source row is m + const)
__device__ __forceinline__ void conv_wgrad_pipe_tile(const Args& a, char* smem, const int bid) {   // (smem: the workgroup's 128 KiB LDS buffer)
  constexpr int BO = 256, BI = 256, NT = 512;
  constexpr int PK = 64, HALF = 16384, TILE = 4 * HALF, ROWB = 256;
  constexpr int EPI_ROWS = 64, EPI_LD = BI + 4;
  const wseg_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tile, split;
  if ((a.nsplit & 7) == 0) {
    const int b = bid, xcd = b & 7, j = b >> 3, q8 = a.nsplit >> 3;
    tile = j % a.ntiles; split = xcd * q8 + j / a.ntiles;
  } else {
    const int l = xcd_remap(bid, a.nwg);
    tile = l % a.ntiles; split = l / a.ntiles;
  }
  const int tap = tile % a.taps;
  const int t2 = tile / a.taps;
  const int ti = t2 % a.nti, to = t2 / a.nti;
  const int oc0 = to * BO, ic0 = ti * BI;
  const int ky = tap / d.KW, kx = tap - ky * d.KW;
  const int m_begin = split * a.pix_per_split;
  const int m_end = min(a.M, m_begin + a.pix_per_split);
  if (m_begin >= m_end) return;
  const char* zero = reinterpret_cast<const char*>(g_wseg_zero_page);
  const char* X = reinterpret_cast<const char*>(d.x);
  const char* DY = reinterpret_cast<const char*>(d.dy);

  // LDS slot (buffer buf, half-tile which = dY0, dY1, X0, X1).  The 2-phase schedules interleave the two buffers at slot
  // granularity so that the buffer offset fits the ds_read immediate (all transposed reads then need no address VALU).
  auto slot_off = [](int buf, int which) { return (which * 2 + buf) * HALF; };
  // staging: thread -> rows r0 = tid>>4 and r0+32 of every half-tile, physical 16-B chunk tid&15.
  // All per-K-tile address work is INCREMENTAL (this loop is VALU-sensitive: 64 MFMAs per wave per K-tile leave
  // ~250 issue slots): dY pointers advance by a constant; the X pixel coordinates (n, oy, ox) advance by 64 rows
  // with one add / compare / subtract each instead of three integer divisions per row.
  const int pch = tid & 15;
  const char* zsrc = zero + (lane & 15) * 16;
  int lc[2];
  const char* ybase[2];                              // dY source of the NEXT Y tile (row rr[k], chunk lc[k], half 0)
  int my[2];                                         // its pixel row
  int xm[2], xoy[2], xox[2], xbase[2];               // NEXT X tile: pixel row, output coordinates, first input row of the image
  const int M1 = d.N * d.OH * d.OW;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int rr = (tid >> 4) + 32 * k;
    const int sw = (rr & 3) | (((rr >> 3) & 1) << 2);
    lc[k] = (((pch >> 1) ^ sw) << 1) | (pch & 1);
    my[k] = m_begin + rr;
    ybase[k] = DY + ((size_t)my[k] * d.ld_dy + oc0 + lc[k] * 8) * 2;
    xm[k] = m_begin + rr;
    const wseg_rowgeo rg = wseg_decode_row(d, min(xm[k], a.M - 1));
    xoy[k] = rg.oy; xox[k] = rg.ox; xbase[k] = (int)rg.in_base;
  }
  const size_t ystep = (size_t)PK * d.ld_dy * 2;
  const bool yok[2][2] = {{oc0 + lc[0] * 8 < d.OC, oc0 + 128 + lc[0] * 8 < d.OC}, {oc0 + lc[1] * 8 < d.OC, oc0 + 128 + lc[1] * 8 < d.OC}};
  const bool xok[2][2] = {{ic0 + lc[0] * 8 < d.IC, ic0 + 128 + lc[0] * 8 < d.IC}, {ic0 + lc[1] * 8 < d.IC, ic0 + 128 + lc[1] * 8 < d.IC}};
  const char* xrow[2];                               // source pixel row (chunk lc[k], half 0) of the X tile being issued, or nullptr
  // UNIT geometry: input pixel of (output row m, tap) = m + dy*W + dx inside its segment, so the source pointer is a running
  // pointer (+64 rows per K-tile, + a constant when crossing into the second segment) and only the VALIDITY needs (oy, ox).
  const int dy = ky * d.dil - d.pad, dx = kx * d.dil - d.pad;
  const char* xptr[2];
  if constexpr (UNIT) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool s2 = d.OH2 != 0 && m_begin >= M1;   // the workgroup's FIRST segment (uniform); the switch adds xcross
      const long shift = (long)dy * (s2 ? d.IW2 : d.IW) + dx;
      xptr[k] = X + (((long)xm[k] + shift) * d.ld_x + ic0 + lc[k] * 8) * 2;
    }
  }
  const size_t xstep = (size_t)PK * d.ld_x * 2;
  const long xcross = (long)dy * (d.IW2 - d.IW) * d.ld_x * 2;      // pointer correction when a row enters segment 2
  // UNIT: the K-tiles never straddle the boundary between the two row segments (views): the pixel range is walked as
  // [m_begin, min(m_end, M1)) then [max(m_begin, M1), m_end), each in 64-row tiles (at most one tile more than a straight walk;
  // rows past a sub-range's end read the zero page on both operands).  All geometry of a tile is then wave-uniform (scalars):
  // per row there remain two bounds tests, the (oy, ox) advance and the source select — this loop is VALU-sensitive.
  const int seg_end = d.OH2 != 0 ? M1 : a.M;
  const int a_end = min(m_end, seg_end), b_begin = max(m_begin, seg_end);
  const int nA = m_begin < a_end ? (a_end - m_begin + PK - 1) / PK : 0;
  const int nB = m_end > b_begin ? (m_end - b_begin + PK - 1) / PK : 0;
  int ty_idx = 0, tx_idx = 0;                        // index of the NEXT dY / X tile to issue
  int y_m = m_begin, x_m = m_begin;                  // its first pixel row
  int y_lim = nA > 0 ? a_end : m_end, x_lim = y_lim; // end of its sub-range
  bool x_s2 = nA == 0 && d.OH2 != 0;                 // the X tile's segment
  int rry[2][2], rrx[2][2];                          // row inside the tile per (k, h); huge where the channel half does not exist
  bool xin[2] = {false, false};                      // the X tile being issued: tap lands inside the image
  if constexpr (UNIT) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        rry[k][h] = yok[k][h] ? (tid >> 4) + 32 * k : 0x40000000;
        rrx[k][h] = xok[k][h] ? (tid >> 4) + 32 * k : 0x40000000;
      }
  }
  auto x_prepare = [&]() {                           // call once per X tile, before its two half issues; advances to the next tile
    if constexpr (UNIT) {
      if (tx_idx == nA && nA > 0 && nB > 0) {        // (wave-uniform, once) enter the second segment: jump the pointers, decode afresh
        const long jump = (long)(b_begin - x_m) * d.ld_x * 2 + xcross;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          xptr[k] += jump;
          const wseg_rowgeo rg = wseg_decode_row(d, min(b_begin + (tid >> 4) + 32 * k, a.M - 1));
          xoy[k] = rg.oy; xox[k] = rg.ox;
        }
        x_m = b_begin; x_lim = m_end; x_s2 = true;
      }
      const int OHs = x_s2 ? d.OH2 : d.OH, OWs = x_s2 ? d.OW2 : d.OW;
      const int r64 = x_s2 ? a.r64_2 : a.r64_1, q64 = x_s2 ? a.q64_2 : a.q64_1;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        xin[k] = (unsigned)(xoy[k] + dy) < (unsigned)OHs && (unsigned)(xox[k] + dx) < (unsigned)OWs;
        int ox = xox[k] + r64, oy = xoy[k] + q64;    // advance 64 rows (host guarantees one row wrap at most: simple_adv)
        const bool cx = ox >= OWs;
        ox = cx ? ox - OWs : ox; oy = cx ? oy + 1 : oy;
        oy = oy >= OHs ? oy - OHs : oy;
        xox[k] = ox; xoy[k] = oy;
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const bool s2 = d.OH2 != 0 && xm[k] >= M1;
      const int OHs = s2 ? d.OH2 : d.OH, OWs = s2 ? d.OW2 : d.OW, IHs = s2 ? d.IH2 : d.IH, IWs = s2 ? d.IW2 : d.IW;
      const int mn = xm[k] + PK;
      xrow[k] = nullptr;
      if (xm[k] < m_end) {
        const int iy = xoy[k] * d.stride + ky * d.dil - d.pad;
        const int ix = xox[k] * d.stride + kx * d.dil - d.pad;
        if (iy >= 0 && iy < IHs && ix >= 0 && ix < IWs)
          xrow[k] = X + ((size_t)(xbase[k] + iy * IWs + ix) * d.ld_x + ic0 + lc[k] * 8) * 2;
      }
      // advance 64 pixel rows
      if (d.OH2 != 0 && xm[k] < M1 && mn >= M1) {     // crosses into the second segment: decode afresh (rare)
        const wseg_rowgeo rg = wseg_decode_row(d, min(mn, a.M - 1));
        xoy[k] = rg.oy; xox[k] = rg.ox; xbase[k] = (int)rg.in_base;
      } else if (a.simple_adv) {                     // branch-free: 64 = q*OW + r (per segment, host-computed), at most one image wrap
        int ox = xox[k] + (s2 ? a.r64_2 : a.r64_1), oy = xoy[k] + (s2 ? a.q64_2 : a.q64_1), bs = xbase[k];
        const bool cx = ox >= OWs;
        ox = cx ? ox - OWs : ox; oy = cx ? oy + 1 : oy;
        const bool cy = oy >= OHs;
        oy = cy ? oy - OHs : oy; bs = cy ? bs + IHs * IWs : bs;
        xox[k] = ox; xoy[k] = oy; xbase[k] = bs;
      } else {                                       // tiny maps (64 pixels span several images)
        int ox = xox[k] + PK, oy = xoy[k], bs = xbase[k];
        while (ox >= OWs) { ox -= OWs; ++oy; }
        while (oy >= OHs) { oy -= OHs; bs += IHs * IWs; }
        xox[k] = ox; xoy[k] = oy; xbase[k] = bs;
      }
      xm[k] = mn;
    }
  };
  auto issue_x = [&](int h, int buf) {
    char* dst = smem + slot_off(buf, 2 + h) + wid * 1024;
    if constexpr (UNIT) {
      const int rem = WG_DIAG(a) >= 4 ? 0 : x_lim - x_m;  // rows of this tile inside its sub-range (diag: X from the zero page)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const char* p = (xin[k] && rrx[k][h] < rem) ? xptr[k] + h * 256 : zsrc;
        glds16(p, dst + k * 8192);
      }
      if (h == 1) {
#pragma unroll
        for (int k = 0; k < 2; ++k) xptr[k] += xstep;
        x_m += PK; ++tx_idx;
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const char* p = (xrow[k] && xok[k][h]) ? xrow[k] + h * 256 : zsrc;
      glds16(p, dst + k * 8192);
    }
  };
  auto issue_y = [&](int h, int buf) {               // half h of the NEXT Y tile; h == 1 advances to the following tile
    char* dst = smem + slot_off(buf, h) + wid * 1024;
    if constexpr (UNIT) {
      if (h == 0 && ty_idx == nA && nA > 0 && nB > 0) {   // (wave-uniform, once) enter the second segment
        const long jump = (long)(b_begin - y_m) * d.ld_dy * 2;
#pragma unroll
        for (int k = 0; k < 2; ++k) ybase[k] += jump;
        y_m = b_begin; y_lim = m_end;
      }
      const int rem = WG_DIAG(a) == 5 ? 0 : y_lim - y_m;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const char* p = rry[k][h] < rem ? ybase[k] + h * 256 : zsrc;
        glds16(p, dst + k * 8192);
      }
      if (h == 1) {
#pragma unroll
        for (int k = 0; k < 2; ++k) ybase[k] += ystep;
        y_m += PK; ++ty_idx;
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const char* p = (my[k] < m_end && yok[k][h] && WG_DIAG(a) != 5) ? ybase[k] + h * 256 : zsrc;
      glds16(p, dst + k * 8192);
    }
    if (h == 1) {
#pragma unroll
      for (int k = 0; k < 2; ++k) { ybase[k] += ystep; my[k] += PK; }
    }
  };

  const int wr = wid >> 2, wc = wid & 3;
  const int fcol = lane & 15, fk = lane >> 4;
  const int q = (lane & 15) >> 2, p = lane & 3;
  // per-lane byte offsets of the transposed reads inside a half-tile: [ks][h] row, swizzled 32-B block per channel tile
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  unsigned rowoff[2][2], rsw[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = ks * 32 + fk * 8 + h * 4 + q;
      rowoff[ks][h] = row * ROWB + p * 8;
      rsw[ks][h] = (row & 3) | (((row >> 3) & 1) << 2);
    }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = UNIT ? nA + nB : (m_end - m_begin + PK - 1) / PK;
  // prologue: tile 0 entirely + the X halves of tile 1
  issue_y(0, 0); issue_y(1, 0);
  x_prepare(); issue_x(0, 0); issue_x(1, 0);
  if (nt > 1) { x_prepare(); issue_x(0, 1); issue_x(1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");

  bf16x4 va[2][2][4], vb0[2][2][2], vb1[2][2][2];   // [ks][h][tile] raw transposed reads
  bf16x8 af[2][4], b0[2][2], b1[2][2];
#define PACK_A()                                                                                                \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)               \
    af[ks][i] = __builtin_shufflevector(va[ks][0][i], va[ks][1][i], 0, 1, 2, 3, 4, 5, 6, 7);
#define PACK_B(BF, VB)                                                                                          \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int j = 0; j < 2; ++j)               \
    BF[ks][j] = __builtin_shufflevector(VB[ks][0][j], VB[ks][1][j], 0, 1, 2, 3, 4, 5, 6, 7);
#define WAIT_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define MFMA_Q(HA, HB, BF)                                                                                      \
  do {                                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)             \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                            \
        acc[(HA) * 4 + i][(HB) * 2 + j] =                                                                       \
            __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], BF[ks][j], acc[(HA) * 4 + i][(HB) * 2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                              \
  } while (0)

  {
    // TWO phases of 32 MFMAs per K-tile (rows 0-63, then 64-127 of the wave tile; both B fragments are read in phase 1 and
    // kept), each a read slot + an MFMA slot, LOCK-STEP (the ping-pong form of the conv kernel — waves 4-7 one slot
    // behind — and the 4-phase forms measured slower here: 15.9 / 14.85 / 17.1 vs 14.06 ms per step, profiles/HISTORY.md).  Refills: dY0/dY1(u+1) in R1, X0/X1(u+2) + the counted wait in R2 — a
    // slot's last reader (the late group's read slot) is always one barrier before the early group's next issue into it.
    // transposed-read addresses = persistent lane term + immediate: the lane's row (fk*8+q) and 8-B column (p) plus the
    // swizzled 32-B block (t ^ s) — s = q | (fk&1)<<2 does not depend on (ks, h) — and the slot of its wave; the buffer
    // (b*HALF) and the (ks, h) row offsets (ks*32 + h*4 rows) are compile-time immediates.  12 registers, no VALU per read.
    unsigned LTA[8], LTB[4];
    {
      const unsigned s_ = (unsigned)q | (((unsigned)fk & 1u) << 2);
      const unsigned rowterm = lds0 + (unsigned)(fk * 8 + q) * ROWB + (unsigned)p * 8;
#pragma unroll
      for (int t = 0; t < 8; ++t) LTA[t] = rowterm + (((unsigned)t ^ s_) << 5) + (unsigned)(wr * 2) * HALF;
#pragma unroll
      for (int t = 0; t < 4; ++t) LTB[t] = rowterm + (((unsigned)((wc & 1) * 4 + t) ^ s_) << 5) + (unsigned)((2 + (wc >> 1)) * 2) * HALF;
    }
    auto ktile = [&](auto BC, int u) {
      constexpr int b = decltype(BC)::value;
      constexpr int BO = b * HALF;
      // ---- R1 / M1: rows 0-63 x all 64 columns
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tr_read<BO + 0>(va[0][0][i], LTA[i]); tr_read<BO + 1024>(va[0][1][i], LTA[i]);
        tr_read<BO + 8192>(va[1][0][i], LTA[i]); tr_read<BO + 9216>(va[1][1][i], LTA[i]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        tr_read<BO + 0>(vb0[0][0][j], LTB[j]); tr_read<BO + 1024>(vb0[0][1][j], LTB[j]);
        tr_read<BO + 8192>(vb0[1][0][j], LTB[j]); tr_read<BO + 9216>(vb0[1][1][j], LTB[j]);
        tr_read<BO + 0>(vb1[0][0][j], LTB[2 + j]); tr_read<BO + 1024>(vb1[0][1][j], LTB[2 + j]);
        tr_read<BO + 8192>(vb1[1][0][j], LTB[2 + j]); tr_read<BO + 9216>(vb1[1][1][j], LTB[2 + j]);
      }
      if (u + 1 < nt && WG_DIAG(a) != 6) { issue_y(0, b ^ 1); issue_y(1, b ^ 1); }   // (diag 6, probe builds: no LDS-DMA request inside the loop — timing only)
      WAIT_LDS();
      PACK_A() PACK_B(b0, vb0) PACK_B(b1, vb1)
      MFMA_Q(0, 0, b0);
      MFMA_Q(0, 1, b1);
      asm volatile("s_barrier" ::: "memory");
      // ---- R2 / M2: rows 64-127; the counted wait publishes tile u+1 (only X0/X1(u+2) may stay in flight)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tr_read<BO + 0>(va[0][0][i], LTA[4 + i]); tr_read<BO + 1024>(va[0][1][i], LTA[4 + i]);
        tr_read<BO + 8192>(va[1][0][i], LTA[4 + i]); tr_read<BO + 9216>(va[1][1][i], LTA[4 + i]);
      }
      if (u + 2 < nt) { if (WG_DIAG(a) != 6) { x_prepare(); issue_x(0, b); issue_x(1, b); } asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WAIT_LDS();
      PACK_A()
      MFMA_Q(1, 1, b1);
      MFMA_Q(1, 0, b0);
      asm volatile("s_barrier" ::: "memory");
    };
    int u = 0;
    for (; u + 1 < nt; u += 2) {
      ktile(std::integral_constant<int, 0>{}, u);
      ktile(std::integral_constant<int, 1>{}, u + 1);
    }
    if (u < nt) ktile(std::integral_constant<int, 0>{}, u);
  }
#undef PACK_A
#undef PACK_B
#undef WAIT_LDS
#undef MFMA_Q
  __syncthreads();                                 // every wave is done with the pipeline buffers

  if (a.wave_epi) {
    // Epilogue, wave-local (default; WSEG_WGRAD_EPI=0 = the block-wide image below): every wave adds its own 128(oc) x 64(ic) accumulator tile to dW through
    // a private 16-row LDS scratch, no workgroup barrier; one atomic wave-instruction = 64 consecutive floats (256 B).
    constexpr int WLD = 64 + 4;
    float* wimg = reinterpret_cast<float*>(smem) + wid * (16 * WLD);
    const size_t row_stride = (size_t)a.taps * d.IC_dw;
    const int ic = ic0 + wc * 64 + lane;
    const bool ic_ok = ic < d.IC_dw;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) wimg[(fk * 4 + e) * WLD + j * 16 + fcol] = acc[i][j][e];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int oc_base = oc0 + wr * 128 + i * 16;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = oc_base + r;
        if (ic_ok && oc < d.OC_dw) atomicAdd(&d.dw[(size_t)oc * row_stride + (size_t)tap * d.IC_dw + ic], wimg[r * WLD + lane]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
  float* img = reinterpret_cast<float*>(smem);
  const size_t row_stride = (size_t)a.taps * d.IC_dw;
#pragma unroll 1
  for (int ps = 0; ps < BO / EPI_ROWS; ++ps) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = wr * 8 + i;
      if (t / 4 == ps) {
        const int row = (t % 4) * 16 + fk * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int col = (wc * 4 + j) * 16 + fcol;
#pragma unroll
          for (int e = 0; e < 4; ++e) img[(row + e) * EPI_LD + col] = acc[i][j][e];
        }
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int idx = tid; idx < EPI_ROWS * BI; idx += NT) {
      const int row = idx / BI, col = idx - row * BI;
      const int oc = oc0 + ps * EPI_ROWS + row, ic = ic0 + col;
      if (oc < d.OC_dw && ic < d.IC_dw) {
        float* dst = &d.dw[(size_t)oc * row_stride + (size_t)tap * d.IC_dw + ic];
        if (WG_DIAG(a) == 0) atomicAdd(dst, img[row * EPI_LD + col]);
        else if (WG_DIAG(a) == 2) *dst = img[row * EPI_LD + col];       // (timing diagnostics only: plain store / nothing)
      }
    }
    __syncthreads();
  }
}

template <int UNIT>
__global__ __launch_bounds__(512, 2) void conv_wgrad_pipe_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 4 * 16384];
  conv_wgrad_pipe_tile<UNIT>(a, smem, blockIdx.x);
}

// host side: validation, tile geometry, split-K choice.  kind: 0 = 256x256 phase-pipelined kernel (bf16, big layers; `unit` picks its UNIT variant),
// 1 = 256x256 plain, 2 = 128x128 (any dtype)
struct Plan { Args a; int kind; bool unit; };
static int wgrad_plan(const wseg_wgrad_desc* d, Plan& pl) {
  Args& a = pl.a;
  WSEG_CHECK(d && d->x && d->dy && d->dw, "conv_wgrad: null pointer");
  WSEG_CHECK(d->dtype == WSEG_F32 || d->dtype == WSEG_BF16 || d->dtype == WSEG_F32X3, "conv_wgrad: bad dtype");
  WSEG_CHECK(d->IC % 8 == 0 && d->OC % 8 == 0 && d->ld_x % 8 == 0 && d->ld_dy % 8 == 0,
             "conv_wgrad: IC/OC/ld must be multiples of 8 (IC=%d OC=%d ld_x=%d ld_dy=%d)", d->IC, d->OC, d->ld_x, d->ld_dy);
  WSEG_CHECK(d->ld_x >= d->IC && d->ld_dy >= d->OC, "conv_wgrad: leading dims too small");
  WSEG_CHECK(d->IC_dw > 0 && d->IC_dw <= d->IC && d->OC_dw > 0 && d->OC_dw <= d->OC, "conv_wgrad: bad dw extents");
  WSEG_CHECK(d->dw_rot >= 0 && d->dw_rot < d->IC_dw, "conv_wgrad: dw_rot=%d must lie in [0, IC_dw)", d->dw_rot);
  WSEG_CHECK(d->N > 0 && d->OH > 0 && d->OW > 0 && d->IH > 0 && d->IW > 0 && d->stride >= 1 && d->dil >= 1, "conv_wgrad: bad shape");
  const long M = (long)d->N * d->OH * d->OW + (long)d->N * d->OH2 * d->OW2;
  WSEG_CHECK(d->OH2 >= 0 && (d->OH2 == 0 || (d->OW2 > 0 && d->IH2 > 0 && d->IW2 > 0)), "conv_wgrad: bad second segment");
  WSEG_CHECK(M < (1L << 31), "conv_wgrad: too many pixels");
  const bool big = d->dtype == WSEG_BF16 && d->OC >= 256 && d->IC >= 256 && d->tile_hint != 128 &&
                   (M >= 16384 || d->tile_hint == 256);   // few pixels: the 128^2 geometry fills the chip better
  const int BO = big ? 256 : 128, BI = BO;
  a.d = *d;
  a.M = (int)M;
  a.taps = d->KH * d->KW;
  a.nto = (d->OC + BO - 1) / BO;
  a.nti = (d->IC + BI - 1) / BI;
  a.ntiles = a.nto * a.nti * a.taps;
  const int pk = d->dtype == WSEG_BF16 ? 64 : 32;       // (the ring variant steps by 32: 64 is a multiple)
  int split = d->split_k;
  if (split <= 0) {
    // pick the split that minimises  rounds x (K-steps per workgroup + epilogue cost):  a tile count that is
    // not a multiple of the resident-workgroup slots otherwise leaves a nearly empty last round
    // (36 tiles x 8 splits = 288 workgroups on 256 single-workgroup CUs ran two rounds).
    const long slots = big ? 256 : 512;
    const double epi = big ? 28.0 : 10.0;          // epilogue (LDS image + float atomics) in K-step units
    const int max_split = (int)std::max(1L, std::min(64L, M / (pk * 8)));
    double best = 1e30;
    split = 1;
    for (int sp = 1; sp <= max_split; ++sp) {
      const long blocks = (long)a.ntiles * sp;
      const long rounds = (blocks + slots - 1) / slots;
      const double steps = (double)((M + sp - 1) / sp + pk - 1) / pk;
      const double cost = rounds * (steps + epi) * (1.0 + 0.002 * sp);     // mild preference for fewer partial sums
      if (cost < best) { best = cost; split = sp; }
    }
  }
  long pps = (M + split - 1) / split;
  pps = (pps + pk - 1) / pk * pk;
  a.pix_per_split = (int)pps;
  split = (int)((M + pps - 1) / pps);
  a.nsplit = split;
  a.nwg = a.ntiles * split;
  a.q64_1 = 64 / d->OW; a.r64_1 = 64 % d->OW;
  a.q64_2 = d->OH2 ? 64 / d->OW2 : 0; a.r64_2 = d->OH2 ? 64 % d->OW2 : 0;
  a.simple_adv = (a.q64_1 + 1 <= d->OH) && (d->OH2 == 0 || a.q64_2 + 1 <= d->OH2);
#ifdef WSEG_PROBES
  static const int diag = getenv("WSEG_WGRAD_DIAG") ? atoi(getenv("WSEG_WGRAD_DIAG")) : 0;
  a.diag = diag;
#else
  a.diag = 0;
#endif
  static const int wave_epi = getenv("WSEG_WGRAD_EPI") ? atoi(getenv("WSEG_WGRAD_EPI")) : 1;   // (same-box A/B: 12.64 vs 12.73 ms/step)
  a.wave_epi = wave_epi;
  static const bool use_pipe = !(getenv("WSEG_WGRAD_PIPE") && getenv("WSEG_WGRAD_PIPE")[0] == '0');
  static const int unit_ok = getenv("WSEG_WGRAD_UNIT") ? atoi(getenv("WSEG_WGRAD_UNIT")) : 1;
  pl.unit = unit_ok && a.simple_adv && d->stride == 1 && d->IH == d->OH && d->IW == d->OW &&
            (d->OH2 == 0 || (d->IH2 == d->OH2 && d->IW2 == d->OW2));
  pl.kind = (big && use_pipe) ? 0 : (big ? 1 : 2);
  WSEG_CHECK(d->dw_rot == 0 || pl.kind == 2, "conv_wgrad: dw_rot is supported by the 128-tile kernel only");
  return 0;
}
}  // namespace
}  // namespace wseg_wg
