// conv_igemm.hip — NHWC convolution forward / data-gradient as an implicit GEMM on MFMA.
//
//   C[m][oc] = sum_{tap} sum_{ic} IN[pix(m,tap)][ic] * W[oc][tap][ic]
//
// Both operands are K-contiguous ("pixel rows" of channels, weight rows of [tap][ic]), so both
// LDS tiles are [128 rows][128 B] and are filled by LDS-DMA (global_load_lds_dwordx4): the
// im2col gather is nothing but the per-lane SOURCE address; zero padding is a 256-B zero page.
// Tile 128(M pixels) x 128(N out-channels) x 128 B of K per step, 4 waves (2x2), each wave a
// 64x64 sub-tile = 4x4 MFMA 16x16 accumulators.  bf16: v_mfma_f32_16x16x32_bf16;
// f32 (parity mode): v_mfma_f32_16x16x4_f32 (bit-exact f32 fma chain).
// LDS rows are XOR-swizzled at 16-B granularity (phys = chunk ^ ((row>>1)&7)) — applied on the
// DMA source side and on the ds_read side (the LDS image itself stays lane-linear).
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "conv_wgrad_kernels.h"   // (wseg_wg::…: the weight-gradient tile body and its host-side plan, for wseg_conv_bwd_pair)

#ifdef WSEG_PROBES   // timing diagnostics of probe builds: bm_hint -1 / -2 feed A / B from the zero page (results wrong by design)
#define WSEG_DIAG_ZERO_A(d) ((d).bm_hint == -1)
#define WSEG_DIAG_ZERO_B(d) ((d).bm_hint == -2)
#else
#define WSEG_DIAG_ZERO_A(d) false
#define WSEG_DIAG_ZERO_B(d) false
#endif

#ifdef WSEG_PROBES
// In-kernel stamps of the 256-tile body (probe builds only: `WSEG_PROBES=1 bash build.sh`): per workgroup 8 x s_memrealtime (100 MHz) — entry, gather
// set-up done, first tiles landed, main loop done (early wave group), tile done; slots 5 / 6: main loop / tile done of the late group (wave 4);
// slot 7: XCC id.  Written to a buffer of their own that no kernel reads (scripts/conv_tile_breakdown.py fetches it).
__device__ unsigned long long g_wseg_stamps[24 * 4096];   // per workgroup: 8 wall-clock stamps, then (WSEG_SLOTS builds) 8 slot sums of wave 0 and 8 of wave 4
#define WSEG_STAMP(slot, wave)                                                                                  \
  do { if (threadIdx.x == (wave) * 64 && bid < 4096) g_wseg_stamps[bid * 24 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int wseg_debug_stamps(void* out, size_t bytes) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wseg_stamps), bytes < sizeof(g_wseg_stamps) ? bytes : sizeof(g_wseg_stamps)) == hipSuccess ? 0 : -1;
}
#define WSEG_CSTAMP(slot, wave)   /* shader-clock stamp (s_memtime): with the wall-clock stamps beside it, the clock the loop ran at */ \
  do { if (threadIdx.x == (wave) * 64 && bid < 4096) g_wseg_stamps[bid * 24 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
// probe-only timing switches (results wrong by design): bit 0 = request the A tile only on every 9th K-tile, bit 1 = no B requests after the prologue,
// bit 2 = no A requests after the prologue, bit 3 = the A pointers never move (every request re-reads the tile's first K-tile: cache-resident), bit 4 = the same for B, bit 5 = the wave-local epilogue stores nothing (its loads and LDS round trips stay), bit 6 = it loads nothing either
__device__ int g_wseg_diag;
extern "C" int wseg_debug_set_diag(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_wseg_diag), &v, sizeof(int)) == hipSuccess ? 0 : -1; }
#define WSEG_DIAG_LOAD() const int diag_ = __builtin_amdgcn_readfirstlane(g_wseg_diag)
#define WSEG_DIAG_A_OK(u) (!(diag_ & 4) && (!(diag_ & 1) || (u) % 9 == 8))
#define WSEG_DIAG_B_OK() (!(diag_ & 2))
#define WSEG_DIAG_A_MOVES() (!(diag_ & 8))
#define WSEG_DIAG_B_MOVES() (!(diag_ & 16))
#define WSEG_DIAG_EPI_STORES() (!(__builtin_amdgcn_readfirstlane(g_wseg_diag) & 32))
#define WSEG_DIAG_EPI_LOADS() (!(__builtin_amdgcn_readfirstlane(g_wseg_diag) & 64))
#else
#define WSEG_STAMP(slot, wave) do { } while (0)
#define WSEG_CSTAMP(slot, wave) do { } while (0)
#define WSEG_DIAG_LOAD() do { } while (0)
#define WSEG_DIAG_A_OK(u) true
#define WSEG_DIAG_B_OK() true
#define WSEG_DIAG_A_MOVES() true
#define WSEG_DIAG_B_MOVES() true
#define WSEG_DIAG_EPI_STORES() true
#define WSEG_DIAG_EPI_LOADS() true
#endif
// WSEG_SLOTS (with WSEG_PROBES): cycles (s_memtime) a wave spends in each slot of the main loop, summed over the K-tiles: read slot 1 (fragment reads
// until they have landed + LDS-DMA issue), barrier, MFMA slot 1, barrier, read slot 2 (+ the counted DMA wait), barrier, MFMA slot 2, barrier.
// The stamps serialise what the real kernel overlaps (each waits for lgkmcnt(0)): read the SHARES, not the run time of this build.
#if defined(WSEG_PROBES) && defined(WSEG_SLOTS)
#define WSEG_SLOT_DECL() unsigned long long sl_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = 0
#define WSEG_SLOT_BEGIN() asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tp_) :: "memory")
#define WSEG_SLOT(i)                                                                                            \
  do {                                                                                                          \
    unsigned long long t_;                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                               \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    sl_[i] += t_ - tp_; tp_ = t_;                                                                               \
  } while (0)
#define WSEG_SLOT_FLUSH()                                                                                       \
  do {                                                                                                          \
    if ((threadIdx.x == 0 || threadIdx.x == 256) && bid < 4096) {                                               \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) g_wseg_stamps[bid * 24 + 8 + (threadIdx.x >> 8) * 8 + i_] = sl_[i_]; \
    }                                                                                                           \
  } while (0)
#else
#define WSEG_SLOT_DECL() do { } while (0)
#define WSEG_SLOT_BEGIN() do { } while (0)
#define WSEG_SLOT(i) do { } while (0)
#define WSEG_SLOT_FLUSH() do { } while (0)
#endif

namespace {

constexpr int BN = 128, ROWB = 128;                // out-channel rows per tile, bytes of K per LDS row
constexpr int B_TILE = BN * ROWB;                  // 16 KiB weight tile
constexpr int EPI_LD = BN + 4;                     // f32 epilogue image row stride (floats)

struct Args {
  wseg_conv_desc d;
  int M;            // N*OH*OW
  int taps;         // KH*KW
  int cpt;          // K-steps per tap = IC*ES/128
  int cpt2;         // two-source 1x1: K-steps of the second source (= IC2*ES/128)
  int krow;         // K elements per weight row (taps*IC, or IC + IC2)
  int ntn;          // column tiles
  int nwg;
  int row0;         // first output row (pixel) of this launch (a launch may cover a row sub-range)
  int perm;         // 256-tile kernel, stride-2 dgrad: rows are taken in PARITY-CLASS order (see perm_decode)
  int Q1, Q2;       // rows per parity class in segment 1 / 2 (= N * OH/2 * OW/2)
};

// Stride-2 data gradient: an output pixel (y, x) only receives the taps with (y + pad - ky*dil) and (x + pad - kx*dil) even —
// a quarter of the 3x3 taps on average, but consecutive pixels alternate parity, so a tile in natural row order needs every
// tap (zero-page rows for the invalid ones: 4x wasted MFMA work).  The launch therefore walks the rows in parity-class order:
// row m' = [segment][class (py,px)][n][i][j] <-> pixel (n, 2i+py, 2j+px); a tile that lies inside one class runs only that
// class's taps.  Only the row <-> pixel bijection changes: gather addresses and the epilogue use the true pixel.
struct PermRow { wseg_rowgeo g; int cls; long true_row; };
__device__ __forceinline__ PermRow perm_decode(const Args& a, int m) {
  const wseg_conv_desc& d = a.d;
  PermRow r;
  int rr = m, H = d.OH, W = d.OW, Q = a.Q1, seg = 0;
  if (d.OH2 != 0 && rr >= 4 * a.Q1) { rr -= 4 * a.Q1; H = d.OH2; W = d.OW2; Q = a.Q2; seg = 1; }
  const int cls = rr / Q, rem = rr - cls * Q;
  const int hw2 = (H >> 1) * (W >> 1), w2 = W >> 1;
  const int n = rem / hw2, rem2 = rem - n * hw2;
  const int i = rem2 / w2, j = rem2 - i * w2;
  r.g.oy = 2 * i + (cls >> 1); r.g.ox = 2 * j + (cls & 1);
  r.g.n_glob = seg ? d.N + n : n;
  r.g.IH = seg ? d.IH2 : d.IH; r.g.IW = seg ? d.IW2 : d.IW;
  r.g.in_base = seg ? (long)d.N * d.IH * d.IW + (long)n * d.IH2 * d.IW2 : (long)n * d.IH * d.IW;
  r.cls = seg * 4 + cls;
  r.true_row = seg ? (long)d.N * d.OH * d.OW + ((long)n * H + r.g.oy) * W + r.g.ox : ((long)n * H + r.g.oy) * W + r.g.ox;
  return r;
}


// Per-thread BN scale / shift of its 8-channel column group (1 / 0 when absent).
__device__ __forceinline__ void epilogue_coeffs(const wseg_conv_desc& d, int n0, int cv, float (&sc)[8], float (&sh)[8]) {
  const int oc_raw = n0 + cv;
  const int oc = oc_raw < d.OC ? oc_raw : 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
  if (d.scale != nullptr) load8<WSEG_F32>(d.scale, oc, sc);
  if (d.shift != nullptr) load8<WSEG_F32>(d.shift, oc, sh);
}

// Fused epilogue over an f32 LDS image of ROWS x COLS accumulators (row stride LD floats), NT threads.
// Every thread owns ONE 8-channel column group.  All global operands of a chunk of rows are loaded before
// any of them is consumed: one memory round trip per chunk instead of one per row (the dependent-load
// chain was the whole cost of short-K layers).  Rows beyond M are clamped to row 0 for the loads and
// predicated off at the stores (no divergent control flow).
template <int DT, int EPI, int ROWS, int COLS, int LD, int NT, int MAXCHK = 4>
__device__ __forceinline__ void epilogue_image(const wseg_conv_desc& d, int M, const float* img, int m0, int n0, int tid,
                                               const float (&sc)[8], const float (&sh)[8]) {
  constexpr int GPR = COLS / 8;                    // column groups per row
  constexpr int RPS = NT / GPR;                    // rows per sweep of the workgroup
  constexpr int SWEEPS = ROWS / RPS;
  constexpr int CHK = SWEEPS < MAXCHK ? SWEEPS : MAXCHK;
  static_assert(NT % GPR == 0 && ROWS % RPS == 0 && SWEEPS % CHK == 0, "epilogue geometry");
  const int cv = (tid % GPR) * 8;
  const int oc_raw = n0 + cv;
  const bool col_ok = oc_raw < d.OC;                        // OC is a multiple of 8 (host-checked)
  const int oc = col_ok ? oc_raw : 0;
  const bool has_pre = d.r_pre != nullptr, has_post = d.r_post != nullptr, has_mask = d.mask != nullptr;
  const bool has_drop = d.drop != nullptr;
#pragma unroll 1
  for (int c0 = 0; c0 < SWEEPS; c0 += CHK) {
    float rpre[CHK][8], rpost[CHK][8], mk[CHK][8], dr[CHK][8];
    size_t mrow[CHK];
    bool ok[CHK];
#pragma unroll
    for (int j = 0; j < CHK; ++j) {
      const int row = ((c0 + j) * NT + tid) / GPR;
      ok[j] = col_ok && (m0 + row) < M;
      mrow[j] = ok[j] ? (size_t)(m0 + row) : 0;
    }
    if (has_pre) {
#pragma unroll
      for (int j = 0; j < CHK; ++j) load8<DT>(d.r_pre, mrow[j] * d.ld_rpre + oc, rpre[j]);
    }
    if (has_post) {
#pragma unroll
      for (int j = 0; j < CHK; ++j) load8<DT>(d.r_post, mrow[j] * d.ld_rpost + oc, rpost[j]);
    }
    if (EPI == 1 && has_mask) {
#pragma unroll
      for (int j = 0; j < CHK; ++j) load8<DT>(d.mask, mrow[j] * d.ld_mask + oc, mk[j]);
    }
    if (EPI != 2 && has_drop) {
#pragma unroll
      for (int j = 0; j < CHK; ++j) load8<WSEG_F32>(d.drop, (size_t)wseg_decode_row(d, (int)mrow[j]).n_glob * d.OC + oc, dr[j]);
    }
#pragma unroll
    for (int j = 0; j < CHK; ++j) {
      const int row = ((c0 + j) * NT + tid) / GPR;
      const size_t m = mrow[j];
      float v[8];
      {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(&img[row * LD + cv]);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(&img[row * LD + cv + 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = p0[e]; v[4 + e] = p1[e]; }
      }
      if (has_pre) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rpre[j][e];
      }
      if constexpr (EPI == 0) {
        if (has_post) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rpost[j][e];
        }
        if (d.relu_lt > 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (oc + e < d.relu_lt) v[e] = fmaxf(v[e], 0.f);
        }
        if (d.out != nullptr && ok[j]) store8<DT>(d.out, m * d.ld_out + oc, v);
        if (d.out2 != nullptr) {
          float t[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = v[e] * sc[e] + sh[e];
            if (d.relu_out2) x = fmaxf(x, 0.f);
            if (has_drop) x *= dr[j][e];
            t[e] = x;
          }
          if (ok[j]) store8<DT>(d.out2, m * d.ld_out2 + oc, t);
        }
      } else if constexpr (EPI == 1) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = v[e] * sc[e];
          if (has_drop) x *= dr[j][e];
          if (has_mask) x = mk[j][e] > 0.f ? x : 0.f;
          if (has_post) x += rpost[j][e];
          o[e] = x;
        }
        if (ok[j]) store8<DT>(d.out, m * d.ld_out + oc, o);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        if (ok[j]) store8<DT>(d.out, m * d.ld_out + oc, v);
      }
    }
  }
}

// ---- wave-local epilogue of the phase-pipelined kernels: every wave turns its own (NI*16) x 64 accumulator tile into
// 8-channel vectors through a private 16-row LDS scratch (4.25 KiB), NI steps, NO workgroup barrier: the waves drift apart, so
// one wave's global-load latency (residual / mask / dropout operands, 16 B per lane) overlaps with the other waves' LDS and
// store work.  (A block-wide LDS image needed 8 barriers and 8 serialised load round trips per 256x256 tile.)  Stores are full
// 128-B lines: 8 lanes x 16 B per output row.  mw0 = first row of the wave's tile, col0 = its first column inside the
// column tile n0.  The caller has made sure (barrier) that no wave still reads the pipeline buffers.
// Operand prefetch: the residual / mask vectors of FOUR steps are requested together, before the first scratch round trip
// (raw 16-B vectors, 64 registers), so a tile's epilogue exposes two global-load latencies instead of one per step
// (the per-step version: 7-8 x ~1.5 us of a ~100 us tile).  When r_pre and r_post are both present (never in this network),
// r_post stays a per-step load; the per-image dropout factors (b6 / b7 only, L2-resident) too.
// raw8<DT>: 8 consecutive channels as they lie in memory (bf16: one 16-B vector; f32 storage: two), requested early, unpacked late
template <int DT> struct raw8;
template <> struct raw8<WSEG_BF16> {
  uint4 q;
  __device__ __forceinline__ void load(const void* p, size_t i) { q = *reinterpret_cast<const uint4*>((const bf16_t*)p + i); }
  __device__ __forceinline__ void unpack(float (&v)[8]) const {
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
  }
};
template <> struct raw8<WSEG_F32X3> {
  f32x4 q0, q1;
  __device__ __forceinline__ void load(const void* p, size_t i) {
    q0 = *reinterpret_cast<const f32x4*>((const float*)p + i); q1 = *reinterpret_cast<const f32x4*>((const float*)p + i + 4);
  }
  __device__ __forceinline__ void unpack(float (&v)[8]) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = q0[e]; v[4 + e] = q1[e]; }
  }
};

template <int EPI, int NI, int I0, int NB, int DT>
__device__ __forceinline__ void wave_local_epilogue_batch(const Args& a, float* wimg, int lane, int mw0, int oc, bool col_ok,
                                                          const float (&sc)[8], const float (&sh)[8], const f32x4 (&acc)[NI][4]) {
  constexpr int WLD = 64 + 4;
  const wseg_conv_desc& d = a.d;
  const int frow = lane & 15, fk = lane >> 4;
  const int vr = lane >> 3, vg = lane & 7;
  const bool ld_ok_ = WSEG_DIAG_EPI_LOADS();       // (probe builds only: compiled out of the product)
  const bool has_pre = d.r_pre != nullptr && ld_ok_, has_post = d.r_post != nullptr && ld_ok_, has_mask = EPI == 1 && d.mask != nullptr && ld_ok_;
  const bool has_drop = EPI != 2 && d.drop != nullptr && ld_ok_;
  const bool res_is_pre = has_pre;                 // the prefetched residual: r_pre when present, else r_post
  const bool post_in_step = has_pre && has_post;
  const void* res_p = res_is_pre ? d.r_pre : d.r_post;
  const size_t res_ld = res_is_pre ? d.ld_rpre : d.ld_rpost;
  const bool has_res = has_pre || has_post;
  int mrow[NB][2]; bool ok[NB][2];
  raw8<DT> qres[NB][2], qmk[NB][2];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int m = mw0 + (I0 + i) * 16 + vr + 8 * t;
      ok[i][t] = col_ok && m < a.M;
      mrow[i][t] = ok[i][t] ? (a.perm ? (int)perm_decode(a, m).true_row : m) : 0;
      ok[i][t] = ok[i][t] && WSEG_DIAG_EPI_STORES();
      if (has_res) qres[i][t].load(res_p, (size_t)mrow[i][t] * res_ld + oc);
      if (has_mask) qmk[i][t].load(d.mask, (size_t)mrow[i][t] * d.ld_mask + oc);
    }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    float rpost[2][8], dr[2][8];
    if (post_in_step) {
#pragma unroll
      for (int t = 0; t < 2; ++t) load8<DT>(d.r_post, (size_t)mrow[i][t] * d.ld_rpost + oc, rpost[t]);
    }
    if (has_drop) {
#pragma unroll
      for (int t = 0; t < 2; ++t) load8<WSEG_F32>(d.drop, (size_t)wseg_decode_row(d, mrow[i][t]).n_glob * d.OC + oc, dr[t]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) wimg[(fk * 4 + e) * WLD + j * 16 + frow] = acc[I0 + i][j][e];   // C/D: col = lane&15, row = (lane>>4)*4 + reg
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-local: the scratch is written and read by this wave only
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float v[8];
      {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(&wimg[(vr + 8 * t) * WLD + vg * 8]);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(&wimg[(vr + 8 * t) * WLD + vg * 8 + 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = p0[e]; v[4 + e] = p1[e]; }
      }
      const size_t m = (size_t)mrow[i][t];
      float res[8];
      if (has_res) qres[i][t].unpack(res);
      if (has_pre) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += res[e];
      }
      if (!post_in_step && has_post) {
#pragma unroll
        for (int e = 0; e < 8; ++e) rpost[t][e] = res[e];
      }
      if constexpr (EPI == 0) {
        if (has_post) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rpost[t][e];
        }
        if (d.relu_lt > 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (oc + e < d.relu_lt) v[e] = fmaxf(v[e], 0.f);
        }
        if (d.out != nullptr && ok[i][t]) store8<DT>(d.out, m * d.ld_out + oc, v);
        if (d.out2 != nullptr) {
          float o2[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = v[e] * sc[e] + sh[e];
            if (d.relu_out2) x = fmaxf(x, 0.f);
            if (has_drop) x *= dr[t][e];
            o2[e] = x;
          }
          if (ok[i][t]) store8<DT>(d.out2, m * d.ld_out2 + oc, o2);
        }
      } else if constexpr (EPI == 1) {
        float mk[8];
        if (has_mask) qmk[i][t].unpack(mk);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = v[e] * sc[e];
          if (has_drop) x *= dr[t][e];
          if (has_mask) x = mk[e] > 0.f ? x : 0.f;
          if (has_post) x += rpost[t][e];
          o[e] = x;
        }
        if (ok[i][t]) store8<DT>(d.out, m * d.ld_out + oc, o);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        if (ok[i][t]) store8<DT>(d.out, m * d.ld_out + oc, v);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // scratch reads done before the next step overwrites it
  }
}

template <int EPI, int NI, int DT, int I0, int NB>
__device__ __forceinline__ void wave_local_epilogue_rec(const Args& a, float* wimg, int lane, int mw0, int oc, bool col_ok,
                                                        const float (&sc)[8], const float (&sh)[8], const f32x4 (&acc)[NI][4]) {
  if constexpr (I0 < NI) {
    wave_local_epilogue_batch<EPI, NI, I0, (NI - I0 < NB ? NI - I0 : NB), DT>(a, wimg, lane, mw0, oc, col_ok, sc, sh, acc);
    wave_local_epilogue_rec<EPI, NI, DT, I0 + NB, NB>(a, wimg, lane, mw0, oc, col_ok, sc, sh, acc);
  }
}

template <int EPI, int NI, int DT = WSEG_BF16>
__device__ __forceinline__ void wave_local_epilogue(const Args& a, char* smem, int wid, int lane, int mw0, int col0, int n0,
                                                    const f32x4 (&acc)[NI][4]) {
  const wseg_conv_desc& d = a.d;
  constexpr int WLD = 64 + 4;                      // scratch row stride (floats): conflict-free for both access patterns
  float* wimg = reinterpret_cast<float*>(smem) + wid * (16 * WLD);
  const int vg = lane & 7;                         // this lane's vectors: rows vr and vr + 8 of the step, column group vg
  const int oc_raw = n0 + col0 + vg * 8;
  const bool col_ok = oc_raw < d.OC;
  const int oc = col_ok ? oc_raw : 0;
  float sc[8], sh[8];
  epilogue_coeffs(d, n0, col0 + vg * 8, sc, sh);
  constexpr int NB = DT == WSEG_BF16 ? 4 : 2;      // operand prefetch depth in steps (f32 storage: vectors are twice as wide)
  wave_local_epilogue_rec<EPI, NI, DT, 0, NB>(a, wimg, lane, mw0, oc, col_ok, sc, sh, acc);
}

// BM = 128 (default) or 64 (few output pixels: twice the workgroups for the same work)
template <int DT, int EPI, int BM>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const Args a) {
  constexpr int AI = BM / 32;                      // A pieces per thread per K-step == 16-row MFMA tiles per wave
  constexpr int A_TILE = BM * ROWB;
  constexpr int STAGE = A_TILE + B_TILE;
  constexpr int SMEM_BYTES = (BM * EPI_LD * 4 > 2 * STAGE) ? BM * EPI_LD * 4 : 2 * STAGE;
  constexpr int ES = elem<DT>::size;
  constexpr int CH = 16 / ES;                      // elements per 16-B chunk
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  const wseg_conv_desc& d = a.d;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, a.nwg);
  const int tm = tile / a.ntn, tn = tile - tm * a.ntn;
  const int m0 = a.row0 + tm * BM, n0 = tn * BN;

  const char* zero = reinterpret_cast<const char*>(g_wseg_zero_page);
  const char* IN = reinterpret_cast<const char*>(d.in);
  const char* Wp = reinterpret_cast<const char*>(d.w);

  // ---- staging assignment: thread -> 4 A rows + 4 B rows, one 16-B chunk each per K-step
  const int srow = lane >> 3;                      // row within an 8-row DMA piece
  const int pch = lane & 7;                        // physical 16-B chunk in the 128-B row
  int a_iy0[AI], a_ix0[AI], a_H[AI], a_W[AI];
  long a_img[AI];                                  // first input row of the image, or -1 when the row is beyond M
  const char* bptr[4];
  int b_inc[4];
  int lch[4];                                      // logical chunk this lane fetches for B piece i
  int lcha[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int r = wid * (BM / 4) + i * 8 + srow;   // row inside the A tile
    lcha[i] = pch ^ ((r >> 1) & 7);
    const int m = m0 + r;
    if (m < a.M) {
      const wseg_rowgeo rg = wseg_decode_row(d, m);
      const int oy = rg.oy, ox = rg.ox;
      a_img[i] = rg.in_base; a_H[i] = rg.IH; a_W[i] = rg.IW;
      if (d.mode == 0) { a_iy0[i] = oy * d.stride - d.pad; a_ix0[i] = ox * d.stride - d.pad; }
      else             { a_iy0[i] = oy + d.pad;            a_ix0[i] = ox + d.pad; }
    } else {
      a_img[i] = -1; a_iy0[i] = 0; a_ix0[i] = 0; a_H[i] = 1; a_W[i] = 1;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wid * 32 + i * 8 + srow;         // row inside the B tile
    lch[i] = pch ^ ((r >> 1) & 7);
    const int oc = n0 + r;
    if (oc < d.OC && !WSEG_DIAG_ZERO_B(d)) {
      bptr[i] = Wp + ((size_t)oc * a.taps * d.IC + (size_t)lch[i] * CH) * ES;
      b_inc[i] = ROWB;
    } else {
      bptr[i] = zero + pch * 16; b_inc[i] = 0;
    }
  }
  const char* aptr[AI];
  int a_inc[AI];
  auto set_tap = [&](int tap) {
    const int ky = tap / d.KW, kx = tap - ky * d.KW;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      int iy, ix; bool ok = a_img[i] >= 0;
      if (d.mode == 0) {
        iy = a_iy0[i] + ky * d.dil; ix = a_ix0[i] + kx * d.dil;
      } else {
        const int ty = a_iy0[i] - ky * d.dil, tx = a_ix0[i] - kx * d.dil;
        ok = ok && ty >= 0 && tx >= 0;
        if (d.stride == 1) { iy = ty; ix = tx; }
        else { iy = ty / d.stride; ix = tx / d.stride; ok = ok && (iy * d.stride == ty) && (ix * d.stride == tx); }
      }
      ok = ok && iy >= 0 && iy < a_H[i] && ix >= 0 && ix < a_W[i] && !WSEG_DIAG_ZERO_A(d);
      if (ok) {
        aptr[i] = IN + ((size_t)(a_img[i] + (long)iy * a_W[i] + ix) * d.ld_in + (size_t)lcha[i] * CH) * ES;
        a_inc[i] = ROWB;
      } else {
        aptr[i] = zero + pch * 16; a_inc[i] = 0;
      }
    }
  };

  auto stage = [&](int buf) {
    char* la = smem + buf * STAGE + wid * (BM / 4) * ROWB;
    char* lb = smem + buf * STAGE + A_TILE + wid * 32 * ROWB;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      glds16(aptr[i], la + i * 8 * ROWB);
      aptr[i] += a_inc[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(bptr[i], lb + i * 8 * ROWB);
      bptr[i] += b_inc[i];
    }
  };

  // ---- MFMA read addressing
  const int wr = wid >> 1, wc = wid & 1;
  const int frow = lane & 15, fk = lane >> 4;
  const int sw = (lane >> 1) & 7;                  // ((row>>1)&7) for row = ...+16*i+frow
  const int a_rd = (wr * (BM / 2) + frow) * ROWB;
  const int b_rd = A_TILE + (wc * 64 + frow) * ROWB;

  f32x4 acc[AI][4];
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = a.taps * a.cpt;
  int tap = 0, cc = 0;
  set_tap(0);
  stage(0);
  __syncthreads();                                 // (emits vmcnt(0): DMA landed)
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      if (++cc == a.cpt) { cc = 0; ++tap; set_tap(tap); }
      stage(cur ^ 1);
    }
    const char* base = smem + cur * STAGE;
    if constexpr (DT == WSEG_F32X3) {
      // split-bf16 products: the A tile is f32 (32 channels per 128-B row; lane (frow, fk) takes channels 8fk..8fk+7 = 16-B chunks
      // 2fk, 2fk+1 and splits them into hi + lo), the B tile is the pre-split pack [32 hi | 32 lo] bf16 of the same 32 channels
      // (chunks fk and 4+fk).  Three MFMAs per accumulator: lo.hi + hi.lo first, hi.hi last.
      bf16x8 ah[AI], al[AI], bh[4], bl[4];
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(base + a_rd + i * 16 * ROWB + (((2 * fk) ^ sw) * 16));
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(base + a_rd + i * 16 * ROWB + (((2 * fk + 1) ^ sw) * 16));
        split_bf16x8(p0, p1, ah[i], al[i]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(base + b_rd + j * 16 * ROWB + ((fk ^ sw) * 16));
        bl[j] = *reinterpret_cast<const bf16x8*>(base + b_rd + j * 16 * ROWB + (((4 + fk) ^ sw) * 16));
      }
#pragma unroll
      for (int i = 0; i < AI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = ((ks * 4 + fk) ^ sw) * 16;
      if constexpr (DT == WSEG_BF16) {
        bf16x8 af[AI], bf[4];
#pragma unroll
        for (int i = 0; i < AI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + a_rd + i * 16 * ROWB + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = *reinterpret_cast<const bf16x8*>(base + b_rd + i * 16 * ROWB + off);
#pragma unroll
        for (int i = 0; i < AI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      } else {
        f32x4 af[AI], bf[4];
#pragma unroll
        for (int i = 0; i < AI; ++i) af[i] = *reinterpret_cast<const f32x4*>(base + a_rd + i * 16 * ROWB + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[i] = *reinterpret_cast<const f32x4*>(base + b_rd + i * 16 * ROWB + off);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < AI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
      }
    }
    }
    __syncthreads();                               // next stage landed, this stage fully read
    cur ^= 1;
  }

  // ---- epilogue: accumulators -> LDS f32 image -> coalesced 8-channel vectors
  float* img = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < AI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wr * (BM / 2) + i * 16 + fk * 4;   // C/D: col = lane&15, row = (lane>>4)*4 + reg
      const int col = wc * 64 + j * 16 + frow;
#pragma unroll
      for (int e = 0; e < 4; ++e) img[(row + e) * EPI_LD + col] = acc[i][j][e];
    }
  __syncthreads();

  float sc[8], sh[8];
  epilogue_coeffs(d, n0, (tid & 15) * 8, sc, sh);
  epilogue_image<DT, EPI, BM, BN, EPI_LD, 256>(d, a.M, img, m0, n0, tid, sc, sh);
}

// ---- 256 x 256 bf16 phase-pipelined variant (large layers) ------------------------------------------------
// The 128^2 kernel above fills a CU at the L2->LDS rate with 64 FLOP per filled byte (a ~1.1 PF ceiling); a
// 256x256 tile doubles that.  Schedule (validated on plain GEMM in csrc/gemm256_probe.hip, in production for
// wgrad): 8 waves as 2(M) x 4(N), wave tile 128 x 64 = 8 x 4 accumulators; LDS = 2 K-tiles x 4 half-tile slots
// {A0, A1, B0, B1}, each [128 rows][128 B] = 16 KiB.  A K-tile (one 128-B slice of one tap) is 4 phases of 16
// MFMAs (one 64 x 32 quadrant of the wave tile); every phase refills ONE slot that all waves have finished
// reading:    p1(u): A0(u+1)   p2(u): A1(u+1)   p3(u): B0(u+2)   p4(u): B1(u+2)
// so LDS-DMA runs 1.5 tiles ahead with two tile buffers; the only DMA wait is ONE counted s_waitcnt vmcnt(4)
// per K-tile (B0/B1(u+2) stay in flight) and one raw s_barrier per phase.  The im2col gather is again only the
// per-lane source address (4 pixel rows per thread, re-derived once per tap); padded taps read the zero page.
// Epilogue: 4 passes of 64 rows through a 65-KiB f32 LDS image, same fused epilogue as the 128^2 kernel.
constexpr int HALF256 = 16384, TILE256 = 4 * HALF256;

//
// NI = 7: a 224-row tile in the same buffers (each wave row owns 112 rows: LDS rows 112..127 of the A half-slots are fed from
// the zero page and never read, phase 2 runs 3 row blocks instead of 4).  For the layers whose 256-row tiles fill the last
// round of 256 CUs badly (424 tiles = 1.66 rounds for the 512-channel 56x56 layers) 486 tiles of 7/8 the work are 12.5 % less
// time per CU; the host picks it when that arithmetic says so.
// DT = WSEG_F32X3 (split-bf16 products on f32 storage): the same pipeline on f32 activation rows (32 channels per 128-B K-tile row,
// split into hi + lo at fragment-read time) and the pre-split weight pack [32 hi | 32 lo]; the two fragment sets af[0] / af[1] and
// b[0] / b[1] that hold the two K halves in bf16 mode hold (hi, lo) here, and a quadrant issues lo.hi + hi.lo + hi.hi.
// The tile body is a device function of (arguments, the workgroup's 128 KiB LDS buffer, block id): `conv_igemm256_kernel` is one workgroup = one
// tile; `conv_bwd_pair_kernel` (below) runs it in the first workgroups of a grid whose other workgroups run the weight-gradient tile body.
template <int EPI, int NI = 8, int DT = WSEG_BF16, bool TAPF = true>
__device__ __forceinline__ void conv_igemm256_tile(const Args& a, char* smem, const int bid) {
  constexpr bool X3 = DT == WSEG_F32X3;
  constexpr int ES = X3 ? 4 : 2, CH = 16 / ES;
  constexpr int RH = NI * 16, BMT = 2 * RH;        // rows per wave row / per tile
  static_assert(NI == 8 || NI == 7, "256- or 224-row tiles");
  const wseg_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(bid, a.nwg);
  const int tm = tile / a.ntn, tn = tile - tm * a.ntn;
  const int m0 = a.row0 + tm * BMT, n0 = tn * 256;
  const int wr = wid >> 2, wc = wid & 3;
  const int frow = lane & 15, fk = lane >> 4, sw = (lane >> 1) & 7;
  WSEG_STAMP(0, 0);
  WSEG_DIAG_LOAD();

  const char* zero = reinterpret_cast<const char*>(g_wseg_zero_page);
  const char* IN = reinterpret_cast<const char*>(d.in);
  const char* Wp = reinterpret_cast<const char*>(d.w);

  // ---- staging: thread -> rows r0 + 64*j (j = 0..3) of the A tile and of the B tile, physical chunk tid&7.
  //      ((row>>1)&7 is the same for all four rows, so one logical chunk per thread.)  Per-row state is kept
  //      small (the accumulators + fragments already take 192 of the 256 registers): a 32-bit first-input-row
  //      (or -1 beyond M) and one packed word {segment | iy0 | ix0}; B rows are a uniform stride apart.
  const int r0 = tid >> 3, pch = tid & 7;
  const int lc = pch ^ ((r0 >> 1) & 7);
  const char* zsrc = zero + pch * 16;
  const char* INl = IN + lc * 16;
  // tap list (4-bit entries): all taps, or — parity-permuted rows, tile inside one class — only the class's valid taps
  unsigned long long tl = 0xFEDCBA9876543210ull;
  int ntaps = a.taps;
  if (a.perm) {
    const int c0 = perm_decode(a, m0).cls, c1 = perm_decode(a, min(m0 + BMT - 1, a.M - 1)).cls;
    if (c0 == c1) {
      const int py = (c0 >> 1) & 1, px = c0 & 1;
      tl = 0ull; ntaps = 0;
      for (int t = 0; t < a.taps; ++t) {
        const int ky = t / d.KW, kx = t - ky * d.KW;
        if ((((py + d.pad - ky * d.dil) | (px + d.pad - kx * d.dil)) & 1) == 0) { tl |= (unsigned long long)t << (4 * ntaps); ++ntaps; }
      }
      if (ntaps == 0) { tl = 0ull; ntaps = 1; }    // (a class without taps: one all-padding tap keeps the pipeline uniform)
    }
  }
  const char* bptr0 = Wp + ((size_t)(n0 + r0) * a.krow + (size_t)lc * CH) * ES;   // the weight pack holds whole 256-row tiles (host-checked: OC % 256 == 0 or w_rows)
  const char* bptr = bptr0 + (size_t)(tl & 15ull) * d.IC * ES;
  int b_ti = 0, b_cc = 0;                          // (perm only) position of the NEXT B tile in the tap list
  const int brs = 64 * a.krow * ES;                // bytes between B rows r0 + 64*j
  auto issue_b = [&](int h, int buf) {
    char* dst = smem + buf * TILE256 + (2 + h) * HALF256 + wid * 1024;
    glds16(bptr + (2 * h) * brs, dst);
    glds16(bptr + (2 * h + 1) * brs, dst + 8192);
  };
  auto advance_b = [&]() {
    bptr += 128;                                   // (the full tap list is contiguous in K: nothing else to do)
    if (a.perm && ++b_cc == a.cpt) { b_cc = 0; ++b_ti; bptr = bptr0 + (size_t)((tl >> (4 * b_ti)) & 15ull) * d.IC * ES; }
  };

  // The weight tiles need no pixel geometry: their LDS-DMA is issued BEFORE the row decode (three integer divisions per row)
  // and the tap set-up, which then run in the shadow of the DMA latency instead of in front of it.
  const int nt = d.in2 != nullptr ? (ntaps - 1) * a.cpt + a.cpt2 : ntaps * a.cpt;
  issue_b(0, 0); issue_b(1, 0); advance_b();
  if (nt > 1) { issue_b(0, 1); issue_b(1, 1); advance_b(); }
  // Per-row gather state, two words.  TAPF (every launch but the strided data gradient): the row's ORIGIN input pixel (tap (0, 0); possibly outside the
  // tensor, never dereferenced then) and a mask of the taps that fall inside the image — bit ky: row iy0 +- ky*dil is in [0, H), bit 8 + kx: the same
  // for the column, bit 16: second row segment.  A tap switch is then a uniform pixel offset per segment, two bit tests and one 64-bit multiply-add per
  // row (the full decode per tap — unpack, four compares, the stride arithmetic — measured ~1900 cycles per switch with every wave of the tile in it at
  // once: 9 % of a 512-channel and 16 % of a 256-channel 3x3 layer's main loop, profiles/r03_conv_tap_setup.txt).  !TAPF (stride-2 data gradient, rows in
  // parity-class order: three launches per step): first input row (-1 beyond M) + packed {segment | iy0 | ix0}, decoded in full at every switch.
  int a_w0[4], a_w1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rho = r0 + 64 * (j & 1);             // LDS row inside half-slot j >> 1 = tile row (j >> 1) * RH + rho
    const int m = m0 + (j >> 1) * RH + rho;
    if constexpr (TAPF) {
      a_w0[j] = 0; a_w1[j] = 0;
      if (rho < RH && m < a.M) {
        const wseg_rowgeo rg = wseg_decode_row(d, m);
        const int sg = d.mode == 0 ? d.dil : -d.dil;
        const int iy0 = d.mode == 0 ? rg.oy * d.stride - d.pad : rg.oy + d.pad, ix0 = d.mode == 0 ? rg.ox * d.stride - d.pad : rg.ox + d.pad;
        unsigned mk = rg.n_glob >= d.N ? 0x10000u : 0u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {              // (k >= KH / KW: a bit no tap tests)
          if ((unsigned)(iy0 + k * sg) < (unsigned)rg.IH) mk |= 1u << k;
          if ((unsigned)(ix0 + k * sg) < (unsigned)rg.IW) mk |= 0x100u << k;
        }
        if (d.KH > 3 || d.KW > 3) {                // (KH, KW <= 8: host-checked; no layer of this network)
          for (int k = 3; k < 8; ++k) {
            if ((unsigned)(iy0 + k * sg) < (unsigned)rg.IH) mk |= 1u << k;
            if ((unsigned)(ix0 + k * sg) < (unsigned)rg.IW) mk |= 0x100u << k;
          }
        }
        a_w0[j] = (int)rg.in_base + iy0 * rg.IW + ix0;
        a_w1[j] = (int)mk;
      }
    } else {
      if (rho < RH && m < a.M) {
        const wseg_rowgeo rg = a.perm ? perm_decode(a, m).g : wseg_decode_row(d, m);
        int iy0, ix0;
        if (d.mode == 0) { iy0 = rg.oy * d.stride - d.pad; ix0 = rg.ox * d.stride - d.pad; }
        else             { iy0 = rg.oy + d.pad;            ix0 = rg.ox + d.pad; }
        a_w0[j] = (int)rg.in_base;
        a_w1[j] = (rg.n_glob >= d.N ? (int)0x80000000 : 0) | ((iy0 + 0x2000) << 16) | (ix0 + 0x2000);
      } else {
        a_w0[j] = -1; a_w1[j] = (0x2000 << 16) | 0x2000;
      }
    }
  }
  const char* aptr[4];
  unsigned a_live = 0;                             // bit j: row j reads real data (pointer advances by 128 B per K-tile)
  int a_tap = 0, a_cc = 0;                         // position of the NEXT A tile to issue
  int a_ky = 0, a_kx = 0;                          // (TAPF) kernel coordinates of tap a_tap
  auto set_tap = [&](int tap) {
    const bool src2 = d.in2 != nullptr && tap == d.KH * d.KW;   // two sources: the extra last "tap" = the output pixel itself in the second input
    const char* INs = src2 ? reinterpret_cast<const char*>(d.in2) + lc * 16 : INl;
    const int lds = src2 ? d.ld_in2 : d.ld_in;
    a_live = 0;
    if constexpr (TAPF) {
      // (32-bit byte offsets: the host gives this instantiation only tensors below 2 GiB; one quarter-rate multiply and ~10 full-rate instructions per row)
      const int tky = src2 ? d.KH / 2 : a_ky, tkx = src2 ? d.KW / 2 : a_kx;          // (taps come in natural order: no division)
      const int sg = d.mode == 0 ? d.dil : -d.dil;
      const int dp1 = sg * (tky * d.IW + tkx), dp2 = sg * (tky * d.IW2 + tkx);      // pixel offset of the tap in segment 1 / 2 (uniform)
      const unsigned tmask = (1u << tky) | (0x100u << tkx);
      const int rowb = lds * ES;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned vm = (unsigned)a_w1[j];
        const bool ok = (vm & tmask) == tmask;
        const int off = (a_w0[j] + ((vm & 0x10000u) ? dp2 : dp1)) * rowb;
        const char* p = INs + (long)off;
        aptr[j] = ok ? p : zsrc;
        a_live |= ok ? 1u << j : 0u;
      }
    } else {
      const int tg = src2 ? (d.KH / 2) * d.KW + d.KW / 2 : tap;    // (its geometry is the centre tap's: same-size convolution, stride 1)
      const int ky = tg / d.KW, kx = tg - ky * d.KW;
      const int (&a_base)[4] = a_w0; const int (&a_yx)[4] = a_w1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int iy0 = ((a_yx[j] >> 16) & 0x7FFF) - 0x2000, ix0 = (a_yx[j] & 0xFFFF) - 0x2000;
      const bool s2 = a_yx[j] < 0;
      const int H = s2 ? d.IH2 : d.IH, W = s2 ? d.IW2 : d.IW;
      int iy, ix; bool ok = a_base[j] >= 0;
      if (d.mode == 0) {
        iy = iy0 + ky * d.dil; ix = ix0 + kx * d.dil;
      } else {
        const int ty = iy0 - ky * d.dil, tx = ix0 - kx * d.dil;
        ok = ok && ty >= 0 && tx >= 0;
        if (d.stride == 1) { iy = ty; ix = tx; }
        else { iy = ty / d.stride; ix = tx / d.stride; ok = ok && (iy * d.stride == ty) && (ix * d.stride == tx); }
      }
      ok = ok && iy >= 0 && iy < H && ix >= 0 && ix < W;
      if (ok) { aptr[j] = INs + (size_t)(a_base[j] + iy * W + ix) * lds * ES; a_live |= 1u << j; }
      else    { aptr[j] = zsrc; }
    }
    }
  };
  auto issue_a = [&](int h, int buf) {             // half h = rows [128h, 128h+128): this thread's rows 2h, 2h+1
    char* dst = smem + buf * TILE256 + h * HALF256 + wid * 1024;
    glds16(aptr[2 * h], dst);
    glds16(aptr[2 * h + 1], dst + 8192);
  };
  auto advance_a = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) aptr[j] += ((a_live >> j) & 1u) << 7;
    if (++a_cc == a.cpt) {
      a_cc = 0;
      if (++a_tap < ntaps) {                       // (two sources: the walk ends after cpt2 steps of source 2)
        if (TAPF) { if (++a_kx == d.KW) { a_kx = 0; ++a_ky; } }
        set_tap((int)((tl >> (4 * a_tap)) & 15ull));
      }
    }
  };
  f32x4 acc[NI][4];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue: tile 0's A halves (B(0), B(1) are already in flight); everything must have landed before the first reads
  set_tap((int)(tl & 15ull));
  issue_a(0, 0); issue_a(1, 0); advance_a();
  WSEG_STAMP(1, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  WSEG_STAMP(2, 0);
#ifndef WSEG_SLOTS
  WSEG_CSTAMP(8, 0);
#endif

  bf16x8 af[2][4], b0[2][2], b1[2][2];             // [ks][tile]: A sub-tile (64 rows), B sub-tiles hb = 0 / 1 (32 cols each)
  auto ldA = [&](const char* aH, int ha) {
    if constexpr (X3) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ha * 4 + i < NI) {
          const char* row = aH + (ha * 64 + i * 16 + frow) * 128;
          const f32x4 p0 = *reinterpret_cast<const f32x4*>(row + (((2 * fk) ^ sw) << 4));
          const f32x4 p1 = *reinterpret_cast<const f32x4*>(row + (((2 * fk + 1) ^ sw) << 4));
          split_bf16x8(p0, p1, af[0][i], af[1][i]);
        }
    } else {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ha * 4 + i < NI)
          af[ks][i] = *reinterpret_cast<const bf16x8*>(aH + (ha * 64 + i * 16 + frow) * 128 + (((ks * 4 + fk) ^ sw) << 4));
    }
  };
  auto ldB = [&](const char* bH, int hb, bf16x8 (&bf)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bf[ks][j] = *reinterpret_cast<const bf16x8*>(bH + ((wc & 1) * 64 + hb * 32 + j * 16 + frow) * 128 + (((ks * 4 + fk) ^ sw) << 4));
  };
#define MFMA_Q(HA, HB, BF)                                                                                   \
  do {                                                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    if constexpr (X3) {                                                                                      \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                        \
          if ((HA) * 4 + i < NI) {                                                                           \
            f32x4& c_ = acc[(HA) * 4 + i][(HB) * 2 + j];                                                     \
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][i], BF[0][j], c_, 0, 0, 0);                   \
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], BF[1][j], c_, 0, 0, 0);                   \
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i], BF[0][j], c_, 0, 0, 0);                   \
          }                                                                                                  \
    } else                                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                        \
          if ((HA) * 4 + i < NI)              /* (unrolled counter: folds at compile time) */                \
            acc[(HA) * 4 + i][(HB) * 2 + j] =                                                                \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], BF[ks][j], acc[(HA) * 4 + i][(HB) * 2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                           \
  } while (0)

  {
    // TWO phases of 32 MFMAs per K-tile (rows 0-63 then 64-127 of the wave tile, both B fragments read in phase 1 and kept), each a read
    // slot (fragment ds_reads, LDS-DMA issue, address work) and an MFMA slot with a raw barrier after each.  PING-PONG: the two waves of every
    // SIMD (wave w and w + 4 = the two M halves) run one slot apart — while one feeds the matrix pipe the other does its reads; every wave executes
    // the same number of barriers (waves 4-7 one extra before the loop, waves 0-3 one after).  Measured alternatives (4 phases of 16 MFMAs,
    // lock-step forms): profiles/HISTORY.md.
    WSEG_SLOT_DECL();
    if (wr == 1) __builtin_amdgcn_s_barrier();
    WSEG_SLOT_BEGIN();
    for (int u = 0; u < nt; ++u) {
      const int b = u & 1;
      const char* aH = smem + b * TILE256 + wr * HALF256;
      const char* bH = smem + b * TILE256 + (2 + (wc >> 1)) * HALF256;
      // (read slot 1 holds 16 fragment reads + the 4 LDS-DMA requests of the next A tile, slot 2 8 reads + the 4 requests of B(u+2) + the counted
      //  wait.  In-kernel slot stamps (profiles/r03_conv_slots.txt) put slot 1 at ~800 cycles against 370 for slot 2 and 570 / 440 for the MFMA slots
      //  beside them — the matrix pipe waits for slot 1.  Re-balancing was measured in round 3 and gains nothing: rows 0-63 of A in slot 1 and rows
      //  64-127 in slot 2, the requests in front of the reads or behind an lgkmcnt(0): 1.29-1.37 us per K-tile against 1.31 — the sum of the two
      //  read slots stays above the sum of the MFMA slots whatever their split; profiles/HISTORY.md.)
      ldA(aH, 0); ldB(bH, 0, b0); ldB(bH, 1, b1);
      if (u + 1 < nt && WSEG_DIAG_A_OK(u)) { issue_a(0, b ^ 1); issue_a(1, b ^ 1); }
      WSEG_SLOT(0);
      __builtin_amdgcn_s_barrier();
      WSEG_SLOT(1);
      // (the A pointers move on — and, at the end of a tap, are set up for the next one — inside an MFMA slot whose partner slot is longer: the early group's
      //  slot 1 runs beside the late group's read slot 1, the late group's slot 2 beside the early group's next read slot 1; in the read slot, where the
      //  pointers were advanced until round 3, every cycle of the tap switch was a cycle of the matrix pipe waiting)
      if (wr == 0 && u + 1 < nt && WSEG_DIAG_A_MOVES()) advance_a();
      MFMA_Q(0, 0, b0);
      MFMA_Q(0, 1, b1);
      WSEG_SLOT(2);
      __builtin_amdgcn_s_barrier();
      WSEG_SLOT(3);
      ldA(aH, 1);
      if (u + 2 < nt) { if (WSEG_DIAG_B_OK()) { issue_b(0, b); issue_b(1, b); } if (WSEG_DIAG_B_MOVES()) advance_b(); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WSEG_SLOT(4);
      __builtin_amdgcn_s_barrier();
      WSEG_SLOT(5);
      if (wr == 1 && u + 1 < nt && WSEG_DIAG_A_MOVES()) advance_a();
      MFMA_Q(1, 1, b1);
      MFMA_Q(1, 0, b0);
      WSEG_SLOT(6);
      __builtin_amdgcn_s_barrier();
      WSEG_SLOT(7);
    }
    WSEG_SLOT_FLUSH();
    if (wr == 0) __builtin_amdgcn_s_barrier();
  }
#undef MFMA_Q

  // ---- epilogue, wave-local (see wave_local_epilogue)
  WSEG_STAMP(3, 0); WSEG_STAMP(5, 4);
#ifndef WSEG_SLOTS
  WSEG_CSTAMP(9, 0);
#endif
  __syncthreads();                                 // every wave is done with the pipeline buffers
  wave_local_epilogue<EPI, NI, DT>(a, smem, wid, lane, m0 + wr * RH, wc * 64, n0, acc);
#ifdef WSEG_PROBES
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the stamp means: this wave's stores have left)
  WSEG_STAMP(4, 0); WSEG_STAMP(6, 4);
  if (threadIdx.x == 0 && bid < 4096) g_wseg_stamps[bid * 24 + 7] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID bits 0..3
#endif
}

template <int EPI, int NI = 8, int DT = WSEG_BF16, bool TAPF = true>
__global__ __launch_bounds__(512, 2) void conv_igemm256_kernel(const Args a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE256];
  conv_igemm256_tile<EPI, NI, DT, TAPF>(a, smem, blockIdx.x);
}





// ---- 512(M) x 128(N) bf16 phase-pipelined variant for OC = 128 layers (the frozen 224x224 prefix) -----------------------
// Every kernel with a 128-wide column tile so far gave those layers ~680 TF/s whatever its pipeline depth, operand source
// or address work; what they share is 32 MFMAs per wave per K-tile — half of the 256^2 kernel's — against the same fixed
// per-K-tile costs (barriers, fragment-read latency, DMA issue).  This variant keeps the 256^2 kernel's wave tile (128 x 64,
// 64 MFMAs per K-tile, 2-phase ping-pong) by stacking FOUR 128-row A half-tiles: 8 waves as 4(M) x 2(N), LDS = 2 K-tiles x
// {A0..A3, B} x 16 KiB = 160 KiB (all of it).  Gather addresses: row pointer + per-tap scalar offset, branch-free validity
// (fast taps only: forward, or stride-1 data gradient).
constexpr int TILE5 = 5 * HALF256;

template <int EPI>
__global__ __launch_bounds__(512, 2) void conv_igemm512x128_kernel(const Args a) {
  constexpr int ES = 2, CH = 8;
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE5];
  const wseg_conv_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, a.nwg);
  const int tm = tile / a.ntn, tn = tile - tm * a.ntn;
  const int m0 = a.row0 + tm * 512, n0 = tn * 128;
  const int wr = wid >> 1, wc = wid & 1;
  const int frow = lane & 15, fk = lane >> 4, sw = (lane >> 1) & 7;

  const char* zero = reinterpret_cast<const char*>(g_wseg_zero_page);
  const char* IN = reinterpret_cast<const char*>(d.in);
  const char* Wp = reinterpret_cast<const char*>(d.w);

  // staging: thread -> rows r0 + 64*j (j = 0..7) of the A tile (slot j>>1), rows r0, r0 + 64 of the B tile
  const int r0 = tid >> 3, pch = tid & 7;
  const int lc = pch ^ ((r0 >> 1) & 7);
  const char* zsrc = zero + pch * 16;
  const char* INl = IN + lc * 16;
  const char* rowptr[8];                           // pixel (iy0, ix0) of the row's image (possibly outside the tensor)
  int a_yx[8];                                     // packed (iy0 + 0x2000) << 16 | (ix0 + 0x2000); rows beyond M: never in bounds
  unsigned segmask = 0;                            // bit j: row j lies in the second row segment
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int m = m0 + r0 + 64 * j;
    rowptr[j] = zsrc; a_yx[j] = 0;
    if (m < a.M) {
      const wseg_rowgeo rg = wseg_decode_row(d, m);
      int iy0, ix0;
      if (d.mode == 0) { iy0 = rg.oy * d.stride - d.pad; ix0 = rg.ox * d.stride - d.pad; }
      else             { iy0 = rg.oy + d.pad;            ix0 = rg.ox + d.pad; }
      a_yx[j] = ((iy0 + 0x2000) << 16) | (ix0 + 0x2000);
      segmask |= rg.n_glob >= d.N ? 1u << j : 0u;
      rowptr[j] = INl + ((long)rg.in_base + (long)iy0 * rg.IW + ix0) * d.ld_in * ES;
    }
  }
  unsigned a_ok = 0;                               // bit j: row j is inside the image for the current tap
  long koff1 = 0, koff2 = 0;                       // tap offset + K offset inside the tap (bytes), per row segment
  const char* bptr = Wp + ((size_t)(n0 + r0) * a.taps * d.IC + (size_t)lc * CH) * ES;   // OC % 128 == 0 (host-checked)
  const int brs = 64 * a.taps * d.IC * ES;
  int t_ky = 0, t_kx = 0;                          // kernel coordinates of the current tap (taps come in natural order: no division)
  auto set_tap = [&]() {
    const int ky = t_ky, kx = t_kx;
    const int dyt = d.mode == 0 ? ky * d.dil : -ky * d.dil, dxt = d.mode == 0 ? kx * d.dil : -kx * d.dil;   // (uniform)
    koff1 = ((long)dyt * d.IW + dxt) * d.ld_in * ES; koff2 = ((long)dyt * d.IW2 + dxt) * d.ld_in * ES;
    a_ok = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool s2 = (segmask >> j) & 1u;
      const unsigned H = s2 ? d.IH2 : d.IH, W = s2 ? d.IW2 : d.IW;
      const int iy = (a_yx[j] >> 16) - 0x2000 + dyt, ix = (a_yx[j] & 0xFFFF) - 0x2000 + dxt;
      a_ok |= ((unsigned)iy < H && (unsigned)ix < W) ? 1u << j : 0u;
    }
  };
  int a_tap = 0, a_cc = 0;
  auto issue_a = [&](int buf) {                    // the NEXT A tile (8 pieces)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const char* p = rowptr[j] + (((segmask >> j) & 1u) ? koff2 : koff1);
      p = ((a_ok >> j) & 1u) ? p : zsrc;
      glds16(p, smem + buf * TILE5 + (j >> 1) * HALF256 + (j & 1) * 8192 + wid * 1024);
    }
  };
  auto advance_a = [&]() {                         // (called in an MFMA slot whose partner slot is longer — see conv_igemm256_tile: these layers change tap every 1-2 K-tiles)
    koff1 += 128; koff2 += 128;
    if (++a_cc == a.cpt) {
      a_cc = 0;
      if (++a_tap < a.taps) { if (++t_kx == d.KW) { t_kx = 0; ++t_ky; } set_tap(); }
    }
  };
  auto issue_b = [&](int buf) {                    // the NEXT B tile (2 pieces), then advance
    char* dst = smem + buf * TILE5 + 4 * HALF256 + wid * 1024;
    glds16(bptr, dst);
    glds16(bptr + brs, dst + 8192);
    bptr += 128;
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = a.taps * a.cpt;
  set_tap();
  issue_a(0); advance_a(); issue_b(0);
  if (nt > 1) { issue_b(1); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 af[2][4], bf[2][4];
  auto ldA = [&](const char* aH, int ha) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[ks][i] = *reinterpret_cast<const bf16x8*>(aH + (ha * 64 + i * 16 + frow) * 128 + (((ks * 4 + fk) ^ sw) << 4));
  };
#define MFMA_H5(HA)                                                                                          \
  do {                                                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
          acc[(HA) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bf[ks][j], acc[(HA) * 4 + i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                           \
  } while (0)

  // 2-phase ping-pong (see conv_igemm256_kernel): waves 4-7 (wr >= 2) run one slot behind waves 0-3
  const int grp = wr >> 1;
  if (grp == 1) __builtin_amdgcn_s_barrier();
  for (int u = 0; u < nt; ++u) {
    const int b = u & 1;
    const char* aH = smem + b * TILE5 + wr * HALF256;
    const char* bH = smem + b * TILE5 + 4 * HALF256;
    ldA(aH, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bf[ks][j] = *reinterpret_cast<const bf16x8*>(bH + (wc * 64 + j * 16 + frow) * 128 + (((ks * 4 + fk) ^ sw) << 4));
    if (u + 1 < nt) issue_a(b ^ 1);
    __builtin_amdgcn_s_barrier();
    if (grp == 0 && u + 1 < nt) advance_a();
    MFMA_H5(0);
    __builtin_amdgcn_s_barrier();
    ldA(aH, 1);
    if (u + 2 < nt) { issue_b(b); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1 && u + 1 < nt) advance_a();
    MFMA_H5(1);
    __builtin_amdgcn_s_barrier();
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
#undef MFMA_H5

  __syncthreads();                                 // every wave is done with the pipeline buffers
  wave_local_epilogue<EPI, 8>(a, smem, wid, lane, m0 + wr * 128, wc * 64, n0, acc);
}

// ---- A layer's data gradient and weight gradient as ONE grid.  Both need only dY; as two kernels on a stream each pays its own partly filled
// last round and its own synchronised epilogue burst (all CUs store at once while the matrix pipes idle).  Here the first `nd_pad` workgroups run the
// 256-tile dgrad body and the others the 256 x 256 weight-gradient body (same 512-thread / 128 KiB shape): the grid order back-fills the dgrad tail with
// wgrad tiles and the two kinds of tile end at different times.  nd_pad is a multiple of 8, so block id % 8 (the XCD) is the same for both bodies' maps.
template <int EPI, int NI, int UNIT>
__global__ __launch_bounds__(512, 2) void conv_bwd_pair_kernel(const Args ad, const wseg_wg::Args aw, const int nd_pad) {
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE256];
  static_assert(2 * TILE256 == 2 * 4 * 16384, "both tile bodies use the same 128 KiB");
  const int b = blockIdx.x;
  if (b < nd_pad) {
    if (b < ad.nwg) conv_igemm256_tile<EPI, NI, WSEG_BF16>(ad, smem, b);
  } else {
    wseg_wg::conv_wgrad_pipe_tile<UNIT>(aw, smem, b - nd_pad);
  }
}

}  // namespace

static int conv_validate(const wseg_conv_desc* d) {
  WSEG_CHECK(d && d->in && d->w && (d->out || d->out2), "conv_igemm: null pointer");
  WSEG_CHECK(d->dtype == WSEG_F32 || d->dtype == WSEG_BF16 || d->dtype == WSEG_F32X3, "conv_igemm: bad dtype %d", d->dtype);
  const int es = d->dtype == WSEG_BF16 ? 2 : 4;
  WSEG_CHECK((d->IC * es) % ROWB == 0, "conv_igemm: IC=%d must be a multiple of %d", d->IC, ROWB / es);
  WSEG_CHECK(d->OC % 8 == 0 && d->ld_in % 8 == 0, "conv_igemm: OC=%d / ld_in=%d must be multiples of 8", d->OC, d->ld_in);
  WSEG_CHECK(d->N > 0 && d->OH > 0 && d->OW > 0 && d->IH > 0 && d->IW > 0, "conv_igemm: empty shape");
  WSEG_CHECK(d->stride >= 1 && d->dil >= 1 && d->KH >= 1 && d->KW >= 1, "conv_igemm: bad geometry");
  WSEG_CHECK(d->mode == 0 || d->mode == 1, "conv_igemm: bad mode");
  WSEG_CHECK(d->epi >= 0 && d->epi <= 2, "conv_igemm: bad epilogue");
  WSEG_CHECK(d->ld_in >= d->IC, "conv_igemm: ld_in < IC");
  if (d->out) WSEG_CHECK(d->ld_out >= d->OC && d->ld_out % 8 == 0, "conv_igemm: bad ld_out");
  if (d->out2) WSEG_CHECK(d->ld_out2 >= d->OC && d->ld_out2 % 8 == 0 && d->epi == 0, "conv_igemm: bad out2");
  if (d->r_pre) WSEG_CHECK(d->ld_rpre % 8 == 0, "conv_igemm: bad ld_rpre");
  if (d->r_post) WSEG_CHECK(d->ld_rpost % 8 == 0, "conv_igemm: bad ld_rpost");
  if (d->mask) WSEG_CHECK(d->ld_mask % 8 == 0, "conv_igemm: bad ld_mask");
  const long M = (long)d->N * d->OH * d->OW + (long)d->N * d->OH2 * d->OW2;
  WSEG_CHECK(d->OH2 >= 0 && (d->OH2 == 0 || (d->OW2 > 0 && d->IH2 > 0 && d->IW2 > 0)), "conv_igemm: bad second segment");
  WSEG_CHECK(d->w_rows == 0 || d->w_rows >= d->OC, "conv_igemm: w_rows=%d < OC=%d", d->w_rows, d->OC);
  WSEG_CHECK(M < (1L << 31) && (long)d->N * d->IH * d->IW * d->ld_in < (1L << 40), "conv_igemm: tensor too large");
  return 0;
}

// ---- host side ------------------------------------------------------------------------------------------------------------
// Tile choice by CU time in units of (32 rows x 256 columns x K) at the 256-tile kernel's rate: a round of NI-block tiles costs NI, a round of the
// 128^2 kernel (two resident workgroups of 2 units each, 0.75 of that rate) 5.33.  Fitted on the training AND the inference geometries
// (scripts/bench_conv_infer.py).
static bool conv_cost_prefers_256(long M, int OC) {
  const long t256 = ((M + 255) / 256) * ((OC + 255) / 256);
  const long rounds = (t256 + 255) / 256;
  const long t224 = ((M + 223) / 224) * ((OC + 255) / 256), t128 = ((M + 127) / 128) * ((OC + 127) / 128);
  const double c_big = std::min((double)rounds * 8.0, (double)((t224 + 255) / 256) * 7.0), c_128 = (double)((t128 + 511) / 512) * 5.33;
  return c_big <= c_128;
}
// 224-row tiles (NI = 7) when they need less CU time than 256-row tiles: rounds(tiles) x rows per tile
static bool conv_rounds_prefer_224(long M, int ntn256) {
  const long t8 = ((M + 255) / 256) * ntn256, t7 = ((M + 223) / 224) * ntn256;
  return ((t7 + 255) / 256) * 7 < ((t8 + 255) / 256) * 8;
}

static long conv_rows(const wseg_conv_desc* d) { return (long)d->N * d->OH * d->OW + (long)d->N * d->OH2 * d->OW2; }

// The common part of every launch's arguments (K walk, two-source form).  < 0: error.
static int conv_fill_args(const wseg_conv_desc* d, Args& a) {
  const int es = d->dtype == WSEG_BF16 ? 2 : 4;
  a.d = *d;
  a.perm = 0; a.Q1 = a.Q2 = 0;
  a.M = (int)conv_rows(d);
  a.row0 = 0;
  a.taps = d->KH * d->KW;
  if (d->in2) {                                    // two sources: the second one is an extra last "tap" of w = [OC][KH*KW*IC + IC2]
    WSEG_CHECK(d->KH == d->KW && (d->KH & 1) && d->KH * d->KW < 15 && d->stride == 1 && d->pad == d->dil * (d->KH / 2) &&
               d->dtype == WSEG_BF16 && d->OC % 256 == 0 && d->ld_in2 % 8 == 0 && (d->bm_hint == 0 || d->bm_hint == 256 || d->bm_hint == 224),
               "conv_igemm: the two-source form is a same-size stride-1 bf16 convolution with OC %% 256 == 0 on the 256-tile kernel");
    a.taps = d->KH * d->KW + 1;
  }
  a.cpt = d->IC * es / ROWB;
  const int ic2 = d->in2 ? (d->IC2 > 0 ? d->IC2 : d->IC) : 0;
  WSEG_CHECK(!d->in2 || ((ic2 * es) % ROWB == 0 && d->ld_in2 >= ic2), "conv_igemm: IC2=%d must be a multiple of %d and <= ld_in2", ic2, ROWB / es);
  a.cpt2 = ic2 * es / ROWB;
  a.krow = d->in2 ? d->KH * d->KW * d->IC + ic2 : a.taps * d->IC;
  a.ntn = (d->OC + BN - 1) / BN;
  return 0;
}

// Does this launch run on the 256-tile kernel, and with which tile height?  ONE function for wseg_conv_igemm and for wseg_conv_bwd_pair (the paired
// grid must choose exactly what the stand-alone launch would).  On return (true): a.ntn / a.nwg / a.perm / a.Q* are set for the 256-tile kernel.
static bool conv_plan_256(const wseg_conv_desc* d, Args& a, bool& ni7) {
  const long M = a.M;
  ni7 = false;
  // (OC % 256 != 0: only with a weight pack zero-padded to whole 256-row tiles, wseg_conv_desc.w_rows — the epilogue masks the columns >= OC)
  const bool oc_ok = d->OC % 256 == 0 || (d->in2 == nullptr && d->w_rows >= ((d->OC + 255) / 256) * 256);
  if (!((d->dtype == WSEG_BF16 || (d->dtype == WSEG_F32X3 && d->in2 == nullptr)) && oc_ok && d->bm_hint != 64 && d->bm_hint != 128 &&
        d->bm_hint != 259 && d->bm_hint >= 0 && d->KH <= 8 && d->KW <= 8 && a.taps <= 16))   // (tap list: 16 four-bit entries; tap masks: 8 + 8 bits)
    return false;
  static const int auto256 = getenv("WSEG_CONV256") ? atoi(getenv("WSEG_CONV256")) : 1;   // (0: A/B switch — 128-tile kernel everywhere)
  if (!(d->bm_hint == 256 || d->bm_hint == 224 || d->in2 != nullptr || (auto256 && conv_cost_prefers_256(M, d->OC)))) return false;
  if (!(d->IH <= 16384 && d->IW <= 16384 && d->OH <= 16384 && d->OW <= 16384 && d->pad <= 4096 &&
        (long)d->N * d->IH * d->IW + (long)d->N * d->IH2 * d->IW2 < (1L << 31))) {
    wseg_set_error("conv_igemm: shape too large for the 256-tile kernel");
    a.nwg = -1;
    return true;
  }
  static const int perm_ok = getenv("WSEG_CONV_PERM") ? atoi(getenv("WSEG_CONV_PERM")) : 1;
  if (perm_ok && d->mode == 1 && d->stride == 2 && d->OH % 2 == 0 && d->OW % 2 == 0 && d->OH2 % 2 == 0 && d->OW2 % 2 == 0 && a.taps <= 9 &&
      (a.taps > 1 || d->bm_hint == 256)) {   // (1x1: nothing to skip in the K loop, the per-vector row mapping only costs — measured)
    a.perm = 1;
    a.Q1 = d->N * (d->OH / 2) * (d->OW / 2);
    a.Q2 = d->N * (d->OH2 / 2) * (d->OW2 / 2);
  }
  a.ntn = (d->OC + 255) / 256;
  static const int auto224 = getenv("WSEG_CONV224") ? atoi(getenv("WSEG_CONV224")) : 1;   // (0: A/B switch; 2: always)
  ni7 = d->bm_hint != 256 && (d->bm_hint == 224 || (auto224 && d->bm_hint == 0 && (auto224 == 2 || conv_rounds_prefer_224(M, a.ntn))));
  const int bmt = ni7 ? 224 : 256;
  a.nwg = (int)(((M + bmt - 1) / bmt) * a.ntn);
  return true;
}

extern "C" int wseg_conv_igemm(const wseg_conv_desc* d, void* stream) {
  if (int rc = conv_validate(d)) return rc;
  const long M = conv_rows(d);
  Args a;
  if (int rc = conv_fill_args(d, a)) return rc;
  // few output pixels (view 2, 16x16 maps): 64-row tiles double the workgroup count
  const bool small = d->bm_hint == 64 || (d->bm_hint != 128 && ((M + 127) / 128) * a.ntn < 384 && M > 64);
  hipStream_t s = (hipStream_t)stream;
  WSEG_CHECK(d->out || d->epi == 0, "conv_igemm: epilogue %d needs `out`", d->epi);
  WSEG_CHECK(d->bm_hint != 257 && d->bm_hint != 258 && d->bm_hint >= 0, "conv_igemm: bm_hint %d was a development hook and no longer exists", d->bm_hint);
#define WSEG_LAUNCH_CONV1(DT_, EPI_, BM_) hipLaunchKernelGGL((conv_igemm_kernel<DT_, EPI_, BM_>), dim3(a.nwg), dim3(256), 0, s, a)
#define WSEG_LAUNCH_CONV(BM_)                                                                                   \
  do {                                                                                                          \
    if (d->dtype == WSEG_BF16) {                                                                                \
      if (d->epi == 0) WSEG_LAUNCH_CONV1(WSEG_BF16, 0, BM_); else if (d->epi == 1) WSEG_LAUNCH_CONV1(WSEG_BF16, 1, BM_); else WSEG_LAUNCH_CONV1(WSEG_BF16, 2, BM_); \
    } else if (d->dtype == WSEG_F32X3) {                                                                        \
      if (d->epi == 0) WSEG_LAUNCH_CONV1(WSEG_F32X3, 0, BM_); else if (d->epi == 1) WSEG_LAUNCH_CONV1(WSEG_F32X3, 1, BM_); else WSEG_LAUNCH_CONV1(WSEG_F32X3, 2, BM_); \
    } else {                                                                                                    \
      if (d->epi == 0) WSEG_LAUNCH_CONV1(WSEG_F32, 0, BM_); else if (d->epi == 1) WSEG_LAUNCH_CONV1(WSEG_F32, 1, BM_); else WSEG_LAUNCH_CONV1(WSEG_F32, 2, BM_); \
    }                                                                                                           \
  } while (0)
  // 256 x 256 (or 224 x 256) phase-pipelined tiles: bf16 / split-bf16, OC % 256 == 0, chosen by the CU-time model
  bool ni7 = false;
  const bool big = conv_plan_256(d, a, ni7);
  if (big && a.nwg < 0) return -1;
  // 512 x 128 phase-pipelined tiles for OC = 128 layers with many pixels (259 forces it); fast taps only
  static const int auto512 = getenv("WSEG_CONV512") ? atoi(getenv("WSEG_CONV512")) : 1;   // (0: A/B switch; measured 649 -> 766 TF/s on 128->128 3x3 224^2)
  const bool tall = !big && d->dtype == WSEG_BF16 && d->OC % 128 == 0 && (d->mode == 0 || d->stride == 1) &&
                    (d->bm_hint == 259 || (auto512 && d->bm_hint == 0 && d->OC == 128 && (M + 511) / 512 >= 512));
  if (tall) {
    WSEG_CHECK(d->IH <= 8000 && d->IW <= 8000 && d->OH <= 8000 && d->OW <= 8000 && d->pad <= 4096 && d->KH * d->dil <= 4096,
               "conv_igemm: shape too large for the 512x128-tile kernel");
    a.ntn = d->OC / 128;
    a.nwg = (int)(((M + 511) / 512) * a.ntn);
    if (d->epi == 0) hipLaunchKernelGGL(conv_igemm512x128_kernel<0>, dim3(a.nwg), dim3(512), 0, s, a);
    else if (d->epi == 1) hipLaunchKernelGGL(conv_igemm512x128_kernel<1>, dim3(a.nwg), dim3(512), 0, s, a);
    else hipLaunchKernelGGL(conv_igemm512x128_kernel<2>, dim3(a.nwg), dim3(512), 0, s, a);
    WSEG_LAUNCH_CHECK();
    return 0;
  }
  if (big) {
#define WSEG_LAUNCH_256(NI_, DT_, TF_)                                                                                         \
  do {                                                                                                                         \
    if (d->epi == 0) hipLaunchKernelGGL((conv_igemm256_kernel<0, NI_, DT_, TF_>), dim3(a.nwg), dim3(512), 0, s, a);             \
    else if (d->epi == 1) hipLaunchKernelGGL((conv_igemm256_kernel<1, NI_, DT_, TF_>), dim3(a.nwg), dim3(512), 0, s, a);        \
    else hipLaunchKernelGGL((conv_igemm256_kernel<2, NI_, DT_, TF_>), dim3(a.nwg), dim3(512), 0, s, a);                         \
  } while (0)
    const long es_ = d->dtype == WSEG_BF16 ? 2 : 4;
    const long in_bytes = ((long)d->N * d->IH * d->IW + (long)d->N * d->IH2 * d->IW2) * std::max(d->ld_in, d->in2 ? d->ld_in2 : 0) * es_;
    const bool tapf = !(d->mode == 1 && d->stride != 1) && in_bytes < (1L << 31) - (1L << 24);   // (else: the full per-tap decode with 64-bit addresses: conv_igemm256_tile)
    if (d->dtype == WSEG_F32X3) {
      if (tapf) { if (ni7) WSEG_LAUNCH_256(7, WSEG_F32X3, true); else WSEG_LAUNCH_256(8, WSEG_F32X3, true); }
      else      { if (ni7) WSEG_LAUNCH_256(7, WSEG_F32X3, false); else WSEG_LAUNCH_256(8, WSEG_F32X3, false); }
    } else {
      if (tapf) { if (ni7) WSEG_LAUNCH_256(7, WSEG_BF16, true); else WSEG_LAUNCH_256(8, WSEG_BF16, true); }
      else      { if (ni7) WSEG_LAUNCH_256(7, WSEG_BF16, false); else WSEG_LAUNCH_256(8, WSEG_BF16, false); }
    }
#undef WSEG_LAUNCH_256
  } else if (small) {
    a.nwg = (int)(((M + 63) / 64) * a.ntn);
    WSEG_LAUNCH_CONV(64);
  } else {
    // (A split into full rounds of 128-row tiles + a short launch of 64-row tiles for the remainder was measured:
    //  no gain — workgroups are back-filled as they finish, rounds are not discrete — so one launch it is.)
    a.nwg = (int)(((M + 127) / 128) * a.ntn);
    WSEG_LAUNCH_CONV(128);
  }
#undef WSEG_LAUNCH_CONV1
#undef WSEG_LAUNCH_CONV
  WSEG_LAUNCH_CHECK();
  return 0;
}

// One launch for `dg` (a stride-1 bf16 data gradient on the 256-tile kernel) and `wg` (a bf16 weight gradient on the 256 x 256 phase-pipelined
// kernel) when both qualify; otherwise the two ordinary launches, in that order.  Results are those of the separate launches.
// 1: the pair qualifies for the joint grid (a / ni7 / pl then hold both plans), 0: two launches, < 0: error
static int conv_bwd_pair_plan(const wseg_conv_desc* dg, const wseg_wgrad_desc* wg, Args& a, bool& ni7, wseg_wg::Plan& pl) {
  WSEG_CHECK(dg && wg, "conv_bwd_pair: null descriptor");
  static const int pair_ok = getenv("WSEG_BWD_PAIR") ? atoi(getenv("WSEG_BWD_PAIR")) : 1;           // (0: A/B switch — two launches)
  const bool cand = pair_ok && dg->dtype == WSEG_BF16 && wg->dtype == WSEG_BF16 && dg->mode == 1 && dg->stride == 1 && dg->bm_hint == 0 &&
                    dg->out != nullptr && dg->out2 == nullptr && dg->epi >= 0 && dg->epi <= 2;
  if (!cand) return 0;
  if (((long)dg->N * dg->IH * dg->IW + (long)dg->N * dg->IH2 * dg->IW2) * std::max(dg->ld_in, dg->in2 ? dg->ld_in2 : 0) * 2L >= (1L << 31) - (1L << 24))
    return 0;                                      // (the joint grid carries the 32-bit tap arithmetic only)
  if (conv_validate(dg)) return 0;                 // (the fall-back launch reports the error)
  {
    // the two-source form errors out of conv_fill_args when it does not qualify: test its conditions first, quietly
    const int ic2 = dg->in2 ? (dg->IC2 > 0 ? dg->IC2 : dg->IC) : 0;
    if (dg->in2 && !(dg->KH == dg->KW && (dg->KH & 1) && dg->KH * dg->KW < 15 && dg->pad == dg->dil * (dg->KH / 2) && dg->OC % 256 == 0 &&
                     dg->ld_in2 % 8 == 0 && ((ic2 * 2) % ROWB) == 0 && dg->ld_in2 >= ic2))
      return 0;
  }
  if (conv_fill_args(dg, a)) return 0;
  if (!conv_plan_256(dg, a, ni7) || a.nwg < 0 || a.perm) return 0;
  if (int rc = wseg_wg::wgrad_plan(wg, pl)) return rc;
  return pl.kind == 0 ? 1 : 0;
}
extern "C" int wseg_conv_bwd_pair_fuses(const wseg_conv_desc* dg, const wseg_wgrad_desc* wg) {
  wseg_wg::Plan pl; Args a; bool ni7;
  return conv_bwd_pair_plan(dg, wg, a, ni7, pl);
}

extern "C" int wseg_conv_bwd_pair(const wseg_conv_desc* dg, const wseg_wgrad_desc* wg, void* stream) {
  wseg_wg::Plan pl; Args a; bool ni7 = false;
  const int fuse = conv_bwd_pair_plan(dg, wg, a, ni7, pl);
  if (fuse < 0) return fuse;
  if (fuse != 1) {
    if (int rc = wseg_conv_igemm(dg, stream)) return rc;
    return wseg_conv_wgrad(wg, stream);
  }
  const int nd_pad = (a.nwg + 7) & ~7;
  const dim3 grid((unsigned)(nd_pad + pl.a.nwg));
  hipStream_t s = (hipStream_t)stream;
#define WSEG_LAUNCH_PAIR(EPI_, NI_)                                                                                              \
  do {                                                                                                                           \
    if (pl.unit) hipLaunchKernelGGL((conv_bwd_pair_kernel<EPI_, NI_, 1>), grid, dim3(512), 0, s, a, pl.a, nd_pad);               \
    else hipLaunchKernelGGL((conv_bwd_pair_kernel<EPI_, NI_, 0>), grid, dim3(512), 0, s, a, pl.a, nd_pad);                       \
  } while (0)
  if (ni7) { if (dg->epi == 0) WSEG_LAUNCH_PAIR(0, 7); else if (dg->epi == 1) WSEG_LAUNCH_PAIR(1, 7); else WSEG_LAUNCH_PAIR(2, 7); }
  else { if (dg->epi == 0) WSEG_LAUNCH_PAIR(0, 8); else if (dg->epi == 1) WSEG_LAUNCH_PAIR(1, 8); else WSEG_LAUNCH_PAIR(2, 8); }
#undef WSEG_LAUNCH_PAIR
  WSEG_LAUNCH_CHECK();
  return 0;
}
