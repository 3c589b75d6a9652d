// gemm256_probe.hip — development probe for the 256x256 phase-pipelined MFMA core (C = A * B^T, both
// operands K-contiguous bf16, f32 out).  Not on the product path: it exists to validate and time the
// pipeline that csrc/conv_igemm.hip's large-layer variant uses.
//
// Geometry: 256x256 tile, BK = 64 (128-B LDS rows), 8 waves as 2(M) x 4(N), wave tile 128 x 64 =
// 8 x 4 MFMA 16x16x32 accumulators.  LDS = 2 K-tiles x 4 half-tile slots (A0,A1,B0,B1; 128 rows x 128 B
// = 16 KiB each) = 128 KiB.  A K-tile is computed in 4 phases (one 64x32 quadrant of the wave tile x K=64
// = 16 MFMAs each); every phase refills ONE half-tile slot that all waves have finished reading:
//     p1(u): A0(u+1)   p2(u): A1(u+1)   p3(u): B0(u+2)   p4(u): B1(u+2)
// so LDS-DMA runs 1.5 tiles ahead with only two tile buffers; the only wait is a counted
// s_waitcnt vmcnt(4) in p4 (B0/B1(u+2) may stay in flight).  Fragment reads per phase: 12, 4, 8, 0.
#ifdef WSEG_PROBES   // development probe: not part of the product library (build with WSEG_PROBES=1 bash build.sh)
#include "common.h"

namespace {

constexpr int HALF = 16384;
constexpr int TILE = 4 * HALF;

template <int BAR1>
__global__ __launch_bounds__(512, 2) void gemm256_probe_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                               float* __restrict__ C, int M, int N, int K, int nwg, int ntn) {
  __shared__ __attribute__((aligned(16))) char smem[2 * TILE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int wr = wid >> 2, wc = wid & 3;
  const int frow = lane & 15, fk = lane >> 4, sw = (lane >> 1) & 7;

  // staging: thread -> rows r0 = tid>>3 and r0+64 of every half-tile, physical chunk tid&7
  const int r0 = tid >> 3, pch = tid & 7;
  const int lc0 = pch ^ ((r0 >> 1) & 7), lc1 = pch ^ (((r0 + 64) >> 1) & 7);
  const char* src[4][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    src[h][0] = (const char*)(A + ((size_t)tm * 256 + h * 128 + r0) * K + lc0 * 8);
    src[h][1] = (const char*)(A + ((size_t)tm * 256 + h * 128 + r0 + 64) * K + lc1 * 8);
    src[2 + h][0] = (const char*)(B + ((size_t)tn * 256 + h * 128 + r0) * K + lc0 * 8);
    src[2 + h][1] = (const char*)(B + ((size_t)tn * 256 + h * 128 + r0 + 64) * K + lc1 * 8);
  }
  auto issue = [&](int which, int buf, int kt) {
    char* dst = smem + buf * TILE + which * HALF + wid * 1024;
    glds16(src[which][0] + (size_t)kt * 128, dst);
    glds16(src[which][1] + (size_t)kt * 128, dst + 8192);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = K / 64;
  // prologue: tile 0 entirely + the B halves of tile 1 (what p3/p4 of a "tile -1" would have issued)
  issue(0, 0, 0); issue(1, 0, 0); issue(2, 0, 0); issue(3, 0, 0);
  if (nt > 1) { issue(2, 1, 1); issue(3, 1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 af[2][4], b0[2][2], b1[2][2];              // [ks][tile]: A sub-tile (64 rows), B sub-tiles h_b = 0 / 1 (32 cols each)
  auto ldA = [&](const char* aH, int ha) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[ks][i] = *reinterpret_cast<const bf16x8*>(aH + (ha * 64 + i * 16 + frow) * 128 + (((ks * 4 + fk) ^ sw) << 4));
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
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                        \
          acc[(HA) * 4 + i][(HB) * 2 + j] =                                                                  \
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], BF[ks][j], acc[(HA) * 4 + i][(HB) * 2 + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                           \
  } while (0)

  for (int u = 0; u < nt; ++u) {
    const int b = u & 1;
    const char* aH = smem + b * TILE + wr * HALF;
    const char* bH = smem + b * TILE + (2 + (wc >> 1)) * HALF;
    // ---- p1: quadrant (0,0)
    ldA(aH, 0); ldB(bH, 0, b0);
    if (u + 1 < nt) issue(0, b ^ 1, u + 1);
    if (BAR1) __builtin_amdgcn_s_barrier();
    MFMA_Q(0, 0, b0);
    __builtin_amdgcn_s_barrier();
    // ---- p2: quadrant (0,1)
    ldB(bH, 1, b1);
    if (u + 1 < nt) issue(1, b ^ 1, u + 1);
    if (BAR1) __builtin_amdgcn_s_barrier();
    MFMA_Q(0, 1, b1);
    __builtin_amdgcn_s_barrier();
    // ---- p3: quadrant (1,1)
    ldA(aH, 1);
    if (u + 2 < nt) issue(2, b, u + 2);
    if (BAR1) __builtin_amdgcn_s_barrier();
    MFMA_Q(1, 1, b1);
    __builtin_amdgcn_s_barrier();
    // ---- p4: quadrant (1,0); the counted wait publishes tile u+1 (only B0/B1(u+2) may stay in flight)
    if (u + 2 < nt) { issue(3, b, u + 2); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (BAR1) __builtin_amdgcn_s_barrier();
    MFMA_Q(1, 0, b0);
    __builtin_amdgcn_s_barrier();
  }
#undef MFMA_Q

  // probe epilogue: straight from the accumulators (col = lane&15, row = 4*(lane>>4)+reg)
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = tm * 256 + wr * 128 + i * 16 + fk * 4 + e;
        const int col = tn * 256 + wc * 64 + j * 16 + frow;
        C[(size_t)row * N + col] = acc[i][j][e];
      }
}

}  // namespace

extern "C" int wseg_gemm256_probe(const void* A, const void* B, float* C, int M, int N, int K, int variant, void* stream) {
  WSEG_CHECK(A && B && C && M % 256 == 0 && N % 256 == 0 && K % 64 == 0 && K >= 64, "gemm256_probe: M,N multiples of 256, K of 64");
  const int ntn = N / 256, nwg = (M / 256) * ntn;
  if (variant == 0) hipLaunchKernelGGL(gemm256_probe_kernel<1>, dim3(nwg), dim3(512), 0, (hipStream_t)stream, (const bf16_t*)A, (const bf16_t*)B, C, M, N, K, nwg, ntn);
  else hipLaunchKernelGGL(gemm256_probe_kernel<0>, dim3(nwg), dim3(512), 0, (hipStream_t)stream, (const bf16_t*)A, (const bf16_t*)B, C, M, N, K, nwg, ntn);
  WSEG_LAUNCH_CHECK();
  return 0;
}
#endif  // WSEG_PROBES
