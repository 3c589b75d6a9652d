// conv_wgrad.hip — the C entry point of the convolution weight gradient; the kernels and the host-side plan live in conv_wgrad_kernels.h
// (shared with conv_igemm.hip, whose wseg_conv_bwd_pair launches the data-gradient and the weight-gradient tiles of a layer as one grid).
#include "conv_wgrad_kernels.h"

using namespace wseg_wg;

extern "C" int wseg_conv_wgrad(const wseg_wgrad_desc* d, void* stream) {
  Plan pl;
  if (int rc = wgrad_plan(d, pl)) return rc;
  const Args& a = pl.a;
  const bool unit = pl.unit;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(a.nwg);
  if (pl.kind == 0) {
    if (unit) hipLaunchKernelGGL((conv_wgrad_pipe_kernel<1>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_pipe_kernel<0>), grid, dim3(512), 0, s, a);
  }
  else if (pl.kind == 1)
    hipLaunchKernelGGL((conv_wgrad_kernel<WSEG_BF16, 256, 256, 2, 4>), grid, dim3(512), 0, s, a);
  else if (d->dtype == WSEG_BF16)
    hipLaunchKernelGGL((conv_wgrad_kernel<WSEG_BF16, 128, 128, 2, 2>), grid, dim3(256), 0, s, a);
  else if (d->dtype == WSEG_F32X3)
    hipLaunchKernelGGL((conv_wgrad_kernel<WSEG_F32X3, 128, 128, 2, 2>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<WSEG_F32, 128, 128, 2, 2>), grid, dim3(256), 0, s, a);
  WSEG_LAUNCH_CHECK();
  return 0;
}
