// tu_taps.hip -- the kernels unrolled for ONE Gaussian tap count: compiled once per K of DPC_TAP_LIST with
// -DDPC_TU_K=K (see the Makefile), each time into its own object, so that the tap counts build side by side.
// Everything here is reached through dpck::TapKernels<K> (host_launch_k.inc); the kernels that do not depend on a
// tap count, the launch logic and the C ABI are in dpc_kernels.hip.
#ifndef DPC_TU_K
#error "compile with -DDPC_TU_K=<tap count>"
#endif
#include "k_prelude.inc"
#include "k_device_common.inc"
#include "k_points.inc"
#include "k_blur_lds.inc"
#include "k_fir.inc"
#include "k_blur_stream.inc"
#include "k_fused.inc"
#include "k_zpass.inc"
#include "host_launch_k.inc"

template struct dpck::TapKernels<DPC_TU_K>;
