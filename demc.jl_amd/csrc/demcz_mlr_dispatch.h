// demcz_mlr_dispatch.h -- window_kernel_ml<LINREG_SSE, d, 16> (demcz_kernels_ml.h: the regression target on sixteen lanes per chain,
// with and without its helper waves) is instantiated for every dimension from 2 to 28: 54 kernels.  They live in two translation
// units of their own (demcz_mlr_inst_<g>.hip, dimension d in unit d % 2; compiled in parallel by demc.jl_amd/_lib.py) behind the
// function below; demcz_capi.hip does not instantiate them.
#pragma once

#include "demcz_kernels.h"

namespace demcz {

constexpr int MLR_D_MIN = 2, MLR_D_MAX = 28;

// launch: 0 = launched (the caller looks at hipGetLastError), 1 = this dimension is not built.  coop: a workgroup is one chain wave
// (four chains) + its helper waves, `blocks` = ceil(N / 4); otherwise `waves` chain waves per workgroup, blocks = ceil(N / (4 waves)).
int32_t mlr_launch_g0(int d, bool coop, unsigned blocks, int waves, hipStream_t s, const WindowParams& P);
int32_t mlr_launch_g1(int d, bool coop, unsigned blocks, int waves, hipStream_t s, const WindowParams& P);

inline int32_t mlr_launch(int d, bool coop, unsigned blocks, int waves, hipStream_t s, const WindowParams& P)
{
    if (d < MLR_D_MIN || d > MLR_D_MAX) return 1;
    return (d % 2 == 0) ? mlr_launch_g0(d, coop, blocks, waves, s, P) : mlr_launch_g1(d, coop, blocks, waves, s, P);
}

}  // namespace demcz
