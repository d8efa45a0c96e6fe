// window_kernel_ml<LINREG_SSE, d, 16> for odd d: see demcz_mlr_inst.inc
#define MLR_GROUP 1
#include "demcz_mlr_inst.inc"
