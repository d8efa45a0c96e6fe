// window_kernel_ml<LINREG_SSE, d, 16> for even d: see demcz_mlr_inst.inc
#define MLR_GROUP 0
#include "demcz_mlr_inst.inc"
