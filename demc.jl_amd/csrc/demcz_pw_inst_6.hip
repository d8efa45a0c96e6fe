// window_kernel_pw for d % 8 == 6: see demcz_pw_inst.inc
#define PW_GROUP 6
#include "demcz_pw_inst.inc"
