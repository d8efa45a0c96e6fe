// window_kernel_pw for d % 8 == 0: see demcz_pw_inst.inc
#define PW_GROUP 0
#include "demcz_pw_inst.inc"
