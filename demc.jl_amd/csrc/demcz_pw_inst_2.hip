// window_kernel_pw for d % 8 == 2: see demcz_pw_inst.inc
#define PW_GROUP 2
#include "demcz_pw_inst.inc"
