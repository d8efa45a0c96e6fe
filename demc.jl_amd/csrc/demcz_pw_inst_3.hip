// window_kernel_pw for d % 8 == 3: see demcz_pw_inst.inc
#define PW_GROUP 3
#include "demcz_pw_inst.inc"
