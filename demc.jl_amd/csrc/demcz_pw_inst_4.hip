// window_kernel_pw for d % 8 == 4: see demcz_pw_inst.inc
#define PW_GROUP 4
#include "demcz_pw_inst.inc"
