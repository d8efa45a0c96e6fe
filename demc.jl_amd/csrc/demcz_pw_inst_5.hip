// window_kernel_pw for d % 8 == 5: see demcz_pw_inst.inc
#define PW_GROUP 5
#include "demcz_pw_inst.inc"
