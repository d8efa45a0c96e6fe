// window_kernel_pw for d % 8 == 7: see demcz_pw_inst.inc
#define PW_GROUP 7
#include "demcz_pw_inst.inc"
