// window_kernel_pw for d % 8 == 1: see demcz_pw_inst.inc
#define PW_GROUP 1
#include "demcz_pw_inst.inc"
