// demcz_pw_dispatch.h -- window_kernel_pw (demcz_kernels_pw.h) is instantiated for every dimension from 6 to 32, both targets it
// evaluates, six forms each: 324 kernels.  They live in eight translation units of their own (demcz_pw_inst_<g>.hip, dimension d
// in unit d % 8; compiled in parallel by demc.jl_amd/_lib.py) behind the four functions below; demcz_capi.hip knows nothing of
// the template.
#pragma once

#include "demcz_kernels.h"

namespace demcz {

constexpr int PW_GROUPS = 8;
constexpr int PW_D_MIN = 6, PW_D_MAX = 32;
enum { PW_FORM_GENERAL = 0, PW_FORM_REGULAR = 1, PW_FORM_MATRIX = 2 };
enum { PW_QUERY_LIVE_BLOCKS_PER_CU = 0, PW_QUERY_LIVE_LDS_BYTES = 1, PW_QUERY_BUILT = 2 };

// launch: 0 = launched (the caller looks at hipGetLastError), 1 = this (target, d, form) is not built
#define DEMCZ_PW_DECL(g)                                                                                                          \
    int32_t pw_launch_g##g(int target, int d, bool live, bool temper, int form, unsigned blocks, hipStream_t s, const WindowParams& P); \
    int pw_query_g##g(int target, int d, int what);
DEMCZ_PW_DECL(0) DEMCZ_PW_DECL(1) DEMCZ_PW_DECL(2) DEMCZ_PW_DECL(3) DEMCZ_PW_DECL(4) DEMCZ_PW_DECL(5) DEMCZ_PW_DECL(6) DEMCZ_PW_DECL(7)
#undef DEMCZ_PW_DECL

inline int32_t pw_launch(int target, int d, bool live, bool temper, int form, unsigned blocks, hipStream_t s, const WindowParams& P)
{
    if (d < PW_D_MIN || d > PW_D_MAX) return 1;
    switch (d % PW_GROUPS) {
    case 0: return pw_launch_g0(target, d, live, temper, form, blocks, s, P);
    case 1: return pw_launch_g1(target, d, live, temper, form, blocks, s, P);
    case 2: return pw_launch_g2(target, d, live, temper, form, blocks, s, P);
    case 3: return pw_launch_g3(target, d, live, temper, form, blocks, s, P);
    case 4: return pw_launch_g4(target, d, live, temper, form, blocks, s, P);
    case 5: return pw_launch_g5(target, d, live, temper, form, blocks, s, P);
    case 6: return pw_launch_g6(target, d, live, temper, form, blocks, s, P);
    default: return pw_launch_g7(target, d, live, temper, form, blocks, s, P);
    }
}

inline int pw_query(int target, int d, int what)
{
    if (d < PW_D_MIN || d > PW_D_MAX) return 0;
    switch (d % PW_GROUPS) {
    case 0: return pw_query_g0(target, d, what);
    case 1: return pw_query_g1(target, d, what);
    case 2: return pw_query_g2(target, d, what);
    case 3: return pw_query_g3(target, d, what);
    case 4: return pw_query_g4(target, d, what);
    case 5: return pw_query_g5(target, d, what);
    case 6: return pw_query_g6(target, d, what);
    default: return pw_query_g7(target, d, what);
    }
}

}  // namespace demcz
