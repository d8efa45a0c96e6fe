// How many clocks does a wave spend per v_mov_b64_dpp row_newbcast / v_fmac_f64_dpp row_newbcast / plain v_add_f64, issued back to
// back with independent destinations (one wave per SIMD)?    hipcc -O3 --offload-arch=gfx950 -o /tmp/r dpp_mov_rate.hip && /tmp/r
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(256) k(const double* in, double* out, unsigned long long* clk, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double w = in[t & 15], a0 = in[t], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            asm volatile("s_nop 1\n\t"
                "v_mov_b64_dpp %0, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b64_dpp %2, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b64_dpp %4, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %5, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b64_dpp %6, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %7, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(w));
        } else if (MODE == 1) {
            asm volatile("s_nop 1\n\t"
                "v_fmac_f64_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f64_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(w));
        } else {
            asm volatile(
                "v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %8\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %8\n\t"
                "v_add_f64 %4, %4, %8\n\tv_add_f64 %5, %5, %8\n\tv_add_f64 %6, %6, %8\n\tv_add_f64 %7, %7, %8\n\t"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[t] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) clk[t >> 6] = t1 - t0;
}
int main()
{
    const int WG = 256, T = WG * 256, iters = 1000;
    double *in, *out; unsigned long long* dc;
    hipMalloc(&in, T * 8); hipMalloc(&out, T * 8); hipMalloc(&dc, T / 64 * 8); hipMemset(in, 0, T * 8);
    std::vector<unsigned long long> c(T / 64);
    const char* names[] = {"v_mov_b64_dpp row_newbcast", "v_fmac_f64_dpp row_newbcast", "v_add_f64"};
    for (int m = 0; m < 3; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(WG), dim3(256), 0, 0, in, out, dc, iters);
            else if (m == 1) hipLaunchKernelGGL(k<1>, dim3(WG), dim3(256), 0, 0, in, out, dc, iters);
            else hipLaunchKernelGGL(k<2>, dim3(WG), dim3(256), 0, 0, in, out, dc, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto x : c) s += (double)x / c.size();
        printf("%-28s %.2f clocks per instruction (8 independent, back to back)\n", names[m], s / iters / 8);
    }
    return 0;
}
