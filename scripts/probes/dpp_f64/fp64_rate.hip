// Clocks per wave64 FP64 vector instruction with N independent accumulators issued round-robin (one wave per SIMD, and four):
// is v_fma_f64 a 4-clock or an 8-clock instruction on this chip?   hipcc -O3 --offload-arch=gfx950 -o /tmp/f fp64_rate.hip && /tmp/f
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NACC, int OP>
__global__ void __launch_bounds__(512) k(const double* in, double* out, unsigned long long* clk, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double a[NACC];
    const double w = in[t & 15], x = in[(t + 1) & 15];
#pragma unroll
    for (int i = 0; i < NACC; ++i) a[i] = in[t] + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(x));
            else if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(w));
            else if (OP == 3) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(w), "v"(x));
            else if (OP == 4) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(w), "v"(x));
            else asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(*(float*)&a[i]) : "v"((float)w), "v"((float)x));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < NACC; ++i) s += a[i];
    out[t] = s;
    if ((threadIdx.x & 63) == 0) clk[t >> 6] = t1 - t0;
}
template <int NACC, int OP> double run(int wg, int threads, double* in, double* out, unsigned long long* dc, int iters)
{
    std::vector<unsigned long long> c((size_t)wg * threads / 64);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<NACC, OP>), dim3(wg), dim3(threads), 0, 0, in, out, dc, iters); hipDeviceSynchronize(); }
    hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto x : c) s += (double)x / c.size();
    return s / iters / NACC;
}
int main()
{
    const int T = 1024 * 256, iters = 2000;
    double *in, *out; unsigned long long* dc;
    hipMalloc(&in, T * 8); hipMalloc(&out, T * 8); hipMalloc(&dc, T / 64 * 8); hipMemset(in, 0, T * 8);
    printf("one wave per SIMD (256 workgroups x 256 threads):\n");
    printf("  v_fma_f64, 4 / 8 / 16 independent accumulators: %.2f / %.2f / %.2f clocks per instruction\n", run<4, 0>(256, 256, in, out, dc, iters), run<8, 0>(256, 256, in, out, dc, iters), run<16, 0>(256, 256, in, out, dc, iters));
    printf("  v_add_f64, 16: %.2f     v_fma_f32, 16: %.2f\n", run<16, 1>(256, 256, in, out, dc, iters), run<16, 2>(256, 256, in, out, dc, iters));
    printf("four waves per SIMD (1024 workgroups x 256 threads), v_fma_f64, 8 accumulators: %.2f clocks per instruction of ONE wave (x 1/4 = SIMD's issue interval)\n", run<8, 0>(1024, 256, in, out, dc, iters));
    // round 5: is the DPP form (v_fmac_f64_dpp ... row_newbcast) issued at the rate of the plain one when SEVERAL waves share a SIMD?
    printf("v_fmac_f64 plain / dpp row_newbcast, 8 accumulators, clocks per instruction of one wave:\n");
    printf("  one wave per SIMD:   %.2f / %.2f\n", run<8, 4>(256, 256, in, out, dc, iters), run<8, 3>(256, 256, in, out, dc, iters));
    printf("  two waves per SIMD:  %.2f / %.2f\n", run<8, 4>(256, 512, in, out, dc, iters), run<8, 3>(256, 512, in, out, dc, iters));
    printf("  four waves per SIMD: %.2f / %.2f\n", run<8, 4>(1024, 256, in, out, dc, iters), run<8, 3>(1024, 256, in, out, dc, iters));
    return 0;
}
