#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#pragma clang fp contract(off)
constexpr int D = 20;
__global__ void __launch_bounds__(256) k_ref(const double* Wp, const double* R, double* out, unsigned long long* clk, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double rr[D];
    for (int j = 0; j < D; ++j) rr[j] = R[(size_t)t * D + j];
    typedef const __attribute__((address_space(4))) double* cptr;
    double qs = 0.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        uint64_t wa = (uint64_t)(uintptr_t)Wp;
        asm volatile("" : "+s"(wa));
        const cptr Wc = (cptr)wa;
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            double acc = Wc[(i * (i + 1)) / 2] * rr[0];
#pragma unroll
            for (int j = 1; j <= i; ++j) acc = fma(Wc[(i * (i + 1)) / 2 + j], rr[j], acc);
            q = (i == 0) ? acc * acc : fma(acc, acc, q);
        }
        qs += q;
        rr[0] += 1e-9 * q;       // dependency between iterations
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[t] = qs;
    if ((threadIdx.x & 63) == 0) clk[t >> 6] = t1 - t0;
}
template <int G, bool HALF = false> __device__ void dpp_body(const double* Wp, const double* R, double* out, unsigned long long* clk, int iters)
{
    __shared__ double Wl[14 * 16];
    for (int e = threadIdx.x; e < 14 * 16; e += blockDim.x) Wl[e] = (e < D * (D + 1) / 2) ? Wp[e] : 0.0;
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    double rr[D];
    for (int j = 0; j < D; ++j) rr[j] = R[(size_t)t * D + j];
    double qs = 0.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        double Wr[14];
#pragma unroll
        for (int r = 0; r < 14; ++r) Wr[r] = Wl[r * 16 + (lane & 15)];
        double q = 0.0;
        if (!HALF || lane < 32) {
        if constexpr (G == 1) {
#include "rows_20_1.inc"
        } else if constexpr (G == 2) {
#include "rows_20_2.inc"
        } else {
#include "rows_20_4.inc"
        }
        }
        qs += q;
        rr[0] += 1e-9 * q;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[t] = qs;
    if ((threadIdx.x & 63) == 0) clk[t >> 6] = t1 - t0;
}
__global__ void __launch_bounds__(256) k_dpp_half(const double* Wp, const double* R, double* out, unsigned long long* clk, int iters) { dpp_body<4, true>(Wp, R, out, clk, iters); }
template <int G> __global__ void __launch_bounds__(256) k_dpp(const double* Wp, const double* R, double* out, unsigned long long* clk, int iters) { dpp_body<G>(Wp, R, out, clk, iters); }
int main()
{
    const int WG = 256, T = WG * 256, iters = 200;
    std::vector<double> W(D * (D + 1) / 2), R((size_t)T * D);
    std::mt19937_64 g(1); std::normal_distribution<double> nd;
    for (auto& x : W) x = nd(g);
    for (auto& x : R) x = nd(g);
    double *dW, *dR, *o1, *o2; unsigned long long* dc;
    hipMalloc(&dW, W.size() * 8); hipMalloc(&dR, R.size() * 8); hipMalloc(&o1, T * 8); hipMalloc(&o2, T * 8); hipMalloc(&dc, T / 64 * 8);
    hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> a(T), b(T); std::vector<unsigned long long> c(T / 64);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_ref, dim3(WG), dim3(256), 0, 0, dW, dR, o1, dc, iters);
        hipDeviceSynchronize(); hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto x : c) m += (double)x / c.size();
        if (rep) printf("scalar-load W : %.0f clocks per evaluation (one wave per SIMD)\n", m / iters);
        for (int gsel = 0; gsel < 3; ++gsel) {
            if (gsel == 0) hipLaunchKernelGGL(k_dpp<1>, dim3(WG), dim3(256), 0, 0, dW, dR, o2, dc, iters);
            else if (gsel == 1) hipLaunchKernelGGL(k_dpp<2>, dim3(WG), dim3(256), 0, 0, dW, dR, o2, dc, iters);
            else hipLaunchKernelGGL(k_dpp<4>, dim3(WG), dim3(256), 0, 0, dW, dR, o2, dc, iters);
            hipDeviceSynchronize(); hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
            m = 0; for (auto x : c) m += (double)x / c.size();
            if (rep) printf("row_newbcast W, %d rows' chains interleaved: %.0f clocks per evaluation\n", gsel == 0 ? 1 : gsel == 1 ? 2 : 4, m / iters);
            hipMemcpy(b.data(), o2, T * 8, hipMemcpyDeviceToHost); hipMemcpy(a.data(), o1, T * 8, hipMemcpyDeviceToHost);
            size_t df = 0; for (int i = 0; i < T; ++i) df += std::memcmp(&a[i], &b[i], 8) != 0;
            if (rep) printf("   outputs that differ in any bit from the scalar-load form: %zu of %d\n", df, T);
        }
    }
    {
        hipLaunchKernelGGL(k_dpp_half, dim3(WG), dim3(256), 0, 0, dW, dR, o2, dc, iters);
        hipDeviceSynchronize(); hipMemcpy(c.data(), dc, c.size() * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto x : c) m += (double)x / c.size();
        printf("row_newbcast W, 4 rows interleaved, lanes 0..31 only (EXEC = low half): %.0f clocks per evaluation\n", m / iters);
        hipLaunchKernelGGL(k_dpp<4>, dim3(WG), dim3(256), 0, 0, dW, dR, o2, dc, iters);
        hipDeviceSynchronize();
    }
    hipMemcpy(a.data(), o1, T * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, T * 8, hipMemcpyDeviceToHost);
    size_t diff = 0; for (int i = 0; i < T; ++i) diff += std::memcmp(&a[i], &b[i], 8) != 0;
    printf("outputs that differ in any bit: %zu of %d\n", diff, T);
    return diff != 0;
}
