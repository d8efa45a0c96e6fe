// Probe: does v_mfma_f64_16x16x4_f64 round like a sequential fma chain over k?  (For round 2: residual tiles of
// the regression target on the FP64 matrix core must reproduce the oracle's dot products bit for bit.)
// build: hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -o mfma_probe scripts/probes/mfma_f64_order.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A, const double* B, const double* C, double* D, int ntile)
{
    // one wave per tile: A[tile][16][4] row-major (i,k), B[tile][4][16] (k,j), C/D[tile][16][16]
    const int t = blockIdx.x, l = threadIdx.x;
    if (t >= ntile) return;
    const double a = A[(size_t)t * 64 + (l % 16) * 4 + l / 16];
    const double b = B[(size_t)t * 64 + (l / 16) * 16 + l % 16];
    d4 c;
    for (int v = 0; v < 4; ++v) c[v] = C[(size_t)t * 256 + (4 * v + l / 16) * 16 + l % 16];
    const d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    // layout found with the search below on raw output: lane l, element v holds D[4 v + l / 16][l % 16]
    for (int v = 0; v < 4; ++v) D[(size_t)t * 256 + (4 * v + l / 16) * 16 + l % 16] = d[v];
}

static uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }

int main()
{
    const int nt = 4096;
    std::vector<double> A(nt * 64), B(nt * 64), C(nt * 256), D(nt * 256);
    srand(12345);
    auto rnd = [] { return ((double)rand() / RAND_MAX - 0.5) * ldexp(1.0, rand() % 9 - 4); };
    for (auto& x : A) x = rnd();
    for (auto& x : B) x = rnd();
    for (int t = 0; t < nt; ++t)
        for (int i = 0; i < 256; ++i) C[(size_t)t * 256 + i] = (t % 2) ? rnd() : 0.0;
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(nt), dim3(64), 0, 0, dA, dB, dC, dD, nt);
    if (hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    long n = 0, eq_fwd = 0, eq_rev = 0, eq_pair = 0, eq_exact = 0, close = 0;
    for (int t = 0; t < nt; ++t)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const double* a = &A[(size_t)t * 64 + i * 4];
                double bk[4];
                for (int k = 0; k < 4; ++k) bk[k] = B[(size_t)t * 64 + k * 16 + j];
                const double c = C[(size_t)t * 256 + i * 16 + j], d = D[(size_t)t * 256 + i * 16 + j];
                double f = c;                                    // sequential fma chain, k ascending
                for (int k = 0; k < 4; ++k) f = fma(a[k], bk[k], f);
                double r = c;                                    // k descending
                for (int k = 3; k >= 0; --k) r = fma(a[k], bk[k], r);
                const double p = fma(a[1], bk[1], a[0] * bk[0]) + fma(a[3], bk[3], a[2] * bk[2]) + c;   // a pairwise guess
                long double e = c;                               // (nearly) exact sum, rounded once
                for (int k = 0; k < 4; ++k) e += (long double)a[k] * (long double)bk[k];
                ++n;
                eq_fwd += bits(f) == bits(d);
                eq_rev += bits(r) == bits(d);
                eq_pair += bits(p) == bits(d);
                eq_exact += bits((double)e) == bits(d);
                close += fabs(f - d) <= 4e-16 * (fabs(f) + 1e-300);
            }
    printf("results %ld: == fma chain k ascending %ld, k descending %ld, pairwise guess %ld, single rounding (long double) %ld; within 2 ulp of the chain %ld\n",
           n, eq_fwd, eq_rev, eq_pair, eq_exact, close);
    return 0;
}
