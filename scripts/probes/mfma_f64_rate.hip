// v_mfma_f64_16x16x4_f64 on gfx950: cycles per instruction on ONE accumulator chain (each instruction needs the result of the one
// before) and on 2 / 4 independent accumulators, one wave alone and four waves of a workgroup (one per SIMD), and the FP64 vector
// fma rate beside it.  What bounds the regression kernel's 48 matrix instructions per step (demcz_kernels_lr.h)?
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/mf scripts/probes/mfma_f64_rate.hip && /tmp/mf
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define REP 128
template <int NACC>
__global__ void k(double* out, unsigned long long* t, double a, double b)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{a, a, a, a};
    const unsigned long long t0 = __builtin_readcyclecounter();
    double av = a + threadIdx.x * 1e-9, bv = b;
    asm volatile("" : "+v"(av), "+v"(bv));
#pragma unroll
    for (int i = 0; i < REP; ++i)
#pragma unroll
        for (int j = 0; j < NACC; ++j) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(av), "v"(bv));
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
#pragma unroll
    for (int j = 0; j < NACC; ++j) asm volatile("" : "+v"(acc[j]));
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
__global__ void kf(double* out, unsigned long long* t, double a, double b)
{
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        x0 = fma(x0, b, a); x1 = fma(x1, b, a); x2 = fma(x2, b, a); x3 = fma(x3, b, a);
        x4 = fma(x4, b, a); x5 = fma(x5, b, a); x6 = fma(x6, b, a); x7 = fma(x7, b, a);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}
template <int NACC>
static void run(const char* what, int threads, int blocks)
{
    double* o; unsigned long long* t;
    (void)hipMalloc(&o, (size_t)threads * blocks * 8); (void)hipMalloc(&t, (size_t)blocks * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, o, t, 1.0, 1.0000001);
    unsigned long long h[4096]; (void)hipMemcpy(h, t, sizeof(unsigned long long) * (blocks < 4096 ? blocks : 4096), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < (blocks < 4096 ? blocks : 4096); ++i) m += h[i];
    m /= (blocks < 4096 ? blocks : 4096);
    printf("  %-46s %2d accumulator(s): %6.1f shader clocks per v_mfma_f64_16x16x4_f64\n", what, NACC, m / (REP * NACC));
    (void)hipFree(o); (void)hipFree(t);
}
int main()
{
    run<1>("one wave alone", 64, 1); run<2>("one wave alone", 64, 1); run<4>("one wave alone", 64, 1);
    run<1>("four waves of one workgroup (one per SIMD)", 256, 1); run<4>("four waves of one workgroup (one per SIMD)", 256, 1);
    run<1>("256 workgroups of four waves (every SIMD of the chip)", 256, 256); run<4>("256 workgroups of four waves (every SIMD of the chip)", 256, 256);
    double* o; unsigned long long* t; (void)hipMalloc(&o, 256 * 256 * 8); (void)hipMalloc(&t, 256 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kf, dim3(256), dim3(256), 0, 0, o, t, 1.0, 1.0000001);
    unsigned long long h[256]; (void)hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
    printf("  v_fma_f64, 8 independent chains, every SIMD busy: %.2f shader clocks per instruction (64 lanes x 2 flop)\n", m / 256 / (REP * 8));
    printf("  (one v_mfma_f64_16x16x4_f64 = 16*16*4*2 = 2048 flop; at C clocks each a SIMD does 2048/C flop per clock; v_fma_f64 at c clocks does 128/c)\n");
    return 0;
}
