// Dependent-chain latencies of the instructions on the wave-per-chain consumer's critical path (one wave per SIMD, nothing else
// running): v_add_f64, v_fma_f64, ds_bpermute, ds_read_b128 (wave-uniform address), v_cmp -> scalar -> v_mov -> ds_read, v_readlane.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/lat scripts/probes/latency_f64.hip && /tmp/lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
__global__ void k(double* out, unsigned long long* t, double a, int one)
{
    __shared__ double sm[512];
    const int lane = threadIdx.x;
    sm[lane] = a + lane; sm[lane + 64] = a; sm[lane + 128] = a;
    __syncthreads();
    double x = a, y = a + 1.0;
    unsigned long long t0, t1;
    // 1. dependent v_add_f64
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(y));
    t1 = __builtin_readcyclecounter(); if (lane == 0) t[0] = t1 - t0;
    // 2. dependent v_fma_f64
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    t1 = __builtin_readcyclecounter(); if (lane == 0) t[1] = t1 - t0;
    // 3. five independent chains of v_add_f64 (issue-bound?)
    double z0 = a, z1 = a, z2 = a, z3 = a, z4 = a;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP / 4; ++i)
        asm volatile("v_add_f64 %0, %0, %5\n v_add_f64 %1, %1, %5\n v_add_f64 %2, %2, %5\n v_add_f64 %3, %3, %5\n v_add_f64 %4, %4, %5"
                     : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3), "+v"(z4) : "v"(y));
    t1 = __builtin_readcyclecounter(); if (lane == 0) t[2] = t1 - t0;     // REP/4*5 instructions
    x += z0 + z1 + z2 + z3 + z4;
    // 4. ds_bpermute round trips (dependent through the data)
    int idx = (lane * 4) & 252; int v = lane;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(idx));
    t1 = __builtin_readcyclecounter(); if (lane == 0) t[3] = t1 - t0;
    // 5. ds_read_b128 dependent chain (address from the data)
    int ad = 0;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i) { double2 q = *reinterpret_cast<double2*>(&sm[ad]); ad = ((int)q.x & 1) * 2 * one; asm volatile("" : "+v"(ad)); }
    t1 = __builtin_readcyclecounter(); if (lane == 0) t[4] = t1 - t0;
    // 6. v_cmp -> s_and -> s_flbit -> s_mul -> v_mov (VALU -> SALU -> VALU round trip), dependent through m
    unsigned int m = lane;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i)
        asm volatile("v_cmp_gt_u32 vcc, 70, %0\n s_and_b32 s20, vcc_lo, -2\n s_or_b32 s20, s20, 1\n s_flbit_i32_b32 s20, s20\n s_xor_b32 s20, s20, 31\n"
                     "s_mul_i32 s20, s20, 48\n v_mov_b32 %0, s20" : "+v"(m) :: "vcc", "s20");
    t1 = __builtin_readcyclecounter(); if (lane == 0) t[5] = t1 - t0;
    // 6b. the same with only v_cmp -> s_and -> v_mov
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP; ++i)
        asm volatile("v_cmp_gt_u32 vcc, 70, %0\n s_and_b32 s20, vcc_lo, 62\n v_mov_b32 %0, s20" : "+v"(m) :: "vcc", "s20");
    t1 = __builtin_readcyclecounter(); if (lane == 0) t[7] = t1 - t0;
    // 7. 12 v_readlane with a scalar lane select + dependent add (the old kernel's state extraction)
    double r = a;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < REP / 4; ++i) {
        unsigned int sel = __builtin_amdgcn_readfirstlane((int)(((unsigned long long)__double_as_longlong(r)) & 31));
        unsigned int lo = __builtin_amdgcn_readlane((int)(unsigned int)__double_as_longlong(x), sel);
        unsigned int hi = __builtin_amdgcn_readlane((int)((unsigned long long)__double_as_longlong(x) >> 32), sel);
        r = __longlong_as_double(((long long)hi << 32) | lo) + r; asm volatile("" : "+v"(r));
    }
    t1 = __builtin_readcyclecounter(); if (lane == 0) t[6] = t1 - t0;
    out[lane] = x + v + ad + m + r;
}
int main()
{
    double* o; unsigned long long* t;
    hipMalloc(&o, 64 * 8); hipMalloc(&t, 8 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, 1.0, 1);
    unsigned long long h[8]; hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
    printf("shader clocks per step (one wave alone):\n");
    printf("  dependent v_add_f64            %.1f\n", h[0] / (double)REP);
    printf("  dependent v_fma_f64            %.1f\n", h[1] / (double)REP);
    printf("  5 independent v_add_f64 chains %.1f per instruction\n", h[2] / (double)(REP / 4 * 5));
    printf("  ds_bpermute round trip         %.1f\n", h[3] / (double)REP);
    printf("  ds_read_b128 round trip        %.1f\n", h[4] / (double)REP);
    printf("  v_cmp -> 6 scalar ops -> v_mov %.1f\n", h[5] / (double)REP);
    printf("  v_cmp -> 1 scalar op -> v_mov  %.1f\n", h[7] / (double)REP);
    printf("  readfirstlane + 2 readlane + add %.1f\n", h[6] / (double)(REP / 4));
    return 0;
}
