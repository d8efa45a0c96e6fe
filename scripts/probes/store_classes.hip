// Are 8-byte stores into rows a multiple of 1 KiB apart equally fast for every 128-byte column of the row?
// (scripts/ps2_stamps.py at 2048 chains: the workgroups whose history columns sit at bytes 384..511 of every KiB do a pass's
//  memory instructions in twice the time of the others, and the slow column moves with DEMCZ_DEBUG_HIST_SKEW -- is that the
//  memory system alone, or something the kernel adds?)
// 256 workgroups x 4 waves, as the two-chain consumer: wave w of workgroup L stores, per pass, 30 rows x 16 bytes (two lanes x 8)
// at byte L*64 + w*16 of rows `stride` bytes apart, with `delay` s_sleep units of "arithmetic" per pass and `gathers` random
// 64-byte reads per lane group per pass beside them; at most 2 passes of stores in flight.  Prints, per column class (address
// bits 9:7), the mean clocks a wave spent ISSUING the stores and waiting for the pass before last's acknowledgements.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/store_classes scripts/probes/store_classes.hip && /tmp/store_classes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error line %d\n", __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned int hash(unsigned int x)
{
    x *= 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    return x;
}

template <int DELAY, int GATHERS>
__global__ void __launch_bounds__(256) k(double* hist, size_t stride8, int passes, const double* pool, unsigned int pool_rows,
                                         unsigned long long* t, double* sink)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, L = blockIdx.x;
    const int half = lane >> 5, l5 = lane & 31;
    double* p = hist + (size_t)L * 8 + w * 2 + half + (size_t)l5 * stride8;
    unsigned long long spent = 0;
    double acc = 0.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < passes; ++it) {
        double g = 0.0;
        if (GATHERS > 0) {
#pragma unroll
            for (int q = 0; q < GATHERS; ++q) {
                const unsigned int row = hash((L * 4 + w) * 7919u + it * 31u + q * 64u + lane / 4) & (pool_rows - 1u);
                g += pool[(size_t)row * 8 + (lane & 3) * 2];
            }
        }
        const unsigned long long a = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 + GATHERS) : "memory");
        if (l5 < 30) *(volatile double*)p = (double)it;
        const unsigned long long b = __builtin_readcyclecounter();
        spent += b - a;
        p += 30 * stride8;
        acc += g;
        if (DELAY > 0) __builtin_amdgcn_s_sleep(DELAY);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { t[(L * 4 + w) * 2] = spent; t[(L * 4 + w) * 2 + 1] = t1 - t0; }
    if (acc == 123.456) sink[0] = acc;
}

// the same work with the stores taken off the gathering waves: a fifth wave of the workgroup issues all four waves' stores
template <int DELAY, int GATHERS, bool WIDE>
__global__ void __launch_bounds__(320) k5(double* hist, size_t stride8, int passes, const double* pool, unsigned int pool_rows,
                                          unsigned long long* t, double* sink)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, L = blockIdx.x;
    const int half = lane >> 5, l5 = lane & 31;
    if (w == 4) {
        // WIDE: lane (row r = lane / 8, chain k = lane % 8) -- a row's eight chains are 64 contiguous bytes, four instructions
        // cover 32 rows; else: as the chain waves would (lane = (half, row), a wave's two chains 16 bytes)
        double* p = WIDE ? hist + (size_t)L * 8 + (lane & 7) + (size_t)(lane >> 3) * stride8 : hist + (size_t)L * 8 + half + (size_t)l5 * stride8;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < passes; ++it) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                if (WIDE) { if (ww * 8 + (lane >> 3) < 30) *(volatile double*)(p + (size_t)ww * 8 * stride8) = (double)it; }
                else if (l5 < 30) *(volatile double*)(p + ww * 2) = (double)it;
            }
            p += 30 * stride8;
            if (DELAY > 0) __builtin_amdgcn_s_sleep(DELAY);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) t[1024 * 2 + L] = __builtin_readcyclecounter() - t0;
        return;
    }
    double acc = 0.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < passes; ++it) {
        double g = 0.0;
#pragma unroll
        for (int q = 0; q < GATHERS; ++q) {
            const unsigned int row = hash((L * 4 + w) * 7919u + it * 31u + q * 64u + lane / 4) & (pool_rows - 1u);
            g += pool[(size_t)row * 8 + (lane & 3) * 2];
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(GATHERS) : "memory");
        acc += g;
        if (DELAY > 0) __builtin_amdgcn_s_sleep(DELAY);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { t[(L * 4 + w) * 2] = t1 - t0; t[(L * 4 + w) * 2 + 1] = t1 - t0; }
    if (acc == 123.456) sink[0] = acc;
}

int main()
{
    const int WG = 256, passes = 400;
    const unsigned int pool_rows = 8u << 20;          // 512 MiB of 64-byte rows
    double* pool; double* sink;
    CK(hipMalloc(&pool, (size_t)pool_rows * 64)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(pool, 0, (size_t)pool_rows * 64));
    for (int variant = 0; variant < 8; ++variant)
    for (size_t stride : {(size_t)16384, (size_t)15360, (size_t)16384 + 128}) {
        const size_t skew = 0;
        const size_t bytes = (size_t)passes * 30 * stride + 65536;
        unsigned char* base; unsigned long long* t;
        CK(hipMalloc(&base, bytes + 4096)); CK(hipMalloc(&t, WG * 4 * 16 + WG * 8)); CK(hipMemset(t, 0, WG * 4 * 16 + WG * 8));
        CK(hipMemset(base, 0, bytes + 4096));
        double* hist = reinterpret_cast<double*>(base + skew);
        std::vector<unsigned long long> ht(WG * 4 * 2 + WG);
        for (int rep = 0; rep < 3; ++rep) {
            switch (variant) {
            case 0: hipLaunchKernelGGL((k<0, 0>), dim3(WG), dim3(256), 0, 0, hist, stride / 8, passes, pool, pool_rows, t, sink); break;
            case 1: hipLaunchKernelGGL((k<8, 0>), dim3(WG), dim3(256), 0, 0, hist, stride / 8, passes, pool, pool_rows, t, sink); break;
            case 2: hipLaunchKernelGGL((k<0, 3>), dim3(WG), dim3(256), 0, 0, hist, stride / 8, passes, pool, pool_rows, t, sink); break;
            case 3: hipLaunchKernelGGL((k<8, 3>), dim3(WG), dim3(256), 0, 0, hist, stride / 8, passes, pool, pool_rows, t, sink); break;
            case 4: hipLaunchKernelGGL((k5<0, 3, false>), dim3(WG), dim3(320), 0, 0, hist, stride / 8, passes, pool, pool_rows, t, sink); break;
            case 5: hipLaunchKernelGGL((k5<8, 3, false>), dim3(WG), dim3(320), 0, 0, hist, stride / 8, passes, pool, pool_rows, t, sink); break;
            case 6: hipLaunchKernelGGL((k5<0, 3, true>), dim3(WG), dim3(320), 0, 0, hist, stride / 8, passes, pool, pool_rows, t, sink); break;
            default: hipLaunchKernelGGL((k5<8, 3, true>), dim3(WG), dim3(320), 0, 0, hist, stride / 8, passes, pool, pool_rows, t, sink); break;
            }
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(ht.data(), t, WG * 4 * 16 + WG * 8, hipMemcpyDeviceToHost));
        double cls[8] = {0}, tot = 0;
        for (int L = 0; L < WG; ++L) for (int w = 0; w < 4; ++w) {
            cls[((L * 64 + skew) / 128) % 8] += (double)ht[(L * 4 + w) * 2] / passes / (WG * 4 / 8);
            tot += (double)ht[(L * 4 + w) * 2 + 1] / passes / (WG * 4);
        }
        const char* names[] = {"stores only          ", "stores + sleep       ", "stores + gathers     ", "stores+gathers+sleep ", "5th wave stores      ", "5th wave stores+sleep", "5th wave, 64-B chunks", "5th wave 64-B + sleep"};
        printf("%s stride %6zu: %6.0f clocks per pass; in the store (issue + wait; 5th-wave variants: the gathering waves' whole pass), by address class (bits 9:7): ", names[variant], stride, tot);
        for (int c = 0; c < 8; ++c) printf("%6.0f", cls[c]);
        printf("\n");
        if (variant >= 4) {
            double c5[8] = {0};
            for (int L = 0; L < WG; ++L) c5[((L * 64 + skew) / 128) % 8] += (double)ht[WG * 4 * 2 + L] / passes / (WG / 8);
            printf("                                     the storing wave's own clocks per pass, by address class:                                              ");
            for (int c = 0; c < 8; ++c) printf("%6.0f", c5[c]);
            printf("\n");
        }
        CK(hipFree(base)); CK(hipFree(t));
    }
    return 0;
}
