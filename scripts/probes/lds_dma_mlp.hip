// How many gathers does a CU keep in flight?  256 workgroups x W waves; every wave does ITER rounds of "one 64-lane gather of
// 16-byte pieces from random 64-byte rows of a buffer, at most DEPTH rounds in flight", either as global_load_lds_dwordx4 (the
// LDS-DMA the wave-per-chain kernels use) or as global_load_dwordx4 into registers.  If a round's time does not fall with DEPTH,
// the gathers of a CU are being served one after another.  Clocks per round, per wave, for a small (L2-resident) and a large buffer.
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -o /tmp/mlp scripts/probes/lds_dma_mlp.hip && /tmp/mlp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITER 256
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int hash(unsigned int x)
{
    x *= 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
    return x;
}

// ROWS distinct rows per round (10 = a pass of the chain kernel: lanes 0..29 three per row; the rest re-read sequential lines)
template <int DEPTH, bool DMA>
__global__ void __launch_bounds__(512) k(const unsigned char* buf, unsigned int rows, unsigned long long* t, unsigned int* sink)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[8][DEPTH][1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned int wid = blockIdx.x * (blockDim.x >> 6) + w;
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&lds[w][0][0]);
    u4 acc = {0, 0, 0, 0};
    u4 r[DEPTH];
    for (int i = 0; i < DEPTH; ++i) r[i] = u4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITER; it += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) {
            const unsigned int g = (lane < 30) ? lane / 3 : 10 + (lane & 1);
            const unsigned int row = hash(wid * 7919u + (it + s) * 31u + g) & (rows - 1u);     // rows: a power of two
            const unsigned int off = row * 64u + (lane % 3) * 16u;
            if constexpr (DMA) {
                // at most DEPTH in flight: the slot's previous DMA is DEPTH rounds old
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH - 1) : "memory");
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(off), "s"(buf), "s"(lbase + (unsigned)s * 1024u) : "memory", "m0");
            } else {
                acc += r[s];                    // (the compiler's own wait: the load of DEPTH rounds ago)
                r[s] = *reinterpret_cast<const u4*>(buf + off);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < DEPTH; ++i) acc += r[i];
    if (DMA) acc.x += lds[w][0][lane * 16];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (lane == 0) t[wid] = t1 - t0;
}

template <int DEPTH, bool DMA>
static double run(const unsigned char* buf, unsigned int rows, int waves, unsigned long long* dt, unsigned int* sink)
{
    const int wgs = 256;
    k<DEPTH, DMA><<<wgs, 64 * waves>>>(buf, rows, dt, sink);
    k<DEPTH, DMA><<<wgs, 64 * waves>>>(buf, rows, dt, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs * waves);
    hipMemcpy(h.data(), dt, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / h.size() / ITER;
}

int main()
{
    const size_t big = (size_t)4 << 20;        // rows of 64 bytes: 256 MB
    unsigned char* buf; unsigned long long* dt; unsigned int* sink;
    if (hipMalloc(&buf, big * 64 + (1 << 20)) != hipSuccess) return 1;
    hipMemset(buf, 1, big * 64 + (1 << 20));
    if (hipMalloc(&dt, 256 * 8 * 8 * 2) != hipSuccess || hipMalloc(&sink, 256 * 512 * 4 * 2) != hipSuccess) return 1;
    for (int waves : {1, 4, 8}) {
        for (unsigned int rows : {16384u, 1u << 20, 4u << 20}) {
            printf("waves/CU %d  rows %8u (%4zu MB):  LDS-DMA depth 1/2/4/8: %6.0f %6.0f %6.0f %6.0f   plain loads depth 1/2/4/8: %6.0f %6.0f %6.0f %6.0f  clocks per round\n",
                   waves, rows, (size_t)rows * 64 >> 20,
                   run<1, true>(buf, rows, waves, dt, sink), run<2, true>(buf, rows, waves, dt, sink), run<4, true>(buf, rows, waves, dt, sink), run<8, true>(buf, rows, waves, dt, sink),
                   run<1, false>(buf, rows, waves, dt, sink), run<2, false>(buf, rows, waves, dt, sink), run<4, false>(buf, rows, waves, dt, sink), run<8, false>(buf, rows, waves, dt, sink));
        }
    }
    return 0;
}
