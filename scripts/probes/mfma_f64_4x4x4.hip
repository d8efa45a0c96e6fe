// Probe: lane layout and rounding order of v_mfma_f64_4x4x4_f64 (4 blocks of 4x4x4) on gfx950.
// build: hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -o mfma4_probe scripts/probes/mfma_f64_4x4x4.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void probe(const double* A, const double* B, const double* C, double* D)
{
    const int t = blockIdx.x, l = threadIdx.x;
    D[t * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[t * 64 + l], B[t * 64 + l], C[t * 64 + l], 0, 0, 0);
}
static uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }

int main()
{
    const int NT = 128 + 1024;
    std::vector<double> A(NT * 64, 0.0), B(NT * 64, 0.0), C(NT * 64, 0.0), D(NT * 64);
    for (int t = 0; t < 64; ++t)
        for (int l = 0; l < 64; ++l) { A[t * 64 + l] = (l == t) ? 1.0 : 0.0; B[t * 64 + l] = 1.0; }
    for (int t = 0; t < 64; ++t)
        for (int l = 0; l < 64; ++l) { A[(64 + t) * 64 + l] = 1.0; B[(64 + t) * 64 + l] = (l == t) ? 1.0 : 0.0; }
    srand(777);
    auto rnd = [] { return ((double)rand() / RAND_MAX - 0.5) * ldexp(1.0, rand() % 9 - 4); };
    for (int t = 128; t < NT; ++t)
        for (int l = 0; l < 64; ++l) { A[t * 64 + l] = rnd(); B[t * 64 + l] = rnd(); C[t * 64 + l] = (t % 2) ? rnd() : 0.0; }
    double *dA, *dB, *dC, *dD;
    (void)hipMalloc(&dA, A.size() * 8); (void)hipMalloc(&dB, B.size() * 8); (void)hipMalloc(&dC, C.size() * 8); (void)hipMalloc(&dD, D.size() * 8);
    (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(NT), dim3(64), 0, 0, dA, dB, dC, dD);
    if (hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    // which A lanes / B lanes feed output lane l
    std::vector<std::vector<int>> fa(64), fb(64);
    for (int t = 0; t < 64; ++t)
        for (int l = 0; l < 64; ++l) {
            if (D[t * 64 + l] != 0.0) fa[l].push_back(t);
            if (D[(64 + t) * 64 + l] != 0.0) fb[l].push_back(t);
        }
    for (int l = 0; l < 64; l += 7) {
        printf("out lane %2d <- A lanes:", l);
        for (int x : fa[l]) printf(" %d", x);
        printf("   B lanes:");
        for (int x : fb[l]) printf(" %d", x);
        printf("\n");
    }
    // rounding order: for every output lane try the 24 orders of its 4 (A lane, B lane) products; A and B lanes are
    // paired by k: the k-th A lane multiplies the B lane of the same k -- find the pairing too (24 x 24)
    long n = 0, ok = 0;
    int best_pa[4] = {0, 1, 2, 3}, best_pb[4] = {0, 1, 2, 3};
    bool found = false;
    int pa[4] = {0, 1, 2, 3};
    do {
        int pb[4] = {0, 1, 2, 3};
        do {
            long good = 0, tot = 0;
            for (int t = 128; t < 128 + 64; ++t)
                for (int l = 0; l < 64; ++l) {
                    if (fa[l].size() != 4 || fb[l].size() != 4) continue;
                    double f = C[t * 64 + l];
                    for (int k = 0; k < 4; ++k) f = fma(A[t * 64 + fa[l][pa[k]]], B[t * 64 + fb[l][pb[k]]], f);
                    ++tot;
                    good += bits(f) == bits(D[t * 64 + l]);
                }
            if (tot > 0 && good == tot) { found = true; memcpy(best_pa, pa, sizeof(pa)); memcpy(best_pb, pb, sizeof(pb)); }
        } while (!found && std::next_permutation(pb, pb + 4));
    } while (!found && std::next_permutation(pa, pa + 4));
    printf("sequential fma chain order found: %s; A-lane order (%d %d %d %d), B-lane order (%d %d %d %d) of the ascending lists above\n",
           found ? "yes" : "NO", best_pa[0], best_pa[1], best_pa[2], best_pa[3], best_pb[0], best_pb[1], best_pb[2], best_pb[3]);
    if (found)
        for (int t = 128; t < NT; ++t)
            for (int l = 0; l < 64; ++l) {
                double f = C[t * 64 + l];
                for (int k = 0; k < 4; ++k) f = fma(A[t * 64 + fa[l][best_pa[k]]], B[t * 64 + fb[l][best_pb[k]]], f);
                ++n;
                ok += bits(f) == bits(D[t * 64 + l]);
            }
    printf("random results %ld, bit-identical to that chain: %ld\n", n, ok);
    return 0;
}
