// demcz_kernels_pc.h -- K1f: the chain update split into a producer and a consumer (small N).
//
// Three quarters of a chain-update's instructions do not depend on the chain state: the Philox
// rounds, log / sqrt / sincos of the Box-Muller normals, the accept uniform's log and the two archive
// row indices are functions of (seed, chain, generation) and of M alone.  At N = 1024 (BASELINE C2)
// doing them inside the serial per-chain loop is what made a generation cost ~380 wave-instructions
// (measured on the lane-cooperative kernels: issue-bound at ~4.7 cycles per instruction, whatever the
// arrangement inside one wave).  They are embarrassingly parallel over chain x generation x role,
// and the chip has 1000 idle SIMDs, so:
//
//   producer workgroups (one lane per Philox block of the NEXT launch's generations) write a draw
//       record per (generation, chain): the D normals, log u of the accept test, the two row indices;
//   consumer workgroups (one lane per chain) read the records of THIS launch (written by the previous
//       launch's producers), gather the rows, form the proposal increments for a chunk of generations
//       up front (all loads in flight together), then run the state-dependent part -- proposal,
//       log-density, accept, history, K-boundary append -- from registers: ~60 instructions a
//       generation.
//
// Both halves are ONE launch (workgroups [0, consumer_blocks) consume, the rest produce), so they
// overlap on different CUs with no events or second stream; the kernel boundary that already
// separates K-windows also orders a window's records before their use.  The next launch's M is known
// to the host when this one is launched (it is this M plus the rows this launch appends).
// Arithmetic and operation order are those of the one-lane kernel: bit-identical results.
#pragma once

#include "demcz_kernels_ml.h"     // philox_blocks: rocRAND's round function as a counter -> block map

#pragma clang fp contract(off)

namespace demcz {


// record layout: rec[(g * (D + 2) + f) * N + c], f = 0..D-1 normals, D log u, D+1 the two row indices
// packed as 32-bit halves (the split layout is only selected while the archive has < 2^32 rows)
template <int D>
__device__ __forceinline__ size_t rec_index(int64_t N, int g, int f, int64_t c) { return ((size_t)g * (D + 2) + f) * (size_t)N + (size_t)c; }

template <int D>
__device__ __forceinline__ void pc_produce(const WindowParams& P, int64_t pb)
{
    constexpr int NPAIRS = (D == 1) ? 1 : (D + 1) / 2;
    constexpr int S = NPAIRS + 2;
    const int64_t nbc = (P.N + 63) / 64;                   // workgroups per (generation, role) plane
    const int64_t plane = pb / nbc;                        // wave-uniform
    const int64_t c = (pb % nbc) * 64 + threadIdx.x;
    const int role = (int)(plane % S), gi = (int)(plane / S);
    if (gi >= P.next_ngen || c >= P.N) return;
    philox_blocks rng;
    uint64_t r1, r2;
    rng.block(P.seed, (uint64_t)(P.chain_id0 + c), (uint64_t)(P.next_g_first + gi - 1) * (uint64_t)S + (uint64_t)role, r1, r2);
    double* rec = P.rec_out;
    if (role == 0) {
        uint64_t i1, i2;
        draw_rows(r1, r2, (uint64_t)P.next_M, i1, i2);
        rec[rec_index<D>(P.N, gi, D + 1, c)] = __longlong_as_double((long long)(i1 | (i2 << 32)));
    } else {
        const double lg = dm_log(u_open(r1));
        if (role == S - 1) {
            rec[rec_index<D>(P.N, gi, D, c)] = lg;
        } else {
            const double R = sqrt(-2.0 * lg);
            double cs, sn;
            dm_sincos2pi(r2 >> 11, cs, sn);
            const int p0 = (D == 1) ? 0 : 2 * (role - 1);
            rec[rec_index<D>(P.N, gi, p0, c)] = R * cs;
            if (p0 + 1 < D) rec[rec_index<D>(P.N, gi, p0 + 1, c)] = R * sn;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The consumer: 8 lanes per chain.  Lane p of a chain's group prefetches what concerns parameter p
// for a whole chunk of generations at once (row elements, normal: 6 loads a generation, so a
// 10-generation chunk fits in registers and its two dependent memory hops -- record, then archive
// row -- are paid once per chunk), forms its increments and shares them through LDS; then every lane
// of the group runs the state-dependent part redundantly from the whole state (no cross-lane traffic
// there) and stores its own element of the history row.
// (A one-lane-per-chain consumer has to hold 3d+1 doubles per prefetched generation: 5-generation
// chunks, 9.0 us per C2 window against 7.0 for this one.)
// ------------------------------------------------------------------------------------------------
constexpr int PC8_CHUNK = 10;

template <int TARGET, int D>
__global__ void __launch_bounds__(64) window_kernel_pc8(const WindowParams P)
{
    static_assert(TARGET == TARGET_MVNORMAL || TARGET == TARGET_ISO_QUAD, "split layout: MvNormal / isotropic targets");
    constexpr int L = 8, G = 64 / L, DP = ((D + 1) / 2) * 2;
    constexpr int NP = (D + L - 1) / L;                    // parameters a lane prefetches: r, r+8, ...
    constexpr int CH = (NP == 1) ? PC8_CHUNK : PC8_CHUNK / 2;
    if ((int64_t)blockIdx.x >= P.consumer_blocks) {
        pc_produce<D>(P, (int64_t)blockIdx.x - P.consumer_blocks);
        return;
    }
    __shared__ __attribute__((aligned(16))) double sdelta[G * CH * DP];
    const int lane = threadIdx.x, r = lane % L, gq = lane / L;
    const int64_t c = (int64_t)blockIdx.x * G + gq;
    if (c >= P.N) return;

    double x[D], muc[D], Wc[(TARGET == TARGET_MVNORMAL) ? D * (D + 1) / 2 : 1];
#pragma unroll
    for (int p = 0; p < D; ++p) { x[p] = P.Xcur[c + P.N * p]; muc[p] = P.tp.mu[p]; }
    if constexpr (TARGET == TARGET_MVNORMAL) {
#pragma unroll
        for (int i = 0; i < D * (D + 1) / 2; ++i) Wc[i] = P.tp.Wp[i];
    }
    int pk[NP];
    double epsv[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) { pk[k] = (r + L * k < D) ? r + L * k : 0; epsv[k] = P.eps[pk[k]]; }
    const double c0c = P.tp.c0;
    double lp = P.lpcur[c];
    const double scale = (D == 1) ? P.gamma : P.gamma / sqrt((double)(2 * D));
    int to_b = P.to_boundary;
    int64_t nb = 0;
    // history pointers of this lane's elements, advanced by a uniform stride per generation
    double* hist[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) hist[k] = P.chain + c + P.N * ((int64_t)pk[k] + (int64_t)D * P.slot_first);
    double* lobj = P.logobj + c + P.N * P.slot_first;
    const int64_t hist_stride = P.N * (int64_t)D;
    // record pointers of this lane's fields; a generation is a uniform stride further on
    const double* rec_z[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) rec_z[k] = P.rec_in + (int64_t)((D == 1) ? 0 : pk[k]) * P.N + c;
    const double* rec_lg = P.rec_in + (int64_t)D * P.N + c;
    const double* rec_ix = P.rec_in + (int64_t)(D + 1) * P.N + c;
    const int64_t rec_gs = (int64_t)(D + 2) * P.N;

    for (int g0 = 0; g0 < P.ngen; g0 += CH) {
        double lgu[CH];
        {
            uint32_t i1[CH], i2[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int g = (g0 + u < P.ngen) ? g0 + u : P.ngen - 1;
                const uint64_t ii = (uint64_t)__double_as_longlong(rec_ix[g * rec_gs]);
                i1[u] = (uint32_t)ii;
                i2[u] = (uint32_t)(ii >> 32);
            }
            const uint32_t zs = (uint32_t)P.ZS;
            double za[CH][NP], zb[CH][NP], zt[CH][NP];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int g = (g0 + u < P.ngen) ? g0 + u : P.ngen - 1;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    za[u][k] = P.Z[(uint64_t)i1[u] * zs + (uint32_t)pk[k]];      // one 32x32->64 multiply-add
                    zb[u][k] = P.Z[(uint64_t)i2[u] * zs + (uint32_t)pk[k]];
                    zt[u][k] = rec_z[k][g * rec_gs];
                }
                lgu[u] = rec_lg[g * rec_gs];
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const double diff = za[u][k] - zb[u][k];
                    const double t1 = scale * diff;
                    const double t2 = epsv[k] * zt[u][k];
                    if (r + L * k < D) sdelta[(gq * CH + u) * DP + r + L * k] = t1 + t2;
                }
            }
        }
        wave_lds_handoff();
        // the whole chunk's increments into registers first: the LDS latency is paid once, not inside
        // every generation's dependent chain
        double dl[CH][DP];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
#pragma unroll
            for (int j = 0; j < DP / 2; ++j) {
                const double2 t = reinterpret_cast<const double2*>(sdelta + (gq * CH + u) * DP)[j];
                dl[u][2 * j] = t.x;
                dl[u][2 * j + 1] = t.y;
            }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (g0 + u < P.ngen) {       // wave-uniform
                const int gi = g0 + u;
                double xp[D];
#pragma unroll
                for (int p = 0; p < D; ++p) xp[p] = x[p] + dl[u][p];
                double lpp;
                if constexpr (TARGET == TARGET_MVNORMAL) {
                    double q = 0.0;
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        double acc = Wc[(i * (i + 1)) / 2] * (xp[0] - muc[0]);
#pragma unroll
                        for (int j = 1; j <= i; ++j) acc = fma(Wc[(i * (i + 1)) / 2 + j], xp[j] - muc[j], acc);
                        q = (i == 0) ? acc * acc : fma(acc, acc, q);
                    }
                    lpp = fma(-0.5, q, c0c);
                } else {
                    double q = 0.0;
#pragma unroll
                    for (int i = 0; i < D; ++i) {
                        const double rr = xp[i] - muc[i];
                        q = (i == 0) ? rr * rr : fma(rr, rr, q);
                    }
                    lpp = -q;
                }
                double dlt = lpp - lp;
                if (P.temperature) dlt = dlt / P.temperature[gi];
                const bool acc = lgu[u] < dlt;
#pragma unroll
                for (int p = 0; p < D; ++p) x[p] = acc ? xp[p] : x[p];
                lp = acc ? lpp : lp;
                const bool boundary = (--to_b == 0);
                if (boundary) to_b = P.K;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    double xs = x[(L * k < D) ? L * k : 0];
#pragma unroll
                    for (int j = 1; j < L; ++j)
                        if (L * k + j < D) xs = (r == j) ? x[L * k + j] : xs;
                    const int p = r + L * k;
                    if (p < D) {
                        if (P.chain) *hist[k] = xs;
                        if (boundary) {      // generation divisible by K: runchain!'s append, demcz.jl:88-91
                            if (P.do_append) P.Zw[(P.M_append + nb * P.N + c) * P.ZS + p] = xs;
                            if (P.snap) P.snap[nb * P.N * D + c + P.N * p] = xs;
                        }
                    }
                    hist[k] += hist_stride;      // next generation's slab: a uniform stride, no per-lane multiply
                }
                if (P.chain && r == L - 1) *lobj = lp;
                lobj += P.N;
                if (boundary) ++nb;
            }
        }
        wave_lds_handoff();      // sdelta is rewritten by the next chunk
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        double xs = x[(L * k < D) ? L * k : 0];
#pragma unroll
        for (int j = 1; j < L; ++j)
            if (L * k + j < D) xs = (r == j) ? x[L * k + j] : xs;
        const int p = r + L * k;
        if (p < D) P.Xcur[c + P.N * p] = xs;
    }
    if (r == 0) P.lpcur[c] = lp;
}

}  // namespace demcz
