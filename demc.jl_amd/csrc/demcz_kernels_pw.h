// demcz_kernels_pw.h -- K1g': the wave-per-chain consumer (demcz_kernels_ps.h) for dimensions whose pass does not fit
// one DMA instruction and whose node rows do not fit the registers (BASELINE C4's per-GPU shard: MvNormal d = 20).
//
// The same scheme: one wavefront per chain, lane n evaluates node n of the tree of accept / reject outcomes of the next
// R <= 5 generations, every candidate formed by the additions the serial order would have made (a rejected generation
// adds -0.0), ONE log-density evaluation for all 31 nodes, all accept tests one compare into a lane mask, the path
// read off the mask -- bit-identical to the serial order.  What differs from the d <= 5 kernel:
//   * a pass's draws are NDMA = ceil((10 * ceil(d/2) + 3 * (d + 2) [+ 3]) / 64) LDS-DMA instructions (d = 20: three) into
//     a slot of NDMA KiB; the counted wait counts them;
//   * increments and history rows take ceil(5 d / 64) and ceil(5 (d + 1) / 64) rounds of lanes;
//   * a node's rows of increments stay in LDS and are read as they are added (100 doubles would not fit the registers);
//     the front end of the next pass therefore only forms increments (after the adds, in program order);
//   * W comes through scalar loads issued inside the log-density every pass (no room to keep 210 doubles), mu from an LDS copy;
//   * the state travels through the LDS candidate table (one table; the winner's row is copied to row 0) instead of
//     40 v_readlanes.
// Everything else -- records, producer half, LIVE hand-off with sentinel and publisher, counted waits -- is the d <= 5
// kernel's; see there for the reasoning.
#pragma once

#include "demcz_kernels_ps.h"

#pragma clang fp contract(off)

namespace demcz {

// MF (round 4; MvNormal, d = 20, LIVE launches): the 31 candidates of a pass and their log-densities on the FP64 matrix
// instruction, v_mfma_f64_16x16x4_f64, whose accumulation IS the sequential fma chain, k ascending (scripts/probes/
// mfma_f64_order.hip) -- so the doubles are the oracle's:
//   candidates   C = x 1' + Delta T      Delta (d x 5: the pass's increments), T (5 x 32: 1.0 where node j's path takes generation u):
//                c_pj = fma(delta_pu, t_uj, c_pj), u ascending = ((x + d_1) + d_2) ... with "+ 0 * delta" where the serial order
//                adds -0.0 (equal unless a coordinate is exactly -0.0);
//   whitening    Y = W R,  R = C - mu 1' : the accumulator layout of C (lane l, element v <-> parameter 4 v + l / 16, candidate l % 16)
//                IS the B-operand layout of the next product, so R never leaves the registers; W's tiles (zero above the
//                diagonal: fma(0, r_j, y) = y) sit in nine registers;
//   q            the DIAGONAL of Y' Y by the same trick (A = B = Y's accumulator elements): q_j = fma(y_ij, y_ij, q_j), i ascending.
// 8 + 18 + 10 matrix instructions (65 clocks each) replace 100 adds behind 50 LDS reads and 210 fmas behind 27 scalar loads, on
// all 64 lanes instead of 31.  A non-finite increment (0 * inf) would poison candidates that do not take it: such a pass flags
// the launch, and the library redoes it with the scalar kernels (live_verify) -- the reason this form is LIVE-only.
// REG (round 4; LIVE launches that start right after a boundary, with K and their length multiples of five): every pass is five
// generations and a boundary falls on the end of every (K/5)-th pass -- the nibble queue of pass lengths, the clamps on the
// generation indices and the per-pass "how long is the pass after next" arithmetic are constants and a counter
// (what window_kernel_ps2 is to window_kernel_ps; profiles/r04p_pw_regular.txt).
// W by lanes (round 4, MvNormal, every scalar form of the kernel): the whitening's 210 fmas used to wait for 27-29 scalar loads of W
// a pass that ~100 SGPRs cannot prefetch more than a row ahead.  W now sits in 14 register pairs, entry e in lane e % 16 of every
// 16-lane row, and each fma takes its entry by DPP row_newbcast (v_fmac_f64_dpp; scripts/gen_pw_wdpp.py generates the chains,
// scripts/probes/dpp_f64 measured them: 1140-1280 clocks per evaluation against 1700 alone, the same doubles).
#ifndef PW_WDPP
#define PW_WDPP 1
#endif
// Round 5: every dimension from 6 to 32, MvNormal and the isotropic quadratic (the reference's own scripts run at d = 10, 26, 30:
// test/test_anneal.jl:7-10, test/example_linreg.jl:9, test/test_anneal_parallel.jl:16) -- the generated DPP texts exist for
// every one of them (scripts/gen_pw_wdpp.py), the kernel's tables were always written for a general D.  What grows with D: a
// slot is NDMA = 2..5 KiB (a workgroup's LDS: 40 KB at d = 8, 70 KB at d = 20, 110 KB at d = 32 -- from d = 23 on ONE workgroup
// fits a CU), a node's candidate is D registers and W ceil(D (D + 1) / 32) more: beyond d = 20 the kernel is compiled for two
// waves per SIMD (256 registers) instead of three -- it never has more than two there.
constexpr int pw_min_waves(int D) { return (D <= 20) ? 3 : 2; }
template <int TARGET, int D, bool LIVE, bool TEMPER, bool MF = false, bool REG = false>
__global__ void __launch_bounds__(64 * (PS_CHAINS + (LIVE ? 1 : 0)), pw_min_waves(D)) window_kernel_pw(const WindowParams P)
{
    static_assert(!REG || (LIVE && PS_R == 5), "regular launches: LIVE, five generations a pass");
    static_assert(TARGET == TARGET_MVNORMAL || TARGET == TARGET_ISO_QUAD, "split layout: MvNormal / isotropic targets");
    static_assert(D >= 6 && D <= 32, "d <= 5: window_kernel_ps");
    static_assert(!MF || (LIVE && TARGET == TARGET_MVNORMAL && D > 16 && D <= 20 && PS_R == 5), "matrix form: MvNormal, 16 < d <= 20, LIVE");
    constexpr int HW = (D + 1) / 2;                        // 16-byte pieces of an archive row
    constexpr int ZSC = ((D + 7) / 8) * 8;                 // archive row stride in doubles (demcz_create: ZS)
    constexpr int DP = ((D + 1) / 2) * 2;                  // increments row in LDS
    constexpr int CR = ((D + 2) / 2) * 2;                  // candidate row in LDS: D doubles, log-density, pad
    // pieces of a pass's DMAs: [0, ROWL) archive rows (generation u, first / second row, piece j); [FL0, TL0) the D + 2
    // record fields, three pieces = six generations each (fields 0..D-1 normals, D log u: this pass; D+1 row indices: the
    // pass two after it); [TL0, TL0 + 3) temperatures; the rest idle
    constexpr int ROWL = PS_R * 2 * HW;
    constexpr int FL0 = ROWL;
    constexpr int TL0 = FL0 + 3 * (D + 2);
    constexpr int NDMA = (TL0 + 3 + 63) / 64;
    constexpr int SLOTB = NDMA * 1024;
    constexpr int NF = (PS_R * D + 63) / 64;               // rounds of lanes that form increments
    constexpr int NH = (PS_R * (D + 1) + 63) / 64;         // rounds of lanes that store history
    constexpr int VMW = NDMA + 4 * NH;                     // vector-memory instructions behind a slot's last DMA when it is waited for
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (!LIVE) {
        if ((int64_t)blockIdx.x >= P.consumer_blocks) {    // short launches: the producer half rides in the grid
            pc_produce<D>(P, ((int64_t)blockIdx.x - P.consumer_blocks) * PS_CHAINS + w, lane);
            return;
        }
    }
    __shared__ __attribute__((aligned(16))) unsigned char raw[PS_CHAINS][PS_SLOTS][SLOTB];
    __shared__ __attribute__((aligned(16))) double sdelta[PS_CHAINS][(PS_R + 1) * DP];       // row PS_R: negative zeros
    __shared__ __attribute__((aligned(16))) double ctab[PS_CHAINS][32 * CR];                   // row 0: the current state
    __shared__ __attribute__((aligned(16))) double mul[DP];          // mu, one copy per workgroup
    constexpr bool WDPP = (PW_WDPP != 0) && !MF && TARGET == TARGET_MVNORMAL;
    constexpr int NWR = (D * (D + 1) / 2 + 15) / 16;                 // register pairs that hold W, sixteen entries each
    // the pass's increments by lanes, the same way (candidate adds): scripts/gen_pw_wdpp.py, gen_adds
#ifndef PW_DDPP
#define PW_DDPP 1
#endif
    constexpr bool DDPP = (PW_DDPP != 0) && !MF && PS_R == 5;
    // per generation of the pass: positions of a 16-lane row that take it in every row with takers, and the register pairs that
    // then hold its D increments (the generator's numbers; checked against each other in the generated text)
    constexpr int DD_KN[5] = {4, 4, 4, 4, 16};
    constexpr int DD_NQ[5] = {(D + 3) / 4, (D + 3) / 4, (D + 3) / 4, (D + 3) / 4, (D + 15) / 16};
    constexpr int DD_NQMAX = (D + 3) / 4;
    __shared__ double Wl[WDPP ? NWR * 16 : 1];                       // W packed lower-triangular, one copy per workgroup
    if constexpr (WDPP) {
        for (int e = threadIdx.x; e < NWR * 16; e += blockDim.x) Wl[e] = (e < D * (D + 1) / 2) ? P.tp.Wp[e] : 0.0;
    }
    __shared__ double pub_rows[LIVE ? PS_CHAINS * PS_PUB * D : 1];
    __shared__ unsigned int pub_seq[PS_CHAINS], pub_done[PS_CHAINS], pub_exit[PS_CHAINS];
    for (int e = threadIdx.x; e < D; e += blockDim.x) mul[e] = P.tp.mu[e];
    if constexpr (LIVE) {
        if (threadIdx.x < PS_CHAINS) { pub_seq[threadIdx.x] = 0u; pub_done[threadIdx.x] = 0u; pub_exit[threadIdx.x] = 0u; }
    }
    __syncthreads();
    if constexpr (LIVE) {
        if (w == PS_CHAINS) {       // the publisher (demcz_kernels_ps.h): element e = (chain wave, p), in rounds of 64
            constexpr int NPL = (PS_CHAINS * D + 63) / 64;
            unsigned int done[NPL];
#pragma unroll
            for (int t = 0; t < NPL; ++t) done[t] = 0u;
            while (true) {
                publisher_wait(P);
                bool any = false, allgone = true;
#pragma unroll
                for (int t = 0; t < NPL; ++t) {
                    const int e = lane + 64 * t;
                    const bool pl = e < PS_CHAINS * D;
                    const int cw = pl ? e / D : 0, pp = pl ? e % D : 0;
                    const int64_t cl = (int64_t)xcd_block(P) * PS_CHAINS + cw;
                    const unsigned int seq = __hip_atomic_load(&pub_seq[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const bool ready = pl && seq != done[t];
                    if (ready) {
                        const double v = pub_rows[(cw * PS_PUB + (int)(done[t] % PS_PUB)) * D + pp];
                        if (cl < P.N && P.do_append) live_publish(P, (int64_t)done[t], cl, pp, v);
                        ++done[t];
                    }
                    asm volatile("" ::: "memory");
                    if (ready && pp == 0) __hip_atomic_store(&pub_done[cw], done[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    any |= __builtin_amdgcn_ballot_w64(ready) != 0ull;
                    const bool gone = !pl || (__hip_atomic_load(&pub_exit[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u &&
                                              __hip_atomic_load(&pub_seq[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == done[t]);
                    allgone &= __builtin_amdgcn_ballot_w64(!gone) == 0ull;
                }
                if (any) continue;
                if (allgone) break;             // (never before its chain waves: demcz_kernels_ps.h)
                __builtin_amdgcn_s_sleep(1);
            }
            return;
        }
    }
    auto leave = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (LIVE) {
            if (lane == 0) __hip_atomic_store(&pub_exit[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    const int64_t c = (int64_t)xcd_block(P) * PS_CHAINS + w;       // (XCD-aware: demcz_kernels.h)
    if (c >= P.N) {
        wave_store_counts(P, c, 0u, 0u);
        leave();
        return;
    }
    if constexpr (LIVE) {
        if (__hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { leave(); return; }
    }
    __builtin_amdgcn_s_setprio(3);
    unsigned char* const raw_w = &raw[w][0][0];
    double* const sd_w = &sdelta[w][0];
    double* const ct_w = &ctab[w][0];
    const unsigned raw_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)raw_w);

    // ---- what this lane is, in each of its parts (demcz_kernels_ps.h) --------------------------------------------
    const int nn = (lane >= 1 && lane < 32) ? lane : 1;
    const int lev = 32 - __builtin_clz((unsigned)nn);
    const double* mrow[PS_R];
    // DDPP: the lanes whose node takes generation j of the pass (a ballot), and where this lane finds, in the wave's block of
    // increments, the entry it holds of register pair 0 of generation j: entry m, m = its rank among the positions of a row that
    // take the generation in every row with takers (pair q: + q * KN entries; lanes at other positions are never read from)
    [[maybe_unused]] uint64_t tmask[PS_R];
    [[maybe_unused]] const double* dptr[PS_R];
#pragma unroll
    for (int j = 1; j <= PS_R; ++j) {
        const bool tk = lane != 0 && ((j == lev) || (j < lev && ((nn >> (lev - 1 - j)) & 1)));
        mrow[j - 1] = sd_w + (tk ? j - 1 : PS_R) * DP;
        if constexpr (DDPP) {
            const uint64_t tm = __builtin_amdgcn_ballot_w64(tk);
            unsigned int common = 0xffffu;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned int rowm = (unsigned int)(tm >> (16 * r)) & 0xffffu;
                if (rowm) common &= rowm;
            }
            const int pos = lane & 15;
            const int m = ((common >> pos) & 1u) ? __builtin_popcount(common & ((1u << pos) - 1u)) : 0;
            tmask[j - 1] = tm;
            dptr[j - 1] = sd_w + (j - 1) * DP + m;
        }
    }
    [[maybe_unused]] double one = 1.0;
    if constexpr (DDPP) asm volatile("" : "+v"(one));       // (in a register: the multiplicand of the adds-as-fmas)
    int anc = nn;
    while (anc > 1 && (anc & 1) == 0) anc >>= 1;
    anc = (anc == 1) ? 0 : (anc >> 1);
    const int anc4 = anc * 4;
    unsigned int need1 = 0u, need0 = 0u;
#pragma unroll
    for (int t = 1; t < PS_R; ++t) {
        if (t < lev) {
            const unsigned int a = (unsigned int)nn >> (lev - t);
            if ((nn >> (lev - 1 - t)) & 1) need1 |= 1u << a; else need0 |= 1u << a;
        }
    }
    const bool nodel = lane >= 1 && lane < 32;
    // MF: what this lane is in the matrix instruction's layouts: g = lane / 16 (k index of the A / B operands, row group of the
    // accumulator), c16 = lane % 16 (row of A, column of B and of the accumulator)
    typedef double mf4 __attribute__((ext_vector_type(4)));
    [[maybe_unused]] const int mg = lane >> 4, mc = lane & 15;
    [[maybe_unused]] double mT[2][2], mW[2][5], mmu0[4], mmu1;
    if constexpr (MF) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int j = 16 * ct + mc, u1 = 4 * ks + mg + 1;           // candidate, generation (1-based) of the pass
                const int lj = j ? 32 - __builtin_clz((unsigned)j) : 0;
                const bool take = j != 0 && u1 <= PS_R && ((u1 == lj) || (u1 < lj && ((j >> (lj - 1 - u1)) & 1)));
                mT[ct][ks] = take ? 1.0 : 0.0;
            }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const int i = 16 * rt + mc, j = 4 * ks + mg;
                mW[rt][ks] = (i < D && j <= i) ? P.tp.Wp[(i * (i + 1)) / 2 + j] : 0.0;
            }
#pragma unroll
        for (int v = 0; v < 4; ++v) mmu0[v] = P.tp.mu[4 * v + mg];
        mmu1 = (16 + mg < D) ? P.tp.mu[16 + mg] : 0.0;
    }
    const int lgo = (FL0 + 3 * D) * 16 + (lev - 1) * 8;
    [[maybe_unused]] const int tko = TL0 * 16 + (lev - 1) * 8;
    const int ixo = (FL0 + 3 * (D + 1)) * 16;
    // increments: element e = lane + 64 t = (generation u, parameter p)
    bool fl[NF];
    int fu[NF], fp[NF], zao[NF], zbo[NF], zto[NF];
    double eps_p[NF];
#pragma unroll
    for (int t = 0; t < NF; ++t) {
        const int e = lane + 64 * t;
        fl[t] = e < PS_R * D;
        fu[t] = fl[t] ? e / D : 0;
        fp[t] = fl[t] ? e % D : 0;
        zao[t] = ((fu[t] * 2) * HW) * 16 + fp[t] * 8;
        zbo[t] = ((fu[t] * 2 + 1) * HW) * 16 + fp[t] * 8;
        zto[t] = (FL0 + 3 * fp[t]) * 16 + fu[t] * 8;
        eps_p[t] = P.eps[fp[t]];
    }
    const double scale = P.gamma / sqrt((double)(2 * D));
    // DMA sources: piece q = lane + 64 k
    bool rowl[NDMA], ixl[NDMA];
    int ru[NDMA], rwhich[NDMA];
    const unsigned char* sbase[NDMA];
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
        const int q = lane + 64 * k;
        rowl[k] = q < ROWL;
        ru[k] = rowl[k] ? q / (2 * HW) : 0;
        rwhich[k] = rowl[k] ? (q / HW) % 2 : 0;
        const int rj = rowl[k] ? q % HW : 0;
        const bool fieldl = q >= FL0 && q < TL0;
        const int ff = fieldl ? (q - FL0) / 3 : 0, fj = fieldl ? (q - FL0) % 3 : 0;
        ixl[k] = fieldl && ff == D + 1;
        const bool templ = TEMPER && q >= TL0 && q < TL0 + 3;
        if (rowl[k]) sbase[k] = reinterpret_cast<const unsigned char*>(P.Z) + rj * 16;
        else if (fieldl) sbase[k] = reinterpret_cast<const unsigned char*>(P.rec_in + ((int64_t)ff * P.N + c) * P.rec_stride) + fj * 16;
        else if (templ) sbase[k] = reinterpret_cast<const unsigned char*>(P.temperature) + (q - TL0) * 16;
        else sbase[k] = reinterpret_cast<const unsigned char*>(P.rec_in);
    }
    // history: element e = lane + 64 t = (generation j of the pass, p; p == D: log_obj)
    bool hl[NH];
    int hp[NH];
    unsigned int hmask[NH];
    uint32_t hx_off[NH], hl_off[NH];
    const bool hist = P.chain != nullptr;
#pragma unroll
    for (int t = 0; t < NH; ++t) {
        const int e = lane + 64 * t;
        hl[t] = e < PS_R * (D + 1);
        const int hj = hl[t] ? e / (D + 1) : 0;
        hp[t] = hl[t] ? e % (D + 1) : 0;
        hmask[t] = (hj + 1 >= 5) ? 0xffffffffu : ((1u << (1u << (hj + 1))) - 1u);
        hx_off[t] = (hl[t] && hp[t] < D) ? (uint32_t)((((int64_t)hj * D + hp[t]) * P.N + c) * 8) : 0x7fffff00u;
        hl_off[t] = (hl[t] && hp[t] == D) ? (uint32_t)(((int64_t)hj * P.N + c) * 8) : 0x7fffff00u;
    }

    // ---- passes of the launch: the nibble queue of demcz_kernels_ps.h
    unsigned int segq = 0u;
    int cg = 0, ctb = P.to_boundary;
    int ngen_s = P.ngen, K_s = P.K;
    asm volatile("" : "+s"(ngen_s), "+s"(K_s));
    auto seg_make = [&]() __attribute__((always_inline)) -> unsigned int {
        int n = ngen_s - cg;
        n = (n < 0) ? 0 : n;
        n = (n < PS_R) ? n : PS_R;
        const int R = (ctb < n) ? ctb : n;
        const int B = (R > 0 && ctb - R == 0) ? 1 : 0;
        cg += R;
        ctb = B ? K_s : ctb - R;
        return (unsigned int)(R | (B << 3));
    };
    if constexpr (!REG) {
#pragma unroll
        for (int k = 0; k < 6; ++k) segq |= seg_make() << (4 * k);
    }
    auto qR = [&](int k) __attribute__((always_inline)) -> int { if constexpr (REG) return PS_R; else return (int)((segq >> (4 * k)) & 7u); };
    // (the clamp stays in the regular form: these record buffers are not the arena's zero-filled ones -- a look-ahead past the
    //  launch's last generation must not turn stale bytes into row indices)
    auto gclamp = [&](int g) __attribute__((always_inline)) { return (g < ngen_s) ? g : ngen_s - 1; };
    int g0 = 0, g3 = qR(0) + qR(1) + qR(2), g5 = g3 + qR(3) + qR(4);
    int npass;
    [[maybe_unused]] int tb = P.to_boundary / PS_R;                 // REG: passes up to and including the next boundary pass
    [[maybe_unused]] const int tbK = P.K / PS_R;
    if constexpr (REG) {
        npass = P.ngen / PS_R;
    } else {
        const int n1 = (P.to_boundary < P.ngen) ? P.to_boundary : P.ngen, rest = P.ngen - n1;
        npass = (n1 + PS_R - 1) / PS_R + (rest / P.K) * ((P.K + PS_R - 1) / PS_R) + (rest % P.K + PS_R - 1) / PS_R;
    }

    // state of the chain: row 0 of the table (a pass starts by reading it; only its log-density is also kept in a register)
    double lp = P.lpcur[c];
    {
        constexpr int NA = (D + 63) / 64;
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int p = lane + 64 * t;
            if (p < D) ct_w[p] = P.Xcur[c + P.N * p];
        }
        if (lane == 0) ct_w[D] = lp;
    }
    if (lane < DP) sd_w[PS_R * DP + lane] = -0.0;

    const double* rec_ix = P.rec_in + ((int64_t)(D + 1) * P.N + c) * P.rec_stride;
    const int gq1 = qR(0), gq2 = gq1 + qR(1), gq4 = g3 + qR(3);
    [[maybe_unused]] uint64_t ixA[NF], ixB[NF];
#pragma unroll
    for (int t = 0; t < NF; ++t) {
        ixA[t] = (uint64_t)__double_as_longlong(rec_ix[gclamp(fu[t])]);
        ixB[t] = (uint64_t)__double_as_longlong(rec_ix[gclamp(gq1 + fu[t])]);
    }
    // the DMAs of a pass (length Rk, first generation gk; gix: first generation of the pass two after it) into a slot;
    // pack[k]: the row indices the row lanes of DMA k use
    auto issue = [&](int Rk, int gk, int gix, int slot, const uint64_t (&pack)[NDMA]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            uint32_t idx = rwhich[k] ? (uint32_t)(pack[k] >> 32) : (uint32_t)pack[k];
            idx = (ru[k] < Rk) ? idx : 0u;
            const uint32_t gsel = (uint32_t)(ixl[k] ? gclamp(gix) : gclamp(gk));
            const uint64_t dyn = rowl[k] ? (uint64_t)idx * (uint64_t)(ZSC * 8) : (uint64_t)(gsel << 3);
            ps_dma16(sbase[k] + dyn, raw_lds + (unsigned)(slot * SLOTB + k * 1024));
        }
    };
    {
        uint64_t p0[NDMA], p1[NDMA];
#pragma unroll
        for (int k = 0; k < NDMA; ++k) {
            p0[k] = (uint64_t)__double_as_longlong(rec_ix[gclamp(ru[k])]);
            p1[k] = (uint64_t)__double_as_longlong(rec_ix[gclamp(gq1 + ru[k])]);
        }
        // everything loaded so far is in registers, and known to be, before the first DMA (demcz_kernels_ps.h)
        asm volatile("" :: "v"(lp));
#pragma unroll
        for (int t = 0; t < NF; ++t) asm volatile("" :: "v"(eps_p[t]), "v"(ixA[t]), "v"(ixB[t]));
#pragma unroll
        for (int k = 0; k < NDMA; ++k) asm volatile("" :: "v"(p0[k]), "v"(p1[k]));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        issue(qR(0), 0, gq2, 0, p0);
        issue(qR(1), gq1, g3, 1, p1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // history: descriptors moved on every pass; the rows of a pass leave one pass later
    const unsigned char* hx_ptr = reinterpret_cast<const unsigned char*>(hist ? P.chain + (int64_t)P.N * D * P.slot_first : P.Z);
    const unsigned char* hl_ptr = reinterpret_cast<const unsigned char*>(hist ? P.logobj + (int64_t)P.N * P.slot_first : P.Z);
    const uint32_t hx_span = hist ? (uint32_t)((int64_t)D * P.N * 8) : 0u, hl_span = hist ? (uint32_t)(P.N * 8) : 0u;
    double hv[NH];
#pragma unroll
    for (int t = 0; t < NH; ++t) hv[t] = 0.0;
    uint32_t hR = 0;
    auto store_history = [&]() __attribute__((always_inline)) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const uint32_t lim_x = hR * hx_span, lim_l = hR * hl_span;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(hx_ptr), 0, (int)lim_x, 0x00020000);
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(hl_ptr), 0, (int)lim_l, 0x00020000);
#pragma unroll
        for (int t = 0; t < NH; ++t) {
            const unsigned long long vb = (unsigned long long)__double_as_longlong(hv[t]);
            const u32x2 vv = {(unsigned int)vb, (unsigned int)(vb >> 32)};
            __builtin_amdgcn_raw_buffer_store_b64(vv, rx, (int)hx_off[t], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(vv, rl, (int)hl_off[t], 0, 0);
        }
        hx_ptr += lim_x;
        hl_ptr += lim_l;
    };

    int64_t nb = 0;
    unsigned int cnt_total = 0, cnt_first = 0;

    // ---- front end of a pass: increments into LDS, the DMAs of the pass two after it; its log u / temperature
    double logu = 0.0;
    [[maybe_unused]] double temp = 1.0;
    // (what a form lane read stays in LDS -- the slot is not refilled before the pass after next -- and is read again by
    //  whoever needs it later: the LIVE re-reads, the row indices kept for them; registers are short at this d)
    auto write_increment = [&](const double (&za_f)[NF], const double (&zb_f)[NF], const double (&zt_f)[NF]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            const double diff = za_f[t] - zb_f[t];
            const double t1 = scale * diff;
            const double t2 = eps_p[t] * zt_f[t];
            if (fl[t]) sd_w[fu[t] * DP + fp[t]] = t1 + t2;
        }
    };
    // (returns the lanes that found a sentinel as a MASK, taken here where the values are fresh: carried to the end of the pass as a
    //  bool it became a v_cndmask + v_cmp there, on a temporary register the allocator took from LDS reads still in flight -- an
    //  s_waitcnt lgkmcnt(0) in every pass: demcz_kernels_ps2.h, round 5)
    auto front = [&](int slot, int Rn, int R2, int g2, int gix, bool counted) __attribute__((always_inline)) -> unsigned long long {
        const unsigned char* rw = raw_w + slot * SLOTB;
        if (counted) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(VMW) : "memory");
        uint64_t pr[NDMA];
#pragma unroll
        for (int k = 0; k < NDMA; ++k) pr[k] = *reinterpret_cast<const uint64_t*>(rw + ixo + ru[k] * 8);
        bool bad = false;
        double za_f[NF], zb_f[NF], zt_f[NF];
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            za_f[t] = *reinterpret_cast<const double*>(rw + zao[t]);
            zb_f[t] = *reinterpret_cast<const double*>(rw + zbo[t]);
            zt_f[t] = *reinterpret_cast<const double*>(rw + zto[t]);
            if constexpr (LIVE) bad |= fl[t] && fu[t] < Rn && (is_sentinel(za_f[t]) | is_sentinel(zb_f[t]));
        }
        logu = *reinterpret_cast<const double*>(rw + lgo);
        if constexpr (TEMPER) temp = *reinterpret_cast<const double*>(rw + tko);
        write_increment(za_f, zb_f, zt_f);
        const int s2 = (slot + 2 >= PS_SLOTS) ? slot + 2 - PS_SLOTS : slot + 2;
        issue(R2, g2, gix, s2, pr);
        if constexpr (LIVE) return __builtin_amdgcn_ballot_w64(bad);
        return 0ull;
    };
    auto reread = [&](int slot, int Rn, int gpass) __attribute__((always_inline)) -> bool {
        const unsigned char* rw = raw_w + slot * SLOTB;
        double za_f[NF], zb_f[NF], zt_f[NF];
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            za_f[t] = *reinterpret_cast<const double*>(rw + zao[t]);
            zb_f[t] = *reinterpret_cast<const double*>(rw + zbo[t]);
            zt_f[t] = *reinterpret_cast<const double*>(rw + zto[t]);
        }
        int spins = 0;
        while (true) {
            bool bad = false;
#pragma unroll
            for (int t = 0; t < NF; ++t) bad |= fl[t] && fu[t] < Rn && (is_sentinel(za_f[t]) | is_sentinel(zb_f[t]));
            if (__builtin_amdgcn_ballot_w64(bad) == 0ull) break;       // wave-uniform
            if (spins > 0) {
                unsigned row = 0;
#pragma unroll
                for (int t = 0; t < NF; ++t) {
                    if (is_sentinel(za_f[t])) row = (uint32_t)ixA[t];
                    if (is_sentinel(zb_f[t])) row = (uint32_t)(ixA[t] >> 32);
                }
                if (live_poll_abandon(P, spins, bad, row, gpass)) return true;
                __builtin_amdgcn_s_sleep(1);
            } else {
                spins = 1;
            }
#pragma unroll
            for (int t = 0; t < NF; ++t) {
                if (fl[t] && fu[t] < Rn) {
                    if (is_sentinel(za_f[t])) za_f[t] = live_reload(P, &P.Z[(int64_t)(uint32_t)ixA[t] * ZSC + fp[t]]);
                    if (is_sentinel(zb_f[t])) zb_f[t] = live_reload(P, &P.Z[(int64_t)(uint32_t)(ixA[t] >> 32) * ZSC + fp[t]]);
                }
            }
        }
        wave_lds_handoff();
        write_increment(za_f, zb_f, zt_f);
        wave_lds_handoff();
        return false;
    };

    {
        const unsigned long long bad0 = front(0, qR(0), qR(2), gq2, gq4, false);
        if constexpr (LIVE) {
            if (bad0 != 0ull) {
                if (reread(0, qR(0), 0)) { leave(); return; }
            }
        }
#pragma unroll
        for (int t = 0; t < NF; ++t) { ixA[t] = ixB[t]; ixB[t] = *reinterpret_cast<const uint64_t*>(raw_w + ixo + fu[t] * 8); }
    }
#ifdef DEMCZ_STAMPS
    // diagnostic build (scripts/pw_stamps.py): shader-clock sums per segment of a pass (a stamp drains the wave's outstanding LDS /
    // scalar-memory operations: read the segments as proportions)
    unsigned long long sa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long sa_start = __builtin_readcyclecounter();
#define PW_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); sa[i] += t_ - sa_t; sa_t = t_; } while (0)
#else
#define PW_T(i) do { } while (0)
#endif
    int slot = 1;
    for (int ip = 0; ip < npass; ++ip) {
#ifdef DEMCZ_STAMPS
        unsigned long long sa_t = __builtin_readcyclecounter();
#endif
        const int R = qR(0);
        bool bnd;                      // a generation divisible by K ends this pass
        if constexpr (REG) { bnd = (--tb == 0); if (bnd) tb = tbK; } else { bnd = (segq & 8u) != 0u; }
        [[maybe_unused]] unsigned int pub_seen = 0u;
        if constexpr (LIVE) {
            if (bnd) pub_seen = __hip_atomic_load(&pub_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const double logu_c = logu;
        [[maybe_unused]] const double temp_c = temp;
        wave_lds_handoff();
        [[maybe_unused]] mf4 macc[2][2];          // MF: candidates, then residuals: [row tile][column tile]
        if constexpr (MF) {
            // C = x 1' + Delta T: the state down the accumulator's rows, the pass's increments as the A operand
            lp = ct_w[D];
            mf4 x0, x1;
#pragma unroll
            for (int v = 0; v < 4; ++v) x0[v] = ct_w[4 * v + mg];
            x1[0] = (16 + mg < D) ? ct_w[16 + mg] : 0.0; x1[1] = 0.0; x1[2] = 0.0; x1[3] = 0.0;
            double da[2][2];
            bool nonfinite = false;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int pp = 16 * rt + mc, u = 4 * ks + mg;
                    const bool ok = pp < D && u < PS_R;
                    da[rt][ks] = ok ? sd_w[(ok ? u : 0) * DP + (ok ? pp : 0)] : 0.0;
                    nonfinite |= !(fabs(da[rt][ks]) < __builtin_inf());
                }
            if (__builtin_amdgcn_ballot_w64(nonfinite) != 0ull) {
                // 0 * inf would reach candidates that never take that generation: not this kernel's case -- the launch is
                // flagged like a timed-out hand-off and redone by the scalar kernels (live_verify)
                if (lane == 0 && atomicCAS(P.live_err, 0u, 1u) == 0u) { P.live_err[1] = (unsigned)g0; P.live_err[2] = 0xffffffffu; P.live_err[3] = blockIdx.x; }
                leave();
                return;
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                macc[0][ct] = x0;
                macc[1][ct] = x1;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    macc[0][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(da[0][ks], mT[ct][ks], macc[0][ct], 0, 0, 0);
                    macc[1][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(da[1][ks], mT[ct][ks], macc[1][ct], 0, 0, 0);
                }
            }
        }
        // every node's candidate: state + its rows, in order, straight from LDS
        double cand[MF ? 1 : D];
        if constexpr (!MF) {
#pragma unroll
        for (int q = 0; q < CR / 2; ++q) {
            const double2 t = reinterpret_cast<const double2*>(ct_w)[q];      // (wave-uniform address)
            if (2 * q < D) cand[2 * q] = t.x;
            if (2 * q == D) lp = t.x;
            if (2 * q + 1 < D) cand[2 * q + 1] = t.y;
            if (2 * q + 1 == D) lp = t.y;
        }
        if constexpr (DDPP) {
            // the increments are the same for every node: per generation they sit in a few register pairs by lanes and are added
            // by the lanes whose path takes the generation, straight out of a taker lane of their row (scripts/gen_pw_wdpp.py,
            // gen_adds) -- 22 8-byte LDS reads a pass instead of 50 16-byte ones, and no instruction for a generation not taken
            double Dg[PS_R][DD_NQMAX];
#pragma unroll
            for (int u = 0; u < PS_R; ++u)
#pragma unroll
                for (int q = 0; q < DD_NQMAX; ++q)
                    if (q < DD_NQ[u]) Dg[u][q] = dptr[u][q * DD_KN[u]];
#pragma unroll
            for (int u = 0; u < PS_R; ++u)
#pragma unroll
                for (int q = 0; q < DD_NQMAX; ++q)
                    if (q < DD_NQ[u]) asm volatile("" : "+v"(Dg[u][q]));
#include "demcz_pw_ddpp_sel.inc"
        } else {
#pragma unroll
        for (int j = 0; j < PS_R; ++j) {
#pragma unroll
            for (int q = 0; q < DP / 2; ++q) {
                const double2 t = reinterpret_cast<const double2*>(mrow[j])[q];
                cand[2 * q] = cand[2 * q] + t.x;
                if (2 * q + 1 < D) cand[2 * q + 1] = cand[2 * q + 1] + t.y;
            }
            // (one generation's row at a time, its adds done before the next row is asked for: left alone the compiler
            //  fetches all five rows first -- 200 registers, spilled -- and adds after the front end; a rolling window of one
            //  row's worth of pieces across the rows ends the same way: 233 registers / 115 spilled)
#pragma unroll
            for (int p = 0; p < D; ++p) asm volatile("" : "+v"(cand[p]));
        }
        }
        }
        PW_T(0);                 // state row + candidate adds straight from LDS
        wave_lds_handoff();      // (the front end below rewrites the increments)
        store_history();
        const unsigned long long bad_n = front(slot, qR(1), qR(3), g3, g5, true);
        PW_T(1);                 // history stores, DMA wait, next pass's increments, DMA issue
        // the candidates go to rows 1..31 of the table (row 0 keeps the state the pass started from); the log-density follows
        double lpp;
        if constexpr (MF) {
            // table rows from the accumulators (element v of row tile rt <-> parameter 16 rt + 4 v + g, candidate 16 ct + c16)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int j = 16 * ct + mc;
                if (j != 0) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) ct_w[j * CR + 4 * v + mg] = macc[0][ct][v];
                    if (16 + mg < D) ct_w[j * CR + 16 + mg] = macc[1][ct][0];
                }
            }
            // R = C - mu 1', in place; Y = W R; q = diag(Y' Y)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
                for (int v = 0; v < 4; ++v) macc[0][ct][v] = macc[0][ct][v] - mmu0[v];
                macc[1][ct][0] = macc[1][ct][0] - mmu1;
            }
            double qd[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                mf4 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(mW[0][ks], macc[0][ct][ks], y0, 0, 0, 0);
                    y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mW[1][ks], macc[0][ct][ks], y1, 0, 0, 0);
                }
                y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(mW[1][4], macc[1][ct][0], y1, 0, 0, 0);
                mf4 qq = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qq = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[ks], y0[ks], qq, 0, 0, 0);
                qq = __builtin_amdgcn_mfma_f64_16x16x4f64(y1[0], y1[0], qq, 0, 0, 0);
                // the diagonal: candidate c16 of this column tile is row 4 v + g of the accumulator when c16 == 4 v + g
                const int vq = mc >> 2;
                qd[ct] = (vq == 0) ? qq[0] : (vq == 1) ? qq[1] : (vq == 2) ? qq[2] : qq[3];
            }
            if ((mc & 3) == mg) {          // a diagonal lane: the log-densities of candidates c16 and 16 + c16 go to the table
                if (mc != 0) ct_w[mc * CR + D] = fma(-0.5, qd[0], P.tp.c0);
                ct_w[(16 + mc) * CR + D] = fma(-0.5, qd[1], P.tp.c0);
            }
            wave_lds_handoff();
            lpp = ct_w[(nodel ? lane : 0) * CR + D];
        } else {
        if (nodel) {
#pragma unroll
            for (int q = 0; q < D / 2; ++q) reinterpret_cast<double2*>(ct_w + lane * CR)[q] = make_double2(cand[2 * q], cand[2 * q + 1]);
            if constexpr (D & 1) ct_w[lane * CR + D - 1] = cand[D - 1];
        }
        // The log-density: target_logp's operation sequence; mu from the workgroup's LDS copy
        {
            // W by lanes: the register pairs are asked for here, in front of the reads of mu, and pinned behind the subtraction --
            // one LDS round trip for all of them (left alone the compiler fetches a block's pairs right in front of the block:
            // five round trips a pass)
            [[maybe_unused]] double Wr[WDPP ? NWR : 1];
            if constexpr (WDPP) {
#pragma unroll
                for (int r = 0; r < NWR; ++r) Wr[r] = Wl[r * 16 + (lane & 15)];
            }
            double (&rr)[D] = cand;           // (the candidate itself is in the table by now)
#pragma unroll
            for (int q = 0; q < DP / 2; ++q) {
                const double2 t = reinterpret_cast<const double2*>(mul)[q];
                rr[2 * q] = cand[2 * q] - t.x;
                if (2 * q + 1 < D) rr[2 * q + 1] = cand[2 * q + 1] - t.y;
            }
            // W through SCALAR loads issued here, every pass: the address is an opaque integer turned into a constant-address-
            // space pointer (as a loop invariant the compiler would hold all D (D + 1) / 2 entries in registers; through an
            // opaque generic pointer it fetches them with per-lane flat loads).  A pass waits ~27 scalar-cache round trips
            // for them.  Measured alternatives at C4's shard, us per K-window: these scalar loads 7.25; W spread over the
            // lanes' registers and read with v_readlane (420 extra vector instructions) 7.7; an LDS copy read with
            // wave-uniform addresses 11.9 (the CU's four chain waves run into the LDS bandwidth); the 16-lane kernel 10.2.
            // Two rows' fma chains interleaved (independent fmas issue every 4 clocks, dependent ones every 8): 7.0 against
            // 6.8-6.9 -- it is the waits for the scalar loads, not the fma latency, that the 2800-3000 clocks of this section
            // are made of (profiles/r03e_pw_stamps.txt), and the second chain cost 40 more SGPR spills.
            typedef const __attribute__((address_space(4))) double* cptr;
            uint64_t wa = (uint64_t)(uintptr_t)P.tp.Wp;
            asm volatile("" : "+s"(wa));
            const cptr Wc = (cptr)wa;
            auto wentry = [&](int e) __attribute__((always_inline)) { return Wc[e]; };
            double q = 0.0;
            if constexpr (WDPP) {
                // (round 4: W by lanes -- see the top of the file; the alternatives above are what it replaced)
#pragma unroll
                for (int r = 0; r < NWR; ++r) asm volatile("" : "+v"(Wr[r]));
#include "demcz_pw_wdpp_sel.inc"
                lpp = fma(-0.5, q, P.tp.c0);
            } else if constexpr (TARGET == TARGET_MVNORMAL) {
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    double acc = wentry((i * (i + 1)) / 2) * rr[0];
#pragma unroll
                    for (int j = 1; j <= i; ++j) acc = fma(wentry((i * (i + 1)) / 2 + j), rr[j], acc);
                    q = (i == 0) ? acc * acc : fma(acc, acc, q);
                }
                lpp = fma(-0.5, q, P.tp.c0);
            } else {
#pragma unroll
                for (int i = 0; i < D; ++i) q = (i == 0) ? rr[i] * rr[i] : fma(rr[i], rr[i], q);
                lpp = -q;
            }
        }
        if (nodel) ct_w[lane * CR + D] = lpp;
        }
        PW_T(2);                 // table write, log-density (W through scalar loads)
        unsigned long long mask, chg_a, chg_r;
        {
            const unsigned long long lb = (unsigned long long)__double_as_longlong((lane == 0) ? lp : lpp);
            const unsigned int blo = (unsigned int)__builtin_amdgcn_ds_bpermute(anc4, (int)(unsigned int)lb);
            const unsigned int bhi = (unsigned int)__builtin_amdgcn_ds_bpermute(anc4, (int)(unsigned int)(lb >> 32));
            const double lpb = __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));
            const double d0 = lpp - lpb;
            double dlt = d0;
            if constexpr (TEMPER) dlt = dlt / temp_c;
            mask = __builtin_amdgcn_ballot_w64(logu_c < dlt);
            chg_a = __builtin_amdgcn_fcmp(d0, 0.0, 14 /* UNE */);
            chg_r = __builtin_amdgcn_fcmp(lpb - lpb, 0.0, 14);
        }
        const unsigned int m32 = (unsigned int)mask;
        const bool onp = nodel && lev <= R && (m32 & need1) == need1 && (m32 & need0) == 0u;
        const unsigned int path = (unsigned int)__builtin_amdgcn_ballot_w64(onp);
        const unsigned int accp = path & m32;
        const unsigned int win = accp ? 31u - (unsigned int)__builtin_clz(accp) : 0u;
        {
            const unsigned int chm = (accp & (unsigned int)chg_a) | (path & ~m32 & (unsigned int)chg_r);
            cnt_total += (unsigned int)__builtin_popcount(chm);
            if (g0 == 0) cnt_first = (chm >> 1) & 1u;
        }
        PW_T(3);                 // bpermute, accept tests, path
        wave_lds_handoff();
        // history rows of the pass (read now, stored during the next pass), then the winner's row becomes row 0
#pragma unroll
        for (int t = 0; t < NH; ++t) {
            const unsigned int wa = accp & hmask[t];
            const unsigned int wj = wa ? 31u - (unsigned int)__builtin_clz(wa) : 0u;
            hv[t] = ct_w[wj * CR + hp[t]];
        }
        hR = (uint32_t)R;
        wave_lds_handoff();
        if (win != 0u) {            // wave-uniform
            if (lane < CR / 2) reinterpret_cast<double2*>(ct_w)[lane] = reinterpret_cast<const double2*>(ct_w + win * CR)[lane];
        }
        wave_lds_handoff();
        PW_T(4);                 // history values, winner's row to row 0
        // a generation divisible by K ended the pass: runchain!'s append, demcz.jl:88-91
        if (bnd) {
            constexpr int NA = (D + 63) / 64;
#pragma unroll
            for (int t = 0; t < NA; ++t) {
                const int p = lane + 64 * t;
                const double v = ct_w[(p < D) ? p : 0];
                if constexpr (LIVE) {
                    if (t == 0) {
                        asm volatile("" : "+v"(pub_seen));
                        while (pub_seen + (unsigned int)PS_PUB <= (unsigned int)nb) {
                            __builtin_amdgcn_s_sleep(1);
                            pub_seen = __hip_atomic_load(&pub_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    if (p < D) pub_rows[(w * PS_PUB + (int)((unsigned int)nb % PS_PUB)) * D + p] = v;
                } else {
                    if (p < D && P.do_append) P.Zw[(P.M_append + nb * P.N + c) * P.ZS + p] = v;
                }
                if (p < D && P.snap) P.snap[nb * P.N * D + c + P.N * p] = v;
            }
            if constexpr (LIVE) {
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_store(&pub_seq[w], (unsigned int)nb + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            ++nb;
        }
        PW_T(5);                 // boundary
        if constexpr (LIVE) {
            // (REG: every pass counts as five generations, the one BEHIND the launch's last too -- its slot holds whatever the
            //  record buffer has past this launch's draws, which after a rollback, a demcz_set_state or a discarded slab are row
            //  indices of a LONGER archive than there is: never waited for.  Found in round 5 by the re-arming tests: a LIVE launch
            //  behind a redo polled 2^18 times for a row nobody was going to write.  Without REG that pass has length 0.)
            if ((!REG || ip + 1 < npass) && __builtin_expect(bad_n != 0ull, 0)) {
                if (reread(slot, qR(1), g0 + R)) { leave(); return; }
            }
        }
        PW_T(6);                 // waits for rows not yet published
        wave_lds_handoff();
        g0 += R;
        g3 += qR(3);
        g5 += qR(5);
        if constexpr (!REG) segq = (segq >> 4) | (seg_make() << 20);
#pragma unroll
        for (int t = 0; t < NF; ++t) { ixA[t] = ixB[t]; ixB[t] = *reinterpret_cast<const uint64_t*>(raw_w + slot * SLOTB + ixo + fu[t] * 8); }
        slot = (slot + 1 == PS_SLOTS) ? 0 : slot + 1;
        PW_T(7);                 // queue bookkeeping
    }
#ifdef DEMCZ_STAMPS
    if (P.stamps && lane == 0 && c < 65536) {
        unsigned long long* o = P.stamps + (size_t)c * 16;
        for (int i = 0; i < 8; ++i) o[i] = sa[i];
        o[8] = __builtin_readcyclecounter() - sa_start; o[14] = (unsigned long long)npass; o[15] = 3;
    }
#endif
#undef PW_T
    store_history();
    {
        constexpr int NA = (D + 63) / 64;
#pragma unroll
        for (int t = 0; t < NA; ++t) {
            const int p = lane + 64 * t;
            const double v = ct_w[(p < D) ? p : 0];
            if (p < D) P.Xcur[c + P.N * p] = v;
        }
        if (lane == 0) P.lpcur[c] = ct_w[D];           // (row 0, not the register: that one is the last pass's starting value)
    }
    wave_store_counts(P, c, cnt_total, cnt_first);
    leave();
}

}  // namespace demcz
