// demcz_kernels_ps3.h -- K1g3: window_kernel_ps2's LIVE launch with the pass's FRONT END on a wave of its own.
//
// What window_kernel_ps2 measured (profiles/r03b_floor_experiments.txt, r03d_helper_wave_probe.txt): a chain wave that issues
// no DMA runs 1000 generations in 99 us whatever the archive's size; with its DMA, 104 us at a 1024-row archive and 157 us at
// 2 M rows -- and not because it WAITS for rows (dropping the waits changes nothing): the DMA instruction itself stalls at
// issue while the rows of earlier passes are still missing in L2, and one wave per SIMD issues in order, so everything behind
// the DMA stalls with it.  The same DMA stream issued by a SECOND wave on the same SIMD costs the chain wave 5 us, not 58.
//
// A DE-MCz proposal's increment gamma (z_i1 - z_i2) + eps n does not depend on the chain's state (demcz.jl:184-188: the state
// enters as X + delta only), so everything up to the increments can run ahead of the chain on another wave:
//
//   helper wave H (one per chain, same SIMD, lower priority)          chain wave C (one per chain)
//   ---------------------------------------------------------          ------------------------------------------------
//   DMA of pass p + AHEAD: 10 archive rows + the record fields         candidate = state + the node's increments in order
//   wait for pass p's DMA; rows, normals, log u out of the raw slot    log-density, ancestor's by bpermute, accept tests,
//   LIVE: rows that show the sentinel polled (sc1) until published     path, winner, new state from the candidate table
//   increments (and log u) into slot p mod 4 of an LDS ring, then      history store, boundary row to the publisher wave
//   the slot's tag = p + 1                                             reads ring slot (p+1) mod 4 in the shadow of its
//   (before overwriting a slot: C has finished reading it)             bpermute; a stale tag -> read again at the pass's end
//
// C issues no vector-memory load at all (one history store per pass): nothing of the memory system's latency is in its
// instruction stream, and a row that is late costs C only the part of the wait H could not take ahead of it.  H polls from the
// moment its DMA lands -- two passes before C needs the increments -- instead of at the end of the pass before.
//
// Order inside LDS: a wave's LDS operations execute in issue order, so H's data writes are in LDS before its tag write, and C's
// data reads, issued behind its tag read, see them if the tag read saw the tag.  No barrier, no atomics.
//
// Deadlock freedom (LIVE): C hands its boundary row to the publisher before it blocks on a tag; H_A waits only for rows of
// boundaries C_A's pass is beyond or at, which their owners publish without waiting for A (induction over boundaries, as in
// demcz_kernels_rec.h).  H blocks on C only when it is NS passes ahead, i.e. when C has NS passes ready: never both ways.
// A wait that is given up (live_poll_abandon) is H's: it raises abort[chain] for C, which leaves with the launch's error word
// set; C leaving early releases H through done[chain].
//
// The arithmetic is window_kernel_ps2's -- the same additions in the same order on the same values -- hence the oracle's.
// Regular LIVE launches only (ps2_applicable + co-residency of nine-wave workgroups: ps3_applicable, demcz_capi.hip).
//
// MEASURED, AND NOT THE DEFAULT (DEMCZ_PS3=1 selects it; profiles/r03d_helper_wave.txt).  Alone on the GPU the launch is 5-9 %
// faster than window_kernel_ps2's at a 1-4 M row archive with AHEAD = 4 (150 vs 162 us, 178 vs 193 us per 1000 generations)
// and equal at 100 k rows, where the waits are for rows not yet PUBLISHED and no wave can take those ahead.  In the bench --
// with the R-hat monitor's kernels and the history copies beside it -- it is SLOWER (182 vs 164 us): a nine-wave workgroup
// fills its CU, so when another kernel's workgroups reach a CU first, that CU's four chains start late and, through the
// hand-off, hold every other chain back.  The premise was wrong too: scripts/probes/lds_dma_mlp.hip shows a CU keeps many
// gathers in flight (one wave: 690 clocks per gather round at depth 1, 125 at depth 8), so the DMA is not served one at a time.
#pragma once

#include "demcz_kernels_ps2.h"

#pragma clang fp contract(off)

namespace demcz {

#ifndef PS3_AHEAD_N
#define PS3_AHEAD_N 4
#endif
constexpr int PS3_AHEAD = PS3_AHEAD_N;      // H's DMA of a pass is issued this many passes before H consumes its slot
constexpr int PS3_NS = 4;                   // ring slots of increments between H and C (the pass loop of C is unrolled by them)
static_assert(PS3_AHEAD >= 1 && PS3_AHEAD <= 10, "record rows are read AHEAD passes past the pass being formed");
#ifndef PS3_PUBS
#define PS3_PUBS 1                          // publisher waves: 1 = one for the workgroup's four chains, PS_CHAINS = one per chain (measured: no difference)
#endif
constexpr int PS3_WAVES = 2 * PS_CHAINS + PS3_PUBS;

template <int TARGET, int D, bool TEMPER>
__global__ void __launch_bounds__(64 * PS3_WAVES, 1) window_kernel_ps3(const WindowParams P)
{
    static_assert(TARGET == TARGET_MVNORMAL || TARGET == TARGET_ISO_QUAD, "split layout: MvNormal / isotropic targets");
    static_assert(D >= 2 && D <= 5, "a pass's rows, normals, log u and indices are one 64-lane DMA");
    constexpr int R = PS2_R;
    constexpr int HW = (D + 1) / 2;                        // 16-byte pieces of an archive row
    constexpr int ZSC = (D <= 2) ? 2 : (D <= 4) ? 4 : 8;   // archive row stride in doubles (demcz_create: ZS)
    constexpr int ZSH = (ZSC == 2) ? 4 : (ZSC == 4) ? 5 : 6;      // log2 of the row stride in bytes
    constexpr int DP = ((D + 1) / 2) * 2;                  // increments row in LDS
    constexpr int CR = ((D + 2) / 2) * 2;                  // candidate row in LDS: D doubles, log-density, pad
    // lanes of H's DMA: [0, ROWL) archive rows (generation u, first / second row, piece j); then NF record fields of three
    // pieces (six generations) each: 0..D-1 normals, D log u, D+1 this pass's row indices, D+2 the row indices of the pass
    // AHEAD after this one, D+3 temperatures; the rest idle (they fetch row 0)
    constexpr int ROWL = R * 2 * HW;
    constexpr int FL0 = ROWL;
    constexpr int NF = D + 3 + (TEMPER ? 1 : 0);
    constexpr int TL0 = FL0 + 3 * NF;
    static_assert(TL0 <= 64, "one DMA instruction per pass");
    constexpr int F_LOGU = D, F_IXOWN = D + 1, F_IXNEXT = D + 2, F_TEMP = D + 3;
    // a ring slot, in doubles: R rows of increments, the row of negative zeros, a pad (idle lanes' writes), six log u,
    // six temperatures, the tag
    constexpr int SL_NEG = R * DP, SL_PAD = SL_NEG + DP, SL_LOGU = SL_PAD + 2, SL_TEMP = SL_LOGU + 6, SL_TAG = SL_TEMP + 6;
    constexpr int SLD = SL_TAG + 2;
    static_assert(SLD % 2 == 0 && SL_LOGU % 2 == 0, "16-byte reads of the rows");

    typedef __attribute__((address_space(3))) unsigned int lds_u32;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#if PS2_XCD_SWIZZLE
    const int bxs = xcd_block(P);          // XCD x runs the x-th eighth of the chains: demcz_kernels.h
#else
    const int bxs = (int)blockIdx.x;
#endif
    __shared__ __attribute__((aligned(16))) unsigned char raw[PS_CHAINS][PS3_AHEAD][1024];
    __shared__ __attribute__((aligned(16))) double ring[PS_CHAINS][PS3_NS][SLD];
    __shared__ __attribute__((aligned(16))) double ctab[PS_CHAINS][64 * CR];       // row l: lane l's candidate (rows 32..63 shadow 0..31)
    __shared__ double pub_rows[PS_CHAINS * PS_PUB * D];
    __shared__ unsigned int pub_seq[PS_CHAINS], pub_done[PS_CHAINS], pub_exit[PS_CHAINS];
    __shared__ int hc_done[PS_CHAINS], hc_abort[PS_CHAINS];
    if (threadIdx.x < PS_CHAINS) {
        pub_seq[threadIdx.x] = 0u; pub_done[threadIdx.x] = 0u; pub_exit[threadIdx.x] = 0u;
        hc_done[threadIdx.x] = 0; hc_abort[threadIdx.x] = 0;
    }
    if (threadIdx.x < PS_CHAINS * PS3_NS) *reinterpret_cast<unsigned int*>(&ring[threadIdx.x / PS3_NS][threadIdx.x % PS3_NS][SL_TAG]) = 0u;
    __syncthreads();

    if (w >= 2 * PS_CHAINS) {
        // the publisher waves: demcz_kernels_ps.h's protocol (a publisher never leaves before its chain waves).  One per chain
        // here: a wave's write-through store holds back its NEXT vector-memory instruction for the store's round trip, so one
        // publisher for four chains sends the rows of chain waves that reach the boundary a little apart one round trip apart.
        constexpr int CPP = PS_CHAINS / PS3_PUBS;            // chains per publisher wave
        const bool pl = lane < CPP * D;
        const int cw = (w - 2 * PS_CHAINS) * CPP + (pl ? lane / D : 0), pp = pl ? lane % D : 0;
        const int64_t cl = (int64_t)bxs * PS_CHAINS + cw;
        unsigned int done = 0u;
        while (true) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int seq = __hip_atomic_load(&pub_seq[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const bool ready = pl && seq != done;
            if (__builtin_amdgcn_ballot_w64(ready) != 0ull) {
                if (ready) {
                    const double v = pub_rows[(cw * PS_PUB + (int)(done % PS_PUB)) * D + pp];
                    if (cl < P.N && P.do_append) live_store(&P.Zw[(P.M_append + (int64_t)done * P.N + cl) * P.ZS + pp], v);
                    ++done;
                }
                asm volatile("" ::: "memory");
                if (ready && pp == 0) __hip_atomic_store(&pub_done[cw], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                continue;
            }
            const bool gone = !pl || (__hip_atomic_load(&pub_exit[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u &&
                                      __hip_atomic_load(&pub_seq[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == done);
            if (__builtin_amdgcn_ballot_w64(!gone) == 0ull) break;
            __builtin_amdgcn_s_sleep(1);
        }
        return;
    }

    const int npass = P.ngen / R;

    if (w >= PS_CHAINS) {
        // ================================================ helper wave H =======================================================
        const int cw = w - PS_CHAINS;
        const int64_t c = (int64_t)bxs * PS_CHAINS + cw;
        if (c >= P.N) return;
        auto give_up = [&]() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&hc_abort[cw], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        if (__hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { give_up(); return; }
        __builtin_amdgcn_s_setprio(1);
        unsigned char* const raw_w = &raw[cw][0][0];
        double* const ring_w = &ring[cw][0][0];
        const unsigned raw_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)raw_w);
        // increments: lane (u, p) forms element p of generation u of the pass
        const bool fl = lane < R * D;
        const int fu = fl ? lane / D : 0, fp = fl ? lane % D : 0;
        const int zao = ((fu * 2) * HW) * 16 + fp * 8;                      // second row: + HW * 16
        const int zto = (FL0 + 3 * fp) * 16 + fu * 8;
        const int ixown = (FL0 + 3 * F_IXOWN) * 16 + fu * 8;                // this pass's row indices (re-reads)
        const double eps_p = P.eps[fp];
        const double scale = P.gamma / sqrt((double)(2 * D));
        const int incw = fl ? fu * DP + fp : SL_PAD;                        // (idle lanes: the pad)
        // log u and temperatures: lanes 0..5 copy the six values of the field
        const int lgr = (FL0 + 3 * F_LOGU) * 16 + ((lane < 6) ? lane : 0) * 8;
        [[maybe_unused]] const int tkr = (FL0 + 3 * F_TEMP) * 16 + ((lane < 6) ? lane : 0) * 8;
        const int lgw = (lane < 6) ? SL_LOGU + lane : SL_PAD + 1;
        [[maybe_unused]] const int tkw = (lane < 6) ? SL_TEMP + lane : SL_PAD + 1;
        // DMA: rows (ru, which, piece), record fields (f, piece)
        const bool rowl = lane < ROWL;
        const int ru = rowl ? lane / (2 * HW) : 0, rwhich = rowl ? (lane / HW) % 2 : 0, rj = rowl ? lane % HW : 0;
        const bool fieldl = lane >= FL0 && lane < TL0;
        const int ff = fieldl ? (lane - FL0) / 3 : 0, fj = fieldl ? (lane - FL0) % 3 : 0;
        const int ixnext = (FL0 + 3 * F_IXNEXT) * 16 + ru * 8;              // where a row lane finds the indices of the pass AHEAD on
        const unsigned int selv = !rowl ? 0x0c0c0c0cu : (rwhich ? 0x07060504u : 0x03020100u);
        const unsigned char* const zbase = reinterpret_cast<const unsigned char*>(P.Z);
        unsigned int dma_off, dma_inc;
        {
            const unsigned int rec_off = (unsigned int)(reinterpret_cast<const unsigned char*>(P.rec_in) - zbase);
            if (rowl) { dma_off = (unsigned int)rj * 16u; dma_inc = 0u; }
            else if (fieldl && ff != F_TEMP) {
                const int rf = (ff == F_IXOWN || ff == F_IXNEXT) ? D + 1 : ff;
                dma_off = rec_off + (unsigned int)((((int64_t)rf * P.N + c) * P.rec_stride + (ff == F_IXNEXT ? PS3_AHEAD * R : 0)) * 8) + (unsigned int)fj * 16u;
                dma_inc = (unsigned int)(R * 8);
            } else if (fieldl) {
                dma_off = (unsigned int)(reinterpret_cast<const unsigned char*>(P.temperature) - zbase) + (unsigned int)fj * 16u;
                dma_inc = (unsigned int)(R * 8);
            } else { dma_off = 0u; dma_inc = 0u; }
        }
        // the row of negative zeros of every ring slot (a node adds it for a generation it does not take: x + (-0.0) == x)
        if (lane < DP) {
#pragma unroll
            for (int s = 0; s < PS3_NS; ++s) ring_w[s * SLD + SL_NEG + lane] = -0.0;
        }
        // the first AHEAD passes' row indices by ordinary loads
        const double* rec_ix = P.rec_in + ((int64_t)(D + 1) * P.N + c) * P.rec_stride;
        uint64_t pp0[PS3_AHEAD];
#pragma unroll
        for (int k = 0; k < PS3_AHEAD; ++k) pp0[k] = (k < npass) ? (uint64_t)__double_as_longlong(rec_ix[k * R + ru]) : 0ull;
#pragma unroll
        for (int k = 0; k < PS3_AHEAD; ++k) asm volatile("" :: "v"(pp0[k]));
        asm volatile("" :: "v"(eps_p));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        auto issue = [&](uint64_t pack, unsigned slot) __attribute__((always_inline)) {
            const unsigned int sel = __builtin_amdgcn_perm((unsigned int)(pack >> 32), (unsigned int)pack, selv);
            const unsigned int off = (sel << ZSH) + dma_off;
            dma_off += dma_inc;
            ps2_dma16(zbase, off, raw_lds + slot * 1024u);
        };
#pragma unroll
        for (int k = 0; k < PS3_AHEAD; ++k)
            if (k < npass) issue(pp0[k], (unsigned)k);

#ifdef DEMCZ_STAMPS
        unsigned long long hs[4] = {0, 0, 0, 0}, hs_n = 0;      // cycles: DMA wait, re-reads, room in the ring, everything
        const unsigned long long hs_start = __builtin_readcyclecounter();
#define PS3_H(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); hs[i] += t_ - hs_t; hs_t = t_; } while (0)
#else
#define PS3_H(i) do { } while (0)
#endif
        unsigned slot = 0u;                  // raw slot of pass hp: hp mod AHEAD
        for (int hp = 0; hp < npass; ++hp) {
#ifdef DEMCZ_STAMPS
            unsigned long long hs_t = __builtin_readcyclecounter();
#endif
            // outstanding: the DMAs of passes hp .. hp + AHEAD - 1 (fewer at the launch's end)
            if (hp + PS3_AHEAD <= npass) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PS3_AHEAD - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PS3_H(0);
            const unsigned char* rw = raw_w + slot * 1024u;
            double za = *reinterpret_cast<const double*>(rw + zao);
            double zb = *reinterpret_cast<const double*>(rw + zao + HW * 16);
            const double zt = *reinterpret_cast<const double*>(rw + zto);
            const uint64_t pr = *reinterpret_cast<const uint64_t*>(rw + ixnext);
            const uint64_t ix = *reinterpret_cast<const uint64_t*>(rw + ixown);
            const double lg = *reinterpret_cast<const double*>(rw + lgr);
            [[maybe_unused]] double tk = 1.0;
            if constexpr (TEMPER) tk = *reinterpret_cast<const double*>(rw + tkr);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" :: "v"(za), "v"(zb), "v"(zt), "v"(pr), "v"(ix), "v"(lg), "v"(tk));
            wave_lds_handoff();
            // the slot's values are in registers: the DMA of pass hp + AHEAD refills it
            if (hp + PS3_AHEAD < npass) issue(pr, slot);
            // rows other waves had not published when the DMA read them: asked for again (sc1 loads) until they are there
            bool bad = fl && (is_sentinel(za) | is_sentinel(zb));
            PS3_H(3);
            if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
#ifdef DEMCZ_STAMPS
                ++hs_n;
#endif
                const uint32_t i1 = (uint32_t)ix, i2 = (uint32_t)(ix >> 32);
                int spins = 0;
                bool first = true;
                while (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
                    if (!first) {
                        if (live_poll_abandon(P, spins, bad, is_sentinel(za) ? i1 : i2, hp * R)) { give_up(); return; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    first = false;
                    if (bad) {
                        if (is_sentinel(za)) za = live_load(&P.Z[(int64_t)i1 * ZSC + fp]);
                        if (is_sentinel(zb)) zb = live_load(&P.Z[(int64_t)i2 * ZSC + fp]);
                        bad = is_sentinel(za) | is_sentinel(zb);
                    }
                }
            }
            PS3_H(1);
            // room in the ring: C has finished reading the slot's previous contents (pass hp - NS)
            while (__hip_atomic_load(&hc_done[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + PS3_NS <= hp) __builtin_amdgcn_s_sleep(1);
            PS3_H(2);
            double* const rs = ring_w + (hp & (PS3_NS - 1)) * SLD;
            const double diff = za - zb;
            const double t1 = scale * diff;
            const double t2 = eps_p * zt;
            rs[incw] = t1 + t2;                                   // (every lane stores: the lanes that form nothing write the pad)
            rs[lgw] = lg;
            if constexpr (TEMPER) rs[tkw] = tk;
            wave_lds_handoff();
            if (lane == 0) *(lds_u32*)(rs + SL_TAG) = (unsigned int)hp + 1u;       // (an LDS store: a volatile generic one is a FLAT store)
            wave_lds_handoff();
            slot = (slot + 1u == (unsigned)PS3_AHEAD) ? 0u : slot + 1u;
            PS3_H(3);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef DEMCZ_STAMPS
        if (P.stamps && lane == 0 && c < 65536) {
            unsigned long long* o = P.stamps + (size_t)c * 16;
            o[7] = hs[0]; o[9] = hs[1]; o[10] = hs[2]; o[12] = hs_n; o[13] = __builtin_readcyclecounter() - hs_start;
        }
#endif
#undef PS3_H
        return;
    }

    // ==================================================== chain wave C ========================================================
    auto leave = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            __hip_atomic_store(&hc_done[w], 0x3fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // (H never waits for room again)
            __hip_atomic_store(&pub_exit[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    const int64_t c = (int64_t)bxs * PS_CHAINS + w;
    if (c >= P.N) {
        wave_store_counts(P, c, 0u, 0u);
        leave();
        return;
    }
    if (__hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { leave(); return; }
    __builtin_amdgcn_s_setprio(3);
    double* const ring_w = &ring[w][0][0];
    double* const ct_w = &ctab[w][0];

    // ---- what this lane is, in each of its parts ----------------------------------------------------------------
    // node of the tree of outcomes: nn = 0 is the state itself (lanes 0 and 32), 1..31 the nodes (lanes 32..63 shadow 0..31)
    const int nn = lane & 31;
    const int lev = nn ? 32 - __builtin_clz((unsigned)nn) : 0;
    const double* mrow[R];                 // rows of ring slot 0 this node adds, in order: an accepted generation on its path or
#pragma unroll                             // its own -> that generation's increments, anything else -> the row of negative zeros
    for (int j = 1; j <= R; ++j) {
        const bool take = nn != 0 && ((j == lev) || (j < lev && ((nn >> (lev - 1 - j)) & 1)));
        mrow[j - 1] = ring_w + (take ? (j - 1) * DP : SL_NEG);
    }
    int anc = nn;
    while (anc > 1 && (anc & 1) == 0) anc >>= 1;
    anc = (anc <= 1) ? 0 : (anc >> 1);
    const int anc4 = anc * 4;
    unsigned int need1 = 0u, need0 = 0u;
#pragma unroll
    for (int t = 1; t < R; ++t) {
        if (t < lev) {
            const unsigned int a = (unsigned int)nn >> (lev - t);
            if ((nn >> (lev - 1 - t)) & 1) need1 |= 1u << a; else need0 |= 1u << a;
        }
    }
    const unsigned int needm = need1 | need0;
    const int levc = lev ? lev : 1;
    const double* const lgp = ring_w + SL_LOGU + (levc - 1);           // its log u, inside ring slot 0
    [[maybe_unused]] const double* const tkp = ring_w + SL_TEMP + (levc - 1);
    // history: lane (j, p) stores element p of generation j's row (p == D: log_obj); chain and log_obj are one allocation
    const bool hl = lane < R * (D + 1);
    const int hj = hl ? lane / (D + 1) : 0, hp = hl ? lane % (D + 1) : 0;
    const unsigned int hmask = (hj + 1 >= 5) ? 0xffffffffu : ((1u << (1u << (hj + 1))) - 1u);     // the state + nodes of generations 1..hj+1
    const bool hist = P.chain != nullptr;
    unsigned int h_off, h_inc;
    if (hl && hist && hp < D) {
        h_off = (unsigned int)((((P.slot_first + hj) * D + hp) * P.N + c) * 8);
        h_inc = (unsigned int)((int64_t)R * D * P.N * 8);
    } else if (hl && hist) {
        h_off = (unsigned int)((reinterpret_cast<const unsigned char*>(P.logobj) - reinterpret_cast<const unsigned char*>(P.chain)) +
                               ((P.slot_first + hj) * P.N + c) * 8);
        h_inc = (unsigned int)((int64_t)R * P.N * 8);
    } else { h_off = 0xffffff00u; h_inc = 0u; }
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(hist ? reinterpret_cast<unsigned char*>(P.chain)
                                                                                : const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(P.Z)),
                                                                           0, hist ? (int)0xfffff000u : 0, 0x00020000);
    const double* const tab_h = ct_w + hp;                 // + winner row * CR

    // target constants
    double muc[D], Wc[(TARGET == TARGET_MVNORMAL) ? D * (D + 1) / 2 : 1];
#pragma unroll
    for (int p = 0; p < D; ++p) muc[p] = P.tp.mu[p];
    if constexpr (TARGET == TARGET_MVNORMAL) {
#pragma unroll
        for (int i = 0; i < D * (D + 1) / 2; ++i) Wc[i] = P.tp.Wp[i];
    }
    const double c0v = P.tp.c0;

    int tb = P.to_boundary / R;                            // passes up to and including the next boundary pass
    const int tbK = P.K / R;

    // state of the chain: every lane holds a copy
    double x[D], xlp;
#pragma unroll
    for (int p = 0; p < D; ++p) x[p] = P.Xcur[c + P.N * p];
    xlp = P.lpcur[c];
    if (P.safe_X) {                        // the state this launch starts from, kept for a redo (WindowParams::safe_X)
        double xv = x[0];
#pragma unroll
        for (int p = 1; p < D; ++p) xv = (lane == p) ? x[p] : xv;
        if (lane < D) P.safe_X[c + P.N * lane] = xv;
        if (lane == 0) P.safe_lp[c] = xlp;
    }
#pragma unroll
    for (int p = 0; p < D; ++p) asm volatile("" :: "v"(x[p]));
    asm volatile("" :: "v"(xlp));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    double hv = 0.0;
    auto store_history = [&](unsigned int off) __attribute__((always_inline)) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const unsigned long long vb = (unsigned long long)__double_as_longlong(hv);
        const u32x2 vv = {(unsigned int)vb, (unsigned int)(vb >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(vv, hrsrc, (int)off, 0, 0);
    };

    double m[R][D];
    constexpr int NPIECE = DP / 2;
    double logu = 0.0, logu_n = 0.0;
    [[maybe_unused]] double temp = 1.0, temp_n = 1.0;
    unsigned int tag_n = 0u;
    // ring slot `slot` (a literal) into registers: the tag FIRST (see the note on LDS order above), then rows and log u
    auto ring_reads = [&](int slot) __attribute__((always_inline)) {
        tag_n = *(const lds_u32*)(ring_w + slot * SLD + SL_TAG);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
            for (int q = 0; q < NPIECE; ++q) {
                const double2 t = reinterpret_cast<const double2*>(mrow[j] + slot * SLD)[q];
                m[j][2 * q] = t.x;
                if (2 * q + 1 < D) m[j][2 * q + 1] = t.y;
            }
        logu_n = lgp[slot * SLD];
        if constexpr (TEMPER) temp_n = tkp[slot * SLD];
    };
    // blocks until H has filled ring slot `slot` with pass `pass_index`; true: H gave up
    auto ring_wait = [&](int slot, int pass_index) __attribute__((always_inline)) -> bool {
        const unsigned int want = (unsigned int)pass_index + 1u;
        while (true) {
            wave_lds_handoff();
            ring_reads(slot);
            if ((unsigned int)__builtin_amdgcn_readfirstlane((int)tag_n) == want) return false;
            if (__hip_atomic_load(&hc_abort[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return true;
            __builtin_amdgcn_s_sleep(1);
        }
    };

    if (ring_wait(0, 0)) { leave(); return; }
    logu = logu_n;
    if constexpr (TEMPER) temp = temp_n;

    int ip = 0;
    int64_t nb = 0;
    unsigned int cnt_total = 0, cnt_first = 0;
#ifdef DEMCZ_STAMPS
    // diagnostic build (scripts/ps2_stamps.py): shader-clock sums per segment of a pass (a stamp drains the wave's outstanding
    // LDS operations, so the segments add up to MORE than an unstamped pass: read them as proportions)
    unsigned long long sa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sa_nbad = 0;
    const unsigned long long sa_start = __builtin_readcyclecounter();
#define PS3_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); sa[i] += t_ - sa_t; sa_t = t_; } while (0)
#else
#define PS3_T(i) do { } while (0)
#endif
    const bool lane_state = nn == 0;
    // one pass; S = ip mod NS (its ring slot).  Returns 0: go on, 1: that was the last pass, 2: a LIVE wait was abandoned.
    auto pass = [&](auto slot_tag, auto first_tag) __attribute__((always_inline)) -> int {
        constexpr int S = decltype(slot_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;       // the launch's first pass: no history of a pass before it to store
        constexpr int SN = (S + 1) % PS3_NS;
        // The pass is written in the order it should ISSUE (window_kernel_ps2):
        //   A  history store of the pass before
        //   B  every node's candidate: state + its rows in order; then the ring slot is H's again
        //   C  log-density of the candidates
        //   D  ancestor's log-density asked for (bpermute); in its shadow the table write and the NEXT pass's ring slot asked for
        //   E  accept tests, the path taken, the winner; its row asked for
        //   F  in that read's shadow: history values, counts, the boundary; the next slot's tag looked at
#ifdef DEMCZ_STAMPS
        unsigned long long sa_t = __builtin_readcyclecounter();
#endif
        const bool boundary = (--tb == 0);
        unsigned int pub_seen = 0u;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FIRST) {
            store_history(0xffffff00u);                   // (nothing yet: out of range)
        } else {
            store_history(h_off);                         // the pass before's
            h_off += h_inc;
        }
        if (boundary) pub_seen = __hip_atomic_load(&pub_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_sched_barrier(0);
        const double logu_c = logu;
        [[maybe_unused]] const double temp_c = temp;
        double cand[D];
#pragma unroll
        for (int p = 0; p < D; ++p) cand[p] = x[p];
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
            for (int p = 0; p < D; ++p) cand[p] = cand[p] + m[j][p];
        PS3_T(0);                      // history store, candidate adds
        __builtin_amdgcn_sched_barrier(0);
        // (the rows are in registers: pass ip's slot may be refilled)
        if (lane == 0) __hip_atomic_store(&hc_done[w], ip + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        double lpp;
        if constexpr (TARGET == TARGET_MVNORMAL) {
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double acc = Wc[(i * (i + 1)) / 2] * (cand[0] - muc[0]);
#pragma unroll
                for (int j = 1; j <= i; ++j) acc = fma(Wc[(i * (i + 1)) / 2 + j], cand[j] - muc[j], acc);
                q = (i == 0) ? acc * acc : fma(acc, acc, q);
            }
            lpp = fma(-0.5, q, c0v);
        } else {
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double rr = cand[i] - muc[i];
                q = (i == 0) ? rr * rr : fma(rr, rr, q);
            }
            lpp = -q;
        }
        PS3_T(1);                      // log-density
        __builtin_amdgcn_sched_barrier(0);
        // the state's own row keeps the state's log-density (lane 0's candidate IS the state)
        const double lb = lane_state ? xlp : lpp;
        unsigned int m32, path, accp;
        unsigned long long chg_a, chg_r;
        const unsigned long long lbb = (unsigned long long)__double_as_longlong(lb);
        unsigned int blo, bhi;
        asm volatile("ds_bpermute_b32 %0, %2, %3\n\tds_bpermute_b32 %1, %2, %4"
                     : "=&v"(blo), "=&v"(bhi) : "v"(anc4), "v"((unsigned int)lbb), "v"((unsigned int)(lbb >> 32)) : "memory");
        {
            double row[CR];
#pragma unroll
            for (int p = 0; p < CR; ++p) row[p] = (p < D) ? cand[p] : ((p == D) ? lb : 0.0);
#pragma unroll
            for (int q = 0; q < CR / 2; ++q) reinterpret_cast<double2*>(ct_w + lane * CR)[q] = make_double2(row[2 * q], row[2 * q + 1]);
        }
        wave_lds_handoff();
        ring_reads(SN);
        PS3_T(2);                      // bpermute asked for, table write, the next pass's ring slot asked for
        __builtin_amdgcn_sched_barrier(0);
        {
            // behind the two bpermutes in this wave's LDS queue: CR/2 table writes, the tag, R * NPIECE row reads, log u; LDS
            // operations complete in order, and the wait's field holds at most 15
            constexpr int BEHIND = CR / 2 + 1 + R * NPIECE + 1 + (TEMPER ? 1 : 0);
            asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(BEHIND < 15 ? BEHIND : 15) : "memory");
            asm volatile("" : "+v"(blo), "+v"(bhi));
            const double lpb = __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));
            const double d0 = lpp - lpb;
            double dlt = d0;
            if constexpr (TEMPER) dlt = dlt / temp_c;
            m32 = (unsigned int)__builtin_amdgcn_ballot_w64(logu_c < dlt);
            chg_a = __builtin_amdgcn_fcmp(d0, 0.0, 14 /* UNE */);
            chg_r = __builtin_amdgcn_fcmp(lpb - lpb, 0.0, 14);
            // on the path actually taken: every ancestor decided the way that leads here
            const bool onp = ((m32 ^ need1) & needm) == 0u;
            path = (unsigned int)__builtin_amdgcn_ballot_w64(onp) & 0xfffffffeu;
            accp = path & m32;
        }
        PS3_T(3);                      // accept tests, path
        const unsigned int accp1 = accp | 1u;                                   // bit 0: the state the pass started from
        const unsigned int win = 31u - (unsigned int)__builtin_clz(accp1);
        wave_lds_handoff();
        // the new state: the winner's row of the table, every lane reading the same address
        {
            const double* wr = ct_w + win * CR;
#pragma unroll
            for (int q = 0; q < CR / 2; ++q) {
                const double2 t = reinterpret_cast<const double2*>(wr)[q];
                if (2 * q < D) x[2 * q] = t.x; else if (2 * q == D) xlp = t.x;
                if (2 * q + 1 < D) x[2 * q + 1] = t.y; else if (2 * q + 1 == D) xlp = t.y;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        logu = logu_n;
        if constexpr (TEMPER) temp = temp_n;
        // history rows of the pass: read now, stored during the next pass
        {
            const unsigned int wa = accp1 & hmask;
            const unsigned int wj = 31u - (unsigned int)__builtin_clz(wa);
            hv = tab_h[wj * CR];
        }
        {
            const unsigned int chm = (accp & (unsigned int)chg_a) | (path & ~m32 & (unsigned int)chg_r);
            cnt_total += (unsigned int)__builtin_popcount(chm);
            if constexpr (FIRST) cnt_first = (chm >> 1) & 1u;
        }
        PS3_T(4);                      // winner's row asked for, history values, counts
        if (boundary) {                // a generation divisible by K ended the pass: runchain!'s append, demcz.jl:88-91
            const double v = ct_w[win * CR + ((lane < D) ? lane : 0)];
            while (pub_seen + (unsigned int)PS_PUB <= (unsigned int)nb) {          // (bounded by the publisher: demcz_kernels_rec.h)
                __builtin_amdgcn_s_sleep(1);
                pub_seen = __hip_atomic_load(&pub_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (lane < D) pub_rows[(w * PS_PUB + (int)((unsigned int)nb % PS_PUB)) * D + lane] = v;
            asm volatile("" ::: "memory");
            if (lane == 0) __hip_atomic_store(&pub_seq[w], (unsigned int)nb + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane < D && P.snap) P.snap[nb * P.N * D + c + P.N * lane] = v;
            ++nb;
            tb = tbK;
        }
        PS3_T(5);                      // boundary: hand the row to the publisher
        if (++ip == npass) return 1;
        // only now -- this wave's own row is on its way -- may it wait for H (which may be waiting for rows of other waves)
        if ((unsigned int)__builtin_amdgcn_readfirstlane((int)tag_n) != (unsigned int)ip + 1u) {
#ifdef DEMCZ_STAMPS
            ++sa_nbad;
#endif
            if (ring_wait(SN, ip)) return 2;
            logu = logu_n;
            if constexpr (TEMPER) temp = temp_n;
        }
        PS3_T(6);                      // waiting for H
        wave_lds_handoff();
        return 0;
    };
    int st = pass(std::integral_constant<int, 0>{}, std::true_type{});
    while (!st) {
        st = pass(std::integral_constant<int, 1>{}, std::false_type{});
        if (st) break;
        st = pass(std::integral_constant<int, 2>{}, std::false_type{});
        if (st) break;
        st = pass(std::integral_constant<int, 3>{}, std::false_type{});
        if (st) break;
        st = pass(std::integral_constant<int, 0>{}, std::false_type{});
    }
    if (st == 2) { leave(); return; }
    store_history(h_off);     // the last pass's
    {
        double xv = x[0];
#pragma unroll
        for (int p = 1; p < D; ++p) xv = (lane == p) ? x[p] : xv;
        if (lane < D) P.Xcur[c + P.N * lane] = xv;
        if (lane == 0) P.lpcur[c] = xlp;
    }
    wave_store_counts(P, c, cnt_total, cnt_first);
#ifdef DEMCZ_STAMPS
    if (P.stamps && lane == 0 && c < 65536) {
        unsigned long long* o = P.stamps + (size_t)c * 16;
        for (int i = 0; i < 7; ++i) o[i] = sa[i];
        o[8] = __builtin_readcyclecounter() - sa_start; o[11] = sa_nbad; o[14] = (unsigned long long)npass; o[15] = 2;
    }
#endif
#undef PS3_T
    leave();
}

}  // namespace demcz
