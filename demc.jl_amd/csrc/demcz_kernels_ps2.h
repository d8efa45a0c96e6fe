// demcz_kernels_ps2.h -- K1g2: the steady state of the wave-per-chain consumer (demcz_kernels_ps.h), written for it alone.
//
// window_kernel_ps handles every launch shape: passes of 1..5 generations, boundaries anywhere, a queue of pass lengths,
// per-pass buffer descriptors, the state in scalar registers.  Its pass is ~330 instructions of which the arithmetic proper
// is a sixth, and one wave per SIMD issues one instruction every ~4 cycles whatever the instruction is: the launch is
// bound by the instruction COUNT.  But the launches that matter (a 1000-generation autostop slab at K = 10; any launch
// that starts right after a boundary, with K and its length multiples of five) are perfectly regular: every pass is five
// generations, a boundary falls on the end of every (K/5)-th pass.  This kernel runs those and nothing else:
//
//   * no pass queue, no lengths, no clamps: the pass loop is unrolled by the three raw slots (slot addresses are literals), a
//     boundary is a counter that reaches zero;
//   * the archive and both record buffers live in ONE allocation (demcz_create), so a DMA source is a 32-bit offset from one
//     scalar base: per pass a lane's source is `(row index << log2(row bytes)) + its own running offset` -- one v_perm (which
//     half of the packed index pair, or zero for the record lanes), one shift-add, one add;
//   * chain and log_obj histories are one allocation too: ONE buffer store per pass covers both, with a descriptor built once
//     and a per-lane offset that advances by a per-lane constant;
//   * the chain's state lives in VECTOR registers, every lane holding a copy: after the accept tests the winner's row is read
//     back from the LDS candidate table with one wave-uniform address (three 16-byte broadcast reads) instead of twelve
//     v_readlane; lane 0 writes its row too (its candidate IS the state), so the table always holds the state at row 0 and
//     the history and boundary reads never special-case "nothing accepted";
//   * a pass's own row indices travel in the slot as well (three more of the 64 DMA lanes), so the LIVE re-read of a row that
//     showed the sentinel needs no register queue of indices.
//
// The arithmetic -- which additions, in which order, on which values -- is window_kernel_ps's, hence the oracle's: results
// are bit-identical (tests: every parity case whose launches are regular runs through this kernel; DEMCZ_NO_PS2=1 selects the
// general kernel for A/B).  Any launch that is not regular takes the general kernel; the host decides per launch
// (ps2_applicable, demcz_capi.hip).
#pragma once

#include "demcz_kernels_ps.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace demcz {

constexpr int PS2_R = 5;             // generations per pass (the tree's depth): fixed
#ifndef PS2_XCD_SWIZZLE
#define PS2_XCD_SWIZZLE 1
#endif
#ifndef PS2_AHEAD_N
#define PS2_AHEAD_N 2
#endif
constexpr int PS2_AHEAD = PS2_AHEAD_N;       // a pass's DMA is issued this many passes before its slot is consumed
constexpr int PS2_SLOTS = PS2_AHEAD + 1;     // raw slots: the one being consumed + those in flight
static_assert(PS2_AHEAD >= 2 && PS2_AHEAD <= 4, "the pass loop is unrolled by the slots");
#ifndef PS2_HSLOTS_N
#define PS2_HSLOTS_N 4
#endif
constexpr int PS2_HSLOTS = PS2_HSLOTS_N;     // history ring: passes a chain wave may be ahead of the publisher wave
// Which LIVE kernels hand their history to the publisher wave through a ring in LDS (the others store it themselves):
#ifndef PS2_DDPP                             // candidate adds out of registers by lanes (DPP row_newbcast: DDPP in the kernel): built,
#define PS2_DDPP 0                           // bit-identical, 159 -> 129 VGPRs and 153 -> 93 LDS instructions a pass -- and no faster at
#endif                                       // d = 5 (140.9-141.8 against 139.4-140.9 us per launch): off; window_kernel_pw is where it pays
#ifndef PS2_HRING_ONE
#define PS2_HRING_ONE 0                      // one chain to a wave: no (the hand-off costs the chain wave more than the store: +8 %)
#endif
#ifndef PS2_HRING_TWO
#define PS2_HRING_TWO 1                      // two chains to a wave: yes (profiles/r04o_history_store.txt)
#endif
// record rows are read up to R * (2 AHEAD + 1) + 1 generations past the launch's last one (never consumed): REC_PAD covers it

// one 16-byte piece per lane from (scalar base + per-lane 32-bit offset) into LDS at lds_dst + 16 * lane (lds_dst: wave-uniform).
// M0 -- the LDS base of the transfer -- is an INPUT of the asm, bound to the physical register ("{m0}"): the compiler itself
// moves lds_dst there and knows what M0 holds.  (Until round 4 the asm wrote M0 in its own text and listed it as a clobber,
// which the compiler does not promise to honour for a reserved register: 112 warnings, and a latent miscompile.  The compiler's
// builtin, __builtin_amdgcn_global_load_lds, selects the same instruction but models the transfer as a write to ALL of LDS: it
// puts s_waitcnt vmcnt(0) in front of every later LDS access that might alias -- the publisher hand-off words of each boundary
// pass -- which drains the very DMA queue the pass structure keeps two passes deep.)  s_nop: one wait state between an SALU
// write of M0 and an LDS-DMA instruction that reads it (the hazard recogniser does not look inside asm text).
__device__ __forceinline__ void ps2_dma16(const void* sbase, unsigned voff, unsigned lds_dst)
{
#ifndef PS2_DMA_POLICY
#define PS2_DMA_POLICY ""
#endif
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" PS2_DMA_POLICY
                 :: "v"(voff), "s"(sbase), "{m0}"(lds_dst) : "memory");
}
// The same transfer through a buffer descriptor (round 5, the default: PS2_DMA_BUFFER).  A global load checks nothing: its address
// is base + a 32-bit offset made from a row index that itself came out of an LDS slot a pass ago -- one mis-counted s_waitcnt and
// the index is whatever the slot held before, the address anywhere in the 4 GB behind the arena: a memory-access fault, as seen
// once in round 4 with an experimental form of this kernel whose vector-memory count per pass was off.  With
// num_records = the arena's size the hardware returns zeros for anything beyond it: a wrong index can give a wrong result -- which
// the parity tests see -- but never touches memory that is not the handle's.
typedef unsigned int ps2_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ps2_rsrc_t ps2_make_rsrc(const void* base, unsigned int bytes)
{
    const unsigned long long b = (unsigned long long)reinterpret_cast<uintptr_t>(base);
    ps2_rsrc_t r;
    r.x = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)b);
    r.y = (unsigned int)__builtin_amdgcn_readfirstlane((int)((unsigned int)(b >> 32) & 0xffffu));      // stride 0: a raw buffer
    r.z = (unsigned int)__builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000u;
    return r;
}
__device__ __forceinline__ void ps2_dma16b(ps2_rsrc_t rsrc, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" PS2_DMA_POLICY
                 :: "v"(voff), "s"(rsrc), "{m0}"(lds_dst) : "memory");
}
#ifndef PS2_DMA_BUFFER
#define PS2_DMA_BUFFER 1
#endif
// The publisher wave's idle sleep (round 5).  It shares a SIMD with a chain wave and, with nothing to publish, used to look at the
// hand-off words every ~150 clocks (s_sleep 1): ~13 % of that SIMD's issue slots -- measured as the difference between the LIVE
// and the non-LIVE instantiation before a single boundary is counted: 111 against 99.5 us per 1000 generations with four
// boundaries a launch (profiles/r05q_live_fixed_cost.txt), and through the hand-off every chain keeps step with the slowest.  Now it
// sleeps PS2_PUB_IDLE_SLEEP x 64 clocks and the chain wave that posts a row (or leaves) wakes it with s_wakeup -- which pings the
// workgroup's sleeping waves and is ignored by the others.  A ping that arrives just before the publisher falls asleep is lost: the
// row then waits out one sleep (0.4 us at 16), no longer.  (Two chains to a wave: the publisher also carries the history, a
// pass's worth every pass -- it keeps the short sleep.)
#ifndef PS2_PUB_IDLE_SLEEP
#define PS2_PUB_IDLE_SLEEP 16
#endif
__device__ __forceinline__ void ps2_wake_publisher()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_wakeup" ::: "memory");      // (the LDS words it will look at are written)
}

template <int TARGET, int D, bool LIVE, bool TEMPER>
__global__ void __launch_bounds__(64 * (PS_CHAINS + (LIVE ? 1 : 0)), 3) window_kernel_ps2(const WindowParams P)
{
    static_assert(TARGET == TARGET_MVNORMAL || TARGET == TARGET_ISO_QUAD, "split layout: MvNormal / isotropic targets");
    static_assert(D >= 2 && D <= 5, "a pass's rows, normals, log u and indices are one 64-lane DMA");
    constexpr int R = PS2_R;
    constexpr int HW = (D + 1) / 2;                        // 16-byte pieces of an archive row
    constexpr int ZSC = (D <= 2) ? 2 : (D <= 4) ? 4 : 8;   // archive row stride in doubles (demcz_create: ZS)
    constexpr int ZSH = (ZSC == 2) ? 4 : (ZSC == 4) ? 5 : 6;      // log2 of the row stride in bytes
    constexpr int DP = ((D + 1) / 2) * 2;                  // increments row in LDS
    constexpr int CR = ((D + 2) / 2) * 2;                  // candidate row in LDS: D doubles, log-density, pad
    // lanes of the DMA: [0, ROWL) archive rows (generation u, first / second row, piece j); then NF fields of three pieces
    // (six generations) each: 0..D-1 normals, D log u, D+1 this pass's row indices, D+2 the row indices of the pass two after
    // this one (the same record field, AHEAD passes on), D+3 temperatures; the rest idle (they fetch row 0)
    constexpr int ROWL = R * 2 * HW;
    constexpr int FL0 = ROWL;
    constexpr int NF = D + 3 + (TEMPER ? 1 : 0);
    constexpr int TL0 = FL0 + 3 * NF;
    static_assert(TL0 <= 64, "one DMA instruction per pass");
    constexpr int F_LOGU = D, F_IXOWN = D + 1, F_IXNEXT = D + 2, F_TEMP = D + 3;
    constexpr int SDN = (R + 1) * DP + 2;                  // sdelta: R rows of increments, the row of negative zeros, pad
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (!LIVE) {
        // short launches: the producer half rides in the same grid (demcz_kernels_ps.h)
        if ((int64_t)blockIdx.x >= P.consumer_blocks) {
            pc_produce<D>(P, ((int64_t)blockIdx.x - P.consumer_blocks) * PS_CHAINS + w, lane);
            return;
        }
    }
#if PS2_XCD_SWIZZLE
    const int bxs = xcd_block(P);          // XCD x runs the x-th eighth of the chains: demcz_kernels.h
#else
    const int bxs = (int)blockIdx.x;
#endif
    __shared__ __attribute__((aligned(16))) unsigned char raw[PS_CHAINS][PS2_SLOTS][1024];
    __shared__ __attribute__((aligned(16))) double sdelta[PS_CHAINS][SDN];
    __shared__ __attribute__((aligned(16))) double ctab[PS_CHAINS][64 * CR];       // row l: lane l's candidate (rows 32..63 shadow 0..31)
    __shared__ double pub_rows[LIVE ? PS_CHAINS * PS_PUB * D : 1];
    __shared__ unsigned int pub_seq[PS_CHAINS], pub_done[PS_CHAINS], pub_exit[PS_CHAINS];
    // The history of a LIVE launch may go through the publisher wave (HRING).  Why: a store in the chain wave's own queue is
    // four 16-byte pieces per row and workgroup (a wave's two chains are neighbours), and history stores into one 128-byte column
    // in eight of the rows' KiBs (address bits 9:7 = 3 on every box measured) are accepted at half the rate of the others when
    // reads run beside them (scripts/probes/store_classes.hip); the CU's memory pipeline is in order, so the chain waves of the
    // workgroups that own those columns run their DMAs late, and everyone waits for their rows.  Through the ring a row's columns
    // of ALL the workgroup's chains leave as ONE contiguous piece (64 bytes for eight chains): a quarter of the requests.
    // Ring: slot = pass mod PS2_HSLOTS; in it row r (lane r of a chain wave: (generation, element)) of chain kk of the workgroup
    // at r * (HKK + 1) + kk -- the odd stride keeps both the chain waves' writes (30 rows of one kk) and the publisher's reads
    // (consecutive kk of consecutive rows) off each other's banks.
    constexpr int NCHR = 1;                                // chains of a chain wave
    constexpr bool HRING = LIVE && (PS2_HRING_ONE != 0);
    constexpr int HKK = PS_CHAINS * NCHR, HROWS = R * (D + 1), HSL = ((HROWS * (HKK + 1) + 1) / 2) * 2;
    __shared__ double hist_ring[HRING ? PS2_HSLOTS * HSL : 1];
    __shared__ unsigned int hist_seq[PS_CHAINS], hist_done[1];
    // where lane (j, p) of a chain's wave stores element p of generation j's row of the history (p == D: log_obj; chain and
    // log_obj are one allocation), and by how much that moves per pass
    const bool hist = P.chain != nullptr;
    auto hist_offsets = [&](bool hl_, int hj_, int hp_, int64_t c_, unsigned int& off, unsigned int& inc) __attribute__((always_inline)) {
        if (hl_ && hist && hp_ < D) {
            off = (unsigned int)((((P.slot_first + hj_) * D + hp_) * P.N + c_) * 8);
            inc = (unsigned int)((int64_t)R * D * P.N * 8);
        } else if (hl_ && hist) {
            off = (unsigned int)((reinterpret_cast<const unsigned char*>(P.logobj) - reinterpret_cast<const unsigned char*>(P.chain)) +
                                 ((P.slot_first + hj_) * P.N + c_) * 8);
            inc = (unsigned int)((int64_t)R * P.N * 8);
        } else { off = 0xffffff00u; inc = 0u; }
    };
    // (num_records = the history's own size, P.hist_bytes < 0xfff00000 -- ps2_applicable: a store whose offset is wrong is dropped,
    //  not written into whatever lies within 4 GB of the history; "nothing to store" = 0xffffff00 lies above any such size)
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(hist ? reinterpret_cast<unsigned char*>(P.chain) : reinterpret_cast<unsigned char*>(const_cast<double*>(P.Z)),
                                                                           0, hist ? (int)P.hist_bytes : 0, 0x00020000);
    // (all 64 lanes execute the store -- lanes with nothing to store point out of range)
    auto hist_store = [&](double v, unsigned int off) __attribute__((always_inline)) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
        const u32x2 vv = {(unsigned int)vb, (unsigned int)(vb >> 32)};
#ifdef PS2_EXP_NOHIST         // (timing experiment only: every history store out of range, dropped by the descriptor -- no history)
        off = 0xffffff00u;
#endif
        __builtin_amdgcn_raw_buffer_store_b64(vv, hrsrc, (int)off, 0, 0);
    };
    if constexpr (LIVE) {
        if (threadIdx.x < PS_CHAINS) {
            pub_seq[threadIdx.x] = 0u; pub_done[threadIdx.x] = 0u; pub_exit[threadIdx.x] = 0u;
            hist_seq[threadIdx.x] = 0u; hist_done[0] = 0u;
        }
        __syncthreads();
        if (w == PS_CHAINS) {
            // the publisher wave: demcz_kernels_ps.h (identical protocol; it never leaves before its chain waves)
#ifdef PS2_PUB_PRIO
            __builtin_amdgcn_s_setprio(PS2_PUB_PRIO);
#endif
            const bool pl = lane < PS_CHAINS * D;
            const int cw = pl ? lane / D : 0, pp = pl ? lane % D : 0;
            const int64_t cl = (int64_t)bxs * PS_CHAINS + cw;
            unsigned int done = 0u;
            // the history (HRING): lane (row prl of the instruction's RPI rows, chain pk of the workgroup's HKK)
            constexpr int RPI = 64 / HKK, NI = (HROWS + RPI - 1) / RPI;
            const int pk = lane % HKK, prl = lane / HKK;
            const int64_t pc = (int64_t)bxs * HKK + pk;
            [[maybe_unused]] unsigned int hoffp[NI], hincp[NI], hq = 0u;
            [[maybe_unused]] int hidx[NI];
            // chain waves that run at all (the others leave at once and never post)
            [[maybe_unused]] const int nact = (int)((P.N - (int64_t)bxs * HKK + NCHR - 1) / NCHR);
            if constexpr (HRING) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int r = i * RPI + prl;
                    hist_offsets(r < HROWS && pc < P.N, r / (D + 1), r % (D + 1), pc, hoffp[i], hincp[i]);
                    hidx[i] = ((r < HROWS) ? r : HROWS - 1) * (HKK + 1) + pk;
                }
            }
#ifdef DEMCZ_STAMPS
            unsigned long long pb_iter = 0, pb_wait = 0, pb_store = 0, pb_sleep = 0, pb_pub = 0;
            const unsigned long long pb_t0 = __builtin_readcyclecounter();
#endif
            while (true) {
#ifdef DEMCZ_STAMPS
                const unsigned long long pb_a = __builtin_readcyclecounter();
#endif
                publisher_wait(P);
#ifdef DEMCZ_STAMPS
                pb_wait += __builtin_readcyclecounter() - pb_a; ++pb_iter;
#endif
                // one look at what the chain waves have posted: boundary rows (lane -> chain wave cw); history and who has left
                // (lane & 3; the exit flags FIRST: a wave seen gone has posted all it ever will)
                [[maybe_unused]] unsigned int hxl = 0u, hsl = 0u;
                if constexpr (HRING) {
                    hxl = __hip_atomic_load(&pub_exit[lane & (PS_CHAINS - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    asm volatile("" ::: "memory");
                    hsl = __hip_atomic_load(&hist_seq[lane & (PS_CHAINS - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                const unsigned int seq = __hip_atomic_load(&pub_seq[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const bool ready = pl && seq != done;
                const bool pany = __builtin_amdgcn_ballot_w64(ready) != 0ull;
                // history passes that can leave now: a pass leaves when every chain wave still running has posted it (a wave that
                // gave up -- the launch will be redone -- no longer holds the others' history back: they would wait for ring space
                // for ever); up to HPR passes a round
                constexpr int HPR = 2;
                int hn = 0;
                if constexpr (HRING) {
                    if (hist) {
                        unsigned int hmin = 0xffffffffu;
#pragma unroll
                        for (int ww = 0; ww < PS_CHAINS; ++ww) {
                            const unsigned int hs = (unsigned int)__builtin_amdgcn_readlane((int)hsl, ww);
                            const bool left = __builtin_amdgcn_readlane((int)hxl, ww) != 0;
                            if (ww < nact && !(left && hs <= hq)) hmin = (hs < hmin) ? hs : hmin;
                        }
                        if (hmin != 0xffffffffu && hmin > hq) hn = (hmin - hq >= (unsigned int)HPR) ? HPR : 1;
                    }
                }
                // all of the round's LDS reads first (the LDS pipeline is busy with the chain waves: a round trip is long), then its
                // stores -- the boundary rows in front: other workgroups' waves may be waiting for them
#ifdef DEMCZ_STAMPS
                const unsigned long long pb_s = __builtin_readcyclecounter();
#endif
                double v = 0.0;
                if (ready) v = pub_rows[(cw * PS_PUB + (int)(done % PS_PUB)) * D + pp];
                [[maybe_unused]] double hvv[HPR][NI];
                if constexpr (HRING) {
#pragma unroll
                    for (int j = 0; j < HPR; ++j) {
                        if (j < hn) {
                            const double* slot = hist_ring + (int)((hq + (unsigned int)j) % PS2_HSLOTS) * HSL;
#pragma unroll
                            for (int i = 0; i < NI; ++i) hvv[j][i] = slot[hidx[i]];
                        }
                    }
                }
                if (pany) {
                    if (ready) {
                        if (cl < P.N && P.do_append) live_publish(P, (int64_t)done, cl, pp, v);
                        ++done;
                    }
                    asm volatile("" ::: "memory");
                    if (ready && pp == 0) __hip_atomic_store(&pub_done[cw], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                const bool hany = hn > 0;
                if constexpr (HRING) {
                    if (hany) {
#pragma unroll
                        for (int j = 0; j < HPR; ++j) {
                            if (j < hn) {
#pragma unroll
                                for (int i = 0; i < NI; ++i) {
#ifndef PS2_EXP_PUBSKIP       // (timing experiment: the publisher acknowledges the posts and stores nothing)
                                    hist_store(hvv[j][i], hoffp[i]);
#endif
                                    hoffp[i] += hincp[i];
                                }
                            }
                        }
                        hq += (unsigned int)hn;
                        asm volatile("" ::: "memory");
                        if (lane == 0) __hip_atomic_store(&hist_done[0], hq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
#ifdef DEMCZ_STAMPS
                if (hany) pb_store += __builtin_readcyclecounter() - pb_s;
#endif
                if (pany) continue;
                if (hany) continue;
                const bool gone = !pl || (__hip_atomic_load(&pub_exit[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u &&
                                          __hip_atomic_load(&pub_seq[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == done);
                if (__builtin_amdgcn_ballot_w64(!gone) == 0ull) {
                    // every chain wave has left: whatever history it posted before leaving is in the ring by now
                    bool hpend = false;
                    if constexpr (HRING) {
                        asm volatile("" ::: "memory");
                        unsigned int hmin = 0xffffffffu;
#pragma unroll
                        for (int ww = 0; ww < PS_CHAINS; ++ww) {
                            const unsigned int hs = __hip_atomic_load(&hist_seq[ww], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (ww < nact && hs > hq) hmin = (hs < hmin) ? hs : hmin;
                        }
                        hpend = hist && hmin != 0xffffffffu;
                    }
                    if (!hpend) break;
                    continue;
                }
#ifdef DEMCZ_STAMPS
                ++pb_sleep;
#endif
                if constexpr (HRING) __builtin_amdgcn_s_sleep(1);
                else __builtin_amdgcn_s_sleep(PS2_PUB_IDLE_SLEEP);
            }
#ifdef DEMCZ_STAMPS
            if (P.stamps && lane == 0 && (int)blockIdx.x < 2048) {       // the publisher's own account: second half of the stamp buffer
                unsigned long long* o = P.stamps + (size_t)(32768 + bxs) * 16;      // (DEMCZ_STAMP_WGS / 2, demcz_capi.hip)
                o[0] = pb_iter; o[1] = pb_wait; o[2] = pb_store; o[3] = pb_sleep; o[4] = __builtin_readcyclecounter() - pb_t0; o[5] = hq; o[6] = done;
            }
#endif
            return;
        }
    }
    auto leave = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (LIVE) {
            if (lane == 0) __hip_atomic_store(&pub_exit[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (!HRING) ps2_wake_publisher();
        }
    };
    const int64_t c = (int64_t)bxs * PS_CHAINS + w;
    if (c >= P.N) {
        wave_store_counts(P, c, 0u, 0u);
        leave();
        return;
    }
    if constexpr (LIVE) {
        if (__hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            // Another wave has already given up.  This chain's row of the redo snapshot is still owed (the host made no copy
            // in front of this launch: launch_window): Xcur / lpcur of chain c are only ever written by this wave, at the
            // launch's end, so they still hold the state the launch started from.
            if (P.safe_X) {
                if (lane < D) P.safe_X[c + P.N * lane] = P.Xcur[c + P.N * lane];
                if (lane == 0) P.safe_lp[c] = P.lpcur[c];
            }
            leave();
            return;
        }
    }
    __builtin_amdgcn_s_setprio(3);
    unsigned char* const raw_w = &raw[w][0][0];
    double* const sd_w = &sdelta[w][0];
    double* const ct_w = &ctab[w][0];
    const unsigned raw_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)raw_w);

    // ---- what this lane is, in each of its parts ----------------------------------------------------------------
    // node of the tree of outcomes: nn = 0 is the state itself (lanes 0 and 32), 1..31 the nodes (lanes 32..63 shadow 0..31)
    const int nn = lane & 31;
    const int lev = nn ? 32 - __builtin_clz((unsigned)nn) : 0;
    // DDPP (round 4; demcz_kernels_pw.h has the long form of this note, scripts/gen_pw_wdpp.py gen_adds the generator): a pass's
    // increments are the same for every node, and a node that does not take a generation added -0.0, the identity -- so the
    // increments sit in a few register pairs by lanes, and per generation the lanes whose path takes it add them (EXEC = that
    // ballot) straight out of a taker lane of their 16-lane row by DPP row_newbcast: 9 8-byte LDS reads a pass instead of 15
    // 16-byte ones, 18 registers instead of 50.
    constexpr bool DDPP = (PS2_DDPP != 0);
    constexpr int DD_KN[5] = {4, 4, 4, 4, 16};
    constexpr int DD_NQ[5] = {(D + 3) / 4, (D + 3) / 4, (D + 3) / 4, (D + 3) / 4, (D + 15) / 16};
    constexpr int DD_NQMAX = (D + 3) / 4;
    [[maybe_unused]] uint64_t tmask[R];
    [[maybe_unused]] const double* dptr[R];
    const double* mrow[R];                 // rows of sdelta this node adds, in order: an accepted generation on its path or its
#pragma unroll                             // own -> that generation's increments, anything else -> the row of negative zeros
    for (int j = 1; j <= R; ++j) {
        const bool take = nn != 0 && ((j == lev) || (j < lev && ((nn >> (lev - 1 - j)) & 1)));
        mrow[j - 1] = sd_w + (take ? j - 1 : R) * DP;
        if constexpr (DDPP) {
            const uint64_t tm = __builtin_amdgcn_ballot_w64(take);
            unsigned int common = 0xffffu;
#pragma unroll
            for (int rr4 = 0; rr4 < 4; ++rr4) {
                const unsigned int rowm = (unsigned int)(tm >> (16 * rr4)) & 0xffffu;
                if (rowm) common &= rowm;
            }
            const int pos = lane & 15;
            int mrank = ((common >> pos) & 1u) ? __builtin_popcount(common & ((1u << pos) - 1u)) : 0;
            mrank = (mrank < D) ? mrank : 0;               // (positions that hold no entry of this generation: anything in bounds)
            tmask[j - 1] = tm;
            dptr[j - 1] = sd_w + (j - 1) * DP + mrank;
        }
    }
    [[maybe_unused]] double one = 1.0;
    if constexpr (DDPP) asm volatile("" : "+v"(one));
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
    const int lgo = (FL0 + 3 * F_LOGU) * 16 + (levc - 1) * 8;          // its log u, inside a raw slot
    [[maybe_unused]] const int tko = (FL0 + 3 * F_TEMP) * 16 + (levc - 1) * 8;
    // increments: lane (u, p) forms element p of generation u of the pass
    const bool fl = lane < R * D;
    const int fu = fl ? lane / D : 0, fp = fl ? lane % D : 0;
    const int zao = ((fu * 2) * HW) * 16 + fp * 8;                      // second row: + HW * 16
    const int zto = (FL0 + 3 * fp) * 16 + fu * 8;
    const int ixown = (FL0 + 3 * F_IXOWN) * 16 + fu * 8;                // this pass's row indices (LIVE re-reads)
    const double eps_p = P.eps[fp];
    const double scale = P.gamma / sqrt((double)(2 * D));
    double* const sdw_p = sd_w + (fl ? fu * DP + fp : R * DP + DP);     // (idle lanes: the pad)
    // DMA: rows (ru, which, piece), record fields (f, piece)
    const bool rowl = lane < ROWL;
    const int ru = rowl ? lane / (2 * HW) : 0, rwhich = rowl ? (lane / HW) % 2 : 0, rj = rowl ? lane % HW : 0;
    const bool fieldl = lane >= FL0 && lane < TL0;
    const int ff = fieldl ? (lane - FL0) / 3 : 0, fj = fieldl ? (lane - FL0) % 3 : 0;
    const int ixnext = (FL0 + 3 * F_IXNEXT) * 16 + ru * 8;              // where a row lane finds the indices of the pass two on
    // which half of the packed index pair a row lane takes (v_perm byte selectors; 0x0c = the constant 0x00)
    const unsigned int selv = !rowl ? 0x0c0c0c0cu : (rwhich ? 0x07060504u : 0x03020100u);
    const unsigned char* const zbase = reinterpret_cast<const unsigned char*>(P.Z);
    [[maybe_unused]] const ps2_rsrc_t zrsrc = ps2_make_rsrc(zbase, P.z_bytes);      // the arena: archive + both record buffers + temperatures
    unsigned int dma_off, dma_inc;
    {
        const unsigned int rec_off = (unsigned int)(reinterpret_cast<const unsigned char*>(P.rec_in) - zbase);
        if (rowl) { dma_off = (unsigned int)rj * 16u; dma_inc = 0u; }
        else if (fieldl && ff != F_TEMP) {
            const int rf = (ff == F_IXOWN || ff == F_IXNEXT) ? D + 1 : ff;
            dma_off = rec_off + (unsigned int)((((int64_t)rf * P.N + c) * P.rec_stride + (ff == F_IXNEXT ? PS2_AHEAD * R : 0)) * 8) + (unsigned int)fj * 16u;
            dma_inc = (unsigned int)(R * 8);
        } else if (fieldl) {
            dma_off = (unsigned int)(reinterpret_cast<const unsigned char*>(P.temperature) - zbase) + (unsigned int)fj * 16u;
            dma_inc = (unsigned int)(R * 8);
        } else { dma_off = 0u; dma_inc = 0u; }
    }
    // history: lane (j, p) stores element p of generation j's row (p == D: log_obj); chain and log_obj are one allocation
    const bool hl = lane < R * (D + 1);
    const int hj = hl ? lane / (D + 1) : 0, hp = hl ? lane % (D + 1) : 0;
    const unsigned int hmask = (hj + 1 >= 5) ? 0xffffffffu : ((1u << (1u << (hj + 1))) - 1u);     // the state + nodes of generations 1..hj+1
    [[maybe_unused]] unsigned int h_off, h_inc;            // (!LIVE: the chain wave stores its history itself)
    hist_offsets(hl, hj, hp, c, h_off, h_inc);
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

    const int npass = P.ngen / R;
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
    if (lane < DP + 2) sd_w[R * DP + lane] = (lane < DP) ? -0.0 : 0.0;

    // the first two passes' row indices by ordinary loads; then everything that was loaded is in registers before the first DMA
    const double* rec_ix = P.rec_in + ((int64_t)(D + 1) * P.N + c) * P.rec_stride;
    uint64_t pp[PS2_AHEAD];
#pragma unroll
    for (int k = 0; k < PS2_AHEAD; ++k) pp[k] = (uint64_t)__double_as_longlong(rec_ix[k * R + ru]);
#pragma unroll
    for (int p = 0; p < D; ++p) asm volatile("" :: "v"(x[p]));
    asm volatile("" :: "v"(xlp), "v"(eps_p));
#pragma unroll
    for (int k = 0; k < PS2_AHEAD; ++k) asm volatile("" :: "v"(pp[k]));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    auto issue = [&](uint64_t pack, int slot) __attribute__((always_inline)) {
        const unsigned int sel = __builtin_amdgcn_perm((unsigned int)(pack >> 32), (unsigned int)pack, selv);
        const unsigned int off = (sel << ZSH) + dma_off;
        dma_off += dma_inc;
#ifndef PS2_EXP_NODMA
#if PS2_DMA_BUFFER
        ps2_dma16b(zrsrc, off, raw_lds + (unsigned)slot * 1024u);
#else
        ps2_dma16(zbase, off, raw_lds + (unsigned)slot * 1024u);
#endif
#else
        asm volatile("" :: "v"(off));
#endif
    };
#pragma unroll
    for (int k = 0; k < PS2_AHEAD; ++k) issue(pp[k], k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    double hv = 0.0;
    // !HRING: all 64 lanes execute the store -- lanes with nothing to store point out of range -- so the count of vector-memory
    // operations between a DMA and its wait is the same on every path.  HRING: the pass's values go into the ring, the chain
    // wave's memory queue holds its DMAs and nothing else (hd_seen: hist_done as last read, a pass ago).
    [[maybe_unused]] unsigned int hposted = 0u, hd_seen = 0u;
#ifdef DEMCZ_STAMPS
    unsigned long long sa_hfull = 0;       // polls of a full history ring
#endif
    [[maybe_unused]] const int hkk = w, hr = lane;         // HRING: its chain among the workgroup's, its row of the pass
    auto store_history = [&](unsigned int off) __attribute__((always_inline)) {
        if constexpr (HRING) {
            if (hist) {
                while (hposted - hd_seen >= (unsigned int)PS2_HSLOTS) {          // (the publisher is PS2_HSLOTS passes behind: rare)
                    __builtin_amdgcn_s_sleep(1);
#ifdef DEMCZ_STAMPS
                    ++sa_hfull;
#endif
                    hd_seen = __hip_atomic_load(&hist_done[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (hr < HROWS) hist_ring[(int)(hposted % PS2_HSLOTS) * HSL + hr * (HKK + 1) + hkk] = hv;
                ++hposted;
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_store(&hist_seq[w], hposted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            hist_store(hv, off);
        }
    };

    [[maybe_unused]] double m[DDPP ? 1 : R][DDPP ? 1 : D];
    [[maybe_unused]] double Dg[R][DD_NQMAX];
    auto mget = [&](int j, int p) __attribute__((always_inline)) -> double { return m[DDPP ? 0 : j][DDPP ? 0 : p]; };
    constexpr int NPIECE = DP / 2;
    double logu = 0.0;
    [[maybe_unused]] double temp = 1.0;
    double za_f = 0.0, zb_f = 0.0, zt_f = 0.0;
    // (Measured and dropped: running the pass on lanes 0..31 only and reading, per node, only the rows it adds -- 22 KB of LDS
    //  traffic per pass down to 8 KB: 103 -> 99.5 us per 1000 generations, for hand-set lane masks the compiler does not know of.)
    auto load_rows = [&]() __attribute__((always_inline)) {
        if constexpr (DDPP) {
#pragma unroll
            for (int u = 0; u < R; ++u)
#pragma unroll
                for (int q = 0; q < DD_NQMAX; ++q)
                    if (q < DD_NQ[u]) Dg[u][q] = dptr[u][q * DD_KN[u]];
        } else {
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
            for (int q = 0; q < DP / 2; ++q) {
                const double2 t = reinterpret_cast<const double2*>(mrow[j])[q];
                m[j][2 * q] = t.x;
                if (2 * q + 1 < D) m[j][2 * q + 1] = t.y;
            }
        }
    };
    auto write_increment = [&]() __attribute__((always_inline)) {
        const double diff = za_f - zb_f;
        const double t1 = scale * diff;
        const double t2 = eps_p * zt_f;
        *sdw_p = t1 + t2;              // (every lane stores: the lanes that form nothing write the pad -- no branch in the pass)
    };
    // Front end of the pass whose raw slot is `slot`, in three parts that the pass loop places where their latencies hide:
    // (1) the raw values out of the slot (LDS reads); (2) the increments into LDS; (3) the DMA of the pass AHEAD after it (into
    // `slot_dma`) and this pass's node rows into registers.
#ifdef DEMCZ_STAMPS
    unsigned long long sa_spins = 0;       // poll iterations of all waits (scripts/ps2_stamps.py)
#endif
    uint64_t pr_f = 0;
    double logu_n = 0.0;
    [[maybe_unused]] double temp_n = 1.0;
    auto front_reads = [&](int slot, bool counted) __attribute__((always_inline)) {
        const unsigned char* rw = raw_w + slot * 1024;
        // behind this slot's DMA in program order: AHEAD - 1 whole passes (a DMA each; !HRING: and a history store) and, !HRING,
        // this pass's store
#ifndef PS2_EXP_NOWAIT        // (timing experiments only -- results are garbage: scripts/ab_ps2.sh)
        if (counted) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HRING ? PS2_AHEAD - 1 : 2 * PS2_AHEAD - 1) : "memory");
#endif
        za_f = *reinterpret_cast<const double*>(rw + zao);
        zb_f = *reinterpret_cast<const double*>(rw + zao + HW * 16);
        zt_f = *reinterpret_cast<const double*>(rw + zto);
        pr_f = *reinterpret_cast<const uint64_t*>(rw + ixnext);
        logu_n = *reinterpret_cast<const double*>(rw + lgo);
        if constexpr (TEMPER) temp_n = *reinterpret_cast<const double*>(rw + tko);
    };
    // The test for rows other waves have not published yet, as a LANE MASK in scalar registers (round 5): two 64-bit compares
    // straight into masks, OR, AND with the lanes that form increments.  As a per-lane bool fed to a ballot it compiled to
    // v_cndmask + v_cmp -- a temporary VGPR, which the register allocator took from the destinations of LDS reads still in flight:
    // an s_waitcnt lgkmcnt(0) at the end of EVERY pass.
    const unsigned long long flmask = __builtin_amdgcn_ballot_w64(fl);
    auto front_bad = [&]() __attribute__((always_inline)) -> unsigned long long {
#ifndef PS2_EXP_NOBADTEST      // (timing experiment only: no test for unpublished rows -- results are garbage where a row was missing)
        if constexpr (LIVE) {
            return (sentinel_lanes(za_f) | sentinel_lanes(zb_f)) & flmask;
        }
#endif
        return 0ull;
    };
    auto front_rest = [&](int slot_dma) __attribute__((always_inline)) {
        issue(pr_f, slot_dma);
        wave_lds_handoff();
        load_rows();
    };
    // LIVE: rows that other waves had not published when the DMA read them -- asked for again (sc1 loads) until they are
    // there, increments and node rows redone.  Returns true when the wait was abandoned.
    auto reread = [&](bool bad, int slot, int gpass) __attribute__((always_inline)) -> bool {
        const uint64_t ix = *reinterpret_cast<const uint64_t*>(raw_w + slot * 1024 + ixown);
        const uint32_t i1 = (uint32_t)ix, i2 = (uint32_t)(ix >> 32);
        int spins = 0;
        while (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
            if (spins > 0) {
                if (live_poll_abandon(P, spins, bad, is_sentinel(za_f) ? i1 : i2, gpass)) return true;
                __builtin_amdgcn_s_sleep(1);
            } else {
                spins = 1;
            }
            if (bad) {
                if (is_sentinel(za_f)) za_f = live_reload(P, &P.Z[(int64_t)i1 * ZSC + fp]);
                if (is_sentinel(zb_f)) zb_f = live_reload(P, &P.Z[(int64_t)i2 * ZSC + fp]);
                bad = is_sentinel(za_f) | is_sentinel(zb_f);
            }
        }
#ifdef DEMCZ_STAMPS
        sa_spins += (unsigned long long)spins;
#endif
        wave_lds_handoff();
        write_increment();
        wave_lds_handoff();
        load_rows();
        return false;
    };

    {
        front_reads(0, false);                           // the first pass's front end (its DMA: that of pass AHEAD)
        const unsigned long long bad0 = front_bad();
        write_increment();
        front_rest(PS2_AHEAD);
        logu = logu_n;
        if constexpr (TEMPER) temp = temp_n;
        if constexpr (LIVE) {
            if (bad0 != 0ull) {
                if (reread(((bad0 >> lane) & 1ull) != 0ull, 0, 0)) { leave(); return; }
            }
        }
    }

    int ip = 0;
    int64_t nb = 0;
    unsigned int cnt_total = 0, cnt_first = 0;
#ifdef DEMCZ_STAMPS
    // diagnostic build (scripts/ps2_stamps.py): shader-clock sums per segment of a pass.  A stamp drains the wave's outstanding
    // LDS / scalar-memory operations, so the segments add up to MORE than an unstamped pass: read them as proportions.
    unsigned long long sa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sa_nbad = 0, sa_wait = 0;
    const unsigned long long sa_start = __builtin_readcyclecounter();
    const unsigned long long sa_rt0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz, the same clock on every CU: start skew
#define PS2_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); sa[i] += t_ - sa_t; sa_t = t_; } while (0)
#else
#define PS2_T(i) do { } while (0)
#endif
    const bool lane_state = nn == 0;
    // one pass; S = ip mod 3 (its own raw slot: the one its front end's DMA refills).  Returns 0: go on, 1: that was the last
    // pass, 2: a LIVE wait was abandoned.
    auto pass = [&](auto slot_tag, auto first_tag) __attribute__((always_inline)) -> int {
        constexpr int S = decltype(slot_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;       // the launch's first pass: no history of a pass before it to store
        constexpr int SN = (S + 1) % PS2_SLOTS;
#ifdef DEMCZ_STAMPS
        unsigned long long sa_t = __builtin_readcyclecounter();
#endif
        // The pass is written in the order it should ISSUE (one wave per SIMD issues in order: a stalled instruction stalls
        // everything behind it), and the scheduling barriers keep the compiler from re-ordering across the phases:
        //   A  history store of the pass before, the next pass's raw values out of its slot      (LDS latency hides under B)
        //   B  every node's candidate: state + its rows in order                                 (25 dependent-in-fives adds)
        //   C  log-density of the candidates; the next pass's increments into LDS between its fmas
        //   D  ancestor's log-density asked for (bpermute); in its shadow the table write, the DMA, the next pass's node rows
        //   E  accept tests, the path taken, the winner; its row asked for
        //   F  in that read's shadow: history values, counts, the boundary, the sentinel check
        const bool boundary = (--tb == 0);
        [[maybe_unused]] unsigned int pub_seen = 0u;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!HRING) {
            if constexpr (FIRST) {
                store_history(0xffffff00u);               // (nothing yet: out of range, but the operation is there to be counted)
            } else {
                store_history(h_off);                     // the pass before's
                h_off += h_inc;
            }
        }
        front_reads(SN, true);
        if constexpr (LIVE) {
            if (boundary) pub_seen = __hip_atomic_load(&pub_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_sched_barrier(0);
        const double logu_c = logu;
        [[maybe_unused]] const double temp_c = temp;
        double cand[D];
#pragma unroll
        for (int p = 0; p < D; ++p) cand[p] = x[p];
        if constexpr (DDPP) {
            if constexpr (D == 5) {
#include "demcz_pw_ddpp_5.inc"
            } else if constexpr (D == 4) {
#include "demcz_pw_ddpp_4.inc"
            } else if constexpr (D == 3) {
#include "demcz_pw_ddpp_3.inc"
            } else {
#include "demcz_pw_ddpp_2.inc"
            }
        } else {
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
            for (int p = 0; p < D; ++p) cand[p] = cand[p] + mget(j, p);
        }
        PS2_T(0);                      // history store, raw values asked for, candidate adds
        __builtin_amdgcn_sched_barrier(0);
        // LIVE: the pass before's history values into the publisher's ring -- here, among the pass's other LDS writes, long after
        // the reads that fetched them (at the head of the pass the wave would wait for those reads, and for these writes)
#ifndef PS2_EXP_NOPOST         // (timing experiment: no history at all)
        if constexpr (HRING && !FIRST) store_history(0u);
#endif
        write_increment();
        const unsigned long long bad_n = front_bad();
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
        PS2_T(1);                      // log-density + the next pass's increments
        __builtin_amdgcn_sched_barrier(0);
        // the state's own row keeps the state's log-density (lane 0's candidate IS the state)
        const double lb = lane_state ? xlp : lpp;
        unsigned int m32, path, accp;
        unsigned long long chg_a, chg_r;
        const unsigned long long lbb = (unsigned long long)__double_as_longlong(lb);
        // The ancestor's log-density: two ds_bpermute, FIRST in the LDS queue, so that the table write, the DMA and the next
        // pass's row reads are issued in their shadow.  Written as asm: the compiler sinks the builtin to its first use (behind
        // those reads: the accept tests would then wait for all of them), and its own wait for the result would be lgkmcnt(0).
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
        issue(pr_f, S);
        wave_lds_handoff();
        load_rows();
        PS2_T(2);                      // bpermute asked for, table write, DMA issue, node rows asked for
        __builtin_amdgcn_sched_barrier(0);
        {
            // behind the two bpermutes in this wave's LDS queue: CR/2 table writes and R * NPIECE row reads; LDS operations
            // complete in order, and the wait's field holds at most 15
            constexpr int BEHIND = CR / 2 + R * NPIECE;
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
        PS2_T(3);                      // accept tests, path
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
            if constexpr (HRING) hd_seen = __hip_atomic_load(&hist_done[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        {
            const unsigned int chm = (accp & (unsigned int)chg_a) | (path & ~m32 & (unsigned int)chg_r);
            cnt_total += (unsigned int)__builtin_popcount(chm);
            if constexpr (FIRST) cnt_first = (chm >> 1) & 1u;
        }
        PS2_T(4);                      // winner's row asked for, history values, counts
        if (boundary) {                // a generation divisible by K ended the pass: runchain!'s append, demcz.jl:88-91
            const double v = ct_w[win * CR + ((lane < D) ? lane : 0)];
            if constexpr (LIVE) {
                while (pub_seen + (unsigned int)PS_PUB <= (unsigned int)nb) {          // (bounded by the publisher: demcz_kernels_rec.h)
                    __builtin_amdgcn_s_sleep(1);
                    pub_seen = __hip_atomic_load(&pub_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if (lane < D) pub_rows[(w * PS_PUB + (int)((unsigned int)nb % PS_PUB)) * D + lane] = v;
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_store(&pub_seq[w], (unsigned int)nb + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if constexpr (!HRING) ps2_wake_publisher();
            } else {
                if (lane < D && P.do_append) P.Zw[(P.M_append + nb * P.N + c) * P.ZS + lane] = v;
            }
            if (lane < D && P.snap) P.snap[nb * P.N * D + c + P.N * lane] = v;
            ++nb;
            tb = tbK;
        }
        PS2_T(5);                      // boundary: hand the row to the publisher
        if (++ip == npass) return 1;
        if constexpr (LIVE) {
            // only now -- this wave's own row is on its way -- may it wait for rows of other waves
            if (__builtin_expect(bad_n != 0ull, 0)) {
#ifdef DEMCZ_STAMPS
                ++sa_nbad;
                const unsigned long long sa_w0 = __builtin_readcyclecounter();
#endif
                if (reread(((bad_n >> lane) & 1ull) != 0ull, SN, ip * R)) return 2;
#ifdef DEMCZ_STAMPS
                sa_wait += __builtin_readcyclecounter() - sa_w0;      // (the wait alone, timed only where there is one: round 5)
#endif
            }
        }
        PS2_T(6);                      // the test for rows not yet published (+ the wait where there is one; sa_wait: the waits alone)
        wave_lds_handoff();
        return 0;
    };
    int st = pass(std::integral_constant<int, 0>{}, std::true_type{});
    while (!st) {
        st = pass(std::integral_constant<int, 1>{}, std::false_type{});
        if (st) break;
        st = pass(std::integral_constant<int, 2>{}, std::false_type{});
        if (st) break;
        if constexpr (PS2_SLOTS >= 4) {
            st = pass(std::integral_constant<int, 3 % PS2_SLOTS>{}, std::false_type{});
            if (st) break;
        }
        if constexpr (PS2_SLOTS >= 5) {
            st = pass(std::integral_constant<int, 4 % PS2_SLOTS>{}, std::false_type{});
            if (st) break;
        }
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
        o[8] = __builtin_readcyclecounter() - sa_start; o[7] = sa_hfull; o[11] = sa_nbad; o[12] = sa_spins; o[14] = (unsigned long long)npass; o[15] = 2;
        o[6] = sa_wait;        // (round 5: the waits alone; the segment between stamps 5 and 6 -- the test, which every pass pays -- is only in the total)
        o[9] = sa_rt0; o[10] = __builtin_amdgcn_s_memrealtime();
        // where the wave ran: HW_REG_HW_ID (register 4: simd 5:4, cu 11:8, sh 12, se 15:13) and HW_REG_XCC_ID (register 20)
        o[13] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
    }
#endif
#undef PS2_T
    leave();
}

}  // namespace demcz
