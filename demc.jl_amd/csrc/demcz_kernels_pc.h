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
//       record per (generation, chain): the D normals, log u of the accept test, the two row indices
//       (against the archive size that generation will see);
//   consumer workgroups (eight lanes per chain) read the records of THIS launch (written by the
//       previous launch's producers) a chunk of generations at a time -- a chunk's archive rows, normals
//       and log u are asked for while the chunk before it computes, its row indices a chunk earlier still --
//       form the proposal increments of the chunk up front, then run the state-dependent part --
//       proposal, log-density, accept, ballot, history -- from registers: ~66 instructions a generation.
//
// Both halves are ONE launch (workgroups [0, consumer_blocks) consume, the rest produce), so they
// overlap on different CUs with no events or second stream; a launch boundary orders a launch's
// records before their use.  The host knows what the next launch will see: this M plus N rows per K
// boundary passed (demcz_capi.hip, pc_prepare).  On one GPU a launch runs through many K boundaries and
// the consumers hand the appended rows to each other inside it (LIVE, demcz_kernels_rec.h).
// Arithmetic and operation order are those of the one-lane kernel: bit-identical results.
#pragma once

#include "demcz_kernels_ml.h"     // philox_blocks: rocRAND's round function as a counter -> block map

#pragma clang fp contract(off)

namespace demcz {


// ------------------------------------------------------------------------------------------------
// The consumer: 8 lanes per chain, two roles.  Front-end of a chunk of <= 10 generations: lane u of a
// chain's group fetches what generation u needs (the two archive rows whole, its normals), forms that
// generation's increments and puts them into LDS -- the memory latency is paid once per chunk.  Then
// every lane of the group runs the state-dependent part of the chunk redundantly from the whole state
// (no cross-lane traffic there), lane p storing element p of the history row (lane d: log_obj).
// (A one-lane-per-chain consumer has to hold 3d+1 doubles per prefetched generation: 5-generation
// chunks, 9.0 us per C2 window against 7.0 for the first 8-lane version.)
// ------------------------------------------------------------------------------------------------
constexpr int PC8_CHUNK = 10;

// LIVE launches: a workgroup is TWO waves.  Wave 0 runs the chains; wave 1 is the PUBLISHER: at a K boundary wave 0
// leaves its eight chains' rows in LDS and goes on, wave 1 picks them up and writes them through to the archive (sc1
// stores) for the other waves to find.  Why a second wave: a wave's vector-memory instructions behind one of its own
// write-through stores wait for that store's round trip to memory (measured: ~0.4 us of every K-window, wherever the
// next load or store of the wave happened to be) -- the publisher has nothing behind its store.
constexpr int PC8_LIVE_WAVES = 2;

template <int TARGET, int D, bool LIVE, bool TEMPER>
__global__ void __launch_bounds__(LIVE ? 64 * PC8_LIVE_WAVES : 64) window_kernel_pc8(const WindowParams P)
{
    static_assert(TARGET == TARGET_MVNORMAL || TARGET == TARGET_ISO_QUAD, "split layout: MvNormal / isotropic targets");
    constexpr int L = 8, G = 64 / L, DP = ((D + 1) / 2) * 2;
    constexpr int NP = (D + L - 1) / L;                    // history elements a lane stores: r, r+8, ...
    constexpr int CH = (D <= 5) ? PC8_CHUNK : PC8_CHUNK / 2;      // (a lane holds the prefetched rows and normals of ceil(CH / 8) generations)
    constexpr int WAVES = LIVE ? PC8_LIVE_WAVES : 1;
    DEMCZ_STAMP(P, 0);
    if ((int64_t)blockIdx.x >= P.consumer_blocks) {        // every wave of a producer workgroup is one 64-lane producer unit
        pc_produce<D>(P, ((int64_t)blockIdx.x - P.consumer_blocks) * WAVES + (int64_t)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
        DEMCZ_STAMP(P, 7);
        return;
    }
    constexpr int DPL = DP + 2;                            // LDS row of a (chain, generation): DP increments, log u, pad
    __shared__ __attribute__((aligned(16))) double sdelta[G * CH * DPL];
    // LIVE: the rows of a boundary on their way from wave 0 to the publisher (two boundaries' worth), and the hand-shake:
    // pub_seq = boundaries wave 0 has left here, pub_done = boundaries the publisher has written out, pub_exit = wave 0 is leaving
    __shared__ double pub_rows[LIVE ? 2 * G * D : 1];
    __shared__ unsigned int pub_seq, pub_done, pub_exit;
    if constexpr (LIVE) {
        if (threadIdx.x == 0) { pub_seq = 0u; pub_done = 0u; pub_exit = 0u; }
        __syncthreads();
        if (threadIdx.x >= 64) {
            const int l = (int)threadIdx.x - 64;
            const int64_t c0 = (int64_t)xcd_block(P) * G;
            unsigned int done = 0u;
            while (true) {
                const unsigned int seq = __hip_atomic_load(&pub_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (seq != done) {
                    const double* rows = pub_rows + (done & 1u) * (G * D);
                    for (int e = l; e < G * D; e += 64) {
                        const int g = e / D, p = e % D;
                        const double v = rows[e];
                        if (c0 + g < P.N && P.do_append) live_publish(P, (int64_t)done, c0 + g, p, v);
                    }
                    asm volatile("" ::: "memory");
                    ++done;
                    if (l == 0) __hip_atomic_store(&pub_done, done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    continue;
                }
                if (__hip_atomic_load(&pub_exit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
                    // (pub_seq is written before pub_exit: what is read now is final)
                    if (__hip_atomic_load(&pub_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == done) break;
                    continue;
                }
                // (never before wave 0 has said so, not even in a launch that is being abandoned: wave 0 waits for room in
                //  pub_rows without a poll limit, bounded by this loop's progress -- demcz_kernels_ps.h)
                __builtin_amdgcn_s_sleep(1);
            }
            return;
        }
    }
    // wave 0 tells the publisher that nothing more is coming (every way out of a LIVE launch passes here)
    auto leave = [&]() {
        if constexpr (LIVE) {
            asm volatile("" ::: "memory");
            if (threadIdx.x == 0) __hip_atomic_store(&pub_exit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    const int lane = threadIdx.x, r = lane % L, gq = lane / L;
    // groups beyond the last chain shadow chain N-1 and store nothing
    const int64_t c_own = (int64_t)xcd_block(P) * G + gq;      // (XCD-aware: demcz_kernels.h)
    const bool live = c_own < P.N;
    const int64_t c = live ? c_own : P.N - 1;

    double x[D], muc[D], Wc[(TARGET == TARGET_MVNORMAL) ? D * (D + 1) / 2 : 1];
#pragma unroll
    for (int p = 0; p < D; ++p) { x[p] = P.Xcur[c + P.N * p]; muc[p] = P.tp.mu[p]; }
    if constexpr (TARGET == TARGET_MVNORMAL) {
#pragma unroll
        for (int i = 0; i < D * (D + 1) / 2; ++i) Wc[i] = P.tp.Wp[i];
    }
    int pk[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) pk[k] = (r + L * k < D) ? r + L * k : 0;
    const double c0c = P.tp.c0;
    double lp = P.lpcur[c];
    const double scale = (D == 1) ? P.gamma : P.gamma / sqrt((double)(2 * D));
    int to_b = P.to_boundary;
    int64_t nb = 0;
    unsigned int cnt_total = 0, cnt_first = 0;             // accept mask by ballot (WindowParams::acc_out)
    const unsigned long long speaks_mask = __builtin_amdgcn_ballot_w64((r == 0) && live);   // the lanes that speak for a chain in the ballot
    // What a lane stores every generation: its own element(s) of the history row -- tracked beside the
    // replicated state (own' = own + its increment: the same addition on the same values) -- and, where
    // the group has a lane to spare (D < 8), lane D stores log_obj in the same instruction.  Pointers
    // advance by a per-lane stride; lanes with nothing to store are masked off once, here.
    constexpr bool LP_MERGED = (D < L);
    double own[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        own[k] = x[(L * k < D) ? L * k : 0];
#pragma unroll
        for (int j = 1; j < L; ++j)
            if (L * k + j < D) own[k] = (r == j) ? x[L * k + j] : own[k];
    }
    const bool lp_lane = LP_MERGED ? (r == D) : (r == L - 1);
    const bool hist_on = (P.chain != nullptr) && live;
    double* sptr[NP];
    int64_t sstride[NP];
    bool son[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const bool xl = (r + L * k < D);
        sptr[k] = P.chain + c + P.N * ((int64_t)pk[k] + (int64_t)D * P.slot_first);
        sstride[k] = P.N * (int64_t)D;
        son[k] = hist_on && xl;
        if (LP_MERGED && k == 0 && lp_lane) {
            sptr[k] = P.logobj + c + P.N * P.slot_first;
            sstride[k] = P.N;
            son[k] = hist_on;
        }
    }
    double* lobj = P.logobj + c + P.N * P.slot_first;      // D >= 8 only: lane 7 stores log_obj separately
    // record rows of this chain; a generation is the next double
    const double* rec_lg = P.rec_in + ((int64_t)D * P.N + c) * P.rec_stride;
    const double* rec_ix = P.rec_in + ((int64_t)(D + 1) * P.N + c) * P.rec_stride;
    // archive addressing: row stride and the whole archive fit 32 bits of byte offset (condition of this
    // layout), the row stride is a power of two for every dimension built: one shift-add per element
    constexpr int ZSC = (D <= 2) ? 2 : (D <= 4) ? 4 : ((D + 7) / 8) * 8;
    static_assert((ZSC & (ZSC - 1)) == 0, "row stride must be a power of two");
    constexpr int ZSHIFT = (ZSC == 2) ? 4 : (ZSC == 4) ? 5 : (ZSC == 8) ? 6 : 7;
    // Front-end roles: lane r of a chain's group fetches what generation u = r (and, second round, 8 + r) of
    // the chunk needs -- the two archive rows whole (16-byte loads), the D normals -- and forms that
    // generation's D increments.  (Fewer, wider vector-memory instructions than one lane per parameter: a
    // chunk is 12 + 10 + 7 of them at d = 5 instead of 60; their issue was a third of the chunk's front-end.)
    constexpr int ROUNDS = (CH + L - 1) / L;
    constexpr int HW = (D + 1) / 2;                       // 16-byte pieces of a row
    static_assert(2 * HW <= ZSC, "a row's last 16-byte piece stays inside the row");
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(P.Z), 0, (int)P.z_bytes, 0x00020000);      // (the archive's own size: a wrong row index reads zeros, not a neighbour)
    // The prefetch reads every row through the ordinary cached path, also in a LIVE launch.  A row this launch appends
    // is published by write-through (sc1) stores, so memory always holds its final doubles or the sentinel; what an
    // ordinary load can add to that is a STALE copy from this XCD's L2 / the CU's L1 -- which, each double of a row being
    // written once, can only show the sentinel where the final value is not yet seen, never a wrong value.  Slots that
    // show a sentinel are asked for again with sc1 loads (served past L1 and L2, demcz_kernels_rec.h) at the start of
    // their chunk.  sc1 loads are kept out of the prefetch because they are slow to ISSUE: measured on the same
    // gathers, 2.31 -> 2.78 us per 10 generations when every row goes through them (they queue behind the wave's own
    // outstanding stores), and even a few lanes' worth per instruction costs the same.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto piece_to = [&](const u32x4 v, double& lo, double& hi) {
        lo = __longlong_as_double((long long)(((uint64_t)v.y << 32) | v.x));
        hi = __longlong_as_double((long long)(((uint64_t)v.w << 32) | v.z));
    };
    auto row_piece_sc1 = [&](uint32_t byte_off, double& lo, double& hi) {
        piece_to(__builtin_amdgcn_raw_buffer_load_b128(zrsrc, (int)byte_off, 0, 16), lo, hi);
    };
    // one whole archive row (HW 16-byte pieces)
    auto row_fetch = [&](uint32_t off, double (&dst)[2 * HW]) {
#pragma unroll
        for (int j = 0; j < HW; ++j) piece_to(__builtin_amdgcn_raw_buffer_load_b128(zrsrc, (int)(off + 16u * j), 0, 0), dst[2 * j], dst[2 * j + 1]);
    };
    const double* rec_n[D];                                // normals of this chain, parameter p
#pragma unroll
    for (int p = 0; p < D; ++p) rec_n[p] = P.rec_in + ((int64_t)((D == 1) ? 0 : p) * P.N + c) * P.rec_stride;
    double epsall[D];
#pragma unroll
    for (int p = 0; p < D; ++p) epsall[p] = P.eps[p];
    // row indices of this lane's generations of the first chunk; from then on fetched one chunk ahead
    double ixn[ROUNDS];        // (bit patterns of the packed 32-bit index pairs)
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) ixn[rd] = rec_ix[rd * L + r];

    // A chunk is up to CH generations whose draws are fetched together.  It ends at the next K boundary:
    // the append then sits between chunks, not inside the generation code, and in a LIVE launch the
    // generation after a boundary draws from rows that are only being written while this chunk computes
    // (K = 1: one generation per chunk).
    //
    // Software pipeline over chunks: the loads of chunk k+1 (archive rows, normals, log u; its row indices
    // were fetched a chunk before) are issued right after chunk k's increments have gone to LDS and fly
    // during chunk k's generations; chunk k+1 begins with the data already in registers.  The generation code
    // reads its increments from LDS one generation ahead instead of holding the whole chunk in registers --
    // that is what frees the registers the prefetch lives in.
    if constexpr (LIVE) {       // an earlier launch of the run already failed: do not wait again
        if (__hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { leave(); return; }
    }
    [[maybe_unused]] const int stamp_g0 = (P.ngen > 5 * CH) ? 5 * CH : 0;      // diagnostic build: the chunk that is timed
#ifdef DEMCZ_STAMPS
    // diagnostic build: shader-clock sums over ALL chunks of the launch (consumer workgroup's slots 8..15 of its stamp row)
    unsigned long long sa_front = 0, sa_issue = 0, sa_gen = 0, sa_tail = 0, sa_waits = 0, sa_polls = 0, sa_chunks = 0, sa_t = 0;
#define DEMCZ_TICK(acc) do { const unsigned long long t_ = __builtin_readcyclecounter(); (acc) += t_ - sa_t; sa_t = t_; } while (0)
#else
#define DEMCZ_TICK(acc) do { } while (0)
#endif
    // in flight for the chunk to come: this lane's generations u = rd * 8 + r
    double za[ROUNDS][2 * HW], zb[ROUNDS][2 * HW], zt[ROUNDS][D], lgv[ROUNDS];
    uint32_t o1[ROUNDS], o2[ROUNDS];
    auto chunk_len = [&](int g0, int tb) {
        int n = (P.ngen - g0 < CH) ? P.ngen - g0 : CH;
        return (tb < n) ? tb : n;
    };
    auto issue = [&](int g0, int len) {          // loads of the chunk [g0, g0 + len); consumes ixn, refills it for the chunk after
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            const int u = rd * L + r;
            // slots past the end of the chunk read row 0: their own rows may not exist yet (LIVE: never wait for them)
            const uint64_t ii = (uint64_t)__double_as_longlong(ixn[rd]);
            o1[rd] = (u < len) ? (uint32_t)ii << ZSHIFT : 0u;
            o2[rd] = (u < len) ? (uint32_t)(ii >> 32) << ZSHIFT : 0u;
            row_fetch(o1[rd], za[rd]);
            row_fetch(o2[rd], zb[rd]);
            const int gu = g0 + ((u < len) ? u : len - 1);
#pragma unroll
            for (int p = 0; p < ((D == 1) ? 1 : D); ++p) zt[rd][p] = rec_n[p][gu];
            lgv[rd] = rec_lg[gu];
        }
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) ixn[rd] = rec_ix[g0 + len + rd * L + r];      // (the buffers are padded for the overshoot)
    };
    // Every launch prefetches across chunks (above).  In a LIVE launch the rows the last boundaries appended may not be
    // there yet when the prefetch reads them -- they read as the sentinel; such a slot is asked for again at the start
    // of its chunk until the row is there.  Every other row's latency has been taken off the chunk by the prefetch.
    auto slot_bad = [&](int rd) {
        bool b = false;
#pragma unroll
        for (int p = 0; p < D; ++p) b |= is_sentinel(za[rd][p]) | is_sentinel(zb[rd][p]);
        return b;
    };
    auto refetch = [&](int rd) {      // a row is written by one wave, 8 bytes at a time: re-read whole rows that still hold a sentinel
        bool ba = false, bb = false;
#pragma unroll
        for (int p = 0; p < D; ++p) { ba |= is_sentinel(za[rd][p]); bb |= is_sentinel(zb[rd][p]); }
        if (ba) {
#pragma unroll
            for (int j = 0; j < HW; ++j) row_piece_sc1(o1[rd] + 16u * j, za[rd][2 * j], za[rd][2 * j + 1]);
        }
        if (bb) {
#pragma unroll
            for (int j = 0; j < HW; ++j) row_piece_sc1(o2[rd] + 16u * j, zb[rd][2 * j], zb[rd][2 * j + 1]);
        }
    };
    // this lane's generation of round rd: increments and log u into the LDS row of (chain, generation)
    auto form_row = [&](int rd) {
        const int u = rd * L + r;
        double dv[DPL];
#pragma unroll
        for (int p = 0; p < DP; ++p) {
            if (p < D) {
                const double diff = za[rd][p] - zb[rd][p];
                const double t1 = scale * diff;
                const double t2 = epsall[p] * zt[rd][(D == 1) ? 0 : p];
                dv[p] = t1 + t2;
            } else {
                dv[p] = 0.0;
            }
        }
        dv[DP] = lgv[rd];
        dv[DP + 1] = 0.0;
        if (u < CH) {
#pragma unroll
            for (int j = 0; j < DPL / 2; ++j)
                reinterpret_cast<double2*>(sdelta + (gq * CH + u) * DPL)[j] = make_double2(dv[2 * j], dv[2 * j + 1]);
        }
    };
    auto read_row = [&](int u, double (&dd)[DPL], double (&mm)[NP]) {
        const double* row = sdelta + (gq * CH + u) * DPL;
#pragma unroll
        for (int j = 0; j < DPL / 2; ++j) {
            const double2 t = reinterpret_cast<const double2*>(row)[j];
            dd[2 * j] = t.x;
            dd[2 * j + 1] = t.y;
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) mm[k] = row[pk[k]];              // this lane's own element(s)
    };
    // one generation (gi: its index in the launch) from its LDS row (increments, log u), this lane's own increment(s)
    // and, tempered, its temperature
    auto generation = [&](int gi, const double (&dd)[DPL], const double (&mm)[NP], [[maybe_unused]] double temp) {
        double xp[D];
#pragma unroll
        for (int p = 0; p < D; ++p) xp[p] = x[p] + dd[p];
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
        if constexpr (TEMPER) dlt = dlt / temp;
        const bool acc = dd[DP] < dlt;
#pragma unroll
        for (int p = 0; p < D; ++p) x[p] = acc ? xp[p] : x[p];
        {
            const double lp_new = acc ? lpp : lp;
            const unsigned int kc = wave_count_changed(lp_new, lp, speaks_mask);
            cnt_total += kc;
            cnt_first = (gi == 0) ? kc : cnt_first;
            lp = lp_new;
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const double ownp = own[k] + mm[k];
            own[k] = acc ? ownp : own[k];
            const double val = (LP_MERGED && k == 0 && lp_lane) ? lp : own[k];
            if (son[k]) *sptr[k] = val;      // (tried: idle lanes storing to a scrap word instead of the exec mask -- 2.24 -> 2.37 us per chunk)
            sptr[k] += sstride[k];       // next generation's slab: a stride, no per-lane multiply
        }
        if constexpr (!LP_MERGED) {
            if (hist_on && lp_lane) *lobj = lp;
            lobj += P.N;
        }
    };
    // Everything loaded so far (state, target constants, eps, the first row indices) is in registers before the pipeline
    // starts: a constant still in flight at the loop's entry can make its first use inside the loop wait for "everything
    // outstanding" on every trip -- the prefetched loads of the next chunk included (seen: 2.25 -> 2.97 us per chunk).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int len = chunk_len(0, to_b);
    issue(0, len);
    for (int g0 = 0; g0 < P.ngen; g0 += len) {
        if (g0 == stamp_g0) DEMCZ_STAMP(P, 5);
#ifdef DEMCZ_STAMPS
        if (P.stamps && threadIdx.x == 0) P.stamps[(size_t)blockIdx.x * 16 + 6] = 1000000ull + (unsigned long long)g0;   // progress
        sa_t = __builtin_readcyclecounter();
        ++sa_chunks;
#endif
        if (g0 > 0) len = chunk_len(g0, to_b);
        if constexpr (LIVE) {
            // Cheap filter first: the sentinel's high word is that of a negative NaN, above every finite value's, -inf's
            // and the canonical NaN's -- a max per value instead of a 64-bit compare
            uint32_t hmax = 0u;
#pragma unroll
            for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
                for (int p = 0; p < D; ++p) {
                    const uint32_t ha = (uint32_t)((uint64_t)__double_as_longlong(za[rd][p]) >> 32);
                    const uint32_t hb = (uint32_t)((uint64_t)__double_as_longlong(zb[rd][p]) >> 32);
                    hmax = max(hmax, max(ha, hb));
                }
            if (__builtin_amdgcn_ballot_w64(hmax >= (uint32_t)(LIVE_SENTINEL >> 32)) != 0ull) {
                // some slot of the wave holds a row that had not been published when the prefetch read it: ask again
                // until it is there (the prefetch has taken the ordinary latency off every other row already)
                bool bad = false;
#pragma unroll
                for (int rd = 0; rd < ROUNDS; ++rd) bad |= slot_bad(rd);
                int spins = 0;
#ifdef DEMCZ_STAMPS
                if (__builtin_amdgcn_ballot_w64(bad) != 0ull) ++sa_waits;
#endif
                while (__builtin_amdgcn_ballot_w64(bad) != 0ull) {           // wave-uniform
#ifdef DEMCZ_STAMPS
                    ++sa_polls;
#endif
                    unsigned waiting_row = 0;
#pragma unroll
                    for (int rd = 0; rd < ROUNDS; ++rd)
#pragma unroll
                        for (int p = 0; p < D; ++p) {
                            if (is_sentinel(za[rd][p])) waiting_row = o1[rd] >> ZSHIFT;
                            if (is_sentinel(zb[rd][p])) waiting_row = o2[rd] >> ZSHIFT;
                        }
                    if (spins > 0) {
                        if (live_poll_abandon(P, spins, bad, waiting_row, g0)) { leave(); return; }      // wave-uniform
                        __builtin_amdgcn_s_sleep(1);
                    } else {
                        spins = 1;
                    }
                    bad = false;
#pragma unroll
                    for (int rd = 0; rd < ROUNDS; ++rd) {
                        refetch(rd);
                        bad |= slot_bad(rd);
                    }
                }
            }
        }
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) form_row(rd);
        if (g0 == stamp_g0) DEMCZ_STAMP(P, 3);
        wave_lds_handoff();
        DEMCZ_TICK(sa_front);
        // the chunk after this one: its loads fly during this chunk's generations
        {
            const int tb_n = (to_b - len == 0) ? P.K : to_b - len;
            const int g0n = g0 + len;
            if (g0n < P.ngen) issue(g0n, chunk_len(g0n, tb_n));
            __builtin_amdgcn_sched_barrier(0);      // the generation code below must not be scheduled in front of the issue
        }
        if (g0 == stamp_g0) DEMCZ_STAMP(P, 2);
        DEMCZ_TICK(sa_issue);
        {
            // generations of the chunk: the LDS row of generation u + 1 is read while generation u computes
            [[maybe_unused]] double tmpr[CH];
            if constexpr (TEMPER) {
#pragma unroll
                for (int u = 0; u < CH; ++u) tmpr[u] = P.temperature[g0 + ((u < len) ? u : len - 1)];
            }
            // (tried: ping-pong buffers named at compile time instead of the copies below -- four instructions fewer per
            //  generation and 7 % SLOWER, 2.17 -> 2.33 us per chunk: the copies keep the next row's LDS reads early)
            double dcur[DPL], dnxt[DPL], mcur[NP], mnxt[NP];
            read_row(0, dcur, mcur);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (u < len) {       // wave-uniform
                    if (u + 1 < CH) read_row(u + 1, dnxt, mnxt);
                    generation(g0 + u, dcur, mcur, TEMPER ? tmpr[u] : 0.0);
#pragma unroll
                    for (int p = 0; p < DPL; ++p) dcur[p] = dnxt[p];
#pragma unroll
                    for (int k = 0; k < NP; ++k) mcur[k] = mnxt[k];
                }
            }
        }
        DEMCZ_TICK(sa_gen);
        to_b -= len;
        if (to_b == 0) {         // the chunk ended on a generation divisible by K: runchain!'s append, demcz.jl:88-91
            if constexpr (LIVE) {
                // the rows go to the publisher through LDS (the half of pub_rows it emptied two boundaries ago)
                while (__hip_atomic_load(&pub_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 2u <= (unsigned int)nb)
                    __builtin_amdgcn_s_sleep(1);
                double* rows = pub_rows + ((unsigned int)nb & 1u) * (G * D);
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int p = r + L * k;
                    if (p < D) rows[gq * D + p] = own[k];
                }
                asm volatile("" ::: "memory");                 // (one wave's LDS operations execute in order)
                if (threadIdx.x == 0) __hip_atomic_store(&pub_seq, (unsigned int)nb + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = r + L * k;
                if (p < D && live) {
                    if constexpr (!LIVE) { if (P.do_append) P.Zw[(P.M_append + nb * P.N + c) * P.ZS + p] = own[k]; }
                    if (P.snap) P.snap[nb * P.N * D + c + P.N * p] = own[k];
                }
            }
            to_b = P.K;
            ++nb;
        }
        if (g0 == stamp_g0) DEMCZ_STAMP(P, 4);
        wave_lds_handoff();      // sdelta is rewritten by the next chunk
        DEMCZ_TICK(sa_tail);
    }
#ifdef DEMCZ_STAMPS
    if (P.stamps && threadIdx.x == 0 && blockIdx.x < 65536u) {
        unsigned long long* o = P.stamps + (size_t)blockIdx.x * 16 + 8;
        o[0] = sa_front; o[1] = sa_issue; o[2] = sa_gen; o[3] = sa_tail; o[4] = sa_waits; o[5] = sa_polls; o[6] = sa_chunks;
    }
#endif
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = r + L * k;
        if (p < D && live) P.Xcur[c + P.N * p] = own[k];
    }
    if (r == 0 && live) P.lpcur[c] = lp;
    wave_store_counts(P, blockIdx.x, cnt_total, cnt_first);
    leave();
    DEMCZ_STAMP(P, 7);
}

}  // namespace demcz
