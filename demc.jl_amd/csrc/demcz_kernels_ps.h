// demcz_kernels_ps.h -- K1g: the consumer of the split layout for the smallest N: one WAVE per chain,
// five generations resolved per pass by evaluating every proposal they could need at once.
//
// At N = 1024 (BASELINE C2) the replicated consumer (demcz_kernels_pc.h) keeps 128 of the chip's 1024 SIMDs busy and
// its wave runs one generation after the other: ~66 instructions each, issue-bound.  What a generation needs from the
// one before it is ONE bit -- accepted or not; the increments of every generation are in the draw records long before.
// So a wave takes ONE chain and lane n of it takes node n of the binary tree of outcomes of the next R <= 5 generations
// (n = 1: generation 1's proposal; 2n / 2n+1: the next generation's proposal after n was rejected / accepted;
// 31 nodes).  Every lane forms its candidate by the very additions the serial order would have made
// (state + increments of the accepted generations on its path, in order, then its own generation's increment; a
// rejected generation adds -0.0, which changes no double) and evaluates the log-density ONCE; the accept tests of all
// 31 nodes are one vector compare into a lane mask, and the path actually taken is read off that mask with scalar
// instructions.  Results are those of the serial order bit for bit: the same operations on the same values, more of
// them (the untaken branches) and in parallel.  Cost of a pass: about what ONE generation costs the replicated
// consumer, for five.
//
// The wave does no ordinary vector-memory load in its loop: the draws and archive rows of a pass arrive by ONE
// LDS-DMA instruction (global_load_lds_dwordx4, 64 lanes x 16 bytes, per-lane source) issued two passes ahead into a
// ring of three 1-KiB slots, and are waited for with a counted s_waitcnt -- the compiler does not see these loads, so it
// cannot drain them early, and there is no register destination it could touch before the data lands
// (cdna_hip_programming.md, "What hipcc does not do").  History rows leave as buffer stores that every lane executes
// (lanes with nothing to store point out of range): the count of vector-memory operations between a DMA and its
// wait is the same on every path.
//
// LIVE launches (demcz_kernels_rec.h): as in the replicated consumer the first read of a row takes the cached path,
// a sentinel is asked for again with sc1 loads, and appended rows go through LDS to a publisher wave (one per
// workgroup of PS_CHAINS chain waves).
#pragma once

#include "demcz_kernels_pc.h"

#pragma clang fp contract(off)

namespace demcz {

constexpr int PS_CHAINS = 4;     // chain waves per workgroup
constexpr int PS_R = 5;          // generations per pass = depth of the tree of outcomes (2^5 - 1 = 31 nodes)
constexpr int PS_SLOTS = 3;      // ring of raw slots: the pass being worked on + two in flight
constexpr int PS_MAX_N = 2048;   // beyond ~2 waves per SIMD the replicated consumer (8 chains per wave) is the faster one

// one 16-byte piece per lane from a per-lane address into LDS at (wave-uniform) lds_dst + 16 * lane
__device__ __forceinline__ void ps_dma16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int TARGET, int D, bool LIVE, bool TEMPER>
__global__ void __launch_bounds__(64 * (PS_CHAINS + (LIVE ? 1 : 0))) window_kernel_ps(const WindowParams P)
{
    static_assert(TARGET == TARGET_MVNORMAL || TARGET == TARGET_ISO_QUAD, "split layout: MvNormal / isotropic targets");
    static_assert(D >= 2 && D <= 5, "a pass's rows, normals, log u and indices are one 64-lane DMA");
    constexpr int WAVES = PS_CHAINS + (LIVE ? 1 : 0);
    constexpr int HW = (D + 1) / 2;                        // 16-byte pieces of an archive row
    constexpr int ZSC = (D <= 2) ? 2 : (D <= 4) ? 4 : 8;   // archive row stride in doubles (demcz_create: ZS)
    constexpr int ZSH = (ZSC == 2) ? 4 : (ZSC == 4) ? 5 : 6;      // log2 of the row stride in bytes
    constexpr int DP = ((D + 1) / 2) * 2;                  // increments row in LDS
    constexpr int CR = ((D + 2) / 2) * 2;                  // candidate row in LDS: D doubles, log-density, pad
    // lanes of the DMA: [0, ROWL) archive rows (generation u, first / second row, piece j); [FL0, TL0) the D + 2 record
    // fields, three pieces = six generations each (fields 0..D-1 normals, D log u: this pass; D+1 row indices: the pass
    // two after it, whose rows are asked for when this slot is consumed); [TL0, TL0 + 3) temperatures; the rest idle
    constexpr int ROWL = PS_R * 2 * HW;
    constexpr int FL0 = ROWL;
    constexpr int TL0 = FL0 + 3 * (D + 2);
    static_assert(TL0 + 3 <= 64, "one DMA instruction per pass");
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if ((int64_t)blockIdx.x >= P.consumer_blocks) {        // every wave of a producer workgroup is one 64-lane producer unit
        pc_produce<D>(P, ((int64_t)blockIdx.x - P.consumer_blocks) * WAVES + w, lane);
        return;
    }
    __shared__ __attribute__((aligned(16))) unsigned char raw[PS_CHAINS][PS_SLOTS][1024];
    __shared__ __attribute__((aligned(16))) double sdelta[PS_CHAINS][(PS_R + 1) * DP];       // row PS_R: negative zeros
    __shared__ __attribute__((aligned(16))) double ctab[PS_CHAINS][2][32 * CR];
    // LIVE: a boundary's row on its way from a chain wave to the publisher (two boundaries' worth per chain wave) and the
    // hand-shake: pub_seq = boundaries the chain wave has left here, pub_done = written out by the publisher, pub_exit = leaving
    __shared__ double pub_rows[LIVE ? PS_CHAINS * 2 * D : 1];
    __shared__ unsigned int pub_seq[PS_CHAINS], pub_done[PS_CHAINS], pub_exit[PS_CHAINS];
    if constexpr (LIVE) {
        if (threadIdx.x < PS_CHAINS) { pub_seq[threadIdx.x] = 0u; pub_done[threadIdx.x] = 0u; pub_exit[threadIdx.x] = 0u; }
        __syncthreads();
        if (w == PS_CHAINS) {       // the publisher (why a wave of its own: demcz_kernels_pc.h, PC8_LIVE_WAVES)
            unsigned int done[PS_CHAINS];
#pragma unroll
            for (int cw = 0; cw < PS_CHAINS; ++cw) done[cw] = 0u;
            unsigned int idle = 0u;
            while (true) {
                bool any = false;
                int gone = 0;
#pragma unroll
                for (int cw = 0; cw < PS_CHAINS; ++cw) {
                    const unsigned int seq = __hip_atomic_load(&pub_seq[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (seq != done[cw]) {
                        const int64_t c = (int64_t)blockIdx.x * PS_CHAINS + cw;
                        const double* rows = pub_rows + (cw * 2 + (int)(done[cw] & 1u)) * D;
                        if (lane < D && c < P.N && P.do_append)
                            live_store(&P.Zw[(P.M_append + (int64_t)done[cw] * P.N + c) * P.ZS + lane], rows[lane]);
                        asm volatile("" ::: "memory");
                        ++done[cw];
                        if (lane == 0) __hip_atomic_store(&pub_done[cw], done[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        any = true;
                    } else if (__hip_atomic_load(&pub_exit[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
                        // (pub_seq is written before pub_exit: what is read now is final)
                        if (__hip_atomic_load(&pub_seq[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == done[cw]) ++gone;
                    }
                }
                if (gone == PS_CHAINS) break;
                if (any) { idle = 0u; continue; }
                // safety net: a launch that is being abandoned drains even if a chain wave could not say so
                if ((++idle & 4095u) == 0u && __hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                __builtin_amdgcn_s_sleep(1);
            }
            return;
        }
    }
    // a chain wave tells the publisher that nothing more is coming (every way out of a LIVE launch passes here), with
    // none of its DMAs still on their way into LDS
    auto leave = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (LIVE) {
            if (lane == 0) __hip_atomic_store(&pub_exit[w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    const int64_t c = (int64_t)blockIdx.x * PS_CHAINS + w;
    if (c >= P.N) {
        wave_store_counts(P, c, 0u, 0u);
        leave();
        return;
    }
    if constexpr (LIVE) {       // an earlier launch of the run already failed: do not wait again
        if (__hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { leave(); return; }
    }
    unsigned char* const raw_w = &raw[w][0][0];
    double* const sd_w = &sdelta[w][0];
    double* const ct_w = &ctab[w][0][0];
    const unsigned raw_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)raw_w);

    // ---- what this lane is, in each of its parts ----------------------------------------------------------------
    // node of the tree: lanes 1..31 (the others shadow node 1; nothing of theirs is ever selected)
    const int nn = (lane >= 1 && lane < 32) ? lane : 1;
    const int lev = 32 - __builtin_clz((unsigned)nn);      // 1..5: the generation of the pass this node proposes for
    // rows of sdelta this node adds, in order: an accepted generation on its path or its own -> that generation's
    // increments, anything else -> the row of negative zeros
    const double* mrow[PS_R];
#pragma unroll
    for (int j = 1; j <= PS_R; ++j) {
        const bool take = (j == lev) || (j < lev && ((nn >> (lev - 1 - j)) & 1));
        mrow[j - 1] = sd_w + (take ? j - 1 : PS_R) * DP;
    }
    // the node whose candidate this node's base state is: the last accepted generation on its path (0: the state
    // the pass starts from), as a ds_bpermute address
    int anc = nn;
    while (anc > 1 && (anc & 1) == 0) anc >>= 1;
    anc = (anc == 1) ? 0 : (anc >> 1);
    const int anc4 = anc * 4;
    // the accept bits this node's place on the path depends on: ancestors that must have accepted / rejected
    unsigned int need1 = 0u, need0 = 0u;
#pragma unroll
    for (int t = 1; t < PS_R; ++t) {
        if (t < lev) {
            const unsigned int a = (unsigned int)nn >> (lev - t);
            if ((nn >> (lev - 1 - t)) & 1) need1 |= 1u << a; else need0 |= 1u << a;
        }
    }
    const bool nodel = lane >= 1 && lane < 32;
    const int lgo = (FL0 + 3 * D) * 16 + (lev - 1) * 8;    // its log u, and its temperature, inside a raw slot
    [[maybe_unused]] const int tko = TL0 * 16 + (lev - 1) * 8;
    // increments: lane (u, p) forms element p of generation u of the pass
    const bool fl = lane < PS_R * D;
    const int fu = fl ? lane / D : 0, fp = fl ? lane % D : 0;
    const int zao = ((fu * 2) * HW) * 16 + fp * 8, zbo = ((fu * 2 + 1) * HW) * 16 + fp * 8;
    const int zto = (FL0 + 3 * fp) * 16 + fu * 8;
    const int ixo = (FL0 + 3 * (D + 1)) * 16;              // the row-index field of a raw slot
    const double eps_p = P.eps[fp];
    const double scale = P.gamma / sqrt((double)(2 * D));
    // DMA source: rows (ru, which, piece), record fields (f, piece), temperatures (piece)
    const bool rowl = lane < ROWL;
    const int ru = rowl ? lane / (2 * HW) : 0, rwhich = rowl ? (lane / HW) % 2 : 0, rj = rowl ? lane % HW : 0;
    const bool fieldl = lane >= FL0 && lane < TL0;
    const int ff = fieldl ? (lane - FL0) / 3 : 0, fj = fieldl ? (lane - FL0) % 3 : 0;
    const bool ixl = fieldl && ff == D + 1;
    const bool templ = TEMPER && lane >= TL0 && lane < TL0 + 3;
    const unsigned char* sbase;
    if (rowl) sbase = reinterpret_cast<const unsigned char*>(P.Z) + rj * 16;
    else if (fieldl) sbase = reinterpret_cast<const unsigned char*>(P.rec_in + ((int64_t)ff * P.N + c) * P.rec_stride) + fj * 16;
    else if (templ) sbase = reinterpret_cast<const unsigned char*>(P.temperature) + (lane - TL0) * 16;
    else sbase = reinterpret_cast<const unsigned char*>(P.rec_in);
    // history: lane (j, p) stores element p of generation j's row (p == D: log_obj)
    const bool hl = lane < PS_R * (D + 1);
    const int hj = hl ? lane / (D + 1) : 0, hp = hl ? lane % (D + 1) : 0;
    const unsigned int hmask = (hj + 1 >= 5) ? 0xffffffffu : ((1u << (1u << (hj + 1))) - 1u);     // nodes of generations 1..hj+1 of the pass
    const bool hist = P.chain != nullptr;

    // target constants (the arithmetic is window_kernel_pc8's)
    double muc[D], Wc[(TARGET == TARGET_MVNORMAL) ? D * (D + 1) / 2 : 1];
#pragma unroll
    for (int p = 0; p < D; ++p) muc[p] = P.tp.mu[p];
    if constexpr (TARGET == TARGET_MVNORMAL) {
#pragma unroll
        for (int i = 0; i < D * (D + 1) / 2; ++i) Wc[i] = P.tp.Wp[i];
    }
    const double c0c = P.tp.c0;

    // ---- passes of the launch: first generation, length (a pass ends at a K boundary, at the launch's end or after PS_R
    //      generations) and whether it ends on a boundary -- of the current pass and the four after it
    int sg[5], sR[5], sB[5];
    int cg = 0, ctb = P.to_boundary;           // where the pass after the last one in the queue starts
    auto seg_next = [&](int& g0, int& R, int& B) {
        int n = P.ngen - cg;
        n = (n < 0) ? 0 : n;
        n = (n < PS_R) ? n : PS_R;
        R = (ctb < n) ? ctb : n;
        g0 = cg;
        B = (R > 0 && ctb - R == 0) ? 1 : 0;
        cg += R;
        ctb = B ? P.K : ctb - R;
    };
#pragma unroll
    for (int k = 0; k < 5; ++k) seg_next(sg[k], sR[k], sB[k]);
    auto gclamp = [&](int g) { return (g < P.ngen) ? g : P.ngen - 1; };

    // state of the chain: row 0 of table 0
    double x[D], lp;
#pragma unroll
    for (int p = 0; p < D; ++p) x[p] = P.Xcur[c + P.N * p];
    lp = P.lpcur[c];
    if (lane == 0) {
#pragma unroll
        for (int p = 0; p < D; ++p) ct_w[p] = x[p];
        ct_w[D] = lp;
    }
    if (lane < DP) sd_w[PS_R * DP + lane] = -0.0;
    const double* cur = ct_w;         // the row the current state lives in
    int cur_tab = 0;

    // row indices: of this pass and the next (form lanes keep theirs for the LIVE re-reads), by ordinary loads once
    const double* rec_ix = P.rec_in + ((int64_t)(D + 1) * P.N + c) * P.rec_stride;
    [[maybe_unused]] uint64_t ixA = (uint64_t)__double_as_longlong(rec_ix[gclamp(sg[0] + fu)]);
    [[maybe_unused]] uint64_t ixB = (uint64_t)__double_as_longlong(rec_ix[gclamp(sg[1] + fu)]);
    // the DMA of pass k (k = 0..4 relative to the current one) into slot s; `pack`: the row indices its row lanes use
    auto issue = [&](int k, int slot, uint64_t pack) {
        const int Rk = sR[k];
        uint32_t idx = rwhich ? (uint32_t)(pack >> 32) : (uint32_t)pack;
        idx = (ru < Rk) ? idx : 0u;             // slots past the end of the pass read row 0 (LIVE: never wait for them)
        const int gk = gclamp(sg[k]);
        const int gi2 = (k + 2 < 5) ? gclamp(sg[(k + 2 < 5) ? k + 2 : 4]) : 0;
        const uint64_t dyn = rowl ? ((uint64_t)idx << ZSH) : ixl ? (uint64_t)gi2 * 8u : (fieldl || templ) ? (uint64_t)gk * 8u : 0u;
        ps_dma16(sbase + dyn, raw_lds + (unsigned)slot * 1024u);
    };
    {
        const uint64_t p0 = (uint64_t)__double_as_longlong(rec_ix[gclamp(sg[0] + ru)]);
        const uint64_t p1 = (uint64_t)__double_as_longlong(rec_ix[gclamp(sg[1] + ru)]);
        // (everything loaded so far is in registers before the first DMA: the compiler's own waits must never sit behind one)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        issue(0, 0, p0);
        issue(1, 1, p1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    // history: one buffer descriptor per array, re-based every pass (offsets stay small); no history = nothing in range
    const double* hx_base = hist ? P.chain + (int64_t)P.N * D * P.slot_first : P.Z;
    const double* hl_base = hist ? P.logobj + (int64_t)P.N * P.slot_first : P.Z;
    const uint32_t hx_off = (hl && hp < D) ? (uint32_t)((((int64_t)hj * D + hp) * P.N + c) * 8) : 0xffffffffu;
    const uint32_t hl_off = (hl && hp == D) ? (uint32_t)(((int64_t)hj * P.N + c) * 8) : 0xffffffffu;
    const uint32_t hx_span = (uint32_t)((int64_t)D * P.N * 8), hl_span = (uint32_t)(P.N * 8);      // one generation of each

    int slot = 0;
    int64_t nb = 0;
    unsigned int cnt_total = 0, cnt_first = 0;
#ifdef DEMCZ_STAMPS
    unsigned long long sa[6] = {0, 0, 0, 0, 0, 0}, sa_t = __builtin_readcyclecounter(), sa_n = 0, sa_bad = 0;
#define PS_TICK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); sa[i] += t_ - sa_t; sa_t = t_; } while (0)
#else
#define PS_TICK(i) do { } while (0)
#endif
    while (true) {
        const int g = sg[0], R = sR[0];
        if (R == 0) break;
        const unsigned char* rw = raw_w + slot * 1024;
        // ---- 1. this pass's slot: behind it in program order are the DMA of the next pass and two passes' history
        //         stores (two instructions each): everything older has landed when at most those five are outstanding
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        double za = *reinterpret_cast<const double*>(rw + zao);
        double zb = *reinterpret_cast<const double*>(rw + zbo);
        const double zt = *reinterpret_cast<const double*>(rw + zto);
        const uint64_t pr = *reinterpret_cast<const uint64_t*>(rw + ixo + ru * 8);      // row indices of the pass two after this one
        const uint64_t pf = *reinterpret_cast<const uint64_t*>(rw + ixo + fu * 8);
        if constexpr (LIVE) {
            // rows appended by other waves since the DMA read them show the sentinel until they are published: ask again
            bool bad = fl && fu < R && (is_sentinel(za) | is_sentinel(zb));
            if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {
#ifdef DEMCZ_STAMPS
                ++sa_bad;
#endif
                const uint32_t i1 = (uint32_t)ixA, i2 = (uint32_t)(ixA >> 32);
                int spins = 0;
                while (__builtin_amdgcn_ballot_w64(bad) != 0ull) {       // wave-uniform
                    if (spins > 0) {
                        if (live_poll_abandon(P, spins, bad, is_sentinel(za) ? i1 : i2, g)) { leave(); return; }
                        __builtin_amdgcn_s_sleep(1);
                    } else {
                        spins = 1;
                    }
                    if (bad) {
                        if (is_sentinel(za)) za = live_load(&P.Z[(int64_t)i1 * ZSC + fp]);
                        if (is_sentinel(zb)) zb = live_load(&P.Z[(int64_t)i2 * ZSC + fp]);
                        bad = is_sentinel(za) | is_sentinel(zb);
                    }
                }
            }
        }
        {
            const double diff = za - zb;
            const double t1 = scale * diff;
            const double t2 = eps_p * zt;
            if (fl) sd_w[fu * DP + fp] = t1 + t2;
        }
        PS_TICK(0);
        // ---- 2. the DMA of the pass two after this one (its row indices came with this slot)
        {
            const int s2 = (slot + 2 >= PS_SLOTS) ? slot + 2 - PS_SLOTS : slot + 2;
            issue(2, s2, pr);
            ixA = ixB;
            ixB = pf;
        }
        PS_TICK(1);
        wave_lds_handoff();
        // ---- 3. every node's candidate and its log-density
        double cand[D];
#pragma unroll
        for (int p = 0; p < D; ++p) cand[p] = x[p];
#pragma unroll
        for (int j = 0; j < PS_R; ++j) {
            double m[DP];
#pragma unroll
            for (int q = 0; q < DP / 2; ++q) {
                const double2 t = reinterpret_cast<const double2*>(mrow[j])[q];
                m[2 * q] = t.x;
                m[2 * q + 1] = t.y;
            }
#pragma unroll
            for (int p = 0; p < D; ++p) cand[p] = cand[p] + m[p];
        }
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
            lpp = fma(-0.5, q, c0c);
        } else {
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double rr = cand[i] - muc[i];
                q = (i == 0) ? rr * rr : fma(rr, rr, q);
            }
            lpp = -q;
        }
        // the candidates go to the table the current state is NOT in
        double* const tnew = ct_w + (cur_tab ^ 1) * (32 * CR);
        if (lane >= 1 && lane < 32) {
            double row[CR];
#pragma unroll
            for (int p = 0; p < CR; ++p) row[p] = (p < D) ? cand[p] : ((p == D) ? lpp : 0.0);
#pragma unroll
            for (int q = 0; q < CR / 2; ++q) reinterpret_cast<double2*>(tnew + lane * CR)[q] = make_double2(row[2 * q], row[2 * q + 1]);
        }
        PS_TICK(2);
        // ---- 4. all accept tests at once; the path taken, from the lane mask
        unsigned long long mask, chg_a, chg_r;
        {
            const unsigned long long lb = (unsigned long long)__double_as_longlong((lane == 0) ? lp : lpp);
            const unsigned int blo = (unsigned int)__builtin_amdgcn_ds_bpermute(anc4, (int)(unsigned int)lb);
            const unsigned int bhi = (unsigned int)__builtin_amdgcn_ds_bpermute(anc4, (int)(unsigned int)(lb >> 32));
            const double lpb = __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));     // log-density of the node's base state
            const double logu = *reinterpret_cast<const double*>(rw + lgo);
            const double d0 = lpp - lpb;
            double dlt = d0;
            if constexpr (TEMPER) dlt = dlt / *reinterpret_cast<const double*>(rw + tko);
            mask = __builtin_amdgcn_ballot_w64(logu < dlt);
            // "log_obj changed" (WindowParams::acc_out): after an accept lp' - lp, after a reject lp - lp (NaN for an infinite lp)
            chg_a = __builtin_amdgcn_fcmp(d0, 0.0, 14 /* UNE */);
            chg_r = __builtin_amdgcn_fcmp(lpb - lpb, 0.0, 14);
        }
        // A node is on the path when every ancestor decided the way that leads to it: one lane mask, one bit per
        // generation of the pass.  The state after generation j is the candidate of the last node on the path, up to
        // generation j, that accepted -- the highest such bit, nodes being numbered generation by generation.
        const unsigned int m32 = (unsigned int)mask;
        const bool onp = nodel && lev <= R && (m32 & need1) == need1 && (m32 & need0) == 0u;
        const unsigned int path = (unsigned int)__builtin_amdgcn_ballot_w64(onp);
        const unsigned int accp = path & m32;
        const unsigned int win = accp ? 31u - (unsigned int)__builtin_clz(accp) : 0u;        // 0: the pass's starting state
        {
            const unsigned int chm = (accp & (unsigned int)chg_a) | (path & ~m32 & (unsigned int)chg_r);
            cnt_total += (unsigned int)__builtin_popcount(chm);
            if (g == 0) cnt_first = (chm >> 1) & 1u;
        }
        PS_TICK(3);
        wave_lds_handoff();
        // ---- 5. history rows of the pass, the new state
        {
            const unsigned int wa = accp & hmask;
            const unsigned int wj = wa ? 31u - (unsigned int)__builtin_clz(wa) : 0u;
            const double* src = (wj == 0u) ? cur : tnew + wj * CR;
            const double v = src[hp];
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
            const u32x2 vv = {(unsigned int)vb, (unsigned int)(vb >> 32)};
            const uint32_t lim_x = hist ? (uint32_t)R * hx_span : 0u, lim_l = hist ? (uint32_t)R * hl_span : 0u;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(hx_base) + (int64_t)g * D * P.N, 0, (int)lim_x, 0x00020000);
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(hl_base) + (int64_t)g * P.N, 0, (int)lim_l, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(vv, rx, (int)hx_off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(vv, rl, (int)hl_off, 0, 0);
        }
        if (win != 0u) { cur = tnew + win * CR; cur_tab ^= 1; }
#pragma unroll
        for (int q = 0; q < CR / 2; ++q) {
            const double2 t = reinterpret_cast<const double2*>(cur)[q];
            if (2 * q < D) x[2 * q] = t.x;
            if (2 * q == D) lp = t.x;
            if (2 * q + 1 < D) x[2 * q + 1] = t.y;
            if (2 * q + 1 == D) lp = t.y;
        }
        PS_TICK(4);
        // ---- 6. a generation divisible by K ended the pass: runchain!'s append, demcz.jl:88-91
        if (sB[0]) {
            const double v = cur[(lane < D) ? lane : 0];
            if constexpr (LIVE) {
                while (__hip_atomic_load(&pub_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 2u <= (unsigned int)nb)
                    __builtin_amdgcn_s_sleep(1);
                if (lane < D) pub_rows[(w * 2 + (int)((unsigned int)nb & 1u)) * D + lane] = v;
                asm volatile("" ::: "memory");                 // (one wave's LDS operations execute in order)
                if (lane == 0) __hip_atomic_store(&pub_seq[w], (unsigned int)nb + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                if (lane < D && P.do_append) P.Zw[(P.M_append + nb * P.N + c) * P.ZS + lane] = v;
            }
            if (lane < D && P.snap) P.snap[nb * P.N * D + c + P.N * lane] = v;
            ++nb;
        }
        wave_lds_handoff();      // sdelta and the other candidate table are rewritten by the next pass
        // next pass
#pragma unroll
        for (int k = 0; k < 4; ++k) { sg[k] = sg[k + 1]; sR[k] = sR[k + 1]; sB[k] = sB[k + 1]; }
        seg_next(sg[4], sR[4], sB[4]);
        slot = (slot + 1 == PS_SLOTS) ? 0 : slot + 1;
        PS_TICK(5);
#ifdef DEMCZ_STAMPS
        ++sa_n;
#endif
    }
    {
        const double v = cur[(lane < D) ? lane : 0];
        if (lane < D) P.Xcur[c + P.N * lane] = v;
        if (lane == 0) P.lpcur[c] = lp;
    }
    wave_store_counts(P, c, cnt_total, cnt_first);
#ifdef DEMCZ_STAMPS
    if (P.stamps && lane == 0 && c < 65536) {      // [wait + increments, DMA issue, candidates + log-density, accept + path, history + state, boundary + bookkeeping]
        unsigned long long* o = P.stamps + (size_t)c * 16;
        for (int i = 0; i < 6; ++i) o[8 + i] = sa[i];
        o[14] = sa_n;
        o[15] = sa_bad;
    }
#endif
    leave();
}

}  // namespace demcz
