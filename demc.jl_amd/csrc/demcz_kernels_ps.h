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
// The loop is a software pipeline over passes: the state-independent front end of pass r+1 (slot wait, increments into
// LDS, the DMA of pass r+3, each node's rows back into registers) sits inside pass r between the candidate adds and the
// log-density; the new state is read straight out of the winning lane's registers (v_readlane; lane 0 shadows the
// starting state); the pass's history values are read from the LDS candidate table at its end and stored during the
// next pass.  Measured steps and what was tried and dropped: DESIGN.md section 4, K1g.
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
constexpr int PS_PUB = 4;        // boundaries' rows a chain wave may have waiting for the publisher
#ifndef PS_AHEAD_N
#define PS_AHEAD_N 2
#endif
constexpr int PS_AHEAD = PS_AHEAD_N;        // a pass's DMA is issued this many passes before its slot is consumed
constexpr int PS_SLOTS = PS_AHEAD + 1;      // ring of raw slots: the one being consumed + those in flight
constexpr int PS_MAX_N = 2048;   // beyond ~2 waves per SIMD the replicated consumer (8 chains per wave) is the faster one; the library's
                                 // own choice also asks that a LIVE launch of this layout fits the chip (demcz_create): 1024 on MI355X

// one 16-byte piece per lane from a per-lane address into LDS at (wave-uniform) lds_dst + 16 * lane
__device__ __forceinline__ void ps_dma16(const void* gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// The producer half on its own (launched beside window_kernel_ps on a side stream, demcz_capi.hip): a kernel with
// the producer's small register budget fills the SIMDs around the one-wave-per-SIMD consumers; as workgroups of the
// consumer's kernel it inherits that kernel's registers and workgroup shape, and at five waves per workgroup
// (LIVE) only one producer workgroup fits a CU beside a consumer workgroup -- the launch then waits for its producers.
constexpr int PRODUCE_WAVES = 4;
constexpr size_t PRODUCE_THROTTLE_LDS = 64 * 1024;     // of a CU's 160 KB, ~27 KB of them a consumer workgroup's: two producer workgroups fit
                                                       // (32 / 48 KB measure the same since the producer's stores are coalesced: profiles/r03e_producer.txt)
template <int D>
__global__ void __launch_bounds__(64 * PRODUCE_WAVES) produce_kernel(const WindowParams P)
{
    pc_produce<D>(P, (int64_t)blockIdx.x * PRODUCE_WAVES + (int64_t)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
}

template <int TARGET, int D, bool LIVE, bool TEMPER>
__global__ void __launch_bounds__(64 * (PS_CHAINS + (LIVE ? 1 : 0)), 3) window_kernel_ps(const WindowParams P)
{
    static_assert(TARGET == TARGET_MVNORMAL || TARGET == TARGET_ISO_QUAD, "split layout: MvNormal / isotropic targets");
    static_assert(D >= 2 && D <= 5, "a pass's rows, normals, log u and indices are one 64-lane DMA");
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
    if constexpr (!LIVE) {
        // launches that end at the next boundary (sharded runs, deferred visibility) are short: their producer half rides
        // in the same launch, as in the replicated consumer -- every wave of a producer workgroup is one 64-lane unit.
        // (LIVE launches: produce_kernel beside them.)
        if ((int64_t)blockIdx.x >= P.consumer_blocks) {
            pc_produce<D>(P, ((int64_t)blockIdx.x - P.consumer_blocks) * PS_CHAINS + w, lane);
            return;
        }
    }
    __shared__ __attribute__((aligned(16))) unsigned char raw[PS_CHAINS][PS_SLOTS][1024];
    __shared__ __attribute__((aligned(16))) double sdelta[PS_CHAINS][(PS_R + 1) * DP];       // row PS_R: negative zeros
    __shared__ __attribute__((aligned(16))) double ctab[PS_CHAINS][2][32 * CR];
    // LIVE: a boundary's row on its way from a chain wave to the publisher (PS_PUB boundaries' worth per chain wave) and the
    // hand-shake: pub_seq = boundaries the chain wave has left here, pub_done = written out by the publisher, pub_exit = leaving
    __shared__ double pub_rows[LIVE ? PS_CHAINS * PS_PUB * D : 1];
    __shared__ unsigned int pub_seq[PS_CHAINS], pub_done[PS_CHAINS], pub_exit[PS_CHAINS];
    if constexpr (LIVE) {
        if (threadIdx.x < PS_CHAINS) { pub_seq[threadIdx.x] = 0u; pub_done[threadIdx.x] = 0u; pub_exit[threadIdx.x] = 0u; }
        __syncthreads();
        if (w == PS_CHAINS) {
            // The publisher (why a wave of its own: demcz_kernels_pc.h, PC8_LIVE_WAVES).  Lane (cw, p) looks after element p
            // of chain wave cw's rows, so the rows of all chain waves that are ready go out in ONE store instruction --
            // the publisher's own next store waits for the round trip of the one before (the same effect it exists to
            // keep away from the chain waves): a store per chain wave could not keep up with a boundary every two passes.
            const bool pl = lane < PS_CHAINS * D;
            const int cw = pl ? lane / D : 0, pp = pl ? lane % D : 0;
            const int64_t cl = (int64_t)xcd_block(P) * PS_CHAINS + cw;
            unsigned int done = 0u;
            while (true) {
                // The chain waves reach their boundaries at different moments.  Looking for ready rows only once the
                // store before this one is complete lets the rows that arrived during its flight leave TOGETHER (a store
                // issued at once would sit behind the earlier one just as long, with one row in it).
                publisher_wait(P);
                const unsigned int seq = __hip_atomic_load(&pub_seq[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const bool ready = pl && seq != done;
                if (__builtin_amdgcn_ballot_w64(ready) != 0ull) {
                    if (ready) {
                        const double v = pub_rows[(cw * PS_PUB + (int)(done % PS_PUB)) * D + pp];
                        if (cl < P.N && P.do_append) live_publish(P, (int64_t)done, cl, pp, v);
                        ++done;
                    }
                    asm volatile("" ::: "memory");
                    if (ready && pp == 0) __hip_atomic_store(&pub_done[cw], done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    continue;
                }
                // (pub_seq is written before pub_exit: what is read after pub_exit shows set is final)
                const bool gone = !pl || (__hip_atomic_load(&pub_exit[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u &&
                                          __hip_atomic_load(&pub_seq[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == done);
                if (__builtin_amdgcn_ballot_w64(!gone) == 0ull) break;
                // (The publisher never leaves before its chain waves, not even in a launch that is being abandoned: a chain
                //  wave that has not seen the error word yet may still be filling the ring, and it waits for room in it
                //  without a poll limit -- that wait is bounded by THIS loop making progress.  Every way out of a chain wave
                //  passes through leave(), so `gone` always comes.)
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
    const int64_t c = (int64_t)xcd_block(P) * PS_CHAINS + w;       // (XCD-aware: demcz_kernels.h)
    if (c >= P.N) {
        wave_store_counts(P, c, 0u, 0u);
        leave();
        return;
    }
    if constexpr (LIVE) {       // an earlier launch of the run already failed: do not wait again
        if (__hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { leave(); return; }
    }
    // The chain waves are the launch's critical path, one per SIMD, and mostly waiting on their own dependent
    // instructions; the producer kernel's waves around them are throughput work.  Whenever a chain wave can issue, it should.
    __builtin_amdgcn_s_setprio(3);
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
        const bool take = lane != 0 && ((j == lev) || (j < lev && ((nn >> (lev - 1 - j)) & 1)));     // (lane 0: the state itself)
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
    double c0v = P.tp.c0;          // (kept in a vector register: see the W entries below)

    // ---- passes of the launch.  A pass ends at a K boundary, at the launch's end or after PS_R generations.  The current
    //      pass and the five after it are a queue of nibbles (bits 0-2 length, bit 3 "ends on a boundary", entry k at bits
    //      4k..4k+3); g0 / g3 / g5 are the first generations of entries 0, 3 and 5.
    unsigned int segq = 0u;
    int cg = 0, ctb = P.to_boundary;           // where the pass after the last one in the queue starts
    // (opaque copies: as plain kernel arguments the compiler re-loads them from the argument segment inside the loop when
    //  scalar registers are short, and the wait for that load is a wait for every LDS read in flight as well)
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
    constexpr int QN = 2 + 2 * PS_AHEAD;           // entries 0 .. 1 + 2 PS_AHEAD
    static_assert(QN <= 8, "the queue is one 32-bit word");
#pragma unroll
    for (int k = 0; k < QN; ++k) segq |= seg_make() << (4 * k);
    auto qR = [&](int k) __attribute__((always_inline)) -> int { return (int)((segq >> (4 * k)) & 7u); };
    auto gclamp = [&](int g) __attribute__((always_inline)) { return (g < ngen_s) ? g : ngen_s - 1; };
    auto gsum = [&](int k) __attribute__((always_inline)) { int g = 0; for (int i = 0; i < k; ++i) g += qR(i); return g; };   // first generation of entry k
    int g0 = 0, g3 = gsum(1 + PS_AHEAD), g5 = gsum(1 + 2 * PS_AHEAD);       // (named for PS_AHEAD = 2: entries 3 and 5)
    int npass;
    {
        const int n1 = (P.to_boundary < P.ngen) ? P.to_boundary : P.ngen, rest = P.ngen - n1;
        npass = (n1 + PS_R - 1) / PS_R + (rest / P.K) * ((P.K + PS_R - 1) / PS_R) + (rest % P.K + PS_R - 1) / PS_R;
    }

    // state of the chain: wave-uniform values (x, lp) for the arithmetic, and row 0 of table 0 for whoever needs one element
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
    // (form lanes keep the row indices of the passes in flight for the LIVE re-reads: ixq[0] = those of the pass whose
    //  slot is consumed next)
    [[maybe_unused]] uint64_t ixq[PS_AHEAD];
#pragma unroll
    for (int k = 0; k < PS_AHEAD; ++k) ixq[k] = (uint64_t)__double_as_longlong(rec_ix[gclamp(gsum(k) + fu)]);
    // The DMA of a pass (length Rk, first generation gk) into a slot; `pack`: the row indices its row lanes use; gix: the
    // first generation of the pass two after it, whose row indices come with it.
    auto issue = [&](int Rk, int gk, int gix, int slot, uint64_t pack) __attribute__((always_inline)) {
        uint32_t idx = rwhich ? (uint32_t)(pack >> 32) : (uint32_t)pack;
        idx = (ru < Rk) ? idx : 0u;             // slots past the end of the pass read row 0 (LIVE: never wait for them)
        const uint32_t gsel = (uint32_t)(ixl ? gclamp(gix) : gclamp(gk));
        const uint64_t dyn = rowl ? ((uint64_t)idx << ZSH) : (uint64_t)(gsel << 3);     // (idle lanes: somewhere inside the records)
        ps_dma16(sbase + dyn, raw_lds + (unsigned)slot * 1024u);
    };
    {
        uint64_t pp[PS_AHEAD];
#pragma unroll
        for (int k = 0; k < PS_AHEAD; ++k) pp[k] = (uint64_t)__double_as_longlong(rec_ix[gclamp(gsum(k) + ru)]);
        // Everything loaded so far is in registers, and the compiler knows it (each value is an operand of an empty
        // statement), before the first DMA: a wait of the compiler's own for one of these, placed inside the loop,
        // would wait for the DMAs in flight as well -- every pass.  (The W entries stay in vector registers: as scalars
        // they and the state crowd the scalar file into spills.)
#pragma unroll
        for (int p = 0; p < D; ++p) asm volatile("" :: "v"(x[p]), "v"(muc[p]));
        if constexpr (TARGET == TARGET_MVNORMAL) {
#pragma unroll
            for (int i = 0; i < D * (D + 1) / 2; ++i) asm volatile("" : "+v"(Wc[i]));
        }
        asm volatile("" : "+v"(c0v));
        asm volatile("" :: "v"(lp), "v"(eps_p));
#pragma unroll
        for (int k = 0; k < PS_AHEAD; ++k) asm volatile("" :: "v"(ixq[k]), "v"(pp[k]));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < PS_AHEAD; ++k) issue(qR(k), gsum(k), gsum(k + PS_AHEAD), k, pp[k]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    auto uniform = [&](double v) __attribute__((always_inline)) {          // a wave-uniform double as a scalar value
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)b);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    auto from_lane = [&](double v, unsigned int l) __attribute__((always_inline)) {       // lane l's v (l wave-uniform)
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)b, (int)l);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(b >> 32), (int)l);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
#pragma unroll
    for (int p = 0; p < D; ++p) x[p] = uniform(x[p]);
    lp = uniform(lp);

    // history: one buffer descriptor per array, moved on every pass (offsets stay small); no history = nothing in range
    const unsigned char* hx_ptr = reinterpret_cast<const unsigned char*>(hist ? P.chain + (int64_t)P.N * D * P.slot_first : P.Z);
    const unsigned char* hl_ptr = reinterpret_cast<const unsigned char*>(hist ? P.logobj + (int64_t)P.N * P.slot_first : P.Z);
    const uint32_t hx_off = (hl && hp < D) ? (uint32_t)((((int64_t)hj * D + hp) * P.N + c) * 8) : 0x7fffff00u;
    const uint32_t hl_off = (hl && hp == D) ? (uint32_t)(((int64_t)hj * P.N + c) * 8) : 0x7fffff00u;
    const uint32_t hx_span = hist ? (uint32_t)((int64_t)D * P.N * 8) : 0u, hl_span = hist ? (uint32_t)(P.N * 8) : 0u;      // one generation of each
    // the history rows of a pass leave one pass later (their LDS reads then have a whole pass to come back): value and
    // length of the pass they belong to
    double hv = 0.0;
    uint32_t hR = 0;
    auto store_history = [&]() __attribute__((always_inline)) {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const unsigned long long vb = (unsigned long long)__double_as_longlong(hv);
        const u32x2 vv = {(unsigned int)vb, (unsigned int)(vb >> 32)};
        const uint32_t lim_x = hR * hx_span, lim_l = hR * hl_span;      // generations beyond the pass: out of range
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(hx_ptr), 0, (int)lim_x, 0x00020000);
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(hl_ptr), 0, (int)lim_l, 0x00020000);
        // (tried: non-temporal stores -- 217 -> 238 us per 1000-generation launch)
        __builtin_amdgcn_raw_buffer_store_b64(vv, rx, (int)hx_off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(vv, rl, (int)hl_off, 0, 0);
        hx_ptr += lim_x;
        hl_ptr += lim_l;
    };

    int64_t nb = 0;
    unsigned int cnt_total = 0, cnt_first = 0;
#ifdef DEMCZ_STAMPS
    // diagnostic build (scripts/ps_stamps.py): shader-clock time of the launch, of the two kinds of LIVE waits, their counts.
    // Only the waits are bracketed: a stamp drains the wave's outstanding LDS and scalar-memory operations.
    const unsigned long long sa_start = __builtin_readcyclecounter();
    unsigned long long sa_reread = 0, sa_ring = 0, sa_n = 0, sa_bad = 0, sa_rings = 0;
#endif

    // ---- front end of a pass (raw slot `slot`): everything about it that does not depend on the chain's state --
    //      increments into LDS, the DMA of the pass two after it, then each node's rows of increments, its log u and
    //      temperature into registers.  It runs INSIDE the pass before (software pipeline: its LDS and issue time hides
    //      behind that pass's dependent arithmetic); rows that still show the sentinel are only flagged here.
    double m[PS_R][D];                               // the node's rows (of the pass at hand, then of the next one)
    double logu = 0.0;
    [[maybe_unused]] double temp = 1.0;
    double za_f = 0.0, zb_f = 0.0, zt_f = 0.0;       // what the form lane read (kept for the LIVE re-reads)
    uint64_t pf_f = 0;
    auto load_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PS_R; ++j)
#pragma unroll
            for (int q = 0; q < DP / 2; ++q) {
                const double2 t = reinterpret_cast<const double2*>(mrow[j])[q];
                m[j][2 * q] = t.x;
                if (2 * q + 1 < D) m[j][2 * q + 1] = t.y;
            }
    };
    auto write_increment = [&]() __attribute__((always_inline)) {
        const double diff = za_f - zb_f;
        const double t1 = scale * diff;
        const double t2 = eps_p * zt_f;
        if (fl) sd_w[fu * DP + fp] = t1 + t2;
    };
    // Rn: length of the front end's pass; (R2, g2, gix): the pass two after it (the DMA it issues)
    auto front = [&](int slot, int Rn, int R2, int g2, int gix, bool counted) __attribute__((always_inline)) -> bool {
        const unsigned char* rw = raw_w + slot * 1024;
        // behind this slot's DMA in program order: the DMA of the pass after it and two passes' history stores (two
        // instructions each) -- it has landed when at most those five are outstanding
        // (PS_AHEAD - 1 whole passes -- two history stores and a DMA each -- and this pass's two stores are behind it)
        if (counted) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * PS_AHEAD - 1) : "memory");
        za_f = *reinterpret_cast<const double*>(rw + zao);
        zb_f = *reinterpret_cast<const double*>(rw + zbo);
        zt_f = *reinterpret_cast<const double*>(rw + zto);
        const uint64_t pr = *reinterpret_cast<const uint64_t*>(rw + ixo + ru * 8);      // row indices of the pass two after this one
        pf_f = *reinterpret_cast<const uint64_t*>(rw + ixo + fu * 8);
        logu = *reinterpret_cast<const double*>(rw + lgo);
        if constexpr (TEMPER) temp = *reinterpret_cast<const double*>(rw + tko);
        bool bad = false;
        if constexpr (LIVE) bad = fl && fu < Rn && (is_sentinel(za_f) | is_sentinel(zb_f));
        write_increment();
        const int s2 = (slot + PS_AHEAD >= PS_SLOTS) ? slot + PS_AHEAD - PS_SLOTS : slot + PS_AHEAD;
        issue(R2, g2, gix, s2, pr);
        wave_lds_handoff();
        load_rows();
        return bad;
    };
    // LIVE: rows of the front end's pass that other waves had not published when the DMA read them -- asked for again
    // (sc1 loads) until they are there, increments and node rows redone.  Returns true when the wait was abandoned.
    auto reread = [&](bool bad, int gpass) __attribute__((always_inline)) -> bool {
#ifdef DEMCZ_STAMPS
        ++sa_bad;
        const unsigned long long sa_t0 = __builtin_readcyclecounter();
#endif
        const uint32_t i1 = (uint32_t)ixq[0], i2 = (uint32_t)(ixq[0] >> 32);
        int spins = 0;
        while (__builtin_amdgcn_ballot_w64(bad) != 0ull) {       // wave-uniform
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
        sa_reread += __builtin_readcyclecounter() - sa_t0;
#endif
        wave_lds_handoff();
        write_increment();
        wave_lds_handoff();
        load_rows();
        return false;
    };

    {
        // the first pass's front end, on its own; at the start of a launch every row it may draw is published
        // (slot 0; the DMA it issues is the third pass's)
        const bool bad0 = front(0, qR(0), qR(PS_AHEAD), gsum(PS_AHEAD), gsum(2 * PS_AHEAD), false);
        if constexpr (LIVE) {
            if (__builtin_amdgcn_ballot_w64(bad0) != 0ull) {
                if (reread(bad0, 0)) { leave(); return; }
            }
        }
#pragma unroll
        for (int k = 0; k + 1 < PS_AHEAD; ++k) ixq[k] = ixq[k + 1];
        ixq[PS_AHEAD - 1] = pf_f;
    }
    int slot = 1;                     // the raw slot the next front end consumes
    for (int ip = 0; ip < npass; ++ip) {
        const int R = qR(0);
        // a pass that ends on a boundary asks early how far the publisher is (the answer only grows)
        [[maybe_unused]] unsigned int pub_seen = 0u;
        if constexpr (LIVE) {
            if (segq & 8u) pub_seen = __hip_atomic_load(&pub_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        const double logu_c = logu;
        [[maybe_unused]] const double temp_c = temp;
        // every node's candidate: state + its rows in order
        double cand[D];
#pragma unroll
        for (int p = 0; p < D; ++p) cand[p] = x[p];
#pragma unroll
        for (int j = 0; j < PS_R; ++j)
#pragma unroll
            for (int p = 0; p < D; ++p) cand[p] = cand[p] + m[j][p];
        // the pass before's history leaves; the next pass's front end
        store_history();
        const bool bad_n = front(slot, qR(1), qR(1 + PS_AHEAD), g3, g5, true);
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
        // the candidates go to the table the current state is NOT in (for whoever needs single elements: history, append)
        double* const tnew = ct_w + (cur_tab ^ 1) * (32 * CR);
        if (nodel) {
            double row[CR];
#pragma unroll
            for (int p = 0; p < CR; ++p) row[p] = (p < D) ? cand[p] : ((p == D) ? lpp : 0.0);
#pragma unroll
            for (int q = 0; q < CR / 2; ++q) reinterpret_cast<double2*>(tnew + lane * CR)[q] = make_double2(row[2 * q], row[2 * q + 1]);
        }
        // all accept tests at once
        unsigned long long mask, chg_a, chg_r;
        {
            const unsigned long long lb = (unsigned long long)__double_as_longlong((lane == 0) ? lp : lpp);
            const unsigned int blo = (unsigned int)__builtin_amdgcn_ds_bpermute(anc4, (int)(unsigned int)lb);
            const unsigned int bhi = (unsigned int)__builtin_amdgcn_ds_bpermute(anc4, (int)(unsigned int)(lb >> 32));
            const double lpb = __longlong_as_double((long long)(((unsigned long long)bhi << 32) | blo));     // log-density of the node's base state
            const double d0 = lpp - lpb;
            double dlt = d0;
            if constexpr (TEMPER) dlt = dlt / temp_c;
            mask = __builtin_amdgcn_ballot_w64(logu_c < dlt);
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
            if (g0 == 0) cnt_first = (chm >> 1) & 1u;
        }
        // the new state, straight from the winner's registers (lane 0 shadows the starting state: all its rows are
        // negative zeros, its candidate IS x)
#pragma unroll
        for (int p = 0; p < D; ++p) x[p] = from_lane(cand[p], win);
        {
            const double lw = from_lane(lpp, win);
            lp = win ? lw : lp;
        }
        wave_lds_handoff();
        // history rows of the pass: read now, stored during the next pass
        {
            const unsigned int wa = accp & hmask;
            const unsigned int wj = wa ? 31u - (unsigned int)__builtin_clz(wa) : 0u;
            const double* src = (wj == 0u) ? cur : tnew + wj * CR;
            hv = src[hp];
            hR = (uint32_t)R;
        }
        if (win != 0u) { cur = tnew + win * CR; cur_tab ^= 1; }
        // a generation divisible by K ended the pass: runchain!'s append, demcz.jl:88-91
        if (segq & 8u) {
            double v = x[0];               // lane p: element p of the new state (x is wave-uniform)
#pragma unroll
            for (int p = 1; p < D; ++p) v = (lane == p) ? x[p] : v;
            if constexpr (LIVE) {
                // the slot of pub_rows this boundary uses was emptied PS_PUB boundaries ago -- almost always
                asm volatile("" : "+v"(pub_seen));          // (not looked at before this point: the read has had the whole pass)
#ifdef DEMCZ_STAMPS
                const unsigned long long sa_t1 = __builtin_readcyclecounter();
                if (pub_seen + (unsigned int)PS_PUB <= (unsigned int)nb) ++sa_rings;
#endif
                while (pub_seen + (unsigned int)PS_PUB <= (unsigned int)nb) {
                    __builtin_amdgcn_s_sleep(1);
                    pub_seen = __hip_atomic_load(&pub_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#ifdef DEMCZ_STAMPS
                sa_ring += __builtin_readcyclecounter() - sa_t1;
#endif
                if (lane < D) pub_rows[(w * PS_PUB + (int)((unsigned int)nb % PS_PUB)) * D + lane] = v;
                asm volatile("" ::: "memory");                 // (one wave's LDS operations execute in order)
                if (lane == 0) __hip_atomic_store(&pub_seq[w], (unsigned int)nb + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                if (lane < D && P.do_append) P.Zw[(P.M_append + nb * P.N + c) * P.ZS + lane] = v;
            }
            if (lane < D && P.snap) P.snap[nb * P.N * D + c + P.N * lane] = v;
            ++nb;
        }
        if constexpr (LIVE) {
            // only now -- this wave's own row is on its way -- may it wait for rows of other waves
            if (__builtin_amdgcn_ballot_w64(bad_n) != 0ull) {
                if (reread(bad_n, g0 + R)) { leave(); return; }
            }
        }
        wave_lds_handoff();      // sdelta and the other candidate table are rewritten by the next pass
        // the queue moves on
        g0 += R;
        g3 += qR(1 + PS_AHEAD);
        g5 += qR(1 + 2 * PS_AHEAD);
        segq = (segq >> 4) | (seg_make() << (4 * (QN - 1)));
        slot = (slot + 1 == PS_SLOTS) ? 0 : slot + 1;
#pragma unroll
        for (int k = 0; k + 1 < PS_AHEAD; ++k) ixq[k] = ixq[k + 1];
        ixq[PS_AHEAD - 1] = pf_f;
#ifdef DEMCZ_STAMPS
        ++sa_n;
#endif
    }
    store_history();          // the last pass's
    {
        const double v = cur[(lane < D) ? lane : 0];
        if (lane < D) P.Xcur[c + P.N * lane] = v;
        if (lane == 0) P.lpcur[c] = lp;
    }
    wave_store_counts(P, c, cnt_total, cnt_first);
#ifdef DEMCZ_STAMPS
    if (P.stamps && lane == 0 && c < 65536) {
        unsigned long long* o = P.stamps + (size_t)c * 16;
        o[8] = __builtin_readcyclecounter() - sa_start; o[9] = sa_reread; o[10] = sa_ring; o[11] = sa_bad; o[12] = sa_rings; o[14] = sa_n;
    }
#endif
    leave();
}

}  // namespace demcz
