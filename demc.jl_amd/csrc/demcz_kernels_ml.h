// demcz_kernels_ml.h -- K1 in the "latency layout": L lanes cooperate on one chain.
//
// The BASELINE metric is quoted at N = 1024 chains: with one lane per chain that is 16 wavefronts
// on a chip with 1024 SIMDs, and a chain-update is a ~1100-instruction serial stream per wave.
// Here a chain is spread over L = 8 (or 16) lanes, so the same N fills 8x (16x) as many SIMDs and
// the per-wave stream shrinks to ~300 instructions:
//   * lane r of a group generates Philox block r of the generation (role 0: the two archive row
//     indices, roles 1..NPAIRS: one Box-Muller pair each, role S-1: the accept uniform), so the 5
//     blocks of a d=5 update are produced side by side instead of one after another;
//   * the draws are staged through LDS (one 16-byte entry per role) and read back by the lanes
//     that need them (broadcast reads inside the group);
//   * lane p owns parameter p: it gathers Z[i1][p], Z[i2][p] (adjacent lanes hit adjacent words of
//     the same 64-byte row), forms its component of the proposal, and row p of the whitened
//     residual y = W (x' - mu); the components travel through LDS and every lane of the group
//     accumulates q = sum y_i^2 in the spec's order, so all lanes reach the same accept decision.
// Arithmetic is the same operation sequence as the one-lane kernel and the oracle: results are
// bit-identical (tests/test_gpu_parity.py runs both layouts against the oracle).
// Restates the same reference functions as window_kernel (src/demcz.jl:80-93,167-203).
#pragma once

#include "demcz_kernels_rec.h"

#include <type_traits>

#pragma clang fp contract(off)

#ifndef ML_LRDPP
#define ML_LRDPP 1        // regression target on sixteen lanes: the chains' proposals by lanes (DPP row_newbcast); 0: wave-uniform copies in registers
#endif
namespace demcz {

// Workgroup geometry: one wave per workgroup (small N spreads over as many CUs as there are waves).  (The regression
// target has its own kernel on the FP64 matrix instruction, demcz_kernels_lr.h.)
// REC: the split form of this layout (demcz_kernels_rec.h) -- workgroups beyond consumer_blocks are the
// producer half (draw records for the next launch), the others read this launch's records instead of
// drawing: what is left per generation is the state-dependent part.  LIVE: the launch runs through K
// boundaries and takes appended rows from other waves through the archive itself (sentinel + sc1).
// Used where the eight-replicated-lanes consumer does not fit (d = 20: 210 whitening coefficients).
// waves per workgroup of window_kernel_ml: the waves share nothing; a workgroup's waves are placed one per SIMD, one-wave
// workgroups now and then two to a SIMD (demcz_kernels_ml.h, MLB_REC_WAVES).  The host launches ML_WAVES-wave workgroups
// where the chain waves are at least one per SIMD of the chip, one-wave workgroups below that (they spread over more CUs:
// d = 20 split form, N = 2048 / 4096: 1.66 / 2.33 x 10^9 updates/s with one wave, 1.58 / 2.62 with four).
#ifndef DEMCZ_ML_WAVES
#define DEMCZ_ML_WAVES 4      // (1: 2.68 x 10^9 updates/s at C4's whole population on one GPU, 4: 2.91 x 10^9; scripts/ab_cfg.sh)
#endif
constexpr int ML_WAVES = DEMCZ_ML_WAVES;

// COOP (regression target, sixteen lanes per chain): a workgroup is ONE chain wave -- four chains, everything as without COOP -- and
// ML_COOP_WAVES - 1 helper waves that take the residuals of its log-density off it: helper w forms the residuals of the observations
// of rounds w - 1, w - 1 + (ML_COOP_WAVES - 1), ... for all four chains and leaves them in LDS; the chain wave then folds them into the spec's sixteen partial sums
// per chain -- lane r of a chain IS partial r and takes its terms in increasing o, as target_logp does.  ML_COOP_WAVES times the
// waves on a chip that a population of 1024 chains otherwise fills to a quarter of its SIMDs with one FP64-issuing wave each.
#ifndef ML_COOP_WAVES_N
#define ML_COOP_WAVES_N 8
#endif
constexpr int ML_COOP_WAVES = ML_COOP_WAVES_N;
constexpr int ML_COOP_MAX_OBS = 1536;       // residuals of a workgroup's four chains in LDS: 4 x 1536 doubles = 48 KB
#ifndef ML_COOP_TILE
#define ML_COOP_TILE 1                      // whole rounds of the design resident in LDS tiles (0: every lane reads its row from memory, every generation)
#endif
#ifndef ML_COOP_RES_BYTES
#define ML_COOP_RES_BYTES 102400            // LDS for the resident tiles: 100 KB beside the 48 KB of residuals
#endif
constexpr int ml_coop_resident_tiles(int D)
{
    const int fit = ML_COOP_RES_BYTES / (512 * D), all = ML_COOP_MAX_OBS / 64;
    return ML_COOP_TILE ? (fit < all ? fit : all) : 0;
}
#if ML_LRDPP
// A round's 64 design rows are 64 * D contiguous doubles (row-major design).  A lane reading ITS row straight from memory makes
// every load instruction touch 64 different cache lines: with seven helper waves a CU that address path sets the pace (a round
// per 650 clocks for the whole workgroup, whatever the number of waves).  Whole rounds therefore come in as the contiguous tile
// they are (a lane's 16-byte pieces lane, lane + 64, ...: eight lines an instruction), go through a tile of the wave's own in LDS,
// and each lane reads its row back from there (row stride D * 8 bytes: conflict-free for even D).
// What a generation then waits for is the CU's share of the L2's bandwidth: every workgroup reads the whole design (208 KB at d = 26,
// nobs = 1000) every generation -- 3.6 us per generation = 54 GB/s per CU, where the guide's L2-served gather reaches 66-73
// (profiles/r05zb_linreg_pmc.txt: 574 MB of L2 traffic a launch at a 98.5 % hit rate, 61 % of wave-cycles in s_waitcnt).
// So the tiles STAY: the first ML_COOP_RES_BYTES / (512 D) rounds' tiles are resident in LDS for the whole launch -- filled in its
// first generation by the helper that owns the round, read from LDS in every later one (7 of 16 rounds at d = 26, nobs = 1000; the
// whole design up to nobs = 1280 at d = 10) -- and only the rounds behind them are fetched again, every lane its row.
// Measured and dropped on top of this (profiles/r05_linreg_coop.txt): asking for a wave's next tile a round ahead (61.1 against
// 60.3 us per K-window); a dedicated folding wave working through the rounds as their flags come up, six helpers (69.7); the chain
// wave itself folding round by round as the flags come up, one barrier a generation (64.4).
template <int D>
__device__ __forceinline__ void lr_coop_rounds(const WindowParams& P, const double* __restrict__ rvec0, double* __restrict__ elds, double* __restrict__ tiles, unsigned int* ready, unsigned int tag, bool first, int lane, int w)
{
    constexpr int NG = 4, DP = ((D + 1) / 2) * 2, NBP = (NG * D + 15) / 16;
    const int pr = lane & (LINREG_PARTIALS - 1);
    double Bp[NBP];                                            // the four proposals by lanes (scripts/gen_ml_lrdpp.py)
#pragma unroll
    for (int k = 0; k < NBP; ++k) {
        const int en = 16 * k + pr;
        const int ec = (en < NG * D) ? en : 0;
        const int eg = ec / D;
        Bp[k] = rvec0[eg * DP + (ec - eg * D)];
    }
    const double* __restrict__ des = P.tp.design;
    const double* __restrict__ yo = P.tp.yobs;
    const int64_t nobs = P.tp.nobs;
    // (helper wave w = 1 .. ML_COOP_WAVES - 1 takes the rounds w - 1, w - 1 + (ML_COOP_WAVES - 1), ...: the chain wave takes none -- between
    //  the two barriers it makes the next generation's draws, about two rounds' worth of instructions)
    constexpr int NRES = ml_coop_resident_tiles(D);
    int rd = w - 1;
    for (int64_t base = 64 * (int64_t)(w - 1); base < nobs; base += 64 * (ML_COOP_WAVES - 1), rd += ML_COOP_WAVES - 1) {
        const int64_t o = base + lane;
        const bool have = o < nobs;
        double rowv[D];
        const double yv = yo[have ? o : 0];
        if (ML_COOP_TILE && rd < NRES && base + 64 <= nobs) {  // (wave-uniform) a resident round: its tile belongs to this wave alone
            double* __restrict__ tile = tiles + rd * (64 * D);
            if (first) {                                       // the launch's first generation: the tile comes in, as the contiguous piece of memory it is
                constexpr int NPC = 32 * D, NK = (NPC + 63) / 64;  // 16-byte pieces of a tile; per lane
                const double2* __restrict__ src = reinterpret_cast<const double2*>(des + base * D);
                double2 pc[NK];
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int pi = lane + 64 * k;
                    if (NPC % 64 == 0 || pi < NPC) pc[k] = src[pi];
                }
                // (all of the tile's loads in flight before the first is waited for: left alone the compiler sinks every load to its
                //  store and runs them one by one through one register quad -- thirteen L2 round trips a round)
#pragma unroll
                for (int k = 0; k < NK; ++k) asm volatile("" :: "v"(pc[k].x), "v"(pc[k].y));
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int pi = lane + 64 * k;
                    if (NPC % 64 == 0 || pi < NPC) reinterpret_cast<double2*>(tile)[pi] = pc[k];
                }
                wave_lds_handoff();
            }
            if constexpr (D % 2 == 0) {
#pragma unroll
                for (int jj = 0; jj < D; jj += 2) {
                    const double2 v = reinterpret_cast<const double2*>(tile + lane * D)[jj / 2];
                    rowv[jj] = v.x;
                    rowv[jj + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < D; ++jj) rowv[jj] = tile[lane * D + jj];
            }
        } else {
        const double* __restrict__ row = des + (have ? o : 0) * D;
        if constexpr (D % 2 == 0) {
#pragma unroll
            for (int jj = 0; jj < D; jj += 2) {
                const double2 v = reinterpret_cast<const double2*>(row)[jj / 2];
                rowv[jj] = v.x;
                rowv[jj + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < D; ++jj) rowv[jj] = row[jj];
        }
        }
        double a0 = -0.0, a1 = -0.0, a2 = -0.0, a3 = -0.0;
#include "demcz_ml_lrdpp_sel.inc"
        if (have) {
            elds[0 * ML_COOP_MAX_OBS + o] = yv - a0;
            elds[1 * ML_COOP_MAX_OBS + o] = yv - a1;
            elds[2 * ML_COOP_MAX_OBS + o] = yv - a2;
            elds[3 * ML_COOP_MAX_OBS + o] = yv - a3;
        }
        if (rd < NRES) {       // a resident round is done early (no trip to L2): the chain wave may fold it while the others are still being fetched
            asm volatile("" ::: "memory");         // (a wave's LDS operations are carried out in order: whoever sees the flag sees the residuals)
            if (lane == 0) __hip_atomic_store(&ready[rd], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
#endif

template <int TARGET, int D, int L, bool REC = false, bool LIVE = false, bool COOP = false>
__global__ void __launch_bounds__(COOP ? 64 * ML_COOP_WAVES : 64 * ML_WAVES) window_kernel_ml(const WindowParams P)
{
    static_assert(!COOP || (TARGET == TARGET_LINREG_SSE && ML_LRDPP), "COOP: the regression target's helper waves");
    constexpr int WPW = COOP ? 1 : ML_WAVES;                    // chain waves per workgroup, at most; the launch says how many (blockDim.x / 64: 1 or ML_WAVES)
    const int wpw = COOP ? 1 : (int)(blockDim.x >> 6);
    [[maybe_unused]] const int wv_raw = (int)(threadIdx.x >> 6);
    const int wv = COOP ? 0 : (int)(threadIdx.x >> 6);
    const int64_t vb = (int64_t)xcd_block(P) * wpw + wv;        // this wave's index among the chain waves (XCD-aware: demcz_kernels.h)
    // Round 5: the regression target at ANY dimension (test/example_linreg.jl:9-32 runs d = 26; the matrix-instruction kernels of
    // demcz_kernels_lr.h are d = 10 with the design resident in LDS).  Sixteen lanes per chain ARE the spec's sixteen interleaved
    // partial sums: lane l takes the observations o = l, l + 16, ... in increasing order -- each residual a
    // sequential fma chain in j, as target_logp does it -- and the partials meet in the spec's tree (l, l+8), (l, l+4), (l, l+2),
    // (0, 1) by lane shuffles.  The design comes through L2 (208 KB at d = 26, nobs = 1000: the same rows for every chain).
    static_assert(TARGET == TARGET_MVNORMAL || TARGET == TARGET_ISO_QUAD || TARGET == TARGET_LINREG_SSE, "lane-cooperative layout");
    static_assert(TARGET != TARGET_LINREG_SSE || (L == LINREG_PARTIALS && !REC), "regression target: sixteen lanes per chain, fused form");
    static_assert(!LIVE || REC, "LIVE launches are a property of the split form");
    if constexpr (REC) {
        if ((int64_t)blockIdx.x >= P.consumer_blocks) {
            pc_produce<D>(P, ((int64_t)blockIdx.x - P.consumer_blocks) * wpw + wv, (int)(threadIdx.x & 63));
            return;
        }
    }
    constexpr int G = 64 / L;                              // chains per wave = per workgroup
    constexpr int NG = G;
    constexpr int NPAIRS = (D == 1) ? 1 : (D + 1) / 2;
    constexpr int S = NPAIRS + 2;                          // Philox blocks per generation (full block)
    static_assert(S <= L, "one Philox block per lane");
    constexpr int NP = (D + L - 1) / L;                    // parameters owned per lane
    constexpr int DP = ((D + 1) / 2) * 2;                  // staging row, 16-byte multiple
    constexpr int YP = DP;                                 // second staging row: y components

    __shared__ double2 rec[WPW * NG * S];
    __shared__ __attribute__((aligned(16))) double rvec[WPW * NG * DP];
    __shared__ __attribute__((aligned(16))) double yvec[WPW * NG * YP];
    [[maybe_unused]] __shared__ double elds[COOP ? NG * ML_COOP_MAX_OBS : 1];
    [[maybe_unused]] __shared__ __attribute__((aligned(16))) double ctile[(COOP && ML_COOP_TILE) ? ml_coop_resident_tiles(D) * 64 * D : 2];
    [[maybe_unused]] __shared__ unsigned int cready[COOP ? ML_COOP_MAX_OBS / 64 : 1];      // a resident round's flag: the tag (gi + 1) of the generation whose residuals are in

    const int lane = threadIdx.x & 63;
    const int r = lane % L;
    const int gq = wv * NG + lane / L;                     // the chain's slot in the workgroup's LDS arrays
    const int64_t c_raw = vb * NG + lane / L;
    // (the regression target's log-density is computed by ALL 64 lanes of a wave for its four chains together: the lanes of chains
    //  beyond N stay -- as shadows of the last chain that store nothing -- while the wave has a chain at all)
    if constexpr (TARGET == TARGET_LINREG_SSE) { if (vb * NG >= P.N) return; }      // (COOP: the whole workgroup)
    else { if (c_raw >= P.N) return; }
#if ML_LRDPP
    if constexpr (COOP) {
        if (wv_raw != 0) {                  // helper wave: its share of every generation's residuals, between the chain wave's two barriers
            if (wv_raw == 1 && lane < ML_COOP_MAX_OBS / 64) cready[lane] = 0u;          // (in front of the first barrier)
            for (int gi = 0; gi < P.ngen; ++gi) {
                __syncthreads();
                lr_coop_rounds<D>(P, rvec, elds, ctile, cready, (unsigned int)gi + 1u, gi == 0, lane, wv_raw);
                __syncthreads();
            }
            return;
        }
    }
#endif
    const bool active = c_raw < P.N;
    const int64_t c = active ? c_raw : P.N - 1;
    const uint64_t chain = (uint64_t)(P.chain_id0 + c);
    const int role = (r < S) ? r : S - 1;

    // per-lane constants: owned parameters, their eps / mu and rows of W
    double x[NP], epsv[NP], muv[NP], Wrow[NP][(TARGET == TARGET_MVNORMAL) ? D : 1];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = r + L * k;
        const bool own = p < D;
        const int pc = own ? p : 0;
        x[k] = own ? P.Xcur[c + P.N * pc] : 0.0;
        epsv[k] = P.eps[pc];
        if constexpr (TARGET == TARGET_LINREG_SSE) muv[k] = 0.0;        // (x - 0.0 is x, bit for bit: the staging row holds the proposal itself)
        else muv[k] = P.tp.mu[pc];
        if constexpr (TARGET == TARGET_MVNORMAL) {
#pragma unroll
            for (int j = 0; j < D; ++j) Wrow[k][j] = (own && j <= pc) ? P.tp.Wp[(pc * (pc + 1)) / 2 + j] : 0.0;
        }
    }
    double lp = P.lpcur[c];
    const double scale = (D == 1) ? P.gamma : P.gamma / sqrt((double)(2 * D));
    [[maybe_unused]] philox_blocks rng;
    // REC: this lane's fields of the draw records (one double per generation, contiguous).  The packed
    // row indices are fetched one generation ahead of the gather that needs them.
    [[maybe_unused]] const double* rq_z[NP];
    [[maybe_unused]] const double* rq_lg = nullptr;
    [[maybe_unused]] const double* rq_ix = nullptr;
    [[maybe_unused]] uint64_t ix_next = 0;
    [[maybe_unused]] uint32_t ra = 0, rb = 0, ra_c = 0, rb_c = 0;     // rows of the gather in flight / being consumed
    if constexpr (REC) {
#pragma unroll
        // (record-major: the record of (generation, chain) is D + 2 contiguous doubles, WindowParams::rec_fields)
        for (int k = 0; k < NP; ++k) {
            const int p = r + L * k;
            rq_z[k] = P.rec_in + c * (D + 2) + ((D == 1) ? 0 : ((p < D) ? p : 0));
        }
        rq_lg = P.rec_in + c * (D + 2) + D;
        rq_ix = P.rec_in + c * (D + 2) + (D + 1);
        ix_next = (uint64_t)__double_as_longlong(rq_ix[0]);
        if constexpr (LIVE) {       // an earlier launch of the run already failed: do not wait again
            if (__hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        }
    }

    // Draws and archive gathers do not depend on the chain state (the row indices come from the
    // counter-based stream), so generation g+1's are issued before generation g's accept resolves:
    // the gather's L2 / Infinity-Cache / HBM latency hides behind a whole generation of work.
    double za[NP], zb[NP], zt[NP], logu_next;
    double za_c[NP], zb_c[NP], zt_c[NP];
    int to_b = P.to_boundary;            // countdown to the next K boundary
    int64_t nb = 0;                      // boundaries passed inside this launch
    unsigned int cnt_total = 0, cnt_first = 0;      // accept mask by ballot (WindowParams::acc_out)
    const unsigned long long speak64 = __builtin_amdgcn_ballot_w64(r == 0 && active);     // one lane speaks for a chain
    auto issue_draws = [&](int gi) {
        if constexpr (REC) {
            const uint64_t ii = ix_next;
            const int gn = (gi + 1 < P.ngen) ? gi + 1 : gi;
            const int64_t rgen = (int64_t)P.N * (D + 2);               // doubles from one generation's records to the next
            ix_next = (uint64_t)__double_as_longlong(rq_ix[rgen * gn]);
            logu_next = rq_lg[rgen * gi];
            ra = (uint32_t)ii;
            rb = (uint32_t)(ii >> 32);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = r + L * k;
                const int pc = (p < D) ? p : 0;
                zt[k] = rq_z[k][rgen * gi];
                // (also in a LIVE launch the first read takes the ordinary cached path: a stale copy can only show the
                //  sentinel where the row's final doubles are not yet seen, and a sentinel is asked for again with sc1
                //  loads below -- sc1 loads are slow to issue, demcz_kernels_pc.h)
                za[k] = P.Z[(int64_t)ra * P.ZS + pc];
                zb[k] = P.Z[(int64_t)rb * P.ZS + pc];
            }
            return;
        }
        uint64_t r1, r2, i1, i2;
        rng.block(P.seed, chain, (uint64_t)(P.g_first + gi - 1) * (uint64_t)S + (uint64_t)role, r1, r2);
        const double lg = dm_log(u_open(r1));
        double z0, z1;
        {
            const double R = sqrt(-2.0 * lg);
            double cs, sn;
            dm_sincos2pi(r2 >> 11, cs, sn);
            z0 = R * cs;
            z1 = R * sn;
        }
        draw_rows(r1, r2, (uint64_t)P.M, i1, i2);
        double2 e;
        e.x = (r == 0) ? __longlong_as_double((long long)i1) : ((r == S - 1) ? lg : z0);
        e.y = (r == 0) ? __longlong_as_double((long long)i2) : z1;
        if (r < S) rec[gq * S + r] = e;
        wave_lds_handoff();
        const double2 ii = rec[gq * S];
        logu_next = rec[gq * S + S - 1].x;
        const int64_t row1 = __double_as_longlong(ii.x), row2 = __double_as_longlong(ii.y);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = r + L * k;
            const int pc = (p < D) ? p : 0;
            const int zi = (D == 1) ? 0 : pc;
            zt[k] = reinterpret_cast<const double*>(rec)[(gq * S + 1 + zi / 2) * 2 + (zi & 1)];
            za[k] = P.Z[row1 * P.ZS + pc];
            zb[k] = P.Z[row2 * P.ZS + pc];
        }
        wave_lds_handoff();      // rec is rewritten by the next call
    };
    issue_draws(0);

    // One generation; PREFETCH issues the next generation's draws in the middle of it.  The last
    // generation of the window is a second instantiation without the prefetch (nothing to draw for).
    auto generation = [&](int gi, auto prefetch) -> bool {
#pragma unroll
        for (int k = 0; k < NP; ++k) { za_c[k] = za[k]; zb_c[k] = zb[k]; zt_c[k] = zt[k]; }
        const double logu = logu_next;
        ra_c = ra; rb_c = rb;
        // (COOP: the next generation's draws are made once the helper waves have this generation's proposals -- beside their work)
        if constexpr (decltype(prefetch)::value && !COOP) issue_draws(gi + 1);
        if constexpr (LIVE) {
            // the gather was issued a generation ago; rows appended since then by other waves read as the
            // sentinel until they are published: ask again (demcz_kernels_rec.h)
            auto sentinel_mask = [&]() __attribute__((always_inline)) {        // (as a scalar lane mask: sentinel_lanes, demcz_kernels_rec.h)
                unsigned long long m = 0ull;
#pragma unroll
                for (int k = 0; k < NP; ++k) m |= sentinel_lanes(za_c[k]) | sentinel_lanes(zb_c[k]);
                return m;
            };
            unsigned long long badm = sentinel_mask();
            int spins = 0;
            while (__builtin_expect(badm != 0ull, 0)) {       // wave-uniform
                if (live_poll_abandon(P, spins, ((badm >> (threadIdx.x & 63)) & 1ull) != 0ull, is_sentinel(za_c[0]) ? ra_c : rb_c, gi)) return true;
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int p = r + L * k;
                    const int pc = (p < D) ? p : 0;
                    if (is_sentinel(za_c[k])) za_c[k] = live_reload(P, &P.Z[(int64_t)ra_c * P.ZS + pc]);
                    if (is_sentinel(zb_c[k])) zb_c[k] = live_reload(P, &P.Z[(int64_t)rb_c * P.ZS + pc]);
                }
                badm = sentinel_mask();
            }
        }
        double delta[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const double diff = za_c[k] - zb_c[k];
            const double t1 = scale * diff;
            const double t2 = epsv[k] * zt_c[k];
            delta[k] = t1 + t2;
        }

        // ---- state-dependent part --------------------------------------------------------------
        double xp[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = r + L * k;
            xp[k] = x[k] + delta[k];
            if (p < D) rvec[gq * DP + p] = xp[k] - muv[k];
        }
        wave_lds_handoff();
        double rj[DP];
#pragma unroll
        for (int j = 0; j < DP / 2; ++j) {
            const double2 t = reinterpret_cast<const double2*>(rvec + gq * DP)[j];
            rj[2 * j] = t.x;
            rj[2 * j + 1] = t.y;
        }
        double lpp;
        if constexpr (TARGET == TARGET_MVNORMAL) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = r + L * k;
                double acc = Wrow[k][0] * rj[0];
#pragma unroll
                for (int j = 1; j < D; ++j) {
                    const double t = fma(Wrow[k][j], rj[j], acc);
                    acc = (j <= p) ? t : acc;
                }
                if (p < D) yvec[gq * YP + p] = acc;
            }
            wave_lds_handoff();
            double q = 0.0;
#pragma unroll
            for (int j = 0; j < DP / 2; ++j) {
                const double2 t = reinterpret_cast<const double2*>(yvec + gq * YP)[j];
                q = (j == 0) ? t.x * t.x : fma(t.x, t.x, q);
                if (2 * j + 1 < D) q = fma(t.y, t.y, q);
            }
            lpp = fma(-0.5, q, P.tp.c0);
        } else if constexpr (TARGET == TARGET_LINREG_SSE) {
            // The regression log-density of the wave's FOUR chains at once, a LANE PER OBSERVATION: lane l takes the observations
            // o = l, l + 64, ... and, with the one row of the design it has loaded, forms the residual of each of the four chains
            // (their proposals come out of the staging rows in LDS: wave-uniform addresses, broadcast reads) -- four times the
            // arithmetic per byte of a lane-per-partial-sum mapping, whose lanes each fetched their own copy of every row: 16 bytes x
            // 64 lanes through the CU's address path per load, 233 us per K-window at d = 26, nobs = 1000 (profiles/r05x_linreg_d26.txt).
            // The spec's order survives: partial p = o mod 16 of a chain takes its terms in increasing o -- within a round of 64
            // observations from lanes p, p + 16, p + 32, p + 48 in that order (four shuffles per chain), every 16-lane row keeping
            // its own copy of the sixteen partials; then the spec's tree over a row, and each row takes its chain's value.
            const double* __restrict__ des = P.tp.design;
            const double* __restrict__ yo = P.tp.yobs;
            const int64_t nobs = P.tp.nobs;
            const int pr = lane & (LINREG_PARTIALS - 1);       // the partial this lane keeps a copy of
            double part[NG];
#if ML_LRDPP
            if constexpr (COOP) {
                __syncthreads();                               // the four proposals are in rvec: helpers start
                if constexpr (decltype(prefetch)::value) issue_draws(gi + 1);
                // lane r of chain g = partial r: its terms in increasing o (target_logp's order), first term fma onto -0.0
                const double* ev = elds + (lane / L) * ML_COOP_MAX_OBS + pr;
                double pt = -0.0;
                const int nt = (int)(nobs / LINREG_PARTIALS);
                int t = 0;
                {   // the resident rounds come first in o and are done first (LDS-fed): folded while the other rounds are still being fetched
                    constexpr int NRES = ml_coop_resident_tiles(D);
                    const int nearly = (int)((nobs / 64 < NRES) ? nobs / 64 : NRES);
                    for (int rdr = 0; rdr < nearly; ++rdr) {
                        while (__hip_atomic_load(&cready[rdr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned int)gi + 1u) __builtin_amdgcn_s_sleep(2);
                        asm volatile("" ::: "memory");
                        double e4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) e4[u] = ev[(t + u) * LINREG_PARTIALS];
#pragma unroll
                        for (int u = 0; u < 4; ++u) pt = fma(e4[u], e4[u], pt);
                        t += 4;
                    }
                }
                __syncthreads();                               // every residual is in elds
                if (t + 8 <= nt) {                             // (the next eight terms are asked for before the chain of eight dependent fmas on these)
                    double ee[8], en[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) ee[u] = ev[(t + u) * LINREG_PARTIALS];
                    for (; t + 8 <= nt; t += 8) {
                        const bool more = t + 16 <= nt;
#pragma unroll
                        for (int u = 0; u < 8; ++u) en[u] = ev[(more ? t + 8 + u : t + u) * LINREG_PARTIALS];
#pragma unroll
                        for (int u = 0; u < 8; ++u) pt = fma(ee[u], ee[u], pt);
#pragma unroll
                        for (int u = 0; u < 8; ++u) ee[u] = en[u];
                    }
                }
                for (; t < nt; ++t) { const double e1 = ev[t * LINREG_PARTIALS]; pt = fma(e1, e1, pt); }
                if ((int64_t)nt * LINREG_PARTIALS + pr < nobs) { const double e1 = ev[nt * LINREG_PARTIALS]; pt = fma(e1, e1, pt); }
                pt = pt + 0.0;                                 // (a partial that never got a term is +0.0, as the spec starts it)
#pragma unroll
                for (int h = LINREG_PARTIALS / 2; h >= 1; h >>= 1) pt = pt + __shfl_down(pt, h, LINREG_PARTIALS);
#pragma unroll
                for (int g = 0; g < NG; ++g) part[g] = 0.0;
                lpp = -0.5 * __shfl(pt, 0, LINREG_PARTIALS);
            } else {
            // The four proposals BY LANES (scripts/gen_ml_lrdpp.py): entry e = g * D + j in lane e % 16 of every 16-lane row of
            // Bp[e / 16]; each fma of a residual names its entry (v_fmac_f64_dpp ... row_newbcast).  7 register pairs at D = 26
            // where wave-uniform copies took 208 registers (70 of them parked in AGPRs, an instruction per use).
            static_assert(NG == 4, "scripts/gen_ml_lrdpp.py: four chains per wave");
            constexpr int NBP = (NG * D + 15) / 16;
            double Bp[NBP];
#pragma unroll
            for (int k = 0; k < NBP; ++k) {
                const int en = 16 * k + pr;
                const int ec = (en < NG * D) ? en : 0;
                const int eg = ec / D;
                Bp[k] = rvec[(wv * NG + eg) * DP + (ec - eg * D)];
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) part[g] = -0.0;       // (fma(e, e, -0.0) is e * e: the first term needs no case of its own)
            auto load_row = [&](const double* __restrict__ row, double (&rowv)[D]) __attribute__((always_inline)) {
                if constexpr (D % 2 == 0) {
#pragma unroll
                    for (int jj = 0; jj < D; jj += 2) {
                        const double2 v = reinterpret_cast<const double2*>(row)[jj / 2];
                        rowv[jj] = v.x;
                        rowv[jj + 1] = v.y;
                    }
                } else {
#pragma unroll
                    for (int jj = 0; jj < D; ++jj) rowv[jj] = row[jj];
                }
            };
            auto residuals = [&](const double (&rowv)[D], double yv, double (&e)[NG]) __attribute__((always_inline)) {
                double a0 = -0.0, a1 = -0.0, a2 = -0.0, a3 = -0.0;
#include "demcz_ml_lrdpp_sel.inc"
                e[0] = yv - a0;
                e[1] = yv - a1;
                e[2] = yv - a2;
                e[3] = yv - a3;
            };
            const int64_t nfull = nobs & ~(int64_t)63;         // whole rounds of 64 observations: no lane or term is missing
            // (Asking for a round's row a round ahead -- a second copy of the row in registers, the loads pinned in front of the round's
            //  arithmetic -- was measured with the proposals by lanes too: 140.8 against 111.1 us per K-window.  The row's L2 round trip is
            //  not what a round waits for.)
            double rowc[D];
            for (int64_t base = 0; base < nfull; base += 64) {
                double e[NG];
                load_row(des + (base + lane) * D, rowc);
                residuals(rowc, yo[base + lane], e);
                // partial pr: the residuals of observations base + pr + 16 q, q = 0..3, in that order
#pragma unroll
                for (int q = 0; q < 64 / LINREG_PARTIALS; ++q) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const double eq = __shfl(e[g], pr + LINREG_PARTIALS * q, 64);
                        part[g] = fma(eq, eq, part[g]);
                    }
                }
            }
            if (nfull < nobs) {
                const int64_t o = nfull + lane;
                const bool have = o < nobs;
                double e[NG];
                load_row(des + (have ? o : 0) * D, rowc);
                residuals(rowc, yo[have ? o : 0], e);
#pragma unroll
                for (int q = 0; q < 64 / LINREG_PARTIALS; ++q) {
                    const bool ok = nfull + pr + LINREG_PARTIALS * q < nobs;
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const double eq = __shfl(e[g], pr + LINREG_PARTIALS * q, 64);
                        const double nv = fma(eq, eq, part[g]);
                        part[g] = ok ? nv : part[g];
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) part[g] = part[g] + 0.0;      // (a partial that never got a term is +0.0, as the spec starts it)
            }
#else
            double bb[NG][D];                                  // the four proposals (this lane's own chain among them)
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int j = 0; j < DP / 2; ++j) {
                    const double2 t = reinterpret_cast<const double2*>(rvec + (wv * NG + g) * DP)[j];
                    bb[g][2 * j] = t.x;
                    if (2 * j + 1 < D) bb[g][2 * j + 1] = t.y;
                }
            bool first = true;                                 // (the same for the four chains: it depends on o alone)
#pragma unroll
            for (int g = 0; g < NG; ++g) part[g] = 0.0;
            // (a round is one L2 round trip for the lane's row, then its arithmetic.  Prefetching the next round's row into registers
            //  was tried: 428 registers + 172 spilled, the same 200 us per K-window -- the round is bound by what it issues, the
            //  four chains' proposals among it: 208 registers of them)
            for (int64_t base = 0; base < nobs; base += 64) {
                const int64_t o = base + lane;
                const bool have = o < nobs;
                const double* row = des + (have ? o : 0) * D;
                double e[NG];
                {
                    double a[NG];
                    if constexpr (D % 2 == 0) {
                        const double2 u = reinterpret_cast<const double2*>(row)[0];
#pragma unroll
                        for (int g = 0; g < NG; ++g) { a[g] = u.x * bb[g][0]; a[g] = fma(u.y, bb[g][1], a[g]); }
#pragma unroll
                        for (int j = 2; j < D; j += 2) {
                            const double2 v = reinterpret_cast<const double2*>(row)[j / 2];
#pragma unroll
                            for (int g = 0; g < NG; ++g) { a[g] = fma(v.x, bb[g][j], a[g]); a[g] = fma(v.y, bb[g][j + 1], a[g]); }
                        }
                    } else {
                        const double u = row[0];
#pragma unroll
                        for (int g = 0; g < NG; ++g) a[g] = u * bb[g][0];
#pragma unroll
                        for (int j = 1; j < D; ++j) {
                            const double v = row[j];
#pragma unroll
                            for (int g = 0; g < NG; ++g) a[g] = fma(v, bb[g][j], a[g]);
                        }
                    }
                    const double yv = yo[have ? o : 0];
#pragma unroll
                    for (int g = 0; g < NG; ++g) e[g] = yv - a[g];
                }
                // partial pr: the residuals of observations base + pr + 16 q, q = 0..3, in that order
#pragma unroll
                for (int q = 0; q < 64 / LINREG_PARTIALS; ++q) {
                    const bool ok = base + pr + LINREG_PARTIALS * q < nobs;
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const double eq = __shfl(e[g], pr + LINREG_PARTIALS * q, 64);
                        const double nv = first ? eq * eq : fma(eq, eq, part[g]);
                        part[g] = ok ? nv : part[g];
                    }
                    first = first && !ok;
                }
            }
#endif
            if constexpr (!COOP) {
                // the spec's tree over a 16-lane row, per chain; then every row takes its own chain's sum
                double sse = 0.0;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
#pragma unroll
                    for (int h = LINREG_PARTIALS / 2; h >= 1; h >>= 1) part[g] = part[g] + __shfl_down(part[g], h, LINREG_PARTIALS);
                    const double sg = __shfl(part[g], 0, LINREG_PARTIALS);
                    sse = (lane / L == g) ? sg : sse;
                }
                lpp = -0.5 * sse;
            }
        } else {
            double q = 0.0;
#pragma unroll
            for (int j = 0; j < D; ++j) q = (j == 0) ? rj[0] * rj[0] : fma(rj[j], rj[j], q);
            lpp = -q;
        }
        double dlt = lpp - lp;
        if (P.temperature) dlt = dlt / P.temperature[gi];
        const bool acc = logu < dlt;
        {
            const double lp_new = acc ? lpp : lp;
            const unsigned int kc = wave_count_changed(lp_new, lp, speak64);
            cnt_total += kc;
            cnt_first = (gi == 0) ? kc : cnt_first;
            lp = lp_new;
        }
        const int64_t slot = P.slot_first + gi;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = r + L * k;
            x[k] = acc ? xp[k] : x[k];
            if (P.chain && p < D && active) P.chain[c + P.N * (p + (int64_t)D * slot)] = x[k];
        }
        if (P.chain && r == 0 && active) P.logobj[c + P.N * slot] = lp;
        if (--to_b == 0) {                  // generation divisible by K: runchain!'s append, demcz.jl:88-91
            to_b = P.K;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = r + L * k;
                if (p < D && active) {
                    if (P.do_append) {
                        if constexpr (LIVE) live_publish(P, nb, c, p, x[k]);
                        else P.Zw[(P.M_append + nb * P.N + c) * P.ZS + p] = x[k];
                    }
                    if (P.snap) P.snap[nb * P.N * D + c + P.N * p] = x[k];
                }
            }
            ++nb;
        }
        wave_lds_handoff();      // rvec / yvec are rewritten by the next generation
        return false;
    };
    // (a true return: a LIVE wait was abandoned -- the launch drains, the host reports live_err)
    for (int gi = 0; gi + 1 < P.ngen; ++gi)
        if (generation(gi, std::true_type{})) return;
    if (generation(P.ngen - 1, std::false_type{})) return;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = r + L * k;
        if (p < D && active) P.Xcur[c + P.N * p] = x[k];
    }
    if (r == 0 && active) P.lpcur[c] = lp;
    wave_store_counts(P, vb, cnt_total, cnt_first);
}

// ------------------------------------------------------------------------------------------------
// Latency layout with block updates (DEMCopt.Nblocks > 1 or a permuted block; update_blocks,
// src/demcz.jl:167-172): the same L-lane cooperation, one block-step at a time.  Block tables
// (position of every parameter inside every block, Philox offset and gamma scale of each block)
// sit in LDS; each block-step proposes only the block's parameters and re-evaluates the full
// log-density, exactly as the reference does.  Every block needs S_ib = 2 + ceil(nn/2) <= L
// Philox blocks (checked on the host).
// ------------------------------------------------------------------------------------------------
constexpr int MLB_MAX_BLOCKS = 64;

// REC / LIVE: the split form (demcz_kernels_rec.h, pcb_produce) -- a block-step's draws are read from the
// records (entries two block-steps ahead, archive rows one ahead) instead of made here.
// The split form runs FOUR waves to a workgroup -- nothing is shared but the block tables: a workgroup's waves are placed one per
// SIMD, whereas one-wave workgroups now and then land two to a SIMD (22-30 of C3's 1024 ran a quarter slower than the rest, and
// in a LIVE launch everybody who draws one of their rows falls back to their pace: scripts/mlb_stamps.py).
constexpr int MLB_REC_WAVES = 4;

// QB > 0 (round 4): the run's blocks are D / QB consecutive ranges of QB parameters each (C3: 20 = 4 x 5), so the sums of the
// quadratic form are cut at their boundaries (TargetParams::ngrp) -- and a block-step, which moves ONE block, recomputes only
// what that block feeds: the partial dot products P_{i,ib} of the rows at or below the block (a QB-long fma chain instead of a
// D-long one), y_i = the sum of row i's partials, and the partial sums of squares Q_b of the blocks from ib on; the partials of the
// other blocks and Q_b of the blocks before ib are kept per chain and committed on accept.  The doubles are those of the full
// re-evaluation in the same order (oracle: target_logp; src/demcz.jl:189).  QB = 0: any block structure, full evaluation.
// GM (QB = 0): the sums are cut at the boundaries of ANY consecutive blocks (TargetParams::gstart), full evaluation, the restarts
// selected by the group-start mask -- an instantiation of its own, so that runs whose sums are not grouped keep their code.
#ifndef MLB_DPP
#define MLB_DPP 1
#endif
#ifndef MLB_DPP_RECORD          // the record's row indices and log u the same way (every lane fetching the entry it needs itself): built,
#define MLB_DPP_RECORD 0        // bit-identical, and SLOWER -- 40.5 against 38.3 us per K-window: that hand-off is issued a block-step ahead,
#endif                          // its LDS latency was already hidden, and the per-lane choice of entry costs more instructions than it saves

template <int TARGET, int D, int L, bool REC = false, bool LIVE = false, int QB = 0, bool GM = false>
__global__ void __launch_bounds__(REC ? 64 * MLB_REC_WAVES : 64) window_kernel_mlb(const WindowParams P)
{
    static_assert(!GM || (QB == 0 && TARGET == TARGET_MVNORMAL), "grouped full evaluation: MvNormal, not the incremental form");
    static_assert(QB == 0 || (TARGET == TARGET_MVNORMAL && D % QB == 0 && D / QB >= 2 && D / QB <= 8), "incremental form: equal consecutive blocks");
    constexpr int QNB = (QB > 0) ? D / QB : 1;
    constexpr int WPW = REC ? MLB_REC_WAVES : 1;                // at most; the launch says how many (blockDim.x / 64: 1 or MLB_REC_WAVES)
    const int wpw = (int)(blockDim.x >> 6);
    const int wv = (int)(threadIdx.x >> 6);
    const int64_t vb = (int64_t)xcd_block(P) * wpw + wv;        // this wave's index among the consumer waves (XCD-aware)
    static_assert(TARGET == TARGET_MVNORMAL || TARGET == TARGET_ISO_QUAD, "block layout: MvNormal / isotropic targets");
    static_assert(!LIVE || REC, "LIVE launches are a property of the split form");
    if constexpr (REC) {
        if ((int64_t)blockIdx.x >= P.consumer_blocks) {
            pcb_produce(P, ((int64_t)blockIdx.x - P.consumer_blocks) * wpw + wv);
            return;
        }
    }
    constexpr int G = 64 / L;
    constexpr int NP = (D + L - 1) / L;
    constexpr int DP = ((D + 1) / 2) * 2;
    __shared__ double2 rec[WPW * G * L];
    __shared__ __attribute__((aligned(16))) double rvec[WPW * G * DP];
    __shared__ __attribute__((aligned(16))) double yvec[WPW * G * DP];
    __shared__ int slot_l[MLB_MAX_BLOCKS * D];             // position of parameter p in block ib, or -1
    __shared__ int blen_l[MLB_MAX_BLOCKS], boff_l[MLB_MAX_BLOCKS];
    __shared__ double bscale_l[MLB_MAX_BLOCKS];

    const int lane = (int)(threadIdx.x & 63);
    const int NB = P.Nblocks;
    for (int i = (int)threadIdx.x; i < NB * D; i += (int)blockDim.x) slot_l[i] = P.slot_of[i];
    if (threadIdx.x == 0) {
        int off = 0;
        for (int ib = 0; ib < NB; ++ib) {
            const int b = P.block_offsets[ib + 1] - P.block_offsets[ib];
            const int nn = (b == 1) ? 1 : b;
            blen_l[ib] = b;
            boff_l[ib] = off;
            bscale_l[ib] = (b == 1) ? P.gamma : P.gamma / sqrt((double)(2 * b));
            off += 1 + (nn + 1) / 2 + 1;
        }
    }
    __syncthreads();                                       // table hand-off only; the waves share nothing else

    const int r = lane % L, gq = wv * G + lane / L;        // gq: the chain's slot in the workgroup's LDS arrays
    const int64_t c = vb * G + lane / L;
    if (c >= P.N) return;
    const uint64_t chain = (uint64_t)(P.chain_id0 + c);

    double x[NP], epsv[NP], muv[NP], Wrow[NP][(TARGET == TARGET_MVNORMAL) ? D : 1];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = r + L * k;
        const bool own = p < D;
        const int pc = own ? p : 0;
        x[k] = own ? P.Xcur[c + P.N * pc] : 0.0;
        epsv[k] = P.eps[pc];
        muv[k] = P.tp.mu[pc];
        if constexpr (TARGET == TARGET_MVNORMAL) {
#pragma unroll
            for (int j = 0; j < D; ++j) Wrow[k][j] = (own && j <= pc) ? P.tp.Wp[(pc * (pc + 1)) / 2 + j] : 0.0;
        }
    }
    double lp = P.lpcur[c];
    [[maybe_unused]] const double qscale = (QB == 1) ? P.gamma : P.gamma / sqrt((double)(2 * (QB > 0 ? QB : 1)));
    // QB > 0: what is kept per chain between block-steps -- Pc[k][b] = P_{p_k, b} of this lane's rows, Qc[b] (every lane of the
    // chain holds the same) -- started from the state the launch begins with (two LDS hand-offs, once)
    [[maybe_unused]] double Pc[NP][QNB], Qc[QNB];
    if constexpr (QB > 0) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = r + L * k;
            if (p < D) rvec[gq * DP + p] = x[k] - muv[k];
        }
        wave_lds_handoff();
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = r + L * k;
            double y = 0.0;
#pragma unroll
            for (int b = 0; b < QNB; ++b) {
                double acc = Wrow[k][b * QB] * rvec[gq * DP + b * QB];
#pragma unroll
                for (int t = 1; t < QB; ++t) {
                    const double tt = fma(Wrow[k][b * QB + t], rvec[gq * DP + b * QB + t], acc);
                    acc = (b * QB + t <= p) ? tt : acc;
                }
                Pc[k][b] = acc;
                y = (b == 0) ? acc : ((b * QB <= p) ? y + acc : y);
            }
            if (p < D) yvec[gq * DP + p] = y;
        }
        wave_lds_handoff();
#pragma unroll
        for (int b = 0; b < QNB; ++b) {
            double qb = 0.0;
#pragma unroll
            for (int t = 0; t < QB; ++t) {
                const double yy = yvec[gq * DP + b * QB + t];
                qb = (t == 0) ? yy * yy : fma(yy, yy, qb);
            }
            Qc[b] = qb;
        }
        wave_lds_handoff();
    }
    [[maybe_unused]] philox_blocks rng;
    // REC: this lane's entry of a block-step's record: slot boff[ib] + role of chain c, generation gi
    [[maybe_unused]] const double2* rec2 = reinterpret_cast<const double2*>(P.rec_in);
    [[maybe_unused]] double2 e_pre = make_double2(0.0, 0.0);        // entry of the block-step issue_draws is called for next
    [[maybe_unused]] int64_t row1_n = 0, row2_n = 0, row1_c = 0, row2_c = 0;
    // (QB > 0: equal blocks -- length, Philox blocks per step, a block's offset and a parameter's place in it are arithmetic,
    //  not look-ups in the LDS tables: four dependent LDS round trips less per block-step, C3 45.9 -> 41.5 us per K-window.
    //  Tried next and dropped: every lane fetching its own pieces of the record -- row pair, log u, its normals: four small loads a
    //  block-step ahead -- instead of one entry per lane handed round through LDS: 43.2-43.7 us, the extra vector-memory
    //  instructions cost more than the LDS write, three reads and two hand-offs they replace.)
    constexpr int QSB = (QB > 0) ? 2 + (QB + 1) / 2 : 0;
    // MDPP (round 4): sixteen lanes per chain = one DPP row: what the chain's lanes hand each other in a block-step -- the record's
    // row indices and log u, the moved block's residuals, the rows' y -- is taken out of the owner lane's register by
    // v_mov_b64_dpp row_newbcast instead of going through LDS (write, wait, read): scripts/gen_mlb_dpp.py, profiles/r04s_dpp.txt
    constexpr bool MDPP = (MLB_DPP != 0) && QB == 5 && D == 20 && L == 16;
    constexpr bool MDPPR = MDPP && REC && (MLB_DPP_RECORD != 0);
    auto load_entry = [&](int gi, int ib) {
        const int b = (QB > 0) ? QB : blen_l[ib];
        const int nn = (b == 1) ? 1 : b;
        const int Sb = 2 + (nn + 1) / 2;
        int role = (r < Sb) ? r : Sb - 1;
        if constexpr (MDPPR) {
            // every lane fetches the entry IT needs: the owner of a parameter of the moved block the entry with its normal
            // (entry 1 + slot / 2), the first lane behind the block the row indices (entry 0), the next one log u (entry Sb - 1)
            role = (r == ((ib * QB + QB + 1) & (L - 1))) ? Sb - 1 : 0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = r + L * k;
                if (p < D && p >= ib * QB && p < ib * QB + QB) role = 1 + (p - ib * QB) / 2;
            }
        }
        const int g = (gi < P.ngen) ? gi : P.ngen - 1;
        const int bo = (QB > 0) ? ib * QSB : boff_l[ib];
        return rec2[((size_t)g * (size_t)P.N + (size_t)c) * (size_t)P.S + (size_t)(bo + role)];
    };
    if constexpr (REC) {
        if constexpr (LIVE) {       // an earlier launch of the run already failed: do not wait again
            if (__hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        }
        e_pre = load_entry(0, 0);
    }

    // draws of block-step (gi, ib), issued one block-step ahead of their use
    double za[NP], zb[NP], zt[NP], logu_next;
    int tslot[NP];
    auto issue_draws = [&](int gi, int ib) {
        const int b = (QB > 0) ? QB : blen_l[ib];
        const int nn = (b == 1) ? 1 : b;
        const int Sb = 2 + (nn + 1) / 2;
        const int role = (r < Sb) ? r : Sb - 1;
        if constexpr (MDPPR) {
            const double2 e = e_pre;
            {   // the entry of the block-step after this one, for the next call
                const int ib2 = (ib + 1 == NB) ? 0 : ib + 1;
                const int gi2 = (ib + 1 == NB) ? gi + 1 : gi;
                e_pre = load_entry(gi2, ib2);
            }
            double ex = e.x, ey = e.y, ix, iy, lg;
#define MLB_DPP_REC
#include "demcz_mlb_dpp_20_5.inc"
            logu_next = lg;
            row1_n = __double_as_longlong(ix);
            row2_n = __double_as_longlong(iy);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = r + L * k;
                const int pc = (p < D) ? p : 0;
                const int ts = (p < D && p >= ib * QB && p < ib * QB + QB) ? p - ib * QB : -1;
                tslot[k] = ts;
                zt[k] = (ts >= 0 && (ts & 1)) ? ey : ex;          // (its own entry holds its normal; unused where ts < 0)
                za[k] = (ts >= 0) ? P.Z[row1_n * P.ZS + pc] : 0.0;
                zb[k] = (ts >= 0) ? P.Z[row2_n * P.ZS + pc] : 0.0;
            }
            return;
        }
        if constexpr (REC) {
            const double2 e = e_pre;
            {   // the entry of the block-step after this one, for the next call
                const int ib2 = (ib + 1 == NB) ? 0 : ib + 1;
                const int gi2 = (ib + 1 == NB) ? gi + 1 : gi;
                e_pre = load_entry(gi2, ib2);
            }
            if (r < Sb) rec[gq * L + r] = e;
            wave_lds_handoff();
            const double2 ii = rec[gq * L];
            logu_next = rec[gq * L + Sb - 1].x;
            row1_n = __double_as_longlong(ii.x);
            row2_n = __double_as_longlong(ii.y);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = r + L * k;
                const int pc = (p < D) ? p : 0;
                const int ts = (QB > 0) ? ((p < D && p >= ib * QB && p < ib * QB + QB) ? p - ib * QB : -1) : ((p < D) ? slot_l[ib * D + pc] : -1);
                tslot[k] = ts;
                const int zi = (b == 1 || ts < 0) ? 0 : ts;
                zt[k] = reinterpret_cast<const double*>(rec)[(gq * L + 1 + zi / 2) * 2 + (zi & 1)];
                // (LIVE launches too: ordinary cached first read, sc1 only when a sentinel shows -- see window_kernel_ml)
                za[k] = (ts >= 0) ? P.Z[row1_n * P.ZS + pc] : 0.0;
                zb[k] = (ts >= 0) ? P.Z[row2_n * P.ZS + pc] : 0.0;
            }
            wave_lds_handoff();
            return;
        }
        uint64_t r1, r2, i1, i2;
        rng.block(P.seed, chain, (uint64_t)(P.g_first + gi - 1) * (uint64_t)P.S + (uint64_t)(((QB > 0) ? ib * QSB : boff_l[ib]) + role), r1, r2);
        const double lg = dm_log(u_open(r1));
        double z0, z1;
        {
            const double R = sqrt(-2.0 * lg);
            double cs, sn;
            dm_sincos2pi(r2 >> 11, cs, sn);
            z0 = R * cs;
            z1 = R * sn;
        }
        draw_rows(r1, r2, (uint64_t)P.M, i1, i2);
        double2 e;
        e.x = (r == 0) ? __longlong_as_double((long long)i1) : ((r == Sb - 1) ? lg : z0);
        e.y = (r == 0) ? __longlong_as_double((long long)i2) : z1;
        if (r < Sb) rec[gq * L + r] = e;
        wave_lds_handoff();
        const double2 ii = rec[gq * L];
        logu_next = rec[gq * L + Sb - 1].x;
        const int64_t row1 = __double_as_longlong(ii.x), row2 = __double_as_longlong(ii.y);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = r + L * k;
            const int pc = (p < D) ? p : 0;
            const int ts = (QB > 0) ? ((p < D && p >= ib * QB && p < ib * QB + QB) ? p - ib * QB : -1) : ((p < D) ? slot_l[ib * D + pc] : -1);
            tslot[k] = ts;
            const int zi = (b == 1 || ts < 0) ? 0 : ts;
            zt[k] = reinterpret_cast<const double*>(rec)[(gq * L + 1 + zi / 2) * 2 + (zi & 1)];
            za[k] = (ts >= 0) ? P.Z[row1 * P.ZS + pc] : 0.0;
            zb[k] = (ts >= 0) ? P.Z[row2 * P.ZS + pc] : 0.0;
        }
        wave_lds_handoff();
    };
    issue_draws(0, 0);

    int ib_n = 0, gi_n = 0;                                // block-step whose draws are in flight
    int to_b = P.to_boundary;
    int64_t nb = 0;
    unsigned int cnt_total = 0, cnt_first = 0;             // accept mask by ballot (WindowParams::acc_out)
    const unsigned long long speak64 = __builtin_amdgcn_ballot_w64(r == 0);
#ifdef DEMCZ_STAMPS
    unsigned long long sb[7] = {0, 0, 0, 0, 0, 0, 0}, sb_t = __builtin_readcyclecounter(), sb_waits = 0, sb_polls = 0;
#define MLB_TICK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); sb[i] += t_ - sb_t; sb_t = t_; } while (0)
#else
#define MLB_TICK(i) do { } while (0)
#endif
    for (int gi = 0; gi < P.ngen; ++gi) {
        const double lp_gen0 = lp;
        for (int ib = 0; ib < NB; ++ib) {
            MLB_TICK(5);
            if constexpr (LIVE) {
                // the gather was issued a block-step ago; rows appended since then by other waves read as the
                // sentinel until they are published: ask again (demcz_kernels_rec.h)
                row1_c = row1_n; row2_c = row2_n;
                // (the lanes that read a sentinel, as a MASK in scalar registers: compares straight into masks, a scalar test -- as a
                //  per-lane bool fed to a ballot it was a v_cndmask + v_cmp on a temporary register in every block-step: round 5,
                //  demcz_kernels_ps2.h)
                auto sentinel_mask = [&]() __attribute__((always_inline)) {
                    unsigned long long m = 0ull;
#pragma unroll
                    for (int k = 0; k < NP; ++k)
                        m |= sentinel_lanes(za[k]) | sentinel_lanes(zb[k]);
                    return m;
                };
                unsigned long long badm = sentinel_mask();
                int spins = 0;
                MLB_TICK(4);
#ifdef DEMCZ_STAMPS
                // (the poll loop is timed only when it is entered: a stamp costs 100-200 clocks, more than an average step waits)
                const bool sb_any = badm != 0ull;
                unsigned long long sb_w0 = 0;
                if (sb_any) { ++sb_waits; sb_w0 = __builtin_readcyclecounter(); }
#endif
                while (__builtin_expect(badm != 0ull, 0)) {       // wave-uniform
#ifdef DEMCZ_STAMPS
                    ++sb_polls;
#endif
                    if (live_poll_abandon(P, spins, ((badm >> lane) & 1ull) != 0ull, (unsigned)(is_sentinel(za[0]) ? row1_c : row2_c), gi)) return;
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const int p = r + L * k;
                        const int pc = (p < D) ? p : 0;
                        if (is_sentinel(za[k])) za[k] = live_reload(P, &P.Z[row1_c * P.ZS + pc]);
                        if (is_sentinel(zb[k])) zb[k] = live_reload(P, &P.Z[row2_c * P.ZS + pc]);
                    }
                    badm = sentinel_mask();
                }
#ifdef DEMCZ_STAMPS
                if (sb_any) sb[6] += __builtin_readcyclecounter() - sb_w0;
#endif
            }
            double delta[NP];
            bool inb[NP];
            const double scale = (QB > 0) ? qscale : bscale_l[ib];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const double diff = za[k] - zb[k];
                const double t1 = scale * diff;
                const double t2 = epsv[k] * zt[k];
                delta[k] = t1 + t2;
                inb[k] = tslot[k] >= 0;
            }
            const double logu = logu_next;
            ib_n = (ib + 1 == NB) ? 0 : ib + 1;
            gi_n = (ib + 1 == NB) ? gi + 1 : gi;
            MLB_TICK(0);
            issue_draws(gi_n, ib_n);                       // the one past the window is unused
            MLB_TICK(1);

            double xp[NP];
            [[maybe_unused]] double rres[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = r + L * k;
                xp[k] = inb[k] ? x[k] + delta[k] : x[k];
                rres[k] = xp[k] - muv[k];
                if constexpr (!MDPP) { if (p < D) rvec[gq * DP + p] = rres[k]; }
            }
            if constexpr (!MDPP) wave_lds_handoff();
            double lpp;
            [[maybe_unused]] double Pn[NP], Qn[QNB];
            if constexpr (QB > 0) {
                // the incremental form: block ib (wave-uniform) selects one of QNB straight-line bodies with literal offsets
                auto body = [&](auto ibc) __attribute__((always_inline)) {
                    constexpr int IB = decltype(ibc)::value;
                    double rb[QB];
                    if constexpr (MDPP) {
#define MLB_DPP_RB
#include "demcz_mlb_dpp_20_5.inc"
                    } else {
#pragma unroll
                    for (int t = 0; t < QB; ++t) rb[t] = rvec[gq * DP + IB * QB + t];
                    }
                    [[maybe_unused]] double ynew[NP];
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
                        const int p = r + L * k;
                        double acc = Wrow[k][IB * QB] * rb[0];
#pragma unroll
                        for (int t = 1; t < QB; ++t) {
                            const double tt = fma(Wrow[k][IB * QB + t], rb[t], acc);
                            acc = (IB * QB + t <= p) ? tt : acc;
                        }
                        Pn[k] = acc;
                        double y = (IB == 0) ? acc : Pc[k][0];
#pragma unroll
                        for (int b = 1; b < QNB; ++b) {
                            const double pb = (b == IB) ? acc : Pc[k][b];
                            y = (b * QB <= p) ? y + pb : y;
                        }
                        ynew[k] = y;
                        if constexpr (!MDPP) { if (p < D && p >= IB * QB) yvec[gq * DP + p] = y; }      // rows above the block keep their y (not read below)
                    }
                    [[maybe_unused]] double yy[QNB][QB];
                    if constexpr (MDPP) {
#define MLB_DPP_YY
#include "demcz_mlb_dpp_20_5.inc"
                    } else {
                        wave_lds_handoff();
                    }
                    double q = 0.0;
#pragma unroll
                    for (int b = 0; b < QNB; ++b) {
                        double qb = Qc[b];
                        if (b >= IB) {                     // (literals: resolved when the body is instantiated)
#pragma unroll
                            for (int t = 0; t < QB; ++t) {
                                double yv;
                                if constexpr (MDPP) yv = yy[b][t]; else yv = yvec[gq * DP + b * QB + t];
                                qb = (t == 0) ? yv * yv : fma(yv, yv, qb);
                            }
                        }
                        Qn[b] = qb;
                        q = (b == 0) ? qb : q + qb;
                    }
                    lpp = fma(-0.5, q, P.tp.c0);
                    // commit what the moved block fed, should the step be accepted (the same test as below, on the same values)
                    double dlq = lpp - lp;
                    if (P.temperature) dlq = dlq / P.temperature[gi];
                    const bool accq = logu < dlq;
#pragma unroll
                    for (int k = 0; k < NP; ++k) Pc[k][IB] = accq ? Pn[k] : Pc[k][IB];
#pragma unroll
                    for (int b = IB; b < QNB; ++b) Qc[b] = accq ? Qn[b] : Qc[b];
                };
                static_assert(QNB <= 8, "switch below");
                switch (ib) {
                case 0: body(std::integral_constant<int, 0>{}); break;
                case 1: body(std::integral_constant<int, (1 < QNB) ? 1 : 0>{}); break;
                case 2: body(std::integral_constant<int, (2 < QNB) ? 2 : 0>{}); break;
                case 3: body(std::integral_constant<int, (3 < QNB) ? 3 : 0>{}); break;
                case 4: body(std::integral_constant<int, (4 < QNB) ? 4 : 0>{}); break;
                case 5: body(std::integral_constant<int, (5 < QNB) ? 5 : 0>{}); break;
                case 6: body(std::integral_constant<int, (6 < QNB) ? 6 : 0>{}); break;
                default: body(std::integral_constant<int, (7 < QNB) ? 7 : 0>{}); break;
                }
            } else {
            double rj[DP];
#pragma unroll
            for (int j = 0; j < DP / 2; ++j) {
                const double2 t = reinterpret_cast<const double2*>(rvec + gq * DP)[j];
                rj[2 * j] = t.x;
                rj[2 * j + 1] = t.y;
            }
            if constexpr (TARGET == TARGET_MVNORMAL) {
              if constexpr (GM) {
                // sums cut at the block boundaries, any consecutive blocks: restarts selected by the group-start mask
                const uint64_t gs = P.tp.gstart;
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int p = r + L * k;
                    double y = 0.0, acc = 0.0;
                    bool have = false;
#pragma unroll
                    for (int j = 0; j < D; ++j) {
                        const bool st = (gs >> j) & 1ull, in = j <= p;
                        const double ysum = have ? y + acc : acc;
                        if (st && j > 0 && in) { y = ysum; have = true; }
                        const double prod = Wrow[k][j] * rj[j], fm = fma(Wrow[k][j], rj[j], acc);
                        acc = in ? (st ? prod : fm) : acc;
                    }
                    y = have ? y + acc : acc;
                    if (p < D) yvec[gq * DP + p] = y;
                }
                wave_lds_handoff();
                double q = 0.0, Qg = 0.0;
                bool haveq = false;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    const double yy = yvec[gq * DP + i];
                    const bool st = (gs >> i) & 1ull;
                    const double qsum = haveq ? q + Qg : Qg;
                    if (st && i > 0) { q = qsum; haveq = true; }
                    const double sq = yy * yy, fq = fma(yy, yy, Qg);
                    Qg = st ? sq : fq;
                }
                q = haveq ? q + Qg : Qg;
                lpp = fma(-0.5, q, P.tp.c0);
              } else {
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int p = r + L * k;
                    double acc = Wrow[k][0] * rj[0];
#pragma unroll
                    for (int j = 1; j < D; ++j) {
                        const double t = fma(Wrow[k][j], rj[j], acc);
                        acc = (j <= p) ? t : acc;
                    }
                    if (p < D) yvec[gq * DP + p] = acc;
                }
                wave_lds_handoff();
                double q = 0.0;
#pragma unroll
                for (int j = 0; j < DP / 2; ++j) {
                    const double2 t = reinterpret_cast<const double2*>(yvec + gq * DP)[j];
                    q = (j == 0) ? t.x * t.x : fma(t.x, t.x, q);
                    if (2 * j + 1 < D) q = fma(t.y, t.y, q);
                }
                lpp = fma(-0.5, q, P.tp.c0);
              }
            } else {
                double q = 0.0;
#pragma unroll
                for (int j = 0; j < D; ++j) q = (j == 0) ? rj[0] * rj[0] : fma(rj[j], rj[j], q);
                lpp = -q;
            }
            }
            double dlt = lpp - lp;
            if (P.temperature) dlt = dlt / P.temperature[gi];
            const bool acc = logu < dlt;
            lp = acc ? lpp : lp;
#pragma unroll
            for (int k = 0; k < NP; ++k) x[k] = acc ? xp[k] : x[k];
            wave_lds_handoff();
            MLB_TICK(2);
        }
        {
            const unsigned int kc = wave_count_changed(lp, lp_gen0, speak64);
            cnt_total += kc;
            cnt_first = (gi == 0) ? kc : cnt_first;
        }
        const int64_t slot = P.slot_first + gi;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int p = r + L * k;
            if (P.chain && p < D) P.chain[c + P.N * (p + (int64_t)D * slot)] = x[k];
        }
        if (P.chain && r == 0) P.logobj[c + P.N * slot] = lp;
        if (--to_b == 0) {
            to_b = P.K;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int p = r + L * k;
                if (p < D) {
                    if (P.do_append) {
                        if constexpr (LIVE) live_publish(P, nb, c, p, x[k]);
                        else P.Zw[(P.M_append + nb * P.N + c) * P.ZS + p] = x[k];
                    }
                    if (P.snap) P.snap[nb * P.N * D + c + P.N * p] = x[k];
                }
            }
            ++nb;
#ifdef DEMCZ_EXP_DRAIN_PUBLISH     // (diagnosis, scripts/mlb_stamps.py: how long the boundary's write-through stores take to be acknowledged)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        }
        MLB_TICK(3);
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = r + L * k;
        if (p < D) P.Xcur[c + P.N * p] = x[k];
    }
    if (r == 0) P.lpcur[c] = lp;
    wave_store_counts(P, vb, cnt_total, cnt_first);
#ifdef DEMCZ_STAMPS
    if (P.stamps && lane == 0 && vb < 65536) {     // [wait+poll+increments, next draws, dependent part, history, -, between]
        unsigned long long* o = P.stamps + (size_t)vb * 16;
        for (int i = 0; i < 6; ++i) o[8 + i] = sb[i];
        o[6] = sb[6]; o[7] = sb_polls;
        o[14] = (unsigned long long)P.ngen * (unsigned long long)NB;
        o[15] = sb_waits;
    }
#endif
}

}  // namespace demcz
