// demcz_kernels_lr.h -- K1b'': the regression target (-0.5 * sum((y - X b)^2), test/example_linreg.jl:32) on the
// FP64 matrix core at its full rate: v_mfma_f64_16x16x4_f64, sixteen chains per workgroup, four waves per workgroup.
//
// The arithmetic spec sums the squared residuals into 16 interleaved partials -- observation o
// goes to partial o mod 16, in increasing o -- and combines them by a fixed tree.  A 16x16x4 instruction produces
// 16 observations x 16 chains.  Fed with one tile of 16 consecutive observations it needs 16 chains in ONE wave to
// be busy, and a 2048-chain population is then 128 waves on 1024 SIMDs (100 us per K-window, round 1).  Here the
// sixteen chains of a workgroup are shared by FOUR waves instead, split by RESIDUE: wave w owns the partials
// 4w .. 4w+3, and one of its instructions covers those four residues of FOUR consecutive tiles -- its 16 output
// rows are (tile 4T+tt, residue 4w+ii), row = 4 tt + ii.  The instruction's result lane l holds rows
// (l >> 4) + 4 reg, reg = 0..3, of chain l & 15: residue 4w + (l >> 4) of the tiles 4T + reg -- so a lane squares and
// adds its four results in reg order and that IS the partial's own order (tiles ascending).  Nothing about the spec
// changes; the results are bit-identical to every other layout and to the oracle (the instruction accumulates like the
// sequential chain fma(a_k, b_k, .), k ascending, from C: scripts/probes/mfma_f64_order.hip).
//
// Per chain-update a wave issues ceil(nobs/64) x ceil(d/4) instructions of 64 cycles -- 48 at nobs = 1000, d = 10 --
// against 189 of the 4x4x4 form (about 50 cycles each), and a row of the design is fetched from LDS once for sixteen
// chains instead of four.  The 16 partials of a chain meet in LDS (one workgroup barrier per generation), every lane
// runs the tree and the accept test redundantly, wave 0 writes the history.
//
// Restates the same reference functions as window_kernel (src/demcz.jl:80-93,167-203, demcz_anneal.jl:172-178).
#pragma once

#include "demcz_kernels_ml.h"

#pragma clang fp contract(off)

namespace demcz {

// the A operand has 4 * ceil(D / 4) columns: a spare one takes the observations (see A_l below)
template <int D> constexpr bool LR_FOLD_Y = (D % 4) != 0;

constexpr int LR16_WAVES = 4;        // waves per workgroup = residue quarters
constexpr int LR16_CHAINS = 16;      // chains per workgroup = columns of the instruction

// dynamic LDS: the design in A-operand order per (tile group, wave, k-step), y per (tile group, wave, lane quarter, reg),
// the partials of two generations (ping-pong, rows padded to 18 doubles), the draws of a generation (fused form), two flag words
template <int D>
__host__ __device__ constexpr size_t lr16_dynamic_lds(int64_t nobs)
{
    constexpr int NMF = (D + 3) / 4, NPAIRS = (D == 1) ? 1 : (D + 1) / 2, S = NPAIRS + 2;
    const size_t ngrp = (size_t)((nobs + 63) / 64);
    return ngrp * LR16_WAVES * NMF * 64 * 8 + ngrp * LR16_WAVES * 16 * 8 + 2 * LR16_CHAINS * 18 * 8 + LR16_CHAINS * S * 16 + 16;
}

typedef double lr_d4 __attribute__((ext_vector_type(4)));
typedef double lr_d2 __attribute__((ext_vector_type(2)));

// Workgroup barrier for data that travels through LDS only: wait for this wave's LDS operations, then s_barrier.
// (__syncthreads() also waits for every outstanding GLOBAL access of the wave -- here that is the next generation's
//  prefetched records and archive rows, asked for a moment earlier precisely so that they can take a generation to
//  arrive: 2.5 us of a 4.3 us generation went into that wait.)
__device__ __forceinline__ void wg_barrier_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int D, bool REC, bool LIVE>
__global__ void __launch_bounds__(64 * LR16_WAVES) window_kernel_lr16(const WindowParams P)
{
    static_assert(!LIVE || REC, "LIVE launches are a property of the split form");
    constexpr int NMF = (D + 3) / 4;                       // k-steps of 4 per dot product
    constexpr int NPAIRS = (D == 1) ? 1 : (D + 1) / 2;
    constexpr int S = NPAIRS + 2;                          // Philox blocks per generation (full block)
    if constexpr (REC) {
        if ((int64_t)blockIdx.x >= P.consumer_blocks) {    // every wave of a producer workgroup is one 64-lane producer unit
            pc_produce<D>(P, ((int64_t)blockIdx.x - P.consumer_blocks) * LR16_WAVES + (int64_t)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
            return;
        }
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char lr_dyn_lds[];
    const int64_t nobs = P.tp.nobs;
    const int ngrp = (int)((nobs + 63) / 64);              // groups of four tiles of 16 observations
    double* A_l = reinterpret_cast<double*>(lr_dyn_lds);
    double* y_l = A_l + (size_t)ngrp * LR16_WAVES * NMF * 64;
    double* part_l = y_l + (size_t)ngrp * LR16_WAVES * 16;
    double2* draw_l = reinterpret_cast<double2*>(part_l + 2 * LR16_CHAINS * 18);
    unsigned int* flag_l = reinterpret_cast<unsigned int*>(draw_l + LR16_CHAINS * S);
    const int tid = threadIdx.x;
    // A_l[((T 4 + w) NMF + m) 64 + l] = X[16 (4T + tt) + 4w + ii][4m + kk], row i = l & 15 = 4 tt + ii, kk = l >> 4 (zero outside
    // the data: a padded observation then adds fma(0, 0, .) to its partial, a padded column fma(0, 0, .) to the dot product)
    for (int i = tid; i < ngrp * LR16_WAVES * NMF * 64; i += 64 * LR16_WAVES) {
        const int ll = i & 63;
        int rest = i >> 6;
        const int m = rest % NMF;
        rest /= NMF;
        const int ww = rest & 3, T = rest >> 2;
        const int row = ll & 15, kk = ll >> 4;
        const int64_t o = 16 * (int64_t)(4 * T + (row >> 2)) + 4 * ww + (row & 3);
        const int col = 4 * m + kk;
        // (LR_FOLD_Y: where D leaves a k slot free, the observation itself is column D of the A operand and the B operand there is
        //  -1: the last fma of the instruction's k chain is fma(y_o, -1, X_o b) = -(y_o - X_o b), rounded once like the
        //  subtraction it replaces, and its square is the same double -- the residual costs no vector instruction of its own)
        A_l[i] = (o < nobs && col < D) ? P.tp.design[o * D + col] : ((LR_FOLD_Y<D> && o < nobs && col == D) ? P.tp.yobs[o] : 0.0);
    }
    // y_l[((T 4 + w) 4 + q) 4 + reg] = y[16 (4T + reg) + 4w + q]
    for (int i = tid; i < ngrp * LR16_WAVES * 16; i += 64 * LR16_WAVES) {
        const int reg = i & 3, qq = (i >> 2) & 3, ww = (i >> 4) & 3, T = i >> 6;
        const int64_t o = 16 * (int64_t)(4 * T + reg) + 4 * ww + qq;
        y_l[i] = (o < nobs) ? P.tp.yobs[o] : 0.0;
    }
    if (tid == 0) {
        flag_l[0] = 0u;
        flag_l[1] = 0u;
        if constexpr (LIVE) flag_l[0] = __hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // an earlier launch failed
    }
    __syncthreads();
    if (flag_l[0] != 0u) return;                           // (workgroup-uniform: nobody is left at a barrier)

    const int w = tid >> 6, l = tid & 63, q = l >> 4, j = l & 15;
    const int64_t c_raw = (int64_t)blockIdx.x * LR16_CHAINS + j;
    const bool active = c_raw < P.N;                       // idle columns of the last workgroup shadow chain N-1
    const int64_t c = active ? c_raw : P.N - 1;
    const bool writer = (w == 0) && active;                // the four waves hold the same state; wave 0 writes it out
    const uint64_t chain = (uint64_t)(P.chain_id0 + c);

    // this lane's parameters: k = 4m + q, m = 0..NMF-1 -- the k the B operand of k-step m wants from this lane
    double x[NMF], epsv[NMF];
    bool own[NMF];
#pragma unroll
    for (int m = 0; m < NMF; ++m) {
        const int k = 4 * m + q;
        own[m] = k < D;
        x[m] = own[m] ? P.Xcur[c + P.N * k] : 0.0;
        epsv[m] = P.eps[own[m] ? k : 0];
    }
    double lp = P.lpcur[c];
    const double scale = (D == 1) ? P.gamma : P.gamma / sqrt((double)(2 * D));
    [[maybe_unused]] philox_blocks rng;
    // REC: the record of (generation g, this chain): D normals, log u, the packed row indices -- D + 2 contiguous doubles
    constexpr int F = D + 2;
    [[maybe_unused]] const double* rq0 = nullptr;
    [[maybe_unused]] const int64_t rq_gen = (int64_t)P.N * F;             // doubles from one generation's records to the next
    if constexpr (REC) rq0 = P.rec_in + c * F;

    // Draws and archive gathers do not depend on the chain state: generation g+1's are asked for while generation g
    // computes.  Program order inside a generation is CONSUME FIRST, THEN REFILL: the wait in front of the consumption
    // ("everything outstanding is back", which is all the compiler can ask for across the loop's back edge) then only
    // covers loads that have had a whole generation to arrive.  (With the refill in front of the consumption that wait
    // covered the loads just issued -- 0.8 us of every generation; a deeper ring did not help for the same reason.)
    double za[NMF], zb[NMF], zt[NMF], logu_next = 0.0;
    int64_t ra = 0, rb = 0;
    [[maybe_unused]] double ixn = 0.0;                     // REC: packed row indices of the generation after the one in flight
    auto gather = [&]() {
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            const int k = own[m] ? 4 * m + q : 0;
            // (LIVE too: the first read takes the ordinary cached path; a sentinel is asked for again with sc1 loads)
            za[m] = P.Z[ra * P.ZS + k];
            zb[m] = P.Z[rb * P.ZS + k];
        }
    };
    // REC: generation g of this launch's records (past the end: a harmless repeat of the last generation's loads, so
    // that every trip of the loop issues the same memory operations)
    auto fill_rec = [&](int g) {
        const uint64_t ii = (uint64_t)__double_as_longlong(ixn);
        const int gn = (g + 1 < P.ngen) ? g + 1 : P.ngen - 1;
        ixn = rq0[rq_gen * gn + (D + 1)];
        g = (g < P.ngen) ? g : P.ngen - 1;
        const double* rg = rq0 + rq_gen * g;
        logu_next = rg[D];
        ra = (int64_t)(uint32_t)ii;
        rb = (int64_t)(uint32_t)(ii >> 32);
#pragma unroll
        for (int m = 0; m < NMF; ++m) zt[m] = rg[own[m] ? 4 * m + q : 0];
        gather();
    };
    // Fused: the workgroup makes the 16 x S Philox blocks of generation g once, in LDS (thread t < 16 S: chain t / S, role
    // t % S); a barrier later every lane takes what it needs.
    auto make_draws = [&](int g) {
        if (tid < LR16_CHAINS * S) {
            const int jj = tid / S, role = tid % S;
            const int64_t cc_raw = (int64_t)blockIdx.x * LR16_CHAINS + jj;
            const uint64_t cch = (uint64_t)(P.chain_id0 + ((cc_raw < P.N) ? cc_raw : P.N - 1));
            uint64_t r1, r2;
            rng.block(P.seed, cch, (uint64_t)(P.g_first + g - 1) * (uint64_t)S + (uint64_t)role, r1, r2);
            double2 e;
            if (role == 0) {
                uint64_t i1, i2;
                draw_rows(r1, r2, (uint64_t)P.M, i1, i2);
                e.x = __longlong_as_double((long long)i1);
                e.y = __longlong_as_double((long long)i2);
            } else {
                const double lg = dm_log(u_open(r1));
                if (role == S - 1) {
                    e.x = lg;
                    e.y = 0.0;
                } else {
                    const double R = sqrt(-2.0 * lg);
                    double cs, sn;
                    dm_sincos2pi(r2 >> 11, cs, sn);
                    e.x = R * cs;
                    e.y = R * sn;
                }
            }
            draw_l[jj * S + role] = e;
        }
    };
    auto take_draws = [&]() {
        const double2 ii = draw_l[j * S];
        logu_next = draw_l[j * S + S - 1].x;
        ra = __double_as_longlong(ii.x);
        rb = __double_as_longlong(ii.y);
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            const int k = own[m] ? 4 * m + q : 0;
            const int zi = (D == 1) ? 0 : k;
            zt[m] = reinterpret_cast<const double*>(draw_l)[(j * S + 1 + zi / 2) * 2 + (zi & 1)];
        }
        gather();
    };
    if constexpr (REC) {
        ixn = rq0[D + 1];
        fill_rec(0);
    } else {
        make_draws(0);
        wg_barrier_lds();
        take_draws();
        wg_barrier_lds();                                  // draw_l is rewritten in the first generation
    }

    int to_b = P.to_boundary;            // countdown to the next K boundary
    int64_t nb = 0;                      // boundaries passed inside this launch
    unsigned int cnt_total = 0, cnt_first = 0;
    const unsigned long long speak64 = __builtin_amdgcn_ballot_w64(q == 0 && (w == LR16_WAVES - 1) && active);
#ifdef DEMCZ_STAMPS
    unsigned long long sa[6] = {0, 0, 0, 0, 0, 0}, sa_t = __builtin_readcyclecounter();
#define LR_TICK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); sa[i] += t_ - sa_t; sa_t = t_; } while (0)
#else
#define LR_TICK(i) do { } while (0)
#endif
    // history row and log_obj of generation `g` of the launch (runchain!, demcz.jl:84-85), from the current state.  The
    // four waves hold the same state; writing it out is shared: wave m stores parameter group m (k = 4m + q), the last
    // wave log_obj.
    auto write_hist = [&](int g) {
        const int64_t slot = P.slot_first + g;
#pragma unroll
        for (int m = 0; m < NMF; ++m)
            if (P.chain && own[m] && active && w == m) P.chain[c + P.N * ((4 * m + q) + (int64_t)D * slot)] = x[m];    // demcz.jl:84
        if (P.chain && q == 0 && active && w == LR16_WAVES - 1) P.logobj[c + P.N * slot] = lp;                         // demcz.jl:85
    };
    constexpr int PROW = 18;             // doubles between two chains' partials in LDS: 16 + 2, so that the sixteen chains'
                                         // rows fall into different banks (a stride of 16 doubles is a 16-way conflict)
    for (int gi = 0; gi < P.ngen; ++gi) {
        LR_TICK(5);
        // ---- consume what was asked for a generation ago ------------------------------------------------------------
        const double logu = logu_next;
        [[maybe_unused]] const int64_t ra_c = ra, rb_c = rb;
        [[maybe_unused]] bool failed = false;
        if constexpr (LIVE) {
            // rows appended since the gather was issued read as the sentinel until they are published: ask again (sc1)
            auto sentinel_mask = [&]() __attribute__((always_inline)) {        // (as a scalar lane mask: sentinel_lanes, demcz_kernels_rec.h)
                unsigned long long mm = 0ull;
#pragma unroll
                for (int m = 0; m < NMF; ++m) mm |= sentinel_lanes(za[m]) | sentinel_lanes(zb[m]);
                return mm;
            };
            unsigned long long badm = sentinel_mask();
            int spins = 0;
            while (__builtin_expect(badm != 0ull, 0)) {       // wave-uniform
                if (live_poll_abandon(P, spins, ((badm >> (threadIdx.x & 63)) & 1ull) != 0ull, (unsigned)(is_sentinel(za[0]) ? ra_c : rb_c), gi)) { failed = true; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int m = 0; m < NMF; ++m) {
                    const int k = own[m] ? 4 * m + q : 0;
                    if (is_sentinel(za[m])) za[m] = live_reload(P, &P.Z[ra_c * P.ZS + k]);
                    if (is_sentinel(zb[m])) zb[m] = live_reload(P, &P.Z[rb_c * P.ZS + k]);
                }
                badm = sentinel_mask();
            }
        }
        LR_TICK(4);
        // proposal (update_demcz_chain_block, demcz.jl:180-188) for this lane's parameters = the B operands
        double xp[NMF], bop[NMF];
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            const double diff = za[m] - zb[m];
            const double t1 = scale * diff;
            const double t2 = epsv[m] * zt[m];
            const double delta = t1 + t2;
            xp[m] = x[m] + delta;
            bop[m] = own[m] ? xp[m] : ((LR_FOLD_Y<D> && 4 * m + q == D) ? -1.0 : 0.0);
        }
        const double temp = P.temperature ? P.temperature[gi] : 1.0;
        __builtin_amdgcn_sched_barrier(0);
        // ---- ask for the next generation's (unconditionally: the same memory operations on every trip) --------------
        if constexpr (REC) fill_rec(gi + 1);
        else make_draws((gi + 1 < P.ngen) ? gi + 1 : gi);        // (taken behind the generation's barrier, below)
        __builtin_amdgcn_sched_barrier(0);
        // ---- the history row of the generation BEFORE this one (x, lp still hold it) ------------------------------------
        // Here rather than at that generation's end: the wait in front of the next consumption covers every outstanding
        // access of the wave, stores included, and a store issued a moment before it costs its whole round trip.  (The
        // APPEND of a boundary generation stays at its generation's end: a wave must publish its rows before it waits for
        // other waves' rows of the same boundary.)
        if (gi > 0) write_hist(gi - 1);
        __builtin_amdgcn_sched_barrier(0);
        LR_TICK(0);
        // residuals of this wave's residues, four tiles per instruction chain; TF chains in flight
        double sacc = 0.0;
        {
            constexpr int TF = 4;       // (8: 3.48 vs 3.42 us per generation)
            const double* Aw = A_l + (size_t)w * NMF * 64 + l;
            const double* yw = y_l + (size_t)(w * 4 + q) * 4;
            auto finish = [&](const lr_d4& a, int T) {
                if constexpr (LR_FOLD_Y<D>) {
                    sacc = fma(a[0], a[0], sacc); sacc = fma(a[1], a[1], sacc);
                    sacc = fma(a[2], a[2], sacc); sacc = fma(a[3], a[3], sacc);
                    (void)T; (void)yw;
                } else {
                    const double2 y01 = reinterpret_cast<const double2*>(yw + (size_t)T * 64)[0];
                    const double2 y23 = reinterpret_cast<const double2*>(yw + (size_t)T * 64)[1];
                    double e;
                    e = y01.x - a[0]; sacc = fma(e, e, sacc);
                    e = y01.y - a[1]; sacc = fma(e, e, sacc);
                    e = y23.x - a[2]; sacc = fma(e, e, sacc);
                    e = y23.y - a[3]; sacc = fma(e, e, sacc);
                }
            };
            int T = 0;
            // (tried: software-pipelining the batches so that the vector pipe squares batch b-1 while the matrix pipe runs
            //  batch b, order forced with sched_group_barrier -- 5300 -> 6200 cycles: on this chip the FP64 matrix
            //  instruction and the FP64 vector instructions share their multipliers, there is nothing to overlap)
            for (; T + TF <= ngrp; T += TF) {
                lr_d4 a[TF];
#pragma unroll
                for (int i = 0; i < TF; ++i) a[i] = lr_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int m = 0; m < NMF; ++m)
#pragma unroll
                    for (int i = 0; i < TF; ++i)
                        a[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aw[((size_t)(T + i) * LR16_WAVES * NMF + m) * 64], bop[m], a[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TF; ++i) finish(a[i], T + i);
            }
            for (; T < ngrp; ++T) {
                lr_d4 a = lr_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int m = 0; m < NMF; ++m)
                    a = __builtin_amdgcn_mfma_f64_16x16x4f64(Aw[((size_t)T * LR16_WAVES * NMF + m) * 64], bop[m], a, 0, 0, 0);
                finish(a, T);
            }
        }
        // the 16 partials of a chain meet in LDS; every lane combines them by the spec's tree
        double* part = part_l + (size_t)(gi & 1) * LR16_CHAINS * PROW;
        part[j * PROW + 4 * w + q] = sacc;
        LR_TICK(1);
        if constexpr (LIVE) { if (failed) flag_l[1] = 1u; }
        wg_barrier_lds();
        LR_TICK(2);
        if constexpr (LIVE) { if (flag_l[1] != 0u) return; }           // workgroup-uniform: the launch is being abandoned
        if constexpr (!REC) take_draws();                              // the next generation's draws, made above
        double pt[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double2 t = reinterpret_cast<const double2*>(part + j * PROW)[i];
            pt[2 * i] = t.x;
            pt[2 * i + 1] = t.y;
        }
#pragma unroll
        for (int h = 8; h >= 1; h >>= 1) {
#pragma unroll
            for (int i = 0; i < h; ++i) pt[i] = pt[i] + pt[i + h];
        }
        const double lpp = -0.5 * pt[0];
        double dlt = lpp - lp;
        if (P.temperature) dlt = dlt / temp;
        const bool acc = logu < dlt;                        // demcz.jl:197-203 / demcz_anneal.jl:172-178
        {
            const double lp_new = acc ? lpp : lp;
            const unsigned int kc = wave_count_changed(lp_new, lp, speak64);
            cnt_total += kc;
            cnt_first = (gi == 0) ? kc : cnt_first;
            lp = lp_new;
        }
#pragma unroll
        for (int m = 0; m < NMF; ++m) x[m] = acc ? xp[m] : x[m];
        if (--to_b == 0) {                  // generation divisible by K: runchain!'s append, demcz.jl:88-91
            to_b = P.K;
#pragma unroll
            for (int m = 0; m < NMF; ++m) {
                if (own[m] && active && w == m) {
                    const int k = 4 * m + q;
                    if (P.do_append) {
                        if constexpr (LIVE) live_publish(P, nb, c, k, x[m]);
                        else P.Zw[(P.M_append + nb * P.N + c) * P.ZS + k] = x[m];
                    }
                    if (P.snap) P.snap[nb * P.N * D + c + P.N * k] = x[m];
                }
            }
            ++nb;
        }
        if constexpr (!REC) wg_barrier_lds();               // draw_l is rewritten in the next generation
        LR_TICK(3);
    }
    write_hist(P.ngen - 1);
#pragma unroll
    for (int m = 0; m < NMF; ++m)
        if (own[m] && writer) P.Xcur[c + P.N * (4 * m + q)] = x[m];
    if (q == 0 && writer) P.lpcur[c] = lp;
    static_assert(NMF < LR16_WAVES, "parameter groups and log_obj each have a wave to write them");
    wave_store_counts(P, (int64_t)blockIdx.x * LR16_WAVES + w, cnt_total, cnt_first);
#ifdef DEMCZ_STAMPS
    if (P.stamps && (tid & 63) == 0 && blockIdx.x < 16384u) {      // per wave: [refill+wait+poll, proposal+matrix, barrier, tail, -, between steps]
        unsigned long long* o = P.stamps + ((size_t)blockIdx.x * LR16_WAVES + w) * 16;
        for (int i = 0; i < 6; ++i) o[8 + i] = sa[i];
        o[14] = (unsigned long long)P.ngen;
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// K1b-3 (the regression kernel's third form): the same instruction fed with EIGHT chains and two
// generations.  At C5's N = 2048 the kernel above is 128
// workgroups on 256 CUs, and a workgroup's log-density pass takes the same time whatever its 16 columns hold.  Here a
// workgroup runs eight chains; column jc holds chain jc's proposal of its next generation g, column jc + 8 its proposal of
// generation g + 1 AS IF GENERATION g WERE REJECTED (x unchanged: the proposal is x + delta(g+1), and delta does not depend
// on the state).  One pass gives both log-densities.  If g is rejected -- most are: the acceptance ratio the reference's
// annealer steers for is 0.1 .. 0.5 (demcz_anneal.jl:48-57) and with a fixed gamma it is a few per cent -- generation g + 1
// is resolved by the same pass and the chain advances by two; if g is accepted the second column is discarded and the chain
// advances by one.  Every generation is still evaluated exactly as the spec says (same operands, same order): results are
// bit-identical to every other layout.  Chains of a workgroup run at their own pace; each lane carries its chain's
// generation counter.  A second column never crosses a K boundary (a chain must publish its row before it may wait for
// rows of the same boundary, as in demcz_kernels_ps.h).
//
// Draws and archive rows.  A lane wants entries k = q, 4 + q, 8 + q of a generation's normals and of both archive rows: read
// from memory like that (window_kernel_lr16) a load instruction is 64 separate requests and costs ~120 cycles to issue, and
// two generations ahead for a chain that may move by one or two makes 25 of them a step.  Instead every WAVE keeps, in LDS of
// its own, a ring of four generations per chain -- record (D normals, log u, row indices) and both archive rows, 16 pieces of
// 16 bytes -- and refills the entries of generations g + 2, g + 3 at every step with four coalesced 16-byte loads per lane
// (asked for at the step's top, written to the ring at its end), the row indices of g + 4, g + 5 alongside.  Nothing is
// shared between waves, so no barrier is added; the lanes read what they need with LDS reads.
// Split form only (the draws come from records), four waves per workgroup split by residue as above.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int LR8_CHAINS = 8;
constexpr int LR8_RING = 4;                                 // generations per chain in a wave's ring
constexpr int LR8_CH_DOUBLES = LR8_RING * 32 + 4;           // (+ 4: the chains' entries fall into different banks)
constexpr int LR8_IXRING = 8;

// the value of the lane eight places on in the same row of sixteen (row_ror:8)
__device__ __forceinline__ double dpp_ror8(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(uint64_t)b, 0x128, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)((uint64_t)b >> 32), 0x128, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo));
}

template <int D>
__host__ __device__ constexpr size_t lr8s_dynamic_lds(int64_t nobs)
{
    return lr16_dynamic_lds<D>(nobs) + (size_t)LR16_WAVES * ((size_t)LR8_CHAINS * LR8_CH_DOUBLES * 8 + (size_t)LR8_CHAINS * LR8_IXRING * 8 + LR8_CHAINS * 4);
}

template <int D, bool LIVE>
__global__ void __launch_bounds__(64 * LR16_WAVES) window_kernel_lr8s(const WindowParams P)
{
    constexpr int NMF = (D + 3) / 4;
    constexpr int F = D + 2;
    static_assert(F % 2 == 0 && F / 2 + 2 * (D / 2) == 16, "a generation's record and two rows are sixteen 16-byte pieces");
    if ((int64_t)blockIdx.x >= P.consumer_blocks) {
        pc_produce<D>(P, ((int64_t)blockIdx.x - P.consumer_blocks) * LR16_WAVES + (int64_t)(threadIdx.x >> 6), (int)(threadIdx.x & 63));
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char lr_dyn_lds[];
    const int64_t nobs = P.tp.nobs;
    const int ngrp = (int)((nobs + 63) / 64);
    double* A_l = reinterpret_cast<double*>(lr_dyn_lds);
    double* y_l = A_l + (size_t)ngrp * LR16_WAVES * NMF * 64;
    double* part_l = y_l + (size_t)ngrp * LR16_WAVES * 16;
    unsigned int* flag_l = reinterpret_cast<unsigned int*>(part_l + 2 * LR16_CHAINS * 18);
    // (where the fused form keeps its draws) what each wave saw missing, two steps' worth: [step & 1][wave][generation g / g + 1]
    [[maybe_unused]] unsigned long long* miss_l = reinterpret_cast<unsigned long long*>(flag_l + 4);
    static_assert(LR16_CHAINS * ((D + 1) / 2 + 2) * 16 >= 2 * LR16_WAVES * 2 * 8, "the draws' LDS holds the waves' masks");
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63, q = l >> 4, j = l & 15, jc = j & 7;
    // this wave's ring, row-index ring and generation table
    double* ring_w = reinterpret_cast<double*>(lr_dyn_lds + lr16_dynamic_lds<D>(nobs)) + (size_t)w * (LR8_CHAINS * LR8_CH_DOUBLES + LR8_CHAINS * LR8_IXRING);
    double* ixr_w = ring_w + LR8_CHAINS * LR8_CH_DOUBLES;
    int* gen_w = reinterpret_cast<int*>(reinterpret_cast<double*>(lr_dyn_lds + lr16_dynamic_lds<D>(nobs)) + (size_t)LR16_WAVES * (LR8_CHAINS * LR8_CH_DOUBLES + LR8_CHAINS * LR8_IXRING)) + w * LR8_CHAINS;
    for (int i = tid; i < ngrp * LR16_WAVES * NMF * 64; i += 64 * LR16_WAVES) {      // (the layouts of window_kernel_lr16)
        const int ll = i & 63;
        int rest = i >> 6;
        const int m = rest % NMF;
        rest /= NMF;
        const int ww = rest & 3, T = rest >> 2;
        const int row = ll & 15, kk = ll >> 4;
        const int64_t o = 16 * (int64_t)(4 * T + (row >> 2)) + 4 * ww + (row & 3);
        const int col = 4 * m + kk;
        // (LR_FOLD_Y: where D leaves a k slot free, the observation itself is column D of the A operand and the B operand there is
        //  -1: the last fma of the instruction's k chain is fma(y_o, -1, X_o b) = -(y_o - X_o b), rounded once like the
        //  subtraction it replaces, and its square is the same double -- the residual costs no vector instruction of its own)
        A_l[i] = (o < nobs && col < D) ? P.tp.design[o * D + col] : ((LR_FOLD_Y<D> && o < nobs && col == D) ? P.tp.yobs[o] : 0.0);
    }
    for (int i = tid; i < ngrp * LR16_WAVES * 16; i += 64 * LR16_WAVES) {
        const int reg = i & 3, qq = (i >> 2) & 3, ww = (i >> 4) & 3, T = i >> 6;
        const int64_t o = 16 * (int64_t)(4 * T + reg) + 4 * ww + qq;
        y_l[i] = (o < nobs) ? P.tp.yobs[o] : 0.0;
    }
    if (tid == 0) {
        flag_l[0] = 0u;
        flag_l[1] = 0u;
        if constexpr (LIVE) flag_l[0] = __hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (l < LR8_CHAINS) gen_w[l] = 0;
    __syncthreads();
    if (flag_l[0] != 0u) return;

    const bool sp = j >= 8;                                // this lane's column: the proposal of generation g (false) / g + 1 (true)
    const int bx8 = xcd_block(P);          // (eight chains = 64 bytes of a history line per workgroup: XCD-aware, demcz_kernels.h)
    const int64_t c_raw = (int64_t)bx8 * LR8_CHAINS + jc;
    const bool active = c_raw < P.N;
    const int64_t c = active ? c_raw : P.N - 1;

    double x[NMF], epsv[NMF];
    bool own[NMF];
    int kq[NMF];
#pragma unroll
    for (int m = 0; m < NMF; ++m) {
        const int k = 4 * m + q;
        own[m] = k < D;
        kq[m] = own[m] ? k : 0;
        x[m] = own[m] ? P.Xcur[c + P.N * k] : 0.0;
        epsv[m] = P.eps[kq[m]];
    }
    double lp = P.lpcur[c];
    const double scale = (D == 1) ? P.gamma : P.gamma / sqrt((double)(2 * D));
    const int last = P.ngen - 1;
    const int64_t rq_gen = (int64_t)P.N * F;

    // ---- the wave's loader: piece pc of generation (chain ch's next) + off + go, for 4 x 64 = 8 chains x 2 generations x 16 pieces.
    // A piece's address is base + n * stride with n the generation (record pieces) or the archive row (row pieces): one
    // 32 x 32 -> 64-bit multiply-add per piece.  (N <= 2048 here -- one workgroup per CU at most -- so N F 8 < 2^32.)
    int ld_ch[4], ld_go[4], ld_at[4];
    bool ld_rec[4], ld_hi[4];
    uint64_t ld_base[4];
    uint32_t ld_stride[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pp = i * 64 + l, pc = pp & 15;
        ld_ch[i] = pp >> 5; ld_go[i] = (pp >> 4) & 1;
        const int64_t cr = (int64_t)bx8 * LR8_CHAINS + ld_ch[i];
        const int64_t cc = cr < P.N ? cr : P.N - 1;
        ld_rec[i] = pc < F / 2;
        ld_hi[i] = pc >= F / 2 + D / 2;
        ld_at[i] = ld_ch[i] * LR8_CH_DOUBLES + pc * 2;
        ld_base[i] = ld_rec[i] ? (uint64_t)(P.rec_in + cc * F + pc * 2) : (uint64_t)(P.Z + (pc - F / 2 - (ld_hi[i] ? D / 2 : 0)) * 2);
        ld_stride[i] = ld_rec[i] ? (uint32_t)(P.N * F * 8) : (uint32_t)(P.ZS * 8);
    }
    lr_d2 dat[4];
    int dat_at[4];                                         // where in the ring the piece goes (doubles)
    auto stage_ask = [&](int off) {
        int gl[4];
        double ixv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gl[i] = gen_w[ld_ch[i]] + off + ld_go[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) ixv[i] = ixr_w[ld_ch[i] * LR8_IXRING + (gl[i] & (LR8_IXRING - 1))];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t ii = (uint64_t)__double_as_longlong(ixv[i]);
            const uint32_t row = ld_hi[i] ? (uint32_t)(ii >> 32) : (uint32_t)ii;
            const uint32_t gc = (uint32_t)(gl[i] < last ? gl[i] : last);      // (past the launch's end: a harmless repeat of its last generation)
            const uint32_t n = ld_rec[i] ? gc : row;
            dat_at[i] = ld_at[i] + (gl[i] & (LR8_RING - 1)) * 32;
            dat[i] = *reinterpret_cast<const lr_d2*>(ld_base[i] + (uint64_t)n * ld_stride[i]);
        }
    };
    auto stage_put = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<lr_d2*>(ring_w + dat_at[i]) = dat[i];
    };
    // packed row indices: lane (chain l >> 3, e = l & 7) of the first fill takes generation e; a step's lanes l < 16 take
    // generations (chain's next) + 4 + (l & 1) of chain l >> 1
    {
        const int ch = l >> 3, e = l & 7;
        const int64_t cr = (int64_t)bx8 * LR8_CHAINS + ch;
        const int64_t cc = cr < P.N ? cr : P.N - 1;
        ixr_w[ch * LR8_IXRING + e] = P.rec_in[rq_gen * (e < last ? e : last) + cc * F + (D + 1)];
    }
    stage_ask(0);
    stage_put();
    const int ix_ch = (l >> 1) & 7, ix_e = l & 1;
    const int64_t ix_c = [&] { const int64_t cr = (int64_t)bx8 * LR8_CHAINS + ix_ch; return cr < P.N ? cr : P.N - 1; }();

    int gen = 0;                         // this chain's next generation (of the launch, 0-based)
    int to_b = P.to_boundary;            // countdown to the chain's next K boundary
    int64_t nb = 0;                      // boundaries it has passed inside this launch
    unsigned int cnt_total = 0, cnt_first = 0;
    const unsigned long long speak64 = __builtin_amdgcn_ballot_w64(q == 0 && (w == LR16_WAVES - 1) && active);
    // the history rows of the step before (written a step later, behind the loads: see window_kernel_lr16)
    double x_mid[NMF], lp_mid = lp;
    int h_gen = 0, h_adv = 0;
#pragma unroll
    for (int m = 0; m < NMF; ++m) x_mid[m] = x[m];
    // column jc writes the first of a step's generations, column jc + 8 the second; wave m parameter group m, the last wave log_obj
    // (one 32 x 32 -> 64-bit multiply-add per address: N D 8 < 2^32 with N <= 2048)
    const bool h_lane = P.chain && active && (w == LR16_WAVES - 1 ? q == 0 : 4 * w + q < D);
    const uint64_t h_base = (w == LR16_WAVES - 1) ? (uint64_t)(P.logobj + c) : (uint64_t)(P.chain + c + P.N * (4 * w + q));
    const uint32_t h_stride = (uint32_t)((w == LR16_WAVES - 1 ? P.N : P.N * D) * 8);
    auto write_hist = [&]() {
        const bool mine = h_lane && (sp ? h_adv == 2 : h_adv >= 1);
        const uint32_t slot = (uint32_t)(P.slot_first + h_gen + (sp ? 1 : 0));
        const bool midrow = !sp && h_adv == 2;            // the first of two generations: the state the rejected one left
        double v = midrow ? lp_mid : lp;                                                   // demcz.jl:85
#pragma unroll
        for (int m = 0; m < NMF; ++m) v = (w == m) ? (midrow ? x_mid[m] : x[m]) : v;       // demcz.jl:84
        if (mine) *reinterpret_cast<double*>(h_base + (uint64_t)slot * h_stride) = v;
    };
    const bool z_lane = active && !sp && w < NMF && 4 * w + q < D;
    const uint64_t z_base = (uint64_t)(P.Zw + (P.M_append + c) * P.ZS + (4 * w + q));
    const uint32_t z_stride = (uint32_t)(P.N * P.ZS * 8);
    constexpr int PROW = 18;
    int step = 0;
    [[maybe_unused]] int spins = 0;
#ifdef DEMCZ_STAMPS
    unsigned long long sa[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sa_t = __builtin_readcyclecounter();
#endif
    while (__builtin_amdgcn_ballot_w64(gen <= last) != 0ull) {       // (the four waves hold the same state: the same decision)
        LR_TICK(5);
        bool a_ok = gen <= last;
        bool b_ok = a_ok && gen < last && to_b != 1;                  // the second generation: inside the launch, not behind a boundary
        // ---- this lane's column's generation (g, or g + 1) of its chain, from the wave's ring
        const int gmine = gen + (sp ? 1 : 0);
        double* rp = ring_w + jc * LR8_CH_DOUBLES + (gmine & (LR8_RING - 1)) * 32;
        double za[NMF], zb[NMF], zt[NMF];
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            zt[m] = rp[kq[m]];
            za[m] = rp[F + kq[m]];
            zb[m] = rp[F + D + kq[m]];
        }
        const double logu = rp[D], ixm = rp[D + 1];
        const double tmine = P.temperature ? P.temperature[gmine < last ? gmine : last] : 1.0;
        // LIVE: rows appended since the gather was issued read as the sentinel until they are published.  They are asked for
        // again (sc1) and the answer is looked at in the NEXT step: a chain with a missing row sits this step out (if only its
        // second generation's is missing, that one is not attempted) while the workgroup's other chains go on -- chains of one
        // workgroup may be boundaries apart, so the workgroup must never stand still for one of them.  The four waves read the
        // archive each at its own moment: what they saw meets in LDS at the step's barrier, so that they decide alike.
        [[maybe_unused]] unsigned long long seen0 = 0ull, seen1 = 0ull;
        [[maybe_unused]] bool smine = false;
        if constexpr (LIVE) {
            bool b = false;
#pragma unroll
            for (int m = 0; m < NMF; ++m) b |= is_sentinel(za[m]) | is_sentinel(zb[m]);
            smine = (sp ? b_ok : a_ok) && b;
            seen0 = __builtin_amdgcn_ballot_w64(smine && !sp);
            seen1 = __builtin_amdgcn_ballot_w64(smine && sp);
        }
        LR_TICK(4);
        // the proposal (update_demcz_chain_block, demcz.jl:180-188) of this lane's column -- both columns' from the SAME x -- is its B operand
        double xp[NMF], bop[NMF];
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            const double diff = za[m] - zb[m];
            const double t1 = scale * diff;
            const double t2 = epsv[m] * zt[m];
            const double delta = t1 + t2;
            xp[m] = x[m] + delta;
            bop[m] = own[m] ? xp[m] : ((LR_FOLD_Y<D> && 4 * m + q == D) ? -1.0 : 0.0);
        }
        __builtin_amdgcn_sched_barrier(0);
        LR_TICK(6);
        if constexpr (LIVE) {
            if ((seen0 | seen1) != 0ull) {      // (behind the proposal: nothing of this step waits for the answer; it goes to the ring)
                if (smine) {
                    const uint64_t ii = (uint64_t)__double_as_longlong(ixm);
                    const int64_t ra = (int64_t)(uint32_t)ii, rb = (int64_t)(uint32_t)(ii >> 32);
#pragma unroll
                    for (int m = 0; m < NMF; ++m) {
                        if (is_sentinel(za[m])) za[m] = live_reload(P, &P.Z[ra * P.ZS + kq[m]]);
                        if (is_sentinel(zb[m])) zb[m] = live_reload(P, &P.Z[rb * P.ZS + kq[m]]);
                    }
                }
            }
        }
        LR_TICK(7);
        // ---- ask for generations g + 2, g + 3 and the row indices of g + 4, g + 5 (the same memory operations on every trip)
        stage_ask(2);
        const int ix_g = gen_w[ix_ch] + 4 + ix_e;
        double ix_new = 0.0;
        if (l < 2 * LR8_CHAINS) ix_new = P.rec_in[rq_gen * (ix_g < last ? ix_g : last) + ix_c * F + (D + 1)];
        __builtin_amdgcn_sched_barrier(0);
        LR_TICK(8);
        if (step > 0) write_hist();
        __builtin_amdgcn_sched_barrier(0);
        LR_TICK(0);
        double sacc = 0.0;
        {
            constexpr int TF = 4;
            const double* Aw = A_l + (size_t)w * NMF * 64 + l;
            const double* yw = y_l + (size_t)(w * 4 + q) * 4;
            auto finish = [&](const lr_d4& a, int T) {
                if constexpr (LR_FOLD_Y<D>) {
                    sacc = fma(a[0], a[0], sacc); sacc = fma(a[1], a[1], sacc);
                    sacc = fma(a[2], a[2], sacc); sacc = fma(a[3], a[3], sacc);
                    (void)T; (void)yw;
                } else {
                    const double2 y01 = reinterpret_cast<const double2*>(yw + (size_t)T * 64)[0];
                    const double2 y23 = reinterpret_cast<const double2*>(yw + (size_t)T * 64)[1];
                    double e;
                    e = y01.x - a[0]; sacc = fma(e, e, sacc);
                    e = y01.y - a[1]; sacc = fma(e, e, sacc);
                    e = y23.x - a[2]; sacc = fma(e, e, sacc);
                    e = y23.y - a[3]; sacc = fma(e, e, sacc);
                }
            };
            int T = 0;
            for (; T + TF <= ngrp; T += TF) {
                lr_d4 a[TF];
#pragma unroll
                for (int i = 0; i < TF; ++i) a[i] = lr_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int m = 0; m < NMF; ++m)
#pragma unroll
                    for (int i = 0; i < TF; ++i)
                        a[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aw[((size_t)(T + i) * LR16_WAVES * NMF + m) * 64], bop[m], a[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TF; ++i) finish(a[i], T + i);
            }
            for (; T < ngrp; ++T) {
                lr_d4 a = lr_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int m = 0; m < NMF; ++m)
                    a = __builtin_amdgcn_mfma_f64_16x16x4f64(Aw[((size_t)T * LR16_WAVES * NMF + m) * 64], bop[m], a, 0, 0, 0);
                finish(a, T);
            }
        }
        double* part = part_l + (size_t)(step & 1) * LR16_CHAINS * PROW;
        part[j * PROW + 4 * w + q] = sacc;
        LR_TICK(1);
        if constexpr (LIVE) {
            if (l == 0) {
                miss_l[((step & 1) * LR16_WAVES + w) * 2] = seen0;
                miss_l[((step & 1) * LR16_WAVES + w) * 2 + 1] = seen1;
            }
            // (a launch another workgroup has given up is left too; looked at now and then, by one lane for all)
            if (tid == 0 && (step & 255) == 255 && __hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) flag_l[1] = 1u;
        }
        wg_barrier_lds();
        LR_TICK(2);
        if constexpr (LIVE) {
            if (flag_l[1] != 0u) return;                       // workgroup-uniform
            unsigned long long m0 = 0ull, m1 = 0ull;
#pragma unroll
            for (int ww = 0; ww < LR16_WAVES; ++ww) {
                m0 |= miss_l[((step & 1) * LR16_WAVES + ww) * 2];
                m1 |= miss_l[((step & 1) * LR16_WAVES + ww) * 2 + 1];
            }
            const unsigned long long mine = 0x0101010101010101ull << jc;        // every lane of this chain, in any wave
            const bool miss0 = (m0 & mine) != 0ull, miss1 = (m1 & mine) != 0ull;
            // a chain that has waited live_spin_limit steps in a row gives the launch up (demcz_kernels_rec.h; the host then
            // redoes it one K-window at a time)
            spins = miss0 ? spins + 1 : 0;
            const bool timeout = miss0 && spins >= P.live_spin_limit;
            if (__builtin_amdgcn_ballot_w64(timeout) != 0ull) {                  // the same in all four waves
                if (timeout && atomicCAS(P.live_err, 0u, 1u) == 0u) {
                    P.live_err[1] = (unsigned)gen; P.live_err[2] = (unsigned)(uint32_t)(uint64_t)__double_as_longlong(ixm); P.live_err[3] = blockIdx.x;
                }
                return;
            }
            a_ok = a_ok && !miss0;
            b_ok = b_ok && !miss0 && !miss1;
        }
        // this column's partials, combined by the spec's tree, and its test (demcz.jl:197-203 / demcz_anneal.jl:172-178) against
        // the chain's CURRENT log_obj: right for generation g, and for g + 1 if g is rejected
        double lpm;
        {
            double pt[16];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double2 tt = reinterpret_cast<const double2*>(part + j * PROW)[i];
                pt[2 * i] = tt.x;
                pt[2 * i + 1] = tt.y;
            }
#pragma unroll
            for (int h = 8; h >= 1; h >>= 1) {
#pragma unroll
                for (int i = 0; i < h; ++i) pt[i] = pt[i] + pt[i + h];
            }
            lpm = -0.5 * pt[0];
        }
        LR_TICK(9);
        double dl = lpm - lp;
        if (P.temperature) dl = dl / tmine;
        const unsigned long long pass = __builtin_amdgcn_ballot_w64(logu < dl);
        // the other column of the chain is eight lanes away, in the same row of sixteen
        const double lpo = dpp_ror8(lpm);
        double xo[NMF];
#pragma unroll
        for (int m = 0; m < NMF; ++m) xo[m] = dpp_ror8(xp[m]);
        const double lpa = sp ? lpo : lpm, lpb = sp ? lpm : lpo;
        const bool acca = a_ok && ((pass >> (l & ~8)) & 1ull) != 0ull;
        const bool two = b_ok && !acca;                      // generation g was rejected: g + 1 started from the same state
        const bool accb = two && ((pass >> (l | 8)) & 1ull) != 0ull;
        const int adv = a_ok ? (two ? 2 : 1) : 0;
        const double lp_a = acca ? lpa : lp;                // log_obj after generation g
        const double lp_new = accb ? lpb : lp_a;            // ... after the step
        {
            // acceptance counts by ballot: column jc speaks for generation g, column jc + 8 for g + 1
            const double df = sp ? (lp_new - lp_a) : (lp_a - lp);
            const bool counted = sp ? two : a_ok;
            const unsigned long long chg = __builtin_amdgcn_fcmp(df, 0.0, 14 /* UNE */) & speak64 & __builtin_amdgcn_ballot_w64(counted);
            cnt_total += (unsigned int)__builtin_popcountll(chg);
            cnt_first += (unsigned int)__builtin_popcountll(chg & __builtin_amdgcn_ballot_w64(!sp && gen == 0));
        }
        h_gen = gen;
        h_adv = adv;
        lp_mid = lp;
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            x_mid[m] = x[m];
            const double xa = sp ? xo[m] : xp[m], xb = sp ? xp[m] : xo[m];
            x[m] = acca ? xa : (accb ? xb : x[m]);
        }
        lp = lp_new;
        LR_TICK(10);
        // ---- what was asked for at the top goes to the ring (the generation table still says where), then the table moves
        stage_put();
        if (l < 2 * LR8_CHAINS) ixr_w[ix_ch * LR8_IXRING + (ix_g & (LR8_IXRING - 1))] = ix_new;
        if constexpr (LIVE) {
            if ((seen0 | seen1) != 0ull) {
#pragma unroll
                for (int m = 0; m < NMF; ++m) {
                    if (smine) { rp[F + kq[m]] = za[m]; rp[F + D + kq[m]] = zb[m]; }
                }
            }
        }
        gen += adv;
        to_b -= adv;
        if (q == 0 && !sp) gen_w[jc] = gen;
        LR_TICK(11);
        if (to_b == 0 && adv != 0) {        // the step ended on a generation divisible by K: runchain!'s append, demcz.jl:88-91
            to_b = P.K;
            if (z_lane) {                   // wave m: parameter group m of column jc's lanes
                double v = 0.0;
#pragma unroll
                for (int m = 0; m < NMF; ++m) v = (w == m) ? x[m] : v;
                if (P.do_append) {
                    double* dst = reinterpret_cast<double*>(z_base + (uint64_t)(uint32_t)nb * z_stride);
                    if constexpr (LIVE) live_publish(P, nb, c, 4 * w + q, v);
                    else *dst = v;
                }
                if (P.snap) P.snap[nb * P.N * D + c + P.N * (4 * w + q)] = v;
            }
            ++nb;
        }
        ++step;
        LR_TICK(3);
    }
    write_hist();
    const bool writer = (w == 0) && active && !sp;
#pragma unroll
    for (int m = 0; m < NMF; ++m)
        if (own[m] && writer) P.Xcur[c + P.N * (4 * m + q)] = x[m];
    if (q == 0 && writer) P.lpcur[c] = lp;
    static_assert(NMF < LR16_WAVES, "parameter groups and log_obj each have a wave to write them");
    wave_store_counts(P, (int64_t)bx8 * LR16_WAVES + w, cnt_total, cnt_first);
#ifdef DEMCZ_STAMPS
    if (P.stamps && (tid & 63) == 0 && blockIdx.x < 16384u) {      // as window_kernel_lr16, per STEP; [15]: steps
        unsigned long long* o = P.stamps + ((size_t)blockIdx.x * LR16_WAVES + w) * 16;
        for (int i = 0; i < 6; ++i) o[8 + i] = sa[i];
        for (int i = 0; i < 6; ++i) o[i] = sa[6 + i];
        o[14] = (unsigned long long)P.ngen;
        o[15] = (unsigned long long)step;
    }
#endif
}

}  // namespace demcz
