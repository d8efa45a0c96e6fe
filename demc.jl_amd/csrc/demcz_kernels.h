// demcz_kernels.h -- HIP kernels of the DEMCz chain update (gfx950).
//
// K1  window_kernel      propose -> log-density -> Metropolis accept -> history/Z write for one
//                        K-window of generations; one lane per chain ("throughput layout").
//                        Restates runchain!/update_blocks/update_demcz_chain_block/accept of
//                        src/demcz.jl:80-93,167-203 and the tempered overloads of
//                        src/demcz_anneal.jl:67-80,142-178 for all N chains at once.
// K4  append_rows_kernel scatter gathered rows into the parameter-major archive
// K5  rhat_*             split-R-hat moments (src/utils.jl:2-20)
// K6  logp_kernel        initial log-densities (src/demcz.jl:17)
// K7  accept_ratio / mean_cov reductions (src/utils.jl:61, 96-111)
#pragma once

#include "demcz_device.h"

#pragma clang fp contract(off)

// Diagnostic build (-DDEMCZ_STAMPS, never the shipped library): lane 0 of a workgroup records the
// shader clock at a few points of the split-layout kernel.
#ifdef DEMCZ_STAMPS
#define DEMCZ_STAMP(P, i) do { if ((P).stamps && threadIdx.x == 0 && blockIdx.x < 65536u) (P).stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define DEMCZ_STAMP(P, i) do { } while (0)
#endif

namespace demcz {

enum { TARGET_MVNORMAL = 0, TARGET_ISO_QUAD = 1, TARGET_LINREG_SSE = 2, TARGET_HOST_CALLBACK = 3 };

constexpr int MAX_D = 64;        // generic (runtime-d) path keeps x / xprop / normals in LDS
constexpr int LINREG_PARTIALS = 16;   // interleaved partial sums of the regression SSE (arithmetic spec)
constexpr int WINDOW_BS = 64;    // one wave per workgroup: small N spreads over as many CUs as waves

struct TargetParams {
    const double* mu;        // d
    const double* Wp;        // MVNORMAL: packed row-major lower triangle, row i at i(i+1)/2
    double c0;
    const double* design;    // LINREG: row-major nobs x d (one observation's regressors contiguous)
    const double* yobs;
    int64_t nobs;
    // MVNORMAL, round 4: the sums of the quadratic form cut at the block boundaries (oracle: mvn_groups).
    // ngrp > 1 iff the run's blocks, in order, are consecutive index ranges covering 0..d-1; goff = their ngrp + 1 offsets
    // (device memory), gstart = bit j set iff parameter j starts a group.  ngrp <= 1: one group, the order of rounds 1-3.
    int32_t ngrp;
    const int32_t* goff;
    uint64_t gstart;
};

constexpr int DEMCZ_MAX_PEERS = 7;      // replicas besides a handle's own: the eight GPUs of a node

struct WindowParams {
    // archive: row-major on the device, row r at Z + r*ZS (ZS = padded row stride in doubles,
    // a multiple of 2 so rows are 16-byte aligned; d=5 -> 8 = one 64-byte line per row).  The
    // boundary keeps Julia's parameter-major layout; set_state/get_state transpose.  A random row
    // then costs one or two cache lines instead of d of them (measured: profiles/r01_*pmc*).
    const double* Z;
    double* Zw;              // same buffer, for the append
    int64_t ZS;
    int64_t M;               // rows visible to every proposal of this window
    // chain state
    double* Xcur;            // N x d (ld N)
    double* lpcur;           // N
    // history (slot s holds generation g0 + 1 + s)
    double* chain;           // N x d x Gcap or nullptr
    double* logobj;          // N x Gcap or nullptr
    const double* temperature;  // per generation of this window, or nullptr
    int64_t N;
    int64_t chain_id0;
    int32_t d;
    int32_t ngen;            // generations in this window
    int64_t g_first;         // 1-based generation index of the first one
    int64_t slot_first;      // history slot of g_first
    double gamma;
    uint64_t seed;
    int64_t S;               // Philox blocks per generation
    int32_t Nblocks;
    int32_t do_append;       // at every K boundary inside the launch the kernel appends rows M_append + b*N + ic
    int64_t M_append;        // first free archive row (== M unless appended rows become visible later)
    double* snap;            // sharded runs with deferred exchange: boundary states, one N x d (ld N) slab per boundary
    int32_t K;
    int32_t to_boundary;     // generations from the first one of the launch to the next K boundary (1 = the first one is it)
    const int32_t* block_offsets;
    const int32_t* slot_of;  // Nblocks x d: position of parameter p inside block ib, or -1
    const double* eps;
    TargetParams tp;
    // split layout (demcz_kernels_pc.h): draw records of THIS launch's generations, and what the
    // producer half of the launch prepares for the NEXT one
    const double* rec_in;
    double* rec_out;
    int64_t rec_stride;      // generations each (field, chain) row of a record buffer holds
    int32_t rec_fields;      // 0: records are generation-major per (field, chain) -- what the replicated consumer reads, a lane
                             // per generation.  F > 0: one record of F doubles per (generation, chain), fields contiguous --
                             // what the lane-per-parameter consumers read (all lanes at the same generation): their loads
                             // then touch a few cache lines instead of one per lane
    const int32_t* slot_role; // block-structured split runs: what Philox block s of a generation is (0 rows, 1 normal pair, 2 log u)
    int64_t next_g_first;    // stream generation index of the next launch's first generation
    int64_t next_M;          // rows its first generation draws from
    int64_t next_rows;       // ... plus this many per K boundary it has passed (0: appended rows become visible later)
    int32_t next_boff;       // K - (generations from its first one to the next boundary)
    int32_t next_ngen;
    int32_t consumer_blocks; // workgroups [0, consumer_blocks) consume, the rest produce
    unsigned int* live_err;  // LIVE launches: set when a row another wave should have appended never showed up
    int32_t live_spin_limit; // LIVE launches: polls of one wait before it is given up (LIVE_SPIN_LIMIT unless a test lowers it)
    // Accept mask by wavefront ballot: every wave counts, per generation, the chains whose log_obj changed
    // ((lp_after - lp_before) != 0: the event diff(log_obj, dims=2) .!= 0 counts, demcz.jl:42, demcz_anneal.jl:50) with
    // one ballot + s_bcnt1 into a scalar register and writes {sum over the launch, count of its first generation}
    // once, at the end, to acc_out[2 * wave] -- no atomics, no second pass over log_obj (nullptr: not wanted).
    unsigned int* acc_out;
    // The first LIVE launch after a verified point keeps the state it started from for a possible redo (live_verify): a kernel that
    // can (window_kernel_ps2) writes it here as it loads it -- N x d (ld N) and N -- instead of two copy launches in front of it.
    double* safe_X;
    double* safe_lp;
    // Replicated archives (a sharded run whose rows are handed over INSIDE the launch, demcz_kernels_rec.h: live_publish): a
    // boundary appends brows rows in all -- this handle's chains are rows row_off .. row_off + N of them -- and every row goes
    // to this handle's own archive AND to the n_peers other replicas (peer GPUs' memory opened over IPC, or other handles of
    // the process).  Unsharded: brows = N, row_off = 0, n_peers = 0.
    int64_t brows;
    int64_t row_off;
    int32_t n_peers;
    double* peer_Z[DEMCZ_MAX_PEERS];
    // What the kernels' raw-buffer descriptors clip at (round 5: every descriptor is sized to its allocation, so that an offset
    // gone wrong is dropped by the hardware instead of written / read somewhere inside the 4 GB a descriptor could span):
    // z_bytes = bytes addressable from Z -- the archive, or with the arena of the wave-per-chain layout the archive, both record
    // buffers and the temperatures behind it; hist_bytes = chain ‖ log_obj as ONE allocation from `chain` (0: two allocations or
    // none).  Both saturate at 0xffffffff; kernels that address through them are only launched when everything lies below that.
    uint32_t z_bytes;
    uint32_t hist_bytes;
#ifdef DEMCZ_STAMPS
    unsigned long long* stamps;   // diagnostic build only (scripts/stamps.py): 16 values per workgroup (8 stamps, 8 sums)
#endif
};

// Which block of chains a workgroup runs.  Workgroups go to the chip's eight XCDs round-robin by index, each XCD with an L2 of
// its own; a 128-byte line of the chain / log_obj history (one generation, one parameter, sixteen consecutive chains) is
// completed by 8-byte stores of the waves that run those chains.  Handed out in index order, a line's chains sit on several XCDs:
// each L2 holds a partly written copy and writes its part back on its own.  Giving XCD x the x-th eighth of the chain blocks
// keeps every line's writers on one XCD, whose L2 merges them (window_kernel_ps2 at C2: 181 -> 153 us per 1000 generations).
// nblocks = the launch's chain-running workgroups (they are the first of the grid); blocks beyond them keep their index.
constexpr int DEMCZ_XCDS = 8;
__device__ __forceinline__ int xcd_block(int nblocks)
{
    const int b = (int)blockIdx.x;
    if (b >= nblocks || nblocks % DEMCZ_XCDS != 0) return b;
    return (b % DEMCZ_XCDS) * (nblocks / DEMCZ_XCDS) + b / DEMCZ_XCDS;
}
__device__ __forceinline__ int xcd_block(const WindowParams& P) { return xcd_block(P.consumer_blocks > 0 ? (int)P.consumer_blocks : (int)gridDim.x); }

// Chains of this wave whose log_obj changed in a generation: a vector compare straight into a lane mask (the wavefront
// ballot), AND the mask of the lanes that speak for a chain (wave-uniform, computed once), popcount.
__device__ __forceinline__ unsigned int wave_count_changed(double lp_after, double lp_before, unsigned long long speak64)
{
    const double df = lp_after - lp_before;          // NaN (from +-Inf - +-Inf, or NaN) counts, as in Julia's diff(.) .!= 0
    const unsigned long long m = __builtin_amdgcn_fcmp(df, 0.0, 14 /* UNE: unordered or not equal */);
    return (unsigned int)__builtin_popcountll(m & speak64);
}
__device__ __forceinline__ void wave_store_counts(const WindowParams& P, int64_t wave, unsigned int total, unsigned int first)
{
    if (P.acc_out && (threadIdx.x & 63) == 0) {
        P.acc_out[2 * wave] = total;
        P.acc_out[2 * wave + 1] = first;
    }
}

// One archive row (16-byte aligned) <-> registers, as 16-byte accesses.
template <int D>
__device__ __forceinline__ void load_row(const double* __restrict__ row, double (&v)[D])
{
    const double2* r2 = reinterpret_cast<const double2*>(row);
#pragma unroll
    for (int k = 0; k < D / 2; ++k) {
        double2 t = r2[k];
        v[2 * k] = t.x;
        v[2 * k + 1] = t.y;
    }
    if constexpr (D & 1) v[D - 1] = row[D - 1];
}

template <int D>
__device__ __forceinline__ void store_row(double* __restrict__ row, const double (&v)[D])
{
    double2* r2 = reinterpret_cast<double2*>(row);
#pragma unroll
    for (int k = 0; k < D / 2; ++k) r2[k] = make_double2(v[2 * k], v[2 * k + 1]);
    if constexpr (D & 1) row[D - 1] = v[D - 1];
}

// ------------------------------------------------------------------------------------------------
// Targets.  `X` is a callable j -> x_j; with a compile-time D every loop unrolls and x stays in
// registers, W / mu / the design row come in through scalar loads (wave-uniform addresses).
// Summation orders are part of the arithmetic spec.
// ------------------------------------------------------------------------------------------------
template <int TARGET, int D, class XF>
__device__ __forceinline__ double target_logp(const TargetParams& tp, int d, XF X)
{
    const int dd = (D > 0) ? D : d;
    if constexpr (TARGET == TARGET_MVNORMAL) {
        if (tp.ngrp > 1) {
            // grouped by the run's blocks: y_i = P_i0 + P_i1 + ..., q = Q_0 + Q_1 + ... (TargetParams; oracle: target_logp)
            if constexpr (D > 0) {
                // compile-time indices for x (it lives in registers); the group starts are a run-time bit mask.  Both forms of
                // every step are computed and one is selected -- three times the arithmetic of the plain order, on the one path
                // (one lane per chain, block updates) where it is rarely the bottleneck.
                double q = 0.0, Qg = 0.0;
                bool haveq = false;
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    const double* wrow = tp.Wp + (i * (i + 1)) / 2;
                    double y = 0.0, acc = 0.0;
                    bool have = false;
#pragma unroll
                    for (int j = 0; j <= i; ++j) {
                        const bool st = (tp.gstart >> j) & 1ull;
                        const double rr = X(j) - tp.mu[j];
                        const double ysum = have ? y + acc : acc;
                        if (st && j > 0) { y = ysum; have = true; }
                        const double prod = wrow[j] * rr, fm = fma(wrow[j], rr, acc);
                        acc = st ? prod : fm;
                    }
                    y = have ? y + acc : acc;
                    const bool sti = (tp.gstart >> i) & 1ull;
                    const double qsum = haveq ? q + Qg : Qg;
                    if (sti && i > 0) { q = qsum; haveq = true; }
                    const double sq = y * y, fq = fma(y, y, Qg);
                    Qg = sti ? sq : fq;
                }
                q = haveq ? q + Qg : Qg;
                return fma(-0.5, q, tp.c0);
            } else {
                double q = 0.0;
                for (int g = 0; g < tp.ngrp; ++g) {
                    const int lo = tp.goff[g], hi = tp.goff[g + 1];
                    double Qg = 0.0;
                    for (int i = lo; i < hi; ++i) {
                        const double* wrow = tp.Wp + (i * (i + 1)) / 2;
                        double y = 0.0;
                        for (int gb = 0; gb <= g; ++gb) {
                            const int jl = tp.goff[gb];
                            int jh = tp.goff[gb + 1] - 1;
                            jh = (jh > i) ? i : jh;
                            double Pv = wrow[jl] * (X(jl) - tp.mu[jl]);
                            for (int j = jl + 1; j <= jh; ++j) Pv = fma(wrow[j], X(j) - tp.mu[j], Pv);
                            y = (gb == 0) ? Pv : y + Pv;
                        }
                        Qg = (i == lo) ? y * y : fma(y, y, Qg);
                    }
                    q = (g == 0) ? Qg : q + Qg;
                }
                return fma(-0.5, q, tp.c0);
            }
        }
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < dd; ++i) {
            const double* wrow = tp.Wp + (i * (i + 1)) / 2;
            double acc = wrow[0] * (X(0) - tp.mu[0]);
#pragma unroll
            for (int j = 1; j <= i; ++j) acc = fma(wrow[j], X(j) - tp.mu[j], acc);
            q = (i == 0) ? acc * acc : fma(acc, acc, q);
        }
        return fma(-0.5, q, tp.c0);
    } else if constexpr (TARGET == TARGET_ISO_QUAD) {
        double q = 0.0;
#pragma unroll
        for (int i = 0; i < dd; ++i) {
            double r = X(i) - tp.mu[i];
            q = (i == 0) ? r * r : fma(r, r, q);
        }
        return -q;
    } else {
        // 16 interleaved partial sums (observation o -> partial o mod 16), then the fixed tree
        // (l, l+8), (l, l+4), (l, l+2), (0, 1): the spec's order, shared with the 16-lane layout.
        double part[LINREG_PARTIALS];
        const int64_t nfull = tp.nobs / LINREG_PARTIALS;       // rounds in which every partial gets a term
#pragma unroll
        for (int l = 0; l < LINREG_PARTIALS; ++l) part[l] = 0.0;
        for (int64_t k = 0; k < nfull; ++k) {
#pragma unroll
            for (int l = 0; l < LINREG_PARTIALS; ++l) {
                const int64_t o = k * LINREG_PARTIALS + l;
                const double* row = tp.design + o * dd;
                double acc = row[0] * X(0);
#pragma unroll
                for (int j = 1; j < dd; ++j) acc = fma(row[j], X(j), acc);
                const double r = tp.yobs[o] - acc;
                part[l] = (k == 0) ? r * r : fma(r, r, part[l]);
            }
        }
#pragma unroll
        for (int l = 0; l < LINREG_PARTIALS; ++l) {
            const int64_t o = nfull * LINREG_PARTIALS + l;
            if (o < tp.nobs) {
                const double* row = tp.design + o * dd;
                double acc = row[0] * X(0);
#pragma unroll
                for (int j = 1; j < dd; ++j) acc = fma(row[j], X(j), acc);
                const double r = tp.yobs[o] - acc;
                part[l] = (nfull == 0) ? r * r : fma(r, r, part[l]);
            }
        }
#pragma unroll
        for (int h = LINREG_PARTIALS / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int l = 0; l < h; ++l) part[l] = part[l] + part[l + h];
        }
        return -0.5 * part[0];
    }
}

// ------------------------------------------------------------------------------------------------
// K1, compile-time D.  FULL: one block covering 0..D-1 in order (the default blockindex = [1:Npar],
// DEMC.jl:41) -- no LDS, no branches.  Otherwise blocks come from the CSR tables and the normals
// of a block-step are staged per lane in LDS (uniform runtime index).
// ------------------------------------------------------------------------------------------------
// A wave-uniform double held in scalar registers (two v_readfirstlane: the compiler then keeps it in SGPRs for the
// whole kernel instead of reloading it through the vector memory path every generation).
__device__ __forceinline__ double uniform_double(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b);
    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int TARGET, int D, bool FULL>
// (forcing 5/6/8 waves per SIMD through __launch_bounds__ spills and measured 0.72x/0.91x/0.81x: not used)
__global__ void __launch_bounds__(WINDOW_BS) window_kernel(const WindowParams P)
{
    __shared__ double zlds[FULL ? 1 : (D + 1) * WINDOW_BS];
    const int64_t c = (int64_t)blockIdx.x * WINDOW_BS + threadIdx.x;
    if (c >= P.N) return;
    const int tid = threadIdx.x;

    double x[D];
#pragma unroll
    for (int p = 0; p < D; ++p) x[p] = P.Xcur[c + P.N * p];
    double lp = P.lpcur[c];
    // target constants of the quadratic-form targets: wave-uniform, read once, kept in scalar registers (they used to be
    // re-read through the vector memory path every generation: the stores to the history may alias them as far as the
    // compiler knows)
    constexpr bool LOCAL_TARGET = (TARGET == TARGET_MVNORMAL && D <= 10) || TARGET == TARGET_ISO_QUAD;
    [[maybe_unused]] double muc[LOCAL_TARGET ? D : 1], Wc[(LOCAL_TARGET && TARGET == TARGET_MVNORMAL) ? D * (D + 1) / 2 : 1], epsc[LOCAL_TARGET ? D : 1];
    if constexpr (LOCAL_TARGET) {
#pragma unroll
        for (int p = 0; p < D; ++p) { muc[p] = uniform_double(P.tp.mu[p]); epsc[p] = uniform_double(P.eps[p]); }
        if constexpr (TARGET == TARGET_MVNORMAL) {
#pragma unroll
            for (int i = 0; i < D * (D + 1) / 2; ++i) Wc[i] = uniform_double(P.tp.Wp[i]);
        }
    }
    [[maybe_unused]] const double c0c = P.tp.c0;
    // the log-density of a point, from the local constants where they exist (the same operation sequence as target_logp)
    auto logp_of = [&](const double (&xq)[D]) -> double {
        if constexpr (!FULL && TARGET == TARGET_MVNORMAL) {
            if (P.tp.ngrp > 1) return target_logp<TARGET, D>(P.tp, D, [&](int j) { return xq[j]; });      // sums grouped by the blocks
        }
        if constexpr (LOCAL_TARGET && TARGET == TARGET_MVNORMAL) {
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                double acc = Wc[(i * (i + 1)) / 2] * (xq[0] - muc[0]);
#pragma unroll
                for (int j = 1; j <= i; ++j) acc = fma(Wc[(i * (i + 1)) / 2 + j], xq[j] - muc[j], acc);
                q = (i == 0) ? acc * acc : fma(acc, acc, q);
            }
            return fma(-0.5, q, c0c);
        } else if constexpr (LOCAL_TARGET) {
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const double rr = xq[i] - muc[i];
                q = (i == 0) ? rr * rr : fma(rr, rr, q);
            }
            return -q;
        } else {
            return target_logp<TARGET, D>(P.tp, D, [&](int j) { return xq[j]; });
        }
    };

    rng_state st;
    rng_seek(st, P.seed, (uint64_t)(P.chain_id0 + c), (uint64_t)(P.g_first - 1) * (uint64_t)P.S);
    int to_b = P.to_boundary;            // countdown to the next K boundary
    int64_t nb = 0;                      // boundaries passed inside this launch
    unsigned int cnt_total = 0, cnt_first = 0;
    const unsigned long long speak64 = __builtin_amdgcn_ballot_w64(true);      // every lane still here runs a chain

    for (int gi = 0; gi < P.ngen; ++gi) {
        const int nblocks = FULL ? 1 : P.Nblocks;
        const double lp_gen0 = lp;
        for (int ib = 0; ib < nblocks; ++ib) {
            uint64_t r1, r2, i1, i2;
            rng_next(st, r1, r2);
            draw_rows(r1, r2, (uint64_t)P.M, i1, i2);
            double xp[D];
            if constexpr (FULL) {
                // the two archive rows are asked for as soon as their indices exist: the gather's latency runs beside
                // the ~600 instructions of the normals and log u below instead of in front of the proposal
                double za[D], zb[D];
                load_row<D>(P.Z + (int64_t)i1 * P.ZS, za);
                load_row<D>(P.Z + (int64_t)i2 * P.ZS, zb);
                __builtin_amdgcn_sched_barrier(0);
                double zn[(D == 1) ? 2 : ((D + 1) / 2) * 2];
                constexpr int NPAIRS = (D == 1) ? 1 : (D + 1) / 2;
#pragma unroll
                for (int pr = 0; pr < NPAIRS; ++pr) {
                    rng_next(st, r1, r2);
                    normal_pair(r1, r2, zn[2 * pr], zn[2 * pr + 1]);
                }
                const double scale = (D == 1) ? P.gamma : P.gamma / sqrt((double)(2 * D));
#pragma unroll
                for (int p = 0; p < D; ++p) {
                    double diff = za[p] - zb[p];
                    double t1 = scale * diff;
                    double t2 = (LOCAL_TARGET ? epsc[p] : P.eps[p]) * zn[(D == 1) ? 0 : p];
                    double delta = t1 + t2;
                    xp[p] = x[p] + delta;
                }
            } else {
                const int b = P.block_offsets[ib + 1] - P.block_offsets[ib];
                const int nn = (b == 1) ? 1 : b;
                const int npairs = (nn + 1) / 2;
                for (int pr = 0; pr < npairs; ++pr) {
                    rng_next(st, r1, r2);
                    double z0, z1;
                    normal_pair(r1, r2, z0, z1);
                    zlds[(2 * pr) * WINDOW_BS + tid] = z0;
                    zlds[(2 * pr + 1) * WINDOW_BS + tid] = z1;
                }
                const double scale = (b == 1) ? P.gamma : P.gamma / sqrt((double)(2 * b));
                const int32_t* so = P.slot_of + ib * D;
                const double* za = P.Z + (int64_t)i1 * P.ZS;
                const double* zb = P.Z + (int64_t)i2 * P.ZS;
#pragma unroll
                for (int p = 0; p < D; ++p) {
                    const int t = so[p];
                    if (t >= 0) {
                        double diff = za[p] - zb[p];
                        double zt = zlds[((b == 1) ? 0 : t) * WINDOW_BS + tid];
                        double t1 = scale * diff;
                        double t2 = P.eps[p] * zt;
                        double delta = t1 + t2;
                        xp[p] = x[p] + delta;
                    } else {
                        xp[p] = x[p];
                    }
                }
            }
            rng_next(st, r1, r2);
            const double logu = dm_log(u_open(r1));
            const double lpp = logp_of(xp);
            double dlt = lpp - lp;
            if (P.temperature) dlt = dlt / P.temperature[gi];
            const bool acc = logu < dlt;
#pragma unroll
            for (int p = 0; p < D; ++p) x[p] = acc ? xp[p] : x[p];
            lp = acc ? lpp : lp;
        }
        {
            const unsigned int k = wave_count_changed(lp, lp_gen0, speak64);
            cnt_total += k;
            cnt_first = (gi == 0) ? k : cnt_first;
        }
        const int64_t slot = P.slot_first + gi;
        if (P.chain) {
#pragma unroll
            for (int p = 0; p < D; ++p) P.chain[c + P.N * (p + (int64_t)D * slot)] = x[p];
            P.logobj[c + P.N * slot] = lp;
        }
        if (--to_b == 0) {                  // generation divisible by K: runchain!'s append, demcz.jl:88-91
            to_b = P.K;
            if (P.do_append) store_row<D>(P.Zw + (P.M_append + nb * P.N + c) * P.ZS, x);
            if (P.snap) {
#pragma unroll
                for (int p = 0; p < D; ++p) P.snap[nb * P.N * D + c + P.N * p] = x[p];
            }
            ++nb;
        }
    }
#pragma unroll
    for (int p = 0; p < D; ++p) P.Xcur[c + P.N * p] = x[p];
    P.lpcur[c] = lp;
    wave_store_counts(P, blockIdx.x, cnt_total, cnt_first);
}

// ------------------------------------------------------------------------------------------------
// K1, runtime d (any d <= MAX_D, any block structure): x, xprop and the normals live in LDS,
// one column per lane (conflict-free: address = index * WINDOW_BS + lane).
// ------------------------------------------------------------------------------------------------
template <int TARGET>
__global__ void __launch_bounds__(WINDOW_BS) window_kernel_generic(const WindowParams P)
{
    extern __shared__ double lds[];
    const int d = P.d;
    double* xs = lds;                         // d x BS
    double* xps = lds + d * WINDOW_BS;        // d x BS
    double* zs = lds + 2 * d * WINDOW_BS;     // (d+1) x BS
    const int64_t c = (int64_t)blockIdx.x * WINDOW_BS + threadIdx.x;
    if (c >= P.N) return;
    const int tid = threadIdx.x;

    for (int p = 0; p < d; ++p) xs[p * WINDOW_BS + tid] = P.Xcur[c + P.N * p];
    double lp = P.lpcur[c];
    rng_state st;
    rng_seek(st, P.seed, (uint64_t)(P.chain_id0 + c), (uint64_t)(P.g_first - 1) * (uint64_t)P.S);
    int to_b = P.to_boundary;            // countdown to the next K boundary
    int64_t nb = 0;                      // boundaries passed inside this launch
    unsigned int cnt_total = 0, cnt_first = 0;
    const unsigned long long speak64 = __builtin_amdgcn_ballot_w64(true);      // every lane still here runs a chain

    for (int gi = 0; gi < P.ngen; ++gi) {
        const double lp_gen0 = lp;
        for (int ib = 0; ib < P.Nblocks; ++ib) {
            uint64_t r1, r2, i1, i2;
            rng_next(st, r1, r2);
            draw_rows(r1, r2, (uint64_t)P.M, i1, i2);
            const int b = P.block_offsets[ib + 1] - P.block_offsets[ib];
            const int nn = (b == 1) ? 1 : b;
            const int npairs = (nn + 1) / 2;
            for (int pr = 0; pr < npairs; ++pr) {
                rng_next(st, r1, r2);
                double z0, z1;
                normal_pair(r1, r2, z0, z1);
                zs[(2 * pr) * WINDOW_BS + tid] = z0;
                zs[(2 * pr + 1) * WINDOW_BS + tid] = z1;
            }
            const double scale = (b == 1) ? P.gamma : P.gamma / sqrt((double)(2 * b));
            const int32_t* so = P.slot_of + ib * d;
            const double* za = P.Z + (int64_t)i1 * P.ZS;
            const double* zb = P.Z + (int64_t)i2 * P.ZS;
            for (int p = 0; p < d; ++p) {
                const int t = so[p];
                double xv = xs[p * WINDOW_BS + tid];
                if (t >= 0) {
                    double diff = za[p] - zb[p];
                    double zt = zs[((b == 1) ? 0 : t) * WINDOW_BS + tid];
                    double t1 = scale * diff;
                    double t2 = P.eps[p] * zt;
                    double delta = t1 + t2;
                    xv = xv + delta;
                }
                xps[p * WINDOW_BS + tid] = xv;
            }
            rng_next(st, r1, r2);
            const double logu = dm_log(u_open(r1));
            const double lpp = target_logp<TARGET, 0>(P.tp, d, [&](int j) { return xps[j * WINDOW_BS + tid]; });
            double dlt = lpp - lp;
            if (P.temperature) dlt = dlt / P.temperature[gi];
            if (logu < dlt) {
                for (int p = 0; p < d; ++p) xs[p * WINDOW_BS + tid] = xps[p * WINDOW_BS + tid];
                lp = lpp;
            }
        }
        {
            const unsigned int k = wave_count_changed(lp, lp_gen0, speak64);
            cnt_total += k;
            cnt_first = (gi == 0) ? k : cnt_first;
        }
        const int64_t slot = P.slot_first + gi;
        if (P.chain) {
            for (int p = 0; p < d; ++p) P.chain[c + P.N * (p + (int64_t)d * slot)] = xs[p * WINDOW_BS + tid];
            P.logobj[c + P.N * slot] = lp;
        }
        if (--to_b == 0) {
            to_b = P.K;
            for (int p = 0; p < d; ++p) {
                const double xv = xs[p * WINDOW_BS + tid];
                if (P.do_append) P.Zw[(P.M_append + nb * P.N + c) * P.ZS + p] = xv;
                if (P.snap) P.snap[nb * P.N * d + c + P.N * p] = xv;
            }
            ++nb;
        }
    }
    for (int p = 0; p < d; ++p) P.Xcur[c + P.N * p] = xs[p * WINDOW_BS + tid];
    P.lpcur[c] = lp;
    wave_store_counts(P, blockIdx.x, cnt_total, cnt_first);
}

// K6: log-density of n points X (n x d, ld ldX) -> out.  Used for the initial log_objcurrent.
template <int TARGET>
__global__ void logp_kernel(TargetParams tp, int d, const double* X, int64_t ldX, int64_t n, double* out)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    out[c] = target_logp<TARGET, 0>(tp, d, [&](int j) { return X[c + ldX * j]; });
}

// ---- the small non-template kernels (archive bookkeeping, R-hat, diagnostics): defined in ONE translation unit, demcz_capi.hip;
// the units that only instantiate window kernels (demcz_pw_inst_<g>.hip) define DEMCZ_NO_AUX_KERNELS ----
#ifndef DEMCZ_NO_AUX_KERNELS
// The part of the archive no row has been appended to yet holds `v` (LIVE launches of the split layout,
// demcz_kernels_pc.h, recognise an unpublished row by it; nothing below row M ever leaves the device).
__global__ void fill_u64_kernel(unsigned long long* p, size_t n, unsigned long long v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// K4: rows (nrows x d column-major, ld ldrows) -> archive rows M .. M+nrows (row-major, stride ZS)
__global__ void append_rows_kernel(double* Z, int64_t ZS, int64_t M, const double* rows, int64_t nrows,
                                   int64_t ldrows, int d)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * d) return;
    const int64_t r = i % nrows, p = i / nrows;
    Z[(M + r) * ZS + p] = rows[r + ldrows * p];
}

// archive rows r0 .. r0+nrows (row-major) -> out (nrows x d column-major, ld ldout)
__global__ void export_rows_kernel(const double* Z, int64_t ZS, int64_t r0, double* out, int64_t nrows, int64_t ldout, int d)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * d) return;
    const int64_t r = i % nrows, p = i / nrows;
    out[r + ldout * p] = Z[(r0 + r) * ZS + p];
}

// Staging slab of an all-gather over R shards, each shard an n_loc x d column-major matrix
// stored contiguously ([R][d][n_loc]) -> archive rows M + r*n_loc + j.
__global__ void append_gathered_kernel(double* Z, int64_t ZS, int64_t M, const double* slab, int64_t n_loc,
                                       int R, int d)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = n_loc * d;
    if (i >= per * R) return;
    const int64_t r = i / per, rem = i % per, p = rem / n_loc, j = rem % n_loc;
    Z[(M + r * n_loc + j) * ZS + p] = slab[i];
}

// All-gather slab of a BATCH of cnt boundaries: [R][cnt][d][n_loc] -> archive rows
// base + (s*R + r)*n_loc + j (boundary-major, then rank: the order an unsharded run appends in).
__global__ void append_batch_kernel(double* Z, int64_t ZS, int64_t base, const double* slab, int64_t n_loc, int R, int cnt, int d)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = n_loc * d;
    if (i >= per * cnt * R) return;
    const int64_t r = i / (per * cnt), rem = i % (per * cnt), sidx = rem / per, rem2 = rem % per, p = rem2 / n_loc, j = rem2 % n_loc;
    Z[(base + (sidx * R + r) * n_loc + j) * ZS + p] = slab[i];
}

// ------------------------------------------------------------------------------------------------
// K5: split-R-hat (src/utils.jl:2-20).  Window of w generations starting at history slot s0,
// n = floor(w/2); split-chain (h, c) covers slots s0 + h n .. s0 + h n + n - 1.
//   rhat_moments_kernel: per (chain, parameter, half, time-chunk) shifted sums
//       S1 = sum (x - x0), S2 = sum (x - x0)^2, x0 = first sample of the half (same shift for
//       every chunk of a half, so chunk partials simply add).
//   rhat_chainstats_kernel: mean_j = x0 + S1/n, s_j^2 = (S2 - S1^2/n)/(n-1) per split-chain.
//   rhat_reduce_kernel stage 0: sum_j mean_j per parameter; stage 1 (given the grand mean):
//       sum_j (mean_j - grand)^2 and sum_j s_j^2.  Fixed-order tree reductions: deterministic.
// ------------------------------------------------------------------------------------------------
__global__ void rhat_moments_kernel(const double* chain, int64_t N, int d, int64_t s0, int64_t n, int nchunk,
                                    double* S1, double* S2)
{
    // grid: x over N*d (chain fastest), y = h * nchunk + chunk
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * d) return;
    const int h = blockIdx.y / nchunk, ck = blockIdx.y % nchunk;
    const int64_t per = (n + nchunk - 1) / nchunk;
    const int64_t t0 = ck * per, t1 = (t0 + per < n) ? t0 + per : n;
    const int64_t stride = N * d;
    const double* base = chain + i + stride * (s0 + h * n);
    const double x0 = base[0];
    double a = 0.0, b = 0.0;
    // (eight loads in flight per thread, added in generation order: a thread's samples are N*d doubles apart, every one a
    //  trip to HBM -- one at a time the loop is that latency times its length)
    int64_t t = t0;
    for (; t + 8 <= t1; t += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[stride * (t + u)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double w = v[u] - x0;
            a += w;
            b = fma(w, w, b);
        }
    }
    for (; t < t1; ++t) {
        double v = base[stride * t] - x0;
        a += v;
        b = fma(v, v, b);
    }
    const int64_t o = i + stride * blockIdx.y;
    S1[o] = a;
    S2[o] = b;
}

__global__ void rhat_chainstats_kernel(const double* chain, int64_t N, int d, int64_t s0, int64_t n, int nchunk,
                                       const double* S1, const double* S2, double* mean_j, double* s2_j)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * d) return;
    const int h = blockIdx.y;
    const int64_t stride = N * d;
    const double x0 = chain[i + stride * (s0 + h * n)];
    double a = 0.0, b = 0.0;
    for (int ck = 0; ck < nchunk; ++ck) {
        a += S1[i + stride * (h * nchunk + ck)];
        b += S2[i + stride * (h * nchunk + ck)];
    }
    const double nn = (double)n;
    mean_j[i + stride * h] = x0 + a / nn;
    s2_j[i + stride * h] = (b - a * a / nn) / (nn - 1.0);
}

// one workgroup of 256 per parameter; out[p] (stage 0) or out[p], out[d + p] (stage 1)
// (stage 1 takes the summed means and divides by `grand_div` = m itself, so no host round trip is
// needed between the stages)
__global__ void __launch_bounds__(256) rhat_reduce_kernel(const double* mean_j, const double* s2_j, int64_t N, int d,
                                                          int stage, const double* grand, double grand_div, double* out)
{
    __shared__ double ra[256], rb[256];
    const int p = blockIdx.x;
    const int64_t stride = N * d;
    double a = 0.0, b = 0.0;
    const double gm = stage ? grand[p] / grand_div : 0.0;
    // k = threadIdx.x, + 256, ...: (h, c) = (k / N, k % N) kept by carrying (a 64-bit division per element was most of the kernel)
    int64_t h = 0, c = threadIdx.x;
    while (c >= N) { c -= N; ++h; }
    for (int64_t k = threadIdx.x; k < 2 * N; k += 256) {
        const int64_t at = c + N * p + stride * h;
        const double mj = mean_j[at];
        if (stage == 0) {
            a += mj;
        } else {
            const double dv = mj - gm;
            a = fma(dv, dv, a);
            b += s2_j[at];
        }
        c += 256;
        while (c >= N) { c -= N; ++h; }
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            ra[threadIdx.x] += ra[threadIdx.x + s];
            rb[threadIdx.x] += rb[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[p] = ra[0];
        if (stage) out[d + p] = rb[0];
    }
}

// Unsharded runs: both reduce stages and utils.jl:13-18 for one parameter in one workgroup -- the same
// per-thread accumulation and tree order as rhat_reduce_kernel (bit-identical), two launches fewer per check.
__global__ void __launch_bounds__(256) rhat_tail_kernel(const double* mean_j, const double* s2_j, int64_t N, int d,
                                                        double n, double m, double* rhat)
{
    __shared__ double ra[256], rb[256];
    const int p = blockIdx.x;
    const int64_t stride = N * d;
    double a = 0.0, b = 0.0;
    int64_t h0 = 0, c0 = threadIdx.x;              // (h, c) = (k / N, k % N) of the thread's first element
    while (c0 >= N) { c0 -= N; ++h0; }
    {
        int64_t h = h0, c = c0;
        for (int64_t k = threadIdx.x; k < 2 * N; k += 256) {
            a += mean_j[c + N * p + stride * h];
            c += 256;
            while (c >= N) { c -= N; ++h; }
        }
    }
    ra[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) ra[threadIdx.x] += ra[threadIdx.x + s];
        __syncthreads();
    }
    const double gm = ra[0] / m;
    __syncthreads();
    a = 0.0;
    {
        int64_t h = h0, c = c0;
        for (int64_t k = threadIdx.x; k < 2 * N; k += 256) {
            const int64_t at = c + N * p + stride * h;
            const double dv = mean_j[at] - gm;
            a = fma(dv, dv, a);
            b += s2_j[at];
            c += 256;
            while (c >= N) { c -= N; ++h; }
        }
    }
    ra[threadIdx.x] = a;
    rb[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            ra[threadIdx.x] += ra[threadIdx.x + s];
            rb[threadIdx.x] += rb[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double B = n / (m - 1.0) * ra[0];                   // utils.jl:13
        const double W = rb[0] / m;                                // utils.jl:15
        const double varhat = (n - 1.0) / n * W + B / n;           // utils.jl:16
        rhat[p] = sqrt(varhat / W);                                // utils.jl:18
    }
}

// utils.jl:13-18 from the reduced sums: in[p] = sum_j (mean_j - grand)^2, in[d+p] = sum_j s_j^2
__global__ void rhat_final_kernel(const double* in, int d, double n, double m, double* rhat)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d) return;
    const double B = n / (m - 1.0) * in[p];                     // utils.jl:13
    const double W = in[d + p] / m;                              // utils.jl:15
    const double varhat = (n - 1.0) / n * W + B / n;             // utils.jl:16
    rhat[p] = sqrt(varhat / W);                                  // utils.jl:18
}

// K7c: changed[s - s_from] = number of chains whose log_obj in history slot s differs from the slot
// before it (slot 0: from lp_origin, the log_obj the chains had when the history window opened) --
// the event sum(diff(log_obj, dims=2) .!= 0) counts (demcz.jl:42, demcz_anneal.jl:50).  Computed
// from the history on demand: a per-generation atomic inside the chain-update loop would put
// N/8 atomics per generation on one cache line (measured: the dominant cost of the window kernel).
__global__ void __launch_bounds__(256) changed_from_history_kernel(const double* logobj, const double* lp_origin, int64_t N,
                                                                   int64_t s_from, long long* out)
{
    __shared__ int cnt[256];
    const int64_t s = s_from + blockIdx.x;
    const double* cur = logobj + N * s;
    const double* prev = (s == 0) ? lp_origin : logobj + N * (s - 1);
    int k = 0;
    for (int64_t c = threadIdx.x; c < N; c += 256) k += ((cur[c] - prev[c]) != 0.0) ? 1 : 0;   // diff(.) .!= 0: a NaN difference counts
    cnt[threadIdx.x] = k;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) cnt[threadIdx.x] += cnt[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = cnt[0];
}

// K7a: per-chain count of generations in slots s0+1 .. s0+w-1 whose log_obj differs from the previous slot
// (src/utils.jl:61).  Grid: x over chains (64 per workgroup), y over time chunks -- the work scales with N x w;
// integer partial counts are added with atomics (order-free, exact), changed_ratio_kernel divides.
__global__ void __launch_bounds__(64) changed_per_chain_kernel(const double* logobj, int64_t N, int64_t s0, int64_t w, int nchunk,
                                                               unsigned int* cnt)
{
    const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= N) return;
    const int64_t per = (w - 1 + nchunk - 1) / nchunk;                 // differences t = 1 .. w-1
    const int64_t t0 = 1 + (int64_t)blockIdx.y * per, t1 = (t0 + per < w) ? t0 + per : w;
    if (t0 >= t1) return;
    unsigned int k = 0;
    double prev = logobj[c + N * (s0 + t0 - 1)];
    for (int64_t t = t0; t < t1; ++t) {
        const double cur = logobj[c + N * (s0 + t)];
        k += ((cur - prev) != 0.0) ? 1u : 0u;      // diff(.) .!= 0 (utils.jl:61): a NaN difference counts
        prev = cur;
    }
    atomicAdd(&cnt[c], k);
}

__global__ void changed_ratio_kernel(const unsigned int* cnt, int64_t N, int64_t w, double* ratio)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < N) ratio[c] = (double)cnt[c] / (double)(w - 1);
}

// K7b: sums for mean_cov_chain (src/utils.jl:96-111), shifted by ref[p] = the first sample of parameter p so that
// the one-pass covariance keeps its digits.  The parameters are cut into tiles of MC_TS; a workgroup takes 256 chains x
// one chunk of generations x one tile pair (I <= J) and every thread accumulates the MC_TS x MC_TS products (and, on
// the diagonal, the MC_TS sums) of its chain in registers: the history is read ceil(d/8)+1 times instead of d+1, by
// (N/256) x chunks x pairs workgroups instead of d(d+1) in all.  Workgroup partials are combined by meancov_final_kernel
// in a fixed order (deterministic).
constexpr int MC_TS = 8;
constexpr int MC_VALS = MC_TS * MC_TS + MC_TS;

__global__ void __launch_bounds__(256) meancov_partial_kernel(const double* chain, int64_t N, int d, int64_t s0, int64_t w, int nchunk,
                                                              double* partial)
{
    __shared__ double red[4][MC_VALS];
    // tile pair of this workgroup: blockIdx.z enumerates (I, J), I <= J
    const int T = (d + MC_TS - 1) / MC_TS;
    int I = 0, J = (int)blockIdx.z;
    while (J >= T - I) { J -= T - I; ++I; }
    J += I;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (w + nchunk - 1) / nchunk;
    const int64_t t0 = (int64_t)blockIdx.y * per, t1 = (t0 + per < w) ? t0 + per : w;
    const int64_t stride = N * d;
    double acc[MC_TS][MC_TS], sm[MC_TS];
#pragma unroll
    for (int a = 0; a < MC_TS; ++a) {
        sm[a] = 0.0;
#pragma unroll
        for (int b = 0; b < MC_TS; ++b) acc[a][b] = 0.0;
    }
    if (c < N) {
        double ri[MC_TS], rj[MC_TS];
#pragma unroll
        for (int a = 0; a < MC_TS; ++a) {
            ri[a] = (I * MC_TS + a < d) ? chain[N * (I * MC_TS + a) + stride * s0] : 0.0;
            rj[a] = (J * MC_TS + a < d) ? chain[N * (J * MC_TS + a) + stride * s0] : 0.0;
        }
        for (int64_t t = t0; t < t1; ++t) {
            const double* base = chain + c + stride * (s0 + t);
            double xi[MC_TS], xj[MC_TS];
#pragma unroll
            for (int a = 0; a < MC_TS; ++a) {
                xi[a] = (I * MC_TS + a < d) ? base[N * (I * MC_TS + a)] - ri[a] : 0.0;
                xj[a] = (J * MC_TS + a < d) ? base[N * (J * MC_TS + a)] - rj[a] : 0.0;
            }
#pragma unroll
            for (int a = 0; a < MC_TS; ++a) {
                sm[a] += xi[a];
#pragma unroll
                for (int b = 0; b < MC_TS; ++b) acc[a][b] = fma(xi[a], xj[b], acc[a][b]);
            }
        }
    }
    // wave reduction of every accumulator, then the four waves through LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < MC_TS; ++a) {
#pragma unroll
        for (int b = 0; b <= MC_TS; ++b) {
            double v = (b < MC_TS) ? acc[a][b] : sm[a];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) red[wv][(b < MC_TS) ? a * MC_TS + b : MC_TS * MC_TS + a] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < MC_VALS) {
        const double v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[blk * MC_VALS + threadIdx.x] = v;
    }
}

// out[p + d*q] = sum (x_p - ref_p)(x_q - ref_q) for q < d, out[p + d*d] = sum (x_p - ref_p): one thread per entry,
// partials added in workgroup order
__global__ void meancov_final_kernel(const double* partial, int d, int nblk_xy, double* out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= d * (d + 1)) return;
    const int p = e % d, q = e / d;
    const int T = (d + MC_TS - 1) / MC_TS;
    int lo = p, hi = (q < d) ? q : p;
    bool swap = false;
    if (q < d && q / MC_TS < p / MC_TS) { lo = q; hi = p; swap = true; }
    const int I = lo / MC_TS, J = (q < d) ? hi / MC_TS : I;
    int z = 0;
    for (int i = 0; i < I; ++i) z += T - i;
    z += J - I;
    int idx;
    if (q == d) idx = MC_TS * MC_TS + p % MC_TS;
    else idx = swap ? (q % MC_TS) * MC_TS + p % MC_TS : (p % MC_TS) * MC_TS + q % MC_TS;
    double a = 0.0;
    for (int b = 0; b < nblk_xy; ++b) a += partial[((size_t)z * nblk_xy + b) * MC_VALS + idx];
    out[e] = a;
}

// Self-test of the draw pipeline: for block index blk0 + i of chain `chain`, the two raw words,
// the Box-Muller pair and log(u_open(r1)).
__global__ void selftest_draws_kernel(uint64_t seed, uint64_t chain, uint64_t blk0, int n, uint64_t* words,
                                      double* normals, double* logu)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rng_state st;
    rng_seek(st, seed, chain, blk0 + (uint64_t)i);
    uint64_t r1, r2;
    rng_next(st, r1, r2);
    words[2 * i] = r1;
    words[2 * i + 1] = r2;
    double z0, z1;
    normal_pair(r1, r2, z0, z1);
    normals[2 * i] = z0;
    normals[2 * i + 1] = z1;
    logu[i] = dm_log(u_open(r1));
}
#endif  // DEMCZ_NO_AUX_KERNELS

}  // namespace demcz
