// demcz_kernels_rec.h -- what the split layouts share (demcz_kernels_pc.h: eight replicated lanes per
// chain; demcz_kernels_ml.h, REC variant: L cooperating lanes per chain):
//   * the draw records and the producer half of a launch that fills them for the NEXT launch;
//   * the in-launch hand-off of appended archive rows (LIVE launches);
//   * Philox as a pure counter -> block map, and the one-wave LDS hand-off.
#pragma once

#include "demcz_kernels.h"

#pragma clang fp contract(off)

namespace demcz {

// rocRAND's Philox4x32-10 round function used as a pure counter -> block map:
// block(seed, chain, blk) == rocrand_init(seed, chain, 4*blk) followed by rocrand4().
struct philox_blocks : rocrand_device::philox4x32_10_engine {
    __device__ __forceinline__ philox_blocks() {}
    __device__ __forceinline__ void block(uint64_t seed, uint64_t chain, uint64_t blk, uint64_t& r1, uint64_t& r2)
    {
        uint4 ctr = {(unsigned)blk, (unsigned)(blk >> 32), (unsigned)chain, (unsigned)(chain >> 32)};
        uint2 key = {(unsigned)seed, (unsigned)(seed >> 32)};
        uint4 w = this->ten_rounds(ctr, key);
        r1 = (uint64_t)w.x | ((uint64_t)w.y << 32);
        r2 = (uint64_t)w.z | ((uint64_t)w.w << 32);
    }
};

// LDS hand-offs between lanes of ONE wave: the hardware keeps a wave's LDS operations in order, so
// all that is needed is to stop the compiler from moving memory operations across the hand-off.
__device__ __forceinline__ void wave_lds_handoff()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// record layout: rec[(f * N + c) * GS + g], f = 0..D-1 normals, D log u, D+1 the two row indices packed
// as 32-bit halves (the split layout is only selected while the archive has < 2^32 rows); GS =
// WindowParams::rec_stride generations.  The generations of one (field, chain) are contiguous: a
// consumer lane fetches a chunk of ten with five 16-byte loads off one address.
__device__ __forceinline__ size_t rec_index(int64_t N, int64_t GS, int g, int f, int64_t c) { return ((size_t)f * (size_t)N + (size_t)c) * (size_t)GS + (size_t)g; }
// record-major layout (WindowParams::rec_fields = F > 0): rec[(g * N + c) * F + f]
__device__ __forceinline__ size_t rec_index_rm(int64_t N, int F, int g, int f, int64_t c) { return ((size_t)g * (size_t)N + (size_t)c) * (size_t)F + (size_t)f; }

// What the 64 lanes of a producer unit are.  Chains: unit = (generation, role, block of 64 chains) -- every lane's store goes to
// a (field, chain) row of its own, rec_stride * 8 bytes from its neighbour's: 64 cache lines per store instruction, each line
// completed by 16 different units at 16 different times (the producer's WRITE_SIZE was twice its bytes and the stores, not the
// arithmetic, were most of its 95 us at C2).  Generations: unit = (chain, role, block of 64 generations) -- a lane per
// generation, one 512-byte run per store.  The records are the same doubles at the same addresses either way (counter-based
// draws: block = generation * S + role of chain c's stream), so consumers do not care; launches shorter than 64 generations
// (one K-window of a sharded run) keep the chain mapping, which fills its lanes.
__host__ __device__ inline bool produce_by_generation(int32_t rec_fields, int32_t ngen) { return rec_fields == 0 && ngen >= 64; }
__host__ __device__ inline int64_t produce_units(int64_t N, int roles, int32_t rec_fields, int32_t ngen)
{
    return produce_by_generation(rec_fields, ngen) ? N * roles * (int64_t)((ngen + 63) / 64) : ((N + 63) / 64) * roles * (int64_t)ngen;
}

// (`lane`: position in the 64-wide producer unit -- a workgroup of one wave, or one wave of a larger workgroup)
template <int D>
__device__ __forceinline__ void pc_produce(const WindowParams& P, int64_t pb, int lane = -1)
{
    if (lane < 0) lane = (int)threadIdx.x;
    constexpr int NPAIRS = (D == 1) ? 1 : (D + 1) / 2;
    constexpr int S = NPAIRS + 2;                          // Philox blocks of a generation = producer roles
    int64_t c;
    int role, gi;
    if (produce_by_generation(P.rec_fields, P.next_ngen)) {
        const int64_t gblocks = (P.next_ngen + 63) / 64;
        const int64_t per_chain = (int64_t)S * gblocks;
        c = pb / per_chain;                                // wave-uniform, like role
        const int64_t rem = pb % per_chain;
        role = (int)(rem / gblocks);
        gi = (int)(rem % gblocks) * 64 + lane;
    } else {
        const int64_t nbc = (P.N + 63) / 64;               // units per (generation, role) plane
        const int64_t plane = pb / nbc;                    // wave-uniform
        c = (pb % nbc) * 64 + lane;
        role = (int)(plane % S);
        gi = (int)(plane / S);
    }
    if (gi >= P.next_ngen || c >= P.N) return;
    philox_blocks rng;
    uint64_t r1, r2;
    rng.block(P.seed, (uint64_t)(P.chain_id0 + c), (uint64_t)(P.next_g_first + gi - 1) * (uint64_t)S + (uint64_t)role, r1, r2);
    double* rec = P.rec_out;
    auto at = [&](int g, int f) -> size_t {
        return P.rec_fields ? rec_index_rm(P.N, P.rec_fields, g, f, c) : rec_index(P.N, P.rec_stride, g, f, c);
    };
    if (role == 0) {
        // rows generation gi of the next launch draws from: those it starts with plus, where appended rows
        // are visible at once, N per K boundary it has passed by then (update_demcz_chain_block, demcz.jl:176-179)
        const int64_t Mg = P.next_M + (int64_t)((gi + P.next_boff) / P.K) * P.next_rows;
        uint64_t i1, i2;
        draw_rows(r1, r2, (uint64_t)Mg, i1, i2);
        rec[at(gi, D + 1)] = __longlong_as_double((long long)(i1 | (i2 << 32)));
    } else {
        const double lg = dm_log(u_open(r1));
        if (role == S - 1) {
            rec[at(gi, D)] = lg;
        } else {
            const double R = sqrt(-2.0 * lg);
            double cs, sn;
            dm_sincos2pi(r2 >> 11, cs, sn);
            const int p0 = (D == 1) ? 0 : 2 * (role - 1);
            rec[at(gi, p0)] = R * cs;
            if (p0 + 1 < D) rec[at(gi, p0 + 1)] = R * sn;
        }
    }
}

// Block-structured runs (DEMCopt.Nblocks > 1, update_blocks demcz.jl:167-172): the record of a generation is
// one 16-byte entry per Philox block s of its S blocks -- the two row indices (as 64-bit integers), a
// Box-Muller pair, or log u -- exactly what the cooperating lanes of the fused kernel hand each other
// through LDS.  rec2[(g * N + c) * S + s].
__device__ __forceinline__ void pcb_produce(const WindowParams& P, int64_t pb)
{
    const int64_t nbc = (P.N + 63) / 64;
    const int64_t plane = pb / nbc;                        // wave-uniform
    const int64_t c = (pb % nbc) * 64 + (threadIdx.x & 63);
    const int s = (int)(plane % P.S), gi = (int)(plane / P.S);
    if (gi >= P.next_ngen || c >= P.N) return;
    const int role = P.slot_role[s];
    philox_blocks rng;
    uint64_t r1, r2;
    rng.block(P.seed, (uint64_t)(P.chain_id0 + c), (uint64_t)(P.next_g_first + gi - 1) * (uint64_t)P.S + (uint64_t)s, r1, r2);
    double2 e;
    if (role == 0) {
        const int64_t Mg = P.next_M + (int64_t)((gi + P.next_boff) / P.K) * P.next_rows;
        uint64_t i1, i2;
        draw_rows(r1, r2, (uint64_t)Mg, i1, i2);
        e.x = __longlong_as_double((long long)i1);
        e.y = __longlong_as_double((long long)i2);
    } else {
        const double lg = dm_log(u_open(r1));
        if (role == 2) {
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
    // record-major: the S entries of (generation, chain) are contiguous -- the consumer's lanes of a chain read neighbouring
    // entries at the same generation
    reinterpret_cast<double2*>(P.rec_out)[((size_t)gi * (size_t)P.N + (size_t)c) * (size_t)P.S + (size_t)s] = e;
}

// LIVE launches (single GPU, the reference's immediate visibility): one launch runs through several
// K boundaries.  The rows a boundary appends are drawn from in the very next generation, by any chain,
// so waves hand rows to each other INSIDE the launch, and they do it through the data itself:
//   * the unwritten part of the archive holds a sentinel (a signalling-NaN pattern no arithmetic
//     produces: results have the quiet bit set);
//   * a wave appends its chains' rows with write-through (sc1) 8-byte stores;
//   * an archive gather that returns the sentinel is repeated as an sc1 load (served past the CU's L1 and
//     the XCD's L2) until the row is there -- each double is its own naturally aligned 8-byte granule
//     and is written once, so nothing needs ordering, flags, fences or a grid barrier
//     (MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores + sc1 loads, data-tagged granules).
//     The FIRST read of a row takes the ordinary cached path (sc1 loads are slow to issue, demcz_kernels_pc.h):
//     a stale cached copy can only show the sentinel where the final double is not yet seen, never a wrong value.
// A wave only ever waits for rows of an EARLIER boundary than the one it is working towards, so the
// waits cannot form a cycle; all consumer workgroups are single waves and co-resident (N <= 8192:
// at most 1024 of them on 256 CUs).  A bounded spin turns a lost row into an error word, not a hang.  The one wait
// without a poll limit -- a chain wave waiting for room in the LDS ring its publisher wave empties -- is bounded by the
// publisher, which waits for nothing but its own stores and never leaves before its chain waves have.
constexpr unsigned long long LIVE_SENTINEL = 0xFFF4DEADC0DE5EEDull;
constexpr int LIVE_SPIN_LIMIT = 1 << 18;      // ~0.1-0.3 s of polling (the default of WindowParams::live_spin_limit)

__device__ __forceinline__ double live_load(const double* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void live_store(double* p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool is_sentinel(double v) { return (unsigned long long)__double_as_longlong(v) == LIVE_SENTINEL; }
// The lanes whose value is the sentinel, as a lane MASK in scalar registers (one v_cmp_eq_u64 straight into an SGPR pair).  Round 5:
// the consumers' per-step test "did any lane read an unpublished row" used to OR per-lane bools and feed them to a ballot, which the
// compiler lowers to v_cndmask + v_cmp on a temporary VGPR -- taken, at that point of a pass, from the destinations of LDS reads still
// in flight: an s_waitcnt lgkmcnt(0) in every pass / block-step.  Masks OR-ed in scalar registers and one scalar test cost none of
// that: window_kernel_ps2's LIVE launch 115-117 -> 108-109 us per 1000 generations, C3's block-step 39.6 -> 37.1 us per K-window.
__device__ __forceinline__ unsigned long long sentinel_lanes(double v)
{
    return __builtin_amdgcn_uicmpl((unsigned long long)__double_as_longlong(v), LIVE_SENTINEL, 32 /* ICMP_EQ */);
}

// Replicated archives (a sharded run: every rank -- a GPU of the node, or for rehearsal another handle on the same GPU -- holds
// the whole archive and runs its own shard of the chains).  The hand-off above needs nothing new on the READING side: a wave
// polls its OWN replica for the sentinel to go away.  The WRITING side stores element p of row `M_append + b * brows + row_off +
// cl` -- boundary b of the launch, local chain cl -- into its own replica and into every peer's: one more write-through store
// per peer, at system scope (sc0 sc1: past this GPU's caches, over xGMI into the peer's memory).  Each double is still its own
// naturally aligned granule written once per replica, so still no flags, fences or ordering; the role the reference gives to
// its shared archive (`Zshared`, one append per chain under pmap, src/demcz.jl:88-91, 137) without its race.
__device__ __forceinline__ void live_store_sys(double* p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void live_publish(const WindowParams& P, int64_t b, int64_t cl, int p, double v)
{
    const int64_t off = (P.M_append + b * P.brows + P.row_off + cl) * P.ZS + p;
    live_store(P.Zw + off, v);
    for (int r = 0; r < P.n_peers; ++r) live_store_sys(P.peer_Z[r] + off, v);
}
// The publisher's wait in front of a round of stores: none for their acknowledgements -- nothing needs them (readers look at
// the data; the launch's end waits for everything) -- only that the number of stores in flight stays within what the wave's
// counter can count (at most 63): up to 40 outstanding before a round of n_peers + 1 stores per lane (pw: two rounds).  With
// peers an acknowledgement crosses xGMI (microseconds): waiting for it would hold the publisher -- and through its ring the
// chain waves -- to one boundary per link round trip.  On one GPU rounds 2-3 waited for ALL earlier stores ("the rows that arrive
// during their flight then leave together in one instruction"); measured in round 4 (scripts/ab_pubwait.sh,
// profiles/r04n_publisher_wait.txt): without the wait C2's launch is 140.1-141.4 us against 142.6-143.8, the other configs
// unchanged within noise -- a row no longer waits behind the acknowledgement of its workgroup neighbours' rows.
__device__ __forceinline__ void publisher_wait(const WindowParams&)
{
#ifdef DEMCZ_PUB_WAIT          // the A/B's other side only (scripts/ab_pubwait.sh): rounds 2-3's wait for all earlier stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
#endif
}
// a re-read of a row that showed the sentinel: past the caches; rows a PEER writes are asked for at system scope
__device__ __forceinline__ double live_reload(const WindowParams& P, const double* p)
{
    if (P.n_peers > 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return live_load(p);
}


// One bounded poll step of a LIVE wait, shared by the consumers: returns true when the wave must give
// up -- its own poll limit, or another wave's (live_err[0]; looked at every 256 polls so that a failed
// launch drains within microseconds).  The first lane to time out records what it was waiting for:
// [1] generation of the launch, [2] archive row, [3] workgroup.
__device__ __forceinline__ bool live_poll_abandon(const WindowParams& P, int& spins, bool lane_waiting, unsigned row, int gi)
{
    const bool timeout = (++spins >= P.live_spin_limit);
    bool abandon = timeout;
    if (!timeout && (spins & 255) == 0)
        abandon = __hip_atomic_load(P.live_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    if (abandon && timeout && lane_waiting) {
        if (atomicCAS(P.live_err, 0u, 1u) == 0u) {
            P.live_err[1] = (unsigned)gi; P.live_err[2] = row; P.live_err[3] = blockIdx.x;
        }
    }
    return abandon;
}

}  // namespace demcz
