/*
 * demcz.h -- C ABI of libdemcz_hip.so, the MI355X (gfx950) DEMCz chain-update engine.
 *
 * The reference (chrished/DEMC.jl) is a pure-Julia package with no FFI of its own; the seam this
 * ABI occupies is the call the drivers make into runchain!:
 *     src/demcz.jl:30-33         for ig = 1:Ngeneration, for ic = 1:N  runchain!(ic, ig, ig, ...)
 *     src/demcz_anneal.jl:39-42  same loop, tempered
 * Everything below that call (runchain! :80-93, update_blocks :167-172,
 * update_demcz_chain_block :174-195, accept :197-203, and the annealer's overloads
 * demcz_anneal.jl:67-80,142-178) runs on the device; the autostop statistic
 * (Rhat_gelman, src/utils.jl:2-20, called at demcz.jl:41) and the acceptance counts
 * (demcz.jl:42, demcz_anneal.jl:50) are device reductions over the on-device history.
 * The Julia-side binding a maintainer would add is in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every function returns an int32 status (0 = OK);
 *     no exceptions or longjmp cross the boundary; demcz_last_error() gives the message.
 *   - All matrices are COLUMN-MAJOR exactly as Julia lays them out (DEMC.jl:10-15), so a
 *     Matrix{Float64} / Array{Float64,3} is passed by pointer without transposition:
 *       X, Xcurrent   N x d        element (ic, ip)      at ic + N*ip
 *       Z             M x d        element (row, ip)     at row + ldZ*ip   ("parameter-major")
 *       chain         N x d x G    element (ic, ip, ig)  at ic + N*(ip + d*ig)
 *       log_obj       N x G        element (ic, ig)      at ic + N*ig
 *   - Indices are 0-based on this side; generations are 1-based like the reference's `ig`
 *     because `ig % K == 0` (demcz.jl:88) and the temperature schedule depend on the value.
 *   - Host pointers unless a name ends in `_device`.  The caller owns every host buffer; the
 *     library owns device memory.
 *   - A handle is bound to one device and one stream and is not thread-safe.
 *   - Work is enqueued asynchronously on the handle's stream; functions that return data to
 *     host memory synchronise that stream first.
 */
#ifndef DEMCZ_H
#define DEMCZ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct demcz_handle demcz_handle;

enum demcz_status {
    DEMCZ_OK = 0,
    DEMCZ_ERR_INVALID_ARGUMENT = 1,
    DEMCZ_ERR_HIP = 2,            /* a HIP runtime call failed (message has hipGetErrorString) */
    DEMCZ_ERR_CAPACITY = 3,       /* Z row capacity or history capacity exceeded               */
    DEMCZ_ERR_STATE = 4,          /* call order: e.g. run before set_state                      */
    DEMCZ_ERR_NO_DEVICE = 5,      /* no gfx950 device visible: there is NO CPU fallback         */
    DEMCZ_ERR_COMM = 6            /* sharded run: an RCCL call failed, RCCL reported an asynchronous error, or a wait on the
                                     exchange made no progress within the deadline (demcz_set_comm_timeout) -- a peer rank is
                                     dead or stalled.  Both communicators have been aborted; the handle only accepts
                                     demcz_destroy / demcz_last_error from now on.  Restart the job in fresh processes.  */
};

/* User log-densities the device can evaluate (the reference takes an arbitrary Julia closure,
 * demcz.jl:189; the BASELINE configs use these three -- SURVEY.md 8(a) a11). */
enum demcz_target_kind {
    DEMCZ_TARGET_MVNORMAL = 0,    /* logpdf(MvNormal(mu, Sigma), x)   test/example_normpdf.jl:13-16 */
    DEMCZ_TARGET_ISO_QUAD = 1,    /* -sum((x - mu).^2)                test/test_anneal.jl:10        */
    DEMCZ_TARGET_LINREG_SSE = 2,  /* -0.5*sum((y - X*b).^2)           test/example_linreg.jl:32     */
    DEMCZ_TARGET_HOST_CALLBACK = 3 /* arbitrary closure on the host: demcz_propose/accept_commit    */
};

/* Mirrors the fields of DEMCopt the hot path reads (src/DEMC.jl:24-39: N, K, Nblocks,
 * blockindex, eps_scale) plus what a device engine needs.  POD; copied by demcz_create. */
typedef struct demcz_config {
    int64_t N;                    /* chains on THIS handle (DEMCopt.N, or the local shard of it)   */
    int64_t chain_id0;            /* global id of local chain 0 (0 when not sharded): the chain's
                                     RNG stream is Philox subsequence chain_id0 + ic, so results do
                                     not depend on how chains are sharded over GPUs                */
    int32_t d;                    /* Npar                                                           */
    int32_t K;                    /* DEMCopt.K: every K-th generation the current states join Z     */
    int64_t Mcap;                 /* row capacity of Z: M0 + ceil(N_total*Ngeneration/K), demcz.jl:11 */
    int64_t Gcap;                 /* generations of chain/log_obj history kept on the device        */
    int32_t Nblocks;              /* DEMCopt.Nblocks                                                */
    const int32_t* block_offsets; /* CSR over DEMCopt.blockindex: Nblocks+1 offsets ...             */
    const int32_t* block_indices; /* ... into 0-based parameter indices                             */
    const double* eps_scale;      /* DEMCopt.eps_scale, d values                                    */
    uint64_t seed;                /* Philox4x32-10 key (rocRAND-compatible stream layout)           */
    int32_t device_id;            /* HIP device ordinal                                             */
    int32_t target_kind;          /* enum demcz_target_kind                                         */
    const double* mu;             /* MVNORMAL / ISO_QUAD: d                                         */
    const double* W;              /* MVNORMAL: d x d column-major lower-triangular inv(chol(Sigma)) */
    double c0;                    /* MVNORMAL: -0.5*(d*log(2pi) + logdet(Sigma))                    */
    const double* design;         /* LINREG_SSE: nobs x d column-major                              */
    const double* yobs;           /* LINREG_SSE: nobs                                               */
    int64_t nobs;
    void* stream;                 /* hipStream_t to enqueue on, or NULL: the library makes its own  */
    int32_t lanes_per_chain;      /* kernel layout: 0 = let the library choose; 1 = one lane per chain,
                                     fused (throughput layout, large N); 8 / 16 = that many lanes
                                     cooperate on a chain; DEMCZ_LAYOUT_SPLIT = producer workgroups
                                     make the state-independent draws one launch ahead, consumer
                                     lanes run the chains, and on one GPU a launch runs through many
                                     K boundaries (small N); DEMCZ_LAYOUT_SPLIT_WAVE = the same with
                                     one wavefront per chain that resolves five generations per pass
                                     (smallest N; MvNormal at every d in 2..32, the isotropic quadratic
                                     at d in 6..32).  The regression target: d = 10 on the FP64 matrix
                                     instruction (split layouts), any other d in 2..28 on sixteen
                                     lanes per chain with helper waves.  Results are bit-identical. */
    int32_t reserved0;
} demcz_config;

#define DEMCZ_LAYOUT_SPLIT 100
#define DEMCZ_LAYOUT_SPLIT_WAVE 164

/* Version of this header's ABI; demcz_abi_version() must return the same number. */
#define DEMCZ_ABI_VERSION 1
int32_t demcz_abi_version(void);

/* Create / destroy.  Replaces the driver's setup, demcz.jl:10-12 and :24 (allocation of the
 * padded Z, M, and the MC history arrays) -- here as device buffers sized from cfg. */
int32_t demcz_create(demcz_handle** out, const demcz_config* cfg);
int32_t demcz_destroy(demcz_handle* h);
const char* demcz_last_error(const demcz_handle* h);   /* h may be NULL: last create error */

/* Upload the start state: demcz.jl:13-22.  X is N x d (ld N); logp is N values or NULL to have
 * the device evaluate the target at X (demcz.jl:17); Z is M0 x d with leading dimension ldZ;
 * 2 <= M0 <= Mcap (two distinct archive rows are needed, demcz.jl:176-179). */
int32_t demcz_set_state(demcz_handle* h, const double* X, const double* logp,
                        const double* Z, int64_t ldZ, int64_t M0);

/* Download the current state: mc.Xcurrent, mc.log_objcurrent (DEMC.jl:13-14), Z[1:M,:] and M
 * (demcz.jl:51).  Any pointer may be NULL.  Z is written with leading dimension ldZ >= M. */
int32_t demcz_get_state(demcz_handle* h, double* X, double* logp, double* Z, int64_t ldZ, int64_t* M);

/* Z[1:M,:] (demcz.jl:51) in a pinned host buffer of the library's pool, M x d column-major with leading dimension M: no page
 * faults of a fresh array under the copy (41 MB at the end of a C2 run).  The caller gives *Z back with demcz_release_host_buffer. */
int32_t demcz_get_archive_pinned(demcz_handle* h, double** Z, int64_t* M);

/* Map generation index -> history slot: generation g is stored at slot g - g0 - 1, so the
 * device keeps generations g0+1 .. g0+Gcap.  Default g0 = 0. */
int32_t demcz_set_history_origin(demcz_handle* h, int64_t g0);

/* Run generations g_from..g_to (inclusive, 1-based) for all N chains: the loop demcz.jl:30-33 /
 * demcz_anneal.jl:39-42 with runchain! and everything below it.  gamma is DEMCopt.gamma (a
 * per-call scalar so the annealer's adaptation, demcz_anneal.jl:48-57, can change it between
 * calls).  temperature is NULL for the sampler or g_to-g_from+1 host values (one per
 * generation, temperaturefun(ig, ...) of demcz_anneal.jl:69) for the tempered accept.
 * Asynchronous.  Chains are updated synchronously within a generation (all proposals of a
 * generation see the same M; the N rows are appended after it -- SURVEY.md Q2).
 * When the handle is one shard of a multi-GPU run (demcz_comm_init was called) the K-boundary
 * append all-gathers the shards' rows over RCCL so every replica of Z stays identical. */
int32_t demcz_run(demcz_handle* h, int64_t g_from, int64_t g_to, double gamma, const double* temperature);

int32_t demcz_synchronize(demcz_handle* h);

/* Copy history out: mc.chain[:, :, g_from:g_to] and mc.log_obj[:, g_from:g_to] (DEMC.jl:11-12).
 * Either pointer may be NULL. */
int32_t demcz_get_history(demcz_handle* h, int64_t g_from, int64_t g_to, double* chain, double* log_obj);

/* Streamed history.  mc.chain / mc.log_obj of a C2 run are half a gigabyte that only exists to be handed back (DEMC.jl:10-12);
 * copied after the run they cost four times what the run does.  With streaming enabled (before demcz_run) the library keeps
 * pinned host mirrors of both arrays -- same layout, generation g at slot g - g0 - 1 -- and every demcz_run call's generations
 * leave on a copy stream while the compute stream goes on with the next call.
 *   demcz_get_history_view   waits for the copies of g_from..g_to (redoing a voided LIVE slab first) and returns pointers INTO the
 *                            mirrors: N x d x G and N x G column-major, no further copy.  Valid until the handle is destroyed ...
 *   demcz_detach_history     ... unless the mirrors are detached: they then belong to the caller (a Julia Array made with
 *                            unsafe_wrap, a NumPy array over the pointer) until demcz_release_host_buffer gives each back.
 * The mirrors come from a process-wide pool of pinned buffers: a second run of the same shape allocates nothing. */
int32_t demcz_history_stream(demcz_handle* h, int32_t enabled);
int32_t demcz_get_history_view(demcz_handle* h, int64_t g_from, int64_t g_to, double** chain, double** log_obj);
int32_t demcz_detach_history(demcz_handle* h, void** chain_base, void** logobj_base);
int32_t demcz_release_host_buffer(void* base);

/* The library keeps the big buffers of destroyed handles (device: history + archive, at most 6 GiB; pinned host: history mirrors,
 * at most 3 GiB) for the next handle of the process.  demcz_pool_trim gives everything cached back to the runtime (it is also
 * what the library does by itself when one of its own allocations fails); either pointer may be NULL. */
int32_t demcz_pool_trim(int64_t* device_bytes_freed, int64_t* pinned_bytes_freed);

/* changed[g - g_from] = number of local chains whose log_obj at generation g differs from the
 * one before it -- the event counted by sum(diff(log_obj, dims=2) .!= 0) at demcz.jl:42 and
 * demcz_anneal.jl:50. */
int32_t demcz_get_changed(demcz_handle* h, int64_t g_from, int64_t g_to, int64_t* changed);

/* *total = sum over g in g_from..g_to of the changed[] counts above -- the acceptance count of the annealer's gamma
 * adaptation (demcz_anneal.jl:50) and of the autostop warning (demcz.jl:42).  The window kernels count it as they
 * go: one wavefront ballot of "this chain's log_obj changed" per generation, popcount into a scalar register, two
 * words per wave per launch (no atomics, no second pass over log_obj); when g_from..g_to is covered by whole
 * launches (it is, for the drivers' windows) the answer is the sum of those words and *from_ballots = 1; otherwise
 * it is counted from the history like demcz_get_changed (*from_ballots = 0).  from_ballots may be NULL.
 * Works on handles without a history window (Gcap = 0) in the first case. */
int32_t demcz_get_changed_total(demcz_handle* h, int64_t g_from, int64_t g_to, int64_t* total, int32_t* from_ballots);

/* Split-chain Gelman-Rubin statistic over generations g_from..g_to of the on-device history:
 * Rhat_gelman(chain[:, :, g_from:g_to], N, g_to-g_from+1, d), src/utils.jl:2-20, as called by
 * the autostop at demcz.jl:41.  rhat receives d values.  On a sharded handle the partial
 * moments are all-reduced so every rank gets the statistic over all N_total chains. */
int32_t demcz_rhat(demcz_handle* h, int64_t g_from, int64_t g_to, double* rhat);

/* The two reduction stages of demcz_rhat over the LOCAL chains only, for hosts that reduce
 * across shards themselves: stage 0 -> out[0..d) = sum_j mean_j over the 2N local split-chains;
 * stage 1 (grand = the global mean per parameter, d values) -> out[0..d) = sum_j (mean_j -
 * grand)^2, out[d..2d) = sum_j s_j^2.  The caller sums `out` over shards between the stages
 * and finishes with utils.jl:13-18. */
int32_t demcz_rhat_partial(demcz_handle* h, int64_t g_from, int64_t g_to, int32_t stage,
                           const double* grand, double* out);

/* Per-chain acceptance ratio over generations g_from..g_to:
 * sum(diff(log_obj, dims=2) .!= 0, dims=2) ./ (G-1), src/utils.jl:61.  ratio: N values. */
int32_t demcz_accept_ratio(demcz_handle* h, int64_t g_from, int64_t g_to, double* ratio);

/* mean_cov_chain over generations g_from..g_to, src/utils.jl:96-111 (1/(N*G) normalisation).
 * mean: d values, cov: d x d column-major. */
int32_t demcz_mean_cov(demcz_handle* h, int64_t g_from, int64_t g_to, double* mean, double* cov);

/* Host-closure mode (DEMCZ_TARGET_HOST_CALLBACK): one block-step of update_demcz_chain_block
 * split at the closure call demcz.jl:189.  demcz_propose draws (i1, i2, normals, log u) for
 * block ib of generation g and returns the N proposals (N x d, ld N); the caller evaluates its
 * closure; demcz_accept_commit applies demcz.jl:190-194 (tempered if temperature != NULL).
 * demcz_end_generation performs runchain!'s bookkeeping demcz.jl:84-91. */
int32_t demcz_propose(demcz_handle* h, int64_t g, int32_t ib, double gamma, double* Xprop);
/* The same round trip without its two copies and its two stream synchronisations (round 5).  demcz_closure_buffers hands out pinned
 * host buffers of the handle -- *Xprop: N x d (ld N), *logp: N -- that the kernels address directly: demcz_propose(h, g, ib, gamma,
 * NULL) returns as soon as the propose kernel has written the proposals INTO *Xprop (its last workgroup raises a flag word the host
 * spins on: no copy, no hipStreamSynchronize); the caller evaluates its closure on *Xprop, leaves the values in *logp and calls
 * demcz_accept_commit(h, NULL, temperature), which only ENQUEUES the commit (the kernel reads *logp from host memory) -- the next
 * demcz_propose goes into the stream behind it.  Do not write *logp between demcz_accept_commit and the return of the next
 * demcz_propose.  Own pointers may still be passed to either call: they are copied from / into the pinned buffers. */
int32_t demcz_closure_buffers(demcz_handle* h, double** Xprop, double** logp);
int32_t demcz_accept_commit(demcz_handle* h, const double* logp_prop, const double* temperature);
int32_t demcz_end_generation(demcz_handle* h, int64_t g);

/* Multi-GPU: one handle per rank (one process per GPU).  `unique_id` is the 128-byte
 * ncclUniqueId obtained from demcz_comm_unique_id on rank 0 and broadcast by the host
 * (torch.distributed / MPI / Distributed.jl -- plumbing).  After this call demcz_run appends
 * through ncclAllGather and demcz_rhat all-reduces its partial moments. */
int32_t demcz_comm_unique_id(void* unique_id_128B);
int32_t demcz_comm_init(demcz_handle* h, const void* unique_id_128B, int32_t nranks, int32_t rank);

/* Rows handed over INSIDE the launches (round 4).  After demcz_comm_init every rank's archive is opened by all other ranks
 * over HIP IPC (fine-grained device memory), and a rank's publisher waves store a boundary's rows into every replica themselves,
 * by write-through stores over xGMI; readers poll their own replica.  The all-gather + scatter per K-window is gone and a launch
 * runs through many K boundaries, as on one GPU: the role of the reference's shared archive under pmap (src/demcz.jl:88-91, 137),
 * without its race.  If IPC or peer access is refused on any rank, or a hand-off times out, the run falls back to the
 * ncclAllGather exchange (same results).  DEMCZ_NO_PEER=1 in the environment keeps the exchange from the start.
 *   demcz_get_peer_status   *mode = 0 exchange through RCCL (or unsharded), 1 replica group of this process, 2 IPC peers, 3 IPC peers
 *                           set up by the host (demcz_peer_export / demcz_peer_attach);
 *                           *peers = replicas this handle publishes into besides its own.
 *   demcz_peer_group        the same schedule for R handles of THIS process on ONE device (each with its own replica and shard:
 *                           chain_id0 = r * N), so that a one-GPU machine can run and test it: the members publish into each
 *                           other's replicas directly.  One host thread drives all members and gives every member the same
 *                           demcz_run calls before it asks any of them for results; demcz_run_checked is not available.
 *                           Destroying one member ends the group. */
int32_t demcz_peer_group(demcz_handle** handles, int32_t R);
int32_t demcz_get_peer_status(const demcz_handle* h, int32_t* mode, int32_t* peers);
/* The IPC set-up with the HOST carrying the 64-byte handles (torch.distributed, MPI, Distributed.jl ...) instead of RCCL -- for
 * hosts that shard without the library's communicator, and what lets two PROCESSES on one GPU test the cross-process half of the
 * path (mode 3 of demcz_get_peer_status).  demcz_peer_export moves the archive into fine-grained memory and returns its IPC
 * handle; the host all-gathers the handles; demcz_peer_attach(handles: nranks x 64 bytes in rank order) opens the others'.  The
 * host then owes two meetings of all ranks (barriers): after demcz_set_state / before the first demcz_run, and after the last
 * synchronising call / before demcz_destroy.  A hand-off that times out is DEMCZ_ERR_STATE on the rank that saw it: there is
 * no automatic redo in this mode. */
int32_t demcz_peer_export(demcz_handle* h, int32_t nranks, int32_t rank, void* handle_64B);
int32_t demcz_peer_attach(demcz_handle* h, const void* handles_64B_each);
/* The orderly end of that mode: an exported archive may only be freed once no other rank has it mapped.  After the host's
 * barrier behind the last synchronising call every rank calls demcz_peer_detach (closes its mappings of the others' archives;
 * results stay readable, demcz_run is refused from then on), the host makes the ranks meet once more, then demcz_destroy.
 * (With the library's own communicator, demcz_comm_init, demcz_destroy holds both meetings itself.) */
int32_t demcz_peer_detach(demcz_handle* h);
/* demcz_comm_init's first-contact check (round 5).  Before the in-launch hand-off is switched on, every rank's kernel stores a
 * token into every peer's archive allocation -- the store a published row uses, through the IPC mapping a row would travel --
 * while it polls its own for the peers' tokens (at most 200 ms, DEMCZ_PING_MS); the outcome and "can this rank issue LIVE
 * launches at all" are min-reduced over the ranks: either every rank hands rows over inside its launches or all of them keep
 * the ncclAllGather exchange.  *ok = 1 passed on all ranks, 0 failed somewhere (the exchange is in use), -1 not made (unsharded,
 * DEMCZ_NO_PEER, IPC refused); *wait_us = how long this rank's kernel waited for the last token (start skew + link latency). */
int32_t demcz_get_peer_ping(const demcz_handle* h, int32_t* ok, double* wait_us);

/* Deadline of every host-side wait of a sharded handle (a stream or event behind an RCCL collective): default 60 000 ms, or
 * the environment variable DEMCZ_COMM_TIMEOUT_MS at demcz_comm_init; 0 = wait for ever.  While it waits the library polls
 * ncclCommGetAsyncError about once a millisecond.  On expiry or on an asynchronous error: ncclCommAbort on the handle's
 * communicators and DEMCZ_ERR_COMM (the reference's counterpart: pmap at src/demcz.jl:137 throws when a worker dies). */
int32_t demcz_set_comm_timeout(demcz_handle* h, int64_t milliseconds);

/* Device-pointer access for hosts that do the exchange themselves (e.g. torch.distributed):
 * copy the current N x d states into caller device memory, and append `nrows` rows given as an
 * nrows x d column-major device matrix (ld = ldrows) to Z, bumping M. */
int32_t demcz_export_current_device(demcz_handle* h, double* X_device);
int32_t demcz_append_rows_device(demcz_handle* h, const double* rows_device, int64_t nrows, int64_t ldrows);
/* Host-pointer form of the append: rows is nrows x d column-major with leading dimension ldrows. */
int32_t demcz_append_rows(demcz_handle* h, const double* rows, int64_t nrows, int64_t ldrows);
/* When set (non-zero), demcz_run stops short of the K-boundary append and leaves it to the
 * caller (demcz_export_current_device + host collective + demcz_append_rows_device). */
int32_t demcz_set_external_append(demcz_handle* h, int32_t enabled);

/* Deferred visibility of appended rows.  E = 0 (default): the rows appended after generation j*K are
 * drawn from in generation j*K + 1 already -- the reference's schedule with all chains of a
 * generation updated against the same archive.  E >= 1: boundaries are grouped in batches of E
 * (batch of boundary j closes at J = ceil(j/E)*E) and a batch's rows are drawn from generation
 * (J + E)*K + 1 on.  On a sharded handle the batch travels in ONE all-gather on a side stream while
 * the next E windows compute, so the latency-bound collective is off the critical path; on an
 * unsharded handle the same rule is applied so results do not depend on the sharding.  Any past
 * state is a legitimate DEMCz archive row (ter Braak & Vrugt 2008); this changes WHEN a row becomes
 * eligible, not the target distribution.  Call after demcz_comm_init, before demcz_run. */
int32_t demcz_set_append_lag(demcz_handle* h, int32_t batches);

/* Resume: the chains' Philox streams have already been advanced by `generations` generations (the
 * length of a previous run, demcz.jl:18-22): generation g of this handle draws what generation
 * generations+g of an uninterrupted run would draw.  K boundaries still follow this handle's own g,
 * like the reference's restarted `ig`. */
int32_t demcz_set_rng_offset(demcz_handle* h, int64_t generations);

/* Stateless diagnostics on caller (host) arrays, for code that post-processes returned histories
 * the way the reference's examples do (test/example_normpdf.jl:35-47): the array is uploaded, reduced
 * on the device, and nothing is kept.
 *   demcz_rhat_array          Rhat_gelman(chain, N, G, d)              src/utils.jl:2-20
 *   demcz_accept_ratio_array  sum(diff(log_obj,dims=2).!=0,dims=2)./(G-1)   src/utils.jl:61
 *   demcz_mean_cov_array      mean_cov_chain(chain, N, G, d)           src/utils.jl:96-111 */
int32_t demcz_rhat_array(int32_t device_id, const double* chain, int64_t N, int32_t d, int64_t G, double* rhat);
int32_t demcz_accept_ratio_array(int32_t device_id, const double* log_obj, int64_t N, int64_t G, double* ratio);
int32_t demcz_mean_cov_array(int32_t device_id, const double* chain, int64_t N, int32_t d, int64_t G, double* mean, double* cov);

/* Introspection for benchmarks and tests. */
int32_t demcz_get_info(const demcz_handle* h, int64_t* M, int64_t* launches_window, int32_t* lanes_per_chain);

/* The driver loop with its autostop test as ONE call (demcz.jl:30-55): generations g_from..g_to, and after
 * every generation g that is a multiple of `every` the split-R-hat of generations (g-every, g]
 * (Rhat_gelman, utils.jl:2-20; demcz.jl:41).  The maximum over parameters of the i-th check goes to
 * rhat_max[i] (i < n_max; may be NULL), the whole vector of the last check to rhat_last[d] (may be NULL).
 * threshold > 0: returns as soon as a check's maximum is below it, with *g_stop = that generation
 * (demcz.jl:43-52; the state is as if nothing after it had run -- the library may run the next slab ahead of the
 * decision and discards it); otherwise *g_stop = g_to.  *n_checks = checks made.
 * Needs a history window (Gcap) that holds the generations of the call.  In a sharded run every rank
 * makes the same call; the decision is the same on all of them.                                      */
int32_t demcz_run_checked(demcz_handle* h, int64_t g_from, int64_t g_to, double gamma, const double* temperature,
                          int64_t every, double threshold, int64_t* g_stop, int32_t* n_checks,
                          double* rhat_max, int32_t n_max, double* rhat_last);

/* ------------------------------------------------------------------------------------------------------
 * DIAGNOSTIC entry points -- NOT part of the drop-in contract; a binding of the reference
 * surface needs none of them.  They exist for bench.py (kernel timing), for tests (self-test of the draw
 * arithmetic, forcing the LIVE hand-off's time-out path) and for integrators who want to look inside.
 * ------------------------------------------------------------------------------------------------------ */

/* Timing of the window kernels on the stream they are launched on: while enabled, every demcz_run call
 * brackets its back-to-back window launches with one HIP event pair (an event between two launches would
 * stall the stream being measured).  demcz_get_kernel_time synchronises, returns the number of window
 * launches covered and the summed duration of the brackets, and clears the record.                     */
int32_t demcz_set_kernel_timing(demcz_handle* h, int32_t enabled);
int32_t demcz_get_kernel_time(demcz_handle* h, int64_t* launches, double* milliseconds);

/* The brackets the last demcz_get_kernel_time call summed up, one per demcz_run call (inside demcz_run_checked: one per slab):
 * start_ms[i] = start of bracket i on the device clock, relative to the first bracket's start; duration_ms[i] = its length.
 * start_ms[i+1] - start_ms[i] is therefore the wall time of step i as the GPU saw it, gaps between launches included -- what
 * bench.py takes its median step time from.  At most `cap` entries are written; *n = entries available. */
int32_t demcz_get_kernel_time_series(demcz_handle* h, int32_t cap, double* start_ms, double* duration_ms, int32_t* n);

/* Device self-test of the draw pipeline: for Philox block blk0+i of the
 * stream of global chain `chain`, words[2i..2i+1] = the two raw 64-bit words,
 * normals[2i..2i+1] = the Box-Muller pair, logu[i] = log(u_open(word 0)).  Lets an integrator
 * check the device arithmetic against any host implementation of the spec, bit for bit. */
int32_t demcz_selftest_draws(int32_t device_id, uint64_t seed, uint64_t chain, uint64_t blk0, int32_t n,
                             uint64_t* words, double* normals, double* logu);

/* LIVE launches (split layout on one GPU: a launch runs through many K boundaries and its waves hand the appended
 * rows to each other through the archive itself).  A wave that polls `polls` times for a row without seeing it
 * gives up; the library then redoes everything since the last verified point with one launch per K-window
 * (results are bit-identical either way).  polls = 0 restores the default (2^18).
 * demcz_get_live_status: *live_enabled = 1 while the handle issues LIVE launches, *redos = times it had to
 * fall back.  Tests lower the limit to 1 to walk the fall-back path.
 * Re-arming (round 5).  A time-out is not for ever: the redo runs one launch per K-window up to and including the demcz_run
 * call (inside demcz_run_checked: the slab) that holds the generation whose row never arrived, and the first call behind it
 * issues LIVE launches again -- in a sharded run on every rank together (the ranks agree on the point through the reduced error
 * words, and on "can go LIVE again" through one more reduction).  A handle does this at most `n` times in its life (default 3;
 * DEMCZ_LIVE_REARMS in the environment at demcz_create; demcz_set_live_rearms), after which a failure leaves it at one launch
 * per K-window, the mode that cannot fail.  demcz_get_live_rearms: *rearms = times it went LIVE again, *left = what remains. */
int32_t demcz_set_live_spin_limit(demcz_handle* h, int32_t polls);
int32_t demcz_get_live_status(const demcz_handle* h, int32_t* live_enabled, int32_t* redos);
int32_t demcz_set_live_rearms(demcz_handle* h, int32_t n);
int32_t demcz_get_live_rearms(const demcz_handle* h, int32_t* rearms, int32_t* left);
/* Diagnostic: window launches so far by the kernel that took them (three values): counts[0] window_kernel_ps2 (the regular
 * launches of the wave-per-chain layout), [1] that layout's general kernels (window_kernel_ps / _pw), [2] launches of every
 * other layout.  Tests use it to know which kernel a parity case exercised. */
int32_t demcz_debug_kernel_counts(const demcz_handle* h, int64_t* counts);
/* Diagnostic: the name (template arguments included, as a profiler prints it) of the window kernel the handle's most recent
 * window launch ran, NUL-terminated into buf[cap].  bench.py labels its roofline objects with it. */
int32_t demcz_debug_kernel_name(const demcz_handle* h, char* buf, int32_t cap);

/* Fault injection for the LIVE hand-off: LIVE launches whose first generation is >= g_from use the poll limit `polls` instead
 * of the handle's (demcz_set_live_spin_limit), so that a test can make a hand-off fail LATE in a long call (e.g. in slab 280 of
 * a 300-slab demcz_run_checked).  polls = 0 switches it off; a fault that has fired (the redo it caused has rolled back) switches
 * itself off, so that the redo's own LIVE launches run undisturbed.  polls = -1: such a launch finds the error word already set -- as if
 * a wave had timed out before the others became resident (another process on the GPU) -- so every wave leaves at once; the redo
 * snapshot buffers are filled with NaN patterns beforehand, so a wave that left without writing its row of the snapshot shows. */
int32_t demcz_debug_set_live_fault(demcz_handle* h, int32_t polls, int64_t g_from);

/* Fault injection for the exchange: the next collective of this (sharded) handle is held back on its stream for `milliseconds`
 * (at most 10 000; a one-thread kernel that watches a host flag and the clock, so it always ends) -- to a waiting rank exactly
 * what a stalled peer looks like.  Lets a one-GPU box walk the deadline -> abort -> DEMCZ_ERR_COMM path. */
int32_t demcz_debug_stall_exchange(demcz_handle* h, int32_t milliseconds);

/* The scatter step of the sharded K-boundary exchange on caller data: `slab` (host) has the layout an all-gather
 * over R ranks delivers, [R][cnt][d][n_loc] doubles with n_loc = the handle's N; its R*cnt*n_loc rows are appended
 * in the order an unsharded run appends them (boundary, then rank, then chain).  batched = 0: the kernel of the
 * synchronous exchange (cnt must be 1); 1: the kernel of the batched one.  Lets a one-GPU machine check the R > 1
 * index arithmetic that otherwise only runs behind ncclAllGather on R GPUs. */
int32_t demcz_debug_append_slab(demcz_handle* h, const double* slab, int32_t R, int32_t cnt, int32_t batched);

#ifdef __cplusplus
}
#endif
#endif /* DEMCZ_H */
