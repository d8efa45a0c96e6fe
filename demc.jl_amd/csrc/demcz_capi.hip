// demcz_capi.hip -- the C ABI of include/demcz.h over the kernels of demcz_kernels.h.
// Host side of the seam src/demcz.jl:30-33 / src/demcz_anneal.jl:39-42 (see the header).
// There is no CPU fallback: without a HIP device every entry point fails with
// DEMCZ_ERR_NO_DEVICE.
#include "../../include/demcz.h"
#include "demcz_kernels.h"
#include "demcz_kernels_ml.h"
#include "demcz_kernels_pc.h"
#include "demcz_kernels_lr.h"
#include "demcz_kernels_ps.h"
#include "demcz_kernels_ps2.h"
#include "demcz_kernels_ps2d.h"
#include "demcz_pw_dispatch.h"
#include "demcz_mlr_dispatch.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <thread>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

using namespace demcz;

static thread_local std::string g_create_error;

// ---- buffer pools ---------------------------------------------------------------------------------------------------------
// The reference's demcz_sample allocates its result arrays per call (demcz.jl:24); behind this ABI that is a device history, an
// archive and -- for the streamed history -- pinned host mirrors of half a gigabyte at C2, per call.  hipMalloc / hipFree /
// hipHostMalloc of that size cost milliseconds each (and hipFree synchronises the device), so buffers a destroyed handle gives
// back are kept (up to a cap) for the next handle of the process: a second demcz_sample call allocates nothing.
namespace {
struct BufPool {
    struct Entry { void* p; size_t bytes; int device; };
    std::mutex mu;
    std::vector<Entry> free_list;
    size_t cached = 0;
    const size_t cap;
    const bool host;
    BufPool(size_t cap_, bool host_) : cap(cap_), host(host_) {}
    hipError_t acquire(void** out, size_t bytes, int device)
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            int best = -1;
            for (int i = 0; i < (int)free_list.size(); ++i) {
                const Entry& e = free_list[i];
                if (e.device == device && e.bytes >= bytes && e.bytes <= bytes + bytes / 4 + (1u << 20) && (best < 0 || e.bytes < free_list[best].bytes)) best = i;
            }
            if (best >= 0) {
                *out = free_list[best].p;
                cached -= free_list[best].bytes;
                sizes[*out] = free_list[best].bytes;
                free_list.erase(free_list.begin() + best);
                return hipSuccess;
            }
        }
        hipError_t e = host ? hipHostMalloc(out, bytes, hipHostMallocDefault) : hipMalloc(out, bytes);
        if (e != hipSuccess) {
            // what this pool keeps for later may be exactly what is missing now (a handle of another shape, another library of
            // the process): give everything cached back to the runtime and ask once more
            (void)hipGetLastError();
            if (trim() > 0) e = host ? hipHostMalloc(out, bytes, hipHostMallocDefault) : hipMalloc(out, bytes);
        }
        if (e == hipSuccess) { std::lock_guard<std::mutex> lk(mu); sizes[*out] = bytes; }
        return e;
    }
    // frees every cached buffer; returns the bytes given back
    size_t trim()
    {
        std::vector<Entry> drop;
        size_t n = 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            drop.swap(free_list);
            n = cached;
            cached = 0;
        }
        for (const Entry& e : drop) { if (host) (void)hipHostFree(e.p); else (void)hipFree(e.p); }
        return n;
    }
    // a buffer something may still be writing (an undrained stream, an aborted communicator's kernels): never cached -- hipFree /
    // hipHostFree wait for the device
    void discard(void* p)
    {
        if (!p) return;
        { std::lock_guard<std::mutex> lk(mu); sizes.erase(p); }
        if (host) (void)hipHostFree(p); else (void)hipFree(p);
    }
    void release(void* p, int device)
    {
        if (!p) return;
        size_t bytes = 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = sizes.find(p);
            if (it != sizes.end()) { bytes = it->second; sizes.erase(it); }
            if (bytes && cached + bytes <= cap && !getenv("DEMCZ_NO_POOL")) {
                free_list.push_back({p, bytes, device});
                cached += bytes;
                return;
            }
        }
        if (host) (void)hipHostFree(p); else (void)hipFree(p);
    }
    std::unordered_map<void*, size_t> sizes;
};
BufPool g_dev_pool(6ull << 30, false);      // device buffers (history, archive arena): at most 6 GiB kept
BufPool g_host_pool(3ull << 30, true);      // pinned host mirrors of the history: at most 3 GiB kept
}  // namespace
// ... and so are its streams (creating and destroying five streams was 4-5 ms of an end-to-end C2 call)
static std::mutex g_stream_mu;
static std::vector<std::pair<int, hipStream_t>> g_stream_pool;
static hipError_t stream_acquire(int device, hipStream_t* out)
{
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        for (size_t i = 0; i < g_stream_pool.size(); ++i)
            if (g_stream_pool[i].first == device) { *out = g_stream_pool[i].second; g_stream_pool.erase(g_stream_pool.begin() + (long)i); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
static void stream_release(int device, hipStream_t s, bool drained)
{
    if (!s) return;
    if (drained && !getenv("DEMCZ_NO_POOL")) {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        if (g_stream_pool.size() < 32) { g_stream_pool.emplace_back(device, s); return; }
    }
    (void)hipStreamDestroy(s);
}
// every device / pinned allocation of a handle goes through the pools (a pointer the pool does not know is simply freed)
static hipError_t dev_malloc(int device, void** out, size_t bytes) { return g_dev_pool.acquire(out, std::max<size_t>(bytes, 16), device); }
static hipError_t dev_free(int device, void* p) { g_dev_pool.release(p, device); return hipSuccess; }
static hipError_t host_malloc(void** out, size_t bytes) { return g_host_pool.acquire(out, std::max<size_t>(bytes, 16), -1); }
static hipError_t host_free(void* p) { g_host_pool.release(p, -1); return hipSuccess; }

struct demcz_handle;
// A replica group of one process (demcz_peer_group): R handles on one device, each with its own archive replica and shard of
// the chains, publishing boundary rows into each other's replicas from inside their launches.  One host thread drives them all.
struct PeerGroup {
    std::vector<demcz_handle*> members;
    bool failed = false;       // a hand-off timed out: calls are logged and executed in lockstep at the next verification ...
    int32_t rearms_left = 3;   // ... after which the group tries its in-launch hand-off again, this many times in its life
    bool lockstep_only = false;   // a layout without the hand-off, or streams that cannot run at the same time: never LIVE
    bool busy = false;         // inside group_verify / group_execute
    bool dead = false;         // a member was destroyed: the others only accept demcz_destroy
};

struct demcz_handle {
    demcz_config cfg{};
    int lanes = 1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // host copies of the small tables
    std::vector<int32_t> block_offsets, block_indices, slot_of;
    std::vector<double> eps;
    int64_t S = 0;            // Philox blocks per generation
    int64_t ZS = 0;           // archive row stride in doubles (row-major on the device)
    bool full_block = false;
    int ngrp = 1;             // MvNormal: groups the sums of the quadratic form are cut into (TargetParams::ngrp; 1 = not grouped)
    uint64_t gstart = 1ull;   // bit j: parameter j starts a group
    int mlb_qb = 0;           // > 0: the groups are equal blocks of this many parameters and the block kernel's incremental form is built for them
    // device buffers
    double* dZ = nullptr;
    double* dX = nullptr;
    double* dlp = nullptr;
    double* dchain = nullptr;
    double* dchain_alloc = nullptr;   // what the pool gave (dchain may sit a few bytes into it: DEMCZ_DEBUG_HIST_SKEW)
    double* dlogobj = nullptr;
    double* dlp_origin = nullptr;   // log_obj of every chain when the history window opened
    bool origin_valid = false;
    double* dtemp = nullptr;
    int64_t temp_cap = 0;
    int32_t* d_block_offsets = nullptr;
    int32_t* d_slot_of = nullptr;
    double* d_eps = nullptr;
    double* d_mu = nullptr;
    double* d_Wp = nullptr;
    double* d_design = nullptr;
    double* d_y = nullptr;
    // scratch for reductions
    double* d_scratch = nullptr;
    int64_t scratch_cap = 0;
    double* d_stage = nullptr;     // host-visible staging for small results / uploads
    int64_t stage_cap = 0;
    // state
    int64_t M = 0;
    int64_t g_done = 0;
    int64_t g0 = 0;           // history origin
    int64_t rng_offset = 0;   // generations already consumed from every chain's stream (resume)
    bool has_state = false;
    int64_t launches = 0;
    mutable int64_t kernel_counts[1] = {0};               // demcz_debug_kernel_counts: launches taken by window_kernel_ps2
    int last_live = 0, last_ps2 = 0, last_temper = 0, last_dual = 0, last_pw_reg = 0;     // demcz_debug_kernel_name: what the most recent window launch was
    bool external_append = false;
    // host-closure mode
    double* dXprop = nullptr;
    double* dlogu = nullptr;
    bool proposal_pending = false;
    bool gen_open = false;
    // host-closure mode, pipelined (round 5: demcz_closure_buffers).  Proposals and their log-densities live in pinned host memory
    // the kernels address directly: propose_kernel writes the N x d proposals there (and the device copy the commit needs), the
    // last workgroup to finish raises a flag word in the same memory, the host spins on the flag -- no D2H copy, no stream
    // synchronisation; accept_commit_kernel reads the N log-densities straight from host memory and is only ENQUEUED: the next
    // demcz_propose goes into the stream behind it while it runs.
    double* hc_X = nullptr;           // pinned, mapped: N x d (ld N)
    double* hc_lp = nullptr;          // pinned, mapped: N
    volatile unsigned int* hc_flag = nullptr;   // pinned, mapped: [0] = sequence number of the last proposal that is complete in hc_X
    unsigned int* hc_count = nullptr; // device: workgroups of the current propose launch that have finished
    unsigned int hc_seq = 0;
    // split layout: draw records, double-buffered (this launch reads one, its producer half fills the other)
    hipStream_t diag_stream = nullptr;   // demcz_run_checked, monitoring: the checks run here beside the next slab
    hipEvent_t diag_ev = nullptr;
    hipEvent_t rhat_side_ev = nullptr;   // behind the latest check enqueued on a side stream (they all work in d_scratch) ...
    bool rhat_side_pending = false;      // ... which a check on the compute stream has to wait for
    double* d_spec_X = nullptr;       // demcz_run_checked with a threshold: state before the slab enqueued ahead of a decision
    double* d_spec_lp = nullptr;
    hipEvent_t spec_ev = nullptr;
    double* pinned_rhat = nullptr;    // demcz_run_checked: pinned host slots the checks' results are copied to
    unsigned int* pinned_err = nullptr;       // demcz_run_checked: the hand-off's error word, copied behind the call's last launch ...
    int64_t pinned_err_launches = -1;         // ... valid while no window launch has been made since (h->launches then)
    int64_t pinned_cap = 0;
    bool timing = false;              // demcz_set_kernel_timing: events around every window-kernel launch
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;
    std::vector<double> series_start_ms, series_dur_ms;     // of the brackets the last demcz_get_kernel_time summed up
    int64_t timed_launches = 0;
    int64_t live_wg_cap = -1;         // consumer workgroups a LIVE launch may have (all must be resident at once); -1: not asked yet
    int split_kind = 0;               // lanes == DEMCZ_LAYOUT_SPLIT: 4 = one wave per chain, speculating (ps); 1 = eight replicated lanes per chain (pc8), 2 = 16 cooperating
                                      // lanes (ml, REC), 3 = cooperating lanes with block updates (mlb, REC)
    int split_lanes = 0;              // kinds 2, 3: lanes per chain of the consumer
    int split_per_wg = 1;             // chains per consumer workgroup
    int32_t* d_slot_role = nullptr;   // kind 3: role of every Philox block of a generation
    double* d_rec[2] = {nullptr, nullptr};
    unsigned int* d_live_err = nullptr;   // device word a LIVE launch sets when an expected row never appears
#ifdef DEMCZ_STAMPS
    unsigned long long* d_stamps = nullptr;
#endif
    // wave-per-chain layout, d <= 5: archive, both record buffers and the launch's temperatures in ONE allocation (dZ), chain
    // and log_obj histories in one too -- what window_kernel_ps2 needs to address everything with 32-bit offsets
    bool arena = false, rec_in_arena = false, hist_joint = false;
    int64_t arena_gens = 0;           // generations each record buffer of the arena holds
    double* arena_rec[2] = {nullptr, nullptr};
    double* arena_temp = nullptr;
    int64_t rec_cap = 0;              // generations each buffer holds
    int rec_cur = 0;
    bool host_paced = false;          // inside demcz_run_checked (a blocking call): see launch_window_pc
    int wpw = 1;                      // waves per consumer workgroup of the lane-cooperative kernels (window_kernel_ml / _mlb): 1 or 4, by population
    bool ps_dual = false;             // split_kind 4, d <= 5: two chains to a wave in regular launches (window_kernel_ps2d)
    bool dual_now = false;            // ... and the launch being prepared is one of those
    bool lr_spec = false;             // split_kind 2, regression target: window_kernel_lr8s (eight chains per workgroup, two generations per pass)
    mutable bool lds_raised = false, lds_raised_spec = false;  // hipFuncAttributeMaxDynamicSharedMemorySize raised on this handle's device (the attribute is per device)
    bool no_live = false;             // a LIVE hand-off failed on this handle: one launch per K-window from then on
    bool live_claimed = false;        // this handle holds its device's LIVE slot (one handle per device at a time)
    unsigned int live_spin_limit = 0; // polls before a LIVE wait gives up (0: the default, demcz_kernels_rec.h)
    int32_t live_fault_polls = 0;     // demcz_debug_set_live_fault: this poll limit instead of live_spin_limit ...
    int64_t live_fault_g = 0;         // ... in launches that start at this generation or later (0 polls: off)
    bool in_checked = false;          // inside demcz_run_checked: the call verifies (and, if need be, redoes) itself as a whole
    // LIVE launches are verified at the next synchronising call; until then the state they started from and the
    // calls made since are kept, so that a failed hand-off is redone with one launch per K-window (live_verify)
    struct RunCall { int64_t g_from, g_to; double gamma; bool tempered; std::vector<double> temperature; };
    std::vector<RunCall> live_log;
    double* d_safe_X = nullptr;
    double* d_safe_lp = nullptr;
    bool snap_pending = false;     // the state copy into d_safe_* is still to be made: by the next launch itself, or in front of it
    int64_t safe_M = 0, safe_M_app = 0, safe_g_done = 0;
    bool replaying = false;
    int32_t live_redos = 0;
    // Re-arming (round 5): a failed hand-off no longer costs the handle its LIVE launches for good.  live_rollback notes the
    // generation whose row never arrived; the redo runs one launch per K-window up to and including the demcz_run call that
    // holds that generation, and the first call that starts behind it goes LIVE again (live_try_rearm) -- at most
    // `live_rearms_left` times in the handle's life (default 3, DEMCZ_LIVE_REARMS, demcz_set_live_rearms), so that a handle
    // whose hand-off fails every time (another process's kernels on the GPU, a link that delays rows for good) still ends in
    // the one-launch-per-K-window mode that cannot fail.
    int32_t live_rearms_left = 3;
    int32_t live_rearms = 0;          // times the handle went LIVE again
    int64_t rearm_from = -1;          // >= 0: a demcz_run call that starts behind this generation re-arms (no_live is set)
    uint32_t fail_row_hint = 0xffffffffu;   // archive row of the wait that failed (live_failed -> live_rollback), 0xffffffff: unknown
    struct RecDesc { bool valid = false; int64_t g_first = 0, M = 0, rows = 0; int32_t ngen = 0, boff = 0; } rec_desc[2];
    // wave-per-chain split layout: the producer half of a launch is a kernel of its own on a side stream (its own, small
    // register budget: it fills the SIMDs beside the one-wave-per-SIMD consumers instead of sharing their workgroup shape)
    hipStream_t prod_stream = nullptr;
    hipEvent_t prod_done[2] = {nullptr, nullptr};   // records of buffer b are complete
    hipEvent_t prod_gate = nullptr;                 // main stream: the consumer that last read the buffer about to be refilled is done
    bool prod_pending[2] = {false, false};          // buffer b was (or is being) filled on the side stream: wait for prod_done[b]
    // An event (not owned here) recorded on the compute stream after the most recent window launch, nullptr if there is none:
    // whoever needs "that launch is done" on another stream -- the R-hat stream, the producer gate of the next launch --
    // waits for it instead of putting a marker of its own into the stream.  (Every marker between two window launches is
    // microseconds of an idle GPU: scripts/step_overhead.py.)
    hipEvent_t after_launch_ev = nullptr;
    // accept mask by ballot: per launch and consumer wave {changed over the launch, changed in its first generation};
    // a ring of launches, newest last (demcz_get_changed_total)
    unsigned int* d_acc = nullptr;
    int64_t acc_waves = 0;            // consumer waves of a window launch of this handle's layout
    int32_t acc_slots = 0, acc_next = 0;
    struct LaunchRec { int64_t g_first, g_last; int32_t slot; };
    std::deque<LaunchRec> acc_log;
    // multi-GPU
    ncclComm_t comm = nullptr;          // collectives on the compute stream: synchronous all-gather, R-hat all-reduces
    ncclComm_t comm_side = nullptr;     // its duplicate (ncclCommSplit) for the batched all-gathers on the side stream: operations
                                        // of ONE communicator are serialised by RCCL whatever stream they are enqueued on
    int nranks = 1, rank = 0;
    double* d_gather = nullptr;
    // deferred visibility of appended rows (demcz_set_append_lag): M counts the rows proposals may
    // draw from, M_app the rows written or reserved; equal when lag == 0
    int lag = 0;
    int64_t M_app = 0;
    struct PendingRows { int64_t visible_from; int64_t M_after; hipEvent_t ev; int64_t xseq = 0; };   // xseq: the exchange that carries the rows (0: none yet)
    std::deque<PendingRows> pending;
    int64_t batch_J = -1;              // boundary index that closes the batch the last pending entry belongs to
    // sharded + lag: snapshots of a batch travel together on a side stream
    hipStream_t comm_stream = nullptr;
    double* d_send[2] = {nullptr, nullptr};
    double* d_recv[2] = {nullptr, nullptr};
    hipEvent_t buf_done[2] = {nullptr, nullptr};
    // exchanges are numbered; the compute stream already waits for exchange number `xseq_waited` and everything before it
    // (admit_pending), so a send buffer last read by one of those needs no wait of its own
    int64_t xseq = 0, xseq_waited = 0, buf_xseq[2] = {0, 0};
    int batch_buf = 0, batch_cnt = 0;
    int64_t batch_base = 0;
    // streamed history (demcz_history_stream): pinned host mirrors of chain / log_obj, filled slab by slab on a copy stream while
    // the next slab computes
    bool hs_on = false;
    double* hs_chain = nullptr;
    double* hs_logobj = nullptr;
    hipStream_t hs_stream = nullptr;
    bool pooled_dev = false;           // dZ / dchain came from (and go back to) the process-wide device pool
    // comm failure path: every host-side wait of a sharded handle has a deadline (demcz_set_comm_timeout); on expiry, or on an
    // asynchronous RCCL error, both communicators are aborted and the handle is dead (DEMCZ_ERR_COMM from every call)
    // exchanges completed on the side stream, written by a one-thread kernel behind each batch's scatter into pinned host memory:
    // what a host-paced wait polls (hipEventQuery of an event recorded behind a stream-wait was seen to report "complete" while
    // the work in front of it had not run -- tests/test_gpu_comm_failure.py, lag 2 -- so the host does not ask the runtime)
    volatile long long* xdone = nullptr;
    // Replicated archives with the rows handed over inside the launches (demcz_kernels_rec.h, live_publish): 0 = off; 1 = a
    // replica group of handles of THIS process on one device (demcz_peer_group: rehearsal of the schedule on a one-GPU box);
    // 2 = the ranks of the RCCL communicator, every rank's archive opened over IPC by all others (demcz_comm_init)
    int peer_mode = 0;
    int n_peers = 0;
    double* peer_Z[DEMCZ_MAX_PEERS] = {nullptr};
    void* ipc_mapped[DEMCZ_MAX_PEERS] = {nullptr};   // mode 2: what hipIpcOpenMemHandle returned (closed at destroy)
    struct PeerGroup* group = nullptr;               // mode 1
    bool err_clean = false;            // the LIVE error word was read as zero and no LIVE launch has been enqueued since
    bool peer_fence = false;           // mode 2: the ranks must meet before the next launch that publishes into peers (this
                                       // replica's unwritten rows were just re-filled with the sentinel: set_state, rollback)
    unsigned int* d_err_all = nullptr; // mode 2: [0] max over ranks of the LIVE error word, [1] barrier scratch
    bool archive_fine = false;         // dZ is a fine-grained allocation of its own (hipExtMallocWithFlags), not the pool's
    size_t dZ_bytes = 0;
    size_t mailbox_off = 0;            // archive_fine: byte offset of the 4 KiB mailbox behind the archive (peer_ping)
    bool peers_closed = false;         // mode 3: demcz_peer_detach has closed the mappings of the peers' archives (no further demcz_run)
    int32_t ping_ok = -1;              // demcz_comm_init's first-contact check: -1 not made, 0 failed (on some rank), 1 passed on all
    double ping_wait_us = 0.0;         // ... and how long THIS rank waited for the last peer's token
    int64_t live_share = 0;            // per-mille of the device's LIVE capacity this handle holds (live_claim)
    int64_t comm_timeout_ms = 60000;
    bool comm_dead = false;
    int* stall_flag = nullptr;         // device word; demcz_debug_stall_exchange: the stall kernel spins until it is set
    int32_t stall_next_ms = 0;
};

#define HIPCHK(h, expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
            return DEMCZ_ERR_HIP;                                                                \
        }                                                                                        \
    } while (0)

#define NCCLCHK(h, expr)                                                                         \
    do {                                                                                         \
        ncclResult_t r_ = (expr);                                                                \
        if (r_ != ncclSuccess) {                                                                 \
            (h)->err = std::string(#expr) + ": " + ncclGetErrorString(r_);                       \
            return DEMCZ_ERR_COMM;                                                               \
        }                                                                                        \
    } while (0)


#define DEADCHK(h)                                                                               \
    do {                                                                                         \
        if ((h)->comm_dead) return DEMCZ_ERR_COMM;                                               \
    } while (0)

static int32_t fail(demcz_handle* h, int32_t code, const std::string& msg)
{
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

// ---- comm failure path ---------------------------------------------------------------------------------------------
// A sharded handle's streams carry RCCL kernels that wait for PEERS: a dead or wedged peer would make every blocking
// wait on such a stream (or on an event behind one) wait for ever, on every rank -- the worst failure mode on a shared
// pool.  So a sharded handle never blocks in the runtime: it polls (hipStreamQuery / hipEventQuery), looks at
// ncclCommGetAsyncError of both communicators about once a millisecond, and gives up at the deadline
// (demcz_set_comm_timeout, default 60 s; DEMCZ_COMM_TIMEOUT_MS): ncclCommAbort on both communicators (their kernels
// leave, the streams drain), the handle is marked dead and DEMCZ_ERR_COMM is returned from this and every later call.
// A fresh process is the only retry.  Unsharded handles wait in the runtime as before.
namespace demcz {
__global__ void mark_kernel(volatile long long* word, long long value)
{
    __hip_atomic_store(const_cast<long long*>(word), value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Are the kernels of R streams able to run at the same time?  Each adds one to a counter and waits (bounded) until all R have.
__global__ void rendezvous_kernel(unsigned int* ctr, unsigned int R, unsigned int* met, unsigned long long max_ticks)
{
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();           // 100 MHz
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < R && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(16);
    if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= R) __hip_atomic_fetch_add(met, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// First contact between the replicas of a sharded run (demcz_comm_init): every rank stores a token into slot `rank` of every
// peer's mailbox -- the same write-through, system-scope 8-byte store live_publish uses for a row, through the same IPC mapping,
// into the same fine-grained allocation (the mailbox is its last 4 KiB) -- and polls its OWN mailbox, at system scope like
// live_reload, until all R tokens are there or the time is up.  What the in-launch hand-off rests on -- a store from another
// GPU's running kernel becoming visible to a polling kernel here -- is thereby checked on the real links before any row
// depends on it; result[0] = 1 if all tokens arrived, result[1] = 100 MHz ticks this rank waited for the last of them.
struct PingBoxes { unsigned long long* box[DEMCZ_MAX_PEERS]; };
__global__ void peer_ping_kernel(unsigned long long* own, PingBoxes peers, int n_peers, int rank, int R, unsigned long long token,
                                 unsigned long long max_ticks, unsigned int* result)
{
    const int lane = (int)threadIdx.x;
    if (lane < n_peers) __hip_atomic_store(peers.box[lane] + rank, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lane == 0) __hip_atomic_store(own + rank, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    bool ok = lane >= R;
    while (!ok && wall_clock64() - t0 < max_ticks) {
        ok = __hip_atomic_load(own + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == token;
        if (!ok) __builtin_amdgcn_s_sleep(8);
    }
    const unsigned long long waited = wall_clock64() - t0;
    const bool all = __builtin_amdgcn_ballot_w64(ok) == __builtin_amdgcn_ballot_w64(true);
    unsigned long long wmax = waited;
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long v = __shfl_xor(wmax, o, 64); wmax = v > wmax ? v : wmax; }
    if (lane == 0) { result[0] = all ? 1u : 0u; result[1] = (unsigned int)(wmax > 0xffffffffull ? 0xffffffffull : wmax); }
}
__global__ void stall_kernel(int* release, unsigned long long max_ticks)
{
    const unsigned long long t0 = wall_clock64();           // 100 MHz
    while (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace demcz

static int32_t comm_fail(demcz_handle* h, const std::string& why)
{
    h->comm_dead = true;
    if (h->stall_flag) {                                      // (a test's stall kernel leaves at once: a device word, set from a stream of its own)
        hipStream_t rs = nullptr;
        if (hipStreamCreateWithFlags(&rs, hipStreamNonBlocking) == hipSuccess) {
            (void)hipMemsetAsync(h->stall_flag, 0xff, sizeof(int), rs);
            (void)hipStreamSynchronize(rs);
            (void)hipStreamDestroy(rs);
        }
    }
    if (h->comm_side) (void)ncclCommAbort(h->comm_side);
    if (h->comm) (void)ncclCommAbort(h->comm);
    h->comm_side = nullptr;                                   // (abort frees them; `comm` stays non-null as the "sharded" mark
    // bounded drain: with the collectives gone the streams should run dry; do not wait for ever for that either
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t ss[5] = {h->stream, h->comm_stream, h->prod_stream, h->diag_stream, h->hs_stream};
    for (hipStream_t st : ss) {
        if (!st) continue;
        while (hipStreamQuery(st) == hipErrorNotReady &&
               std::chrono::steady_clock::now() - t0 < std::chrono::seconds(1)) std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    h->err = "communication failure (" + why + "): both RCCL communicators aborted; the handle is dead -- restart the job in fresh processes";
    return DEMCZ_ERR_COMM;
}

template <class Query>
static int32_t wait_deadline(demcz_handle* h, Query query, const char* what)
{
    static const bool dbg = getenv("DEMCZ_DEBUG_COMM") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    auto last_check = t0;
    for (unsigned long long polls = 0;; ++polls) {
        const hipError_t e = query();
        if (e == hipSuccess) {
            if (dbg) fprintf(stderr, "[demcz] wait %s: done after %llu polls, %.3f ms\n", what, polls,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            return DEMCZ_OK;
        }
        if (e != hipErrorNotReady) { h->err = std::string(what) + ": " + hipGetErrorString(e); return DEMCZ_ERR_HIP; }
        const auto now = std::chrono::steady_clock::now();
        if (now - last_check >= std::chrono::milliseconds(1)) {
            last_check = now;
            for (ncclComm_t c : {h->comm, h->comm_side}) {
                if (!c) continue;
                ncclResult_t ar = ncclSuccess;
                const ncclResult_t qr = ncclCommGetAsyncError(c, &ar);
                if (qr != ncclSuccess || (ar != ncclSuccess && ar != ncclInProgress))
                    return comm_fail(h, std::string(what) + ": RCCL reports " + ncclGetErrorString(qr != ncclSuccess ? qr : ar));
            }
            if (h->comm_timeout_ms > 0 && now - t0 >= std::chrono::milliseconds(h->comm_timeout_ms))
                return comm_fail(h, std::string(what) + ": no progress within " + std::to_string(h->comm_timeout_ms) + " ms -- a peer rank is dead or stalled");
        }
        if (polls > 2000) std::this_thread::sleep_for(std::chrono::microseconds(20));      // (spin first: most waits are short)
    }
}

static int32_t sync_stream(demcz_handle* h, hipStream_t s, const char* what)
{
    if (h->comm_dead) return DEMCZ_ERR_COMM;
    if (!h->comm) {
        const hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) { h->err = std::string(what) + ": hipStreamSynchronize: " + hipGetErrorString(e); return DEMCZ_ERR_HIP; }
        return DEMCZ_OK;
    }
    return wait_deadline(h, [s]() { return hipStreamQuery(s); }, what);
}

static int32_t sync_event(demcz_handle* h, hipEvent_t ev, const char* what)
{
    if (h->comm_dead) return DEMCZ_ERR_COMM;
    if (!h->comm) {
        const hipError_t e = hipEventSynchronize(ev);
        if (e != hipSuccess) { h->err = std::string(what) + ": hipEventSynchronize: " + hipGetErrorString(e); return DEMCZ_ERR_HIP; }
        return DEMCZ_OK;
    }
    return wait_deadline(h, [ev]() { return hipEventQuery(ev); }, what);
}

#define SYNCCHK(h, s)                                                                            \
    do {                                                                                         \
        int32_t rcs_ = sync_stream((h), (s), __func__);                                          \
        if (rcs_) return rcs_;                                                                   \
    } while (0)

// hipFree / hipMalloc synchronise the whole device: before either, every stream of a sharded handle is waited for HERE, with
// the deadline -- so that a stalled peer surfaces as DEMCZ_ERR_COMM instead of a runtime call that never returns.
static int32_t quiesce_all(demcz_handle* h)
{
    if (h->comm_dead) return DEMCZ_ERR_COMM;
    if (!h->comm) return DEMCZ_OK;
    for (hipStream_t st : {h->stream, h->comm_stream, h->prod_stream, h->diag_stream, h->hs_stream})
        if (st) { int32_t rc = sync_stream(h, st, "quiesce"); if (rc) return rc; }
    return DEMCZ_OK;
}

// test hook (demcz_debug_stall_exchange): a kernel that holds the stream the next collective goes to
static int32_t maybe_stall(demcz_handle* h, hipStream_t s)
{
    if (h->stall_next_ms <= 0) return DEMCZ_OK;
    if (getenv("DEMCZ_DEBUG_COMM")) fprintf(stderr, "[demcz] stall kernel of %d ms enqueued\n", h->stall_next_ms);
    if (!h->stall_flag) HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->stall_flag, sizeof(int)));
    HIPCHK(h, hipMemsetAsync(h->stall_flag, 0, sizeof(int), s));
    hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(1), 0, s, h->stall_flag, (unsigned long long)h->stall_next_ms * 100000ull);
    HIPCHK(h, hipGetLastError());
    h->stall_next_ms = 0;
    return DEMCZ_OK;
}

static int64_t blockstep_nblk(int b)
{
    const int nn = (b == 1) ? 1 : b;
    return 1 + (nn + 1) / 2 + 1;
}

constexpr size_t ML_MAX_DYNAMIC_LDS = 160 * 1024;      // LDS per CU on gfx950
constexpr size_t PEER_MAILBOX_BYTES = 4096;            // behind a fine-grained archive: peer_ping_kernel's tokens
constexpr size_t REC_PAD = 64;       // doubles behind a record buffer: a consumer's fetches may run past its last row's last generation
constexpr int PC_CONSUMER_CHAINS = 8;       // chains per consumer workgroup of the replicated split layout: 8 lanes per chain
static int ml_lanes_available(int target_kind, int d, bool full_block, int64_t nobs, int max_blocklen, int nblocks);
static bool pc_available(int target_kind, int d, bool full_block);
static bool mlb32_wanted(int64_t N);
static bool split_ml_available(int target_kind, int d, bool full_block, int64_t nobs);
static bool ps_available(int target_kind, int d);
static int64_t live_wg_capacity(demcz_handle* h);
// the draw records on the device no longer match what the next launch will need (or are about to be freed): nothing of
// the side-stream producer may still be writing them
static void rec_invalidate(demcz_handle* h)
{
    if (h->prod_stream) (void)hipStreamSynchronize(h->prod_stream);
    h->prod_pending[0] = h->prod_pending[1] = false;
    h->rec_desc[0].valid = h->rec_desc[1].valid = false;
}
// The archive is about to get SHORTER (a rollback, a new demcz_set_state, a discarded speculative slab): the record buffers may hold
// row indices drawn against the longer one.  No launch consumes them -- rec_invalidate makes the next launch draw its own -- but
// the wave-per-chain kernels prefetch a pass or two past a launch's last generation, out of whatever the buffer holds there,
// and a prefetched index of an unwritten row reads the sentinel.  Zeros ("row 0: always a legal index") again, as at allocation.
static int32_t rec_scrub(demcz_handle* h)
{
    rec_invalidate(h);
    if (h->lanes != DEMCZ_LAYOUT_SPLIT || h->rec_cap <= 0 || getenv("DEMCZ_NO_SCRUB")) return DEMCZ_OK;
    for (int b = 0; b < 2; ++b) {
        if (!h->d_rec[b]) continue;
        const size_t per = h->rec_in_arena ? (size_t)(h->cfg.d + 2) : (size_t)((h->split_kind == 3) ? 2 * h->S : (int64_t)h->cfg.d + 2);
        const size_t nd = (size_t)h->rec_cap * per * (size_t)h->cfg.N + REC_PAD;
        HIPCHK(h, hipMemsetAsync(h->d_rec[b], 0, nd * sizeof(double), h->stream));
    }
    return DEMCZ_OK;
}
static int32_t flush_exchanges(demcz_handle* h);
static void peer_detach(demcz_handle* h);
static void peer_no_dual(demcz_handle* h);
static int32_t peer_setup_ipc(demcz_handle* h);
static bool peer_capable(const demcz_handle* h);
static int32_t check_live_err(demcz_handle* h);
static int32_t live_verify(demcz_handle* h);
static int32_t group_verify(demcz_handle* h);
static int32_t live_failed(demcz_handle* h, bool& failed);
static void live_release(demcz_handle* h);
static int64_t live_span(demcz_handle* h);
static int32_t rec_reserve(demcz_handle* h, int64_t gens);
static int32_t check_hist_range(demcz_handle* h, int64_t g_from, int64_t g_to, const char* who);
static bool ps2_applicable(const demcz_handle* h, const WindowParams& P);

extern "C" int32_t demcz_abi_version(void) { return DEMCZ_ABI_VERSION; }

extern "C" const char* demcz_last_error(const demcz_handle* h)
{
    return h ? h->err.c_str() : g_create_error.c_str();
}

template <class T>
static hipError_t dev_alloc_copy(T** dst, const T* src, size_t n, hipStream_t s, int device)
{
    hipError_t e = dev_malloc(device, (void**)dst, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (n) e = hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, s);
    return e;
}

static void free_all(demcz_handle* h)
{
    // Buffers only go back to the pools when nothing can still be touching them: every stream of the handle has run dry
    // (demcz_destroy waits for them) and no communicator was aborted under it.  Otherwise they are freed -- hipFree / hipHostFree
    // wait for the device, which is what the pools exist to avoid but is the only safe thing then.
    bool drained = !h->comm_dead;
    for (hipStream_t st : {h->stream, h->prod_stream, h->diag_stream, h->comm_stream, h->hs_stream})
        if (st && hipStreamQuery(st) != hipSuccess) drained = false;
    if (h->d_acc) (void)dev_free(h->cfg.device_id, h->d_acc);
    if (h->d_err_all) (void)dev_free(h->cfg.device_id, h->d_err_all);
    if (h->archive_fine) { if (h->dZ) (void)hipFree(h->dZ); h->dZ = nullptr; }     // (an allocation of its own: demcz_comm_init)
    if (h->pooled_dev) {               // (the two big ones)
        if (h->archive_fine) {
        } else if (drained) {
            g_dev_pool.release(h->dZ, h->cfg.device_id);
        } else {
            g_dev_pool.discard(h->dZ);
        }
        if (h->hist_joint) { if (drained) g_dev_pool.release(h->dchain_alloc, h->cfg.device_id); else g_dev_pool.discard(h->dchain_alloc); }
        h->dZ = nullptr;
        if (h->hist_joint) { h->dchain = nullptr; h->dchain_alloc = nullptr; }
    }
    if (h->hs_stream) stream_release(h->cfg.device_id, h->hs_stream, drained);
    if (drained) { g_host_pool.release(h->hs_chain, -1); g_host_pool.release(h->hs_logobj, -1); }
    else { g_host_pool.discard(h->hs_chain); g_host_pool.discard(h->hs_logobj); }
    void* bufs[] = {h->dZ, h->dX, h->dlp, h->dchain, h->hist_joint ? nullptr : h->dlogobj, h->dlp_origin, h->dtemp, h->d_block_offsets,
                    h->d_slot_of, h->d_slot_role, h->d_eps, h->d_mu, h->d_Wp, h->d_design, h->d_y, h->d_scratch, h->dXprop,
                    h->dlogu, h->d_gather};
    for (void* b : bufs)
        if (b) (void)dev_free(h->cfg.device_id, b);
    for (auto& pe : h->pending) if (pe.ev) (void)hipEventDestroy(pe.ev);
    h->pending.clear();
    for (int b = 0; b < 2; ++b) {
        if (h->d_rec[b] && !h->rec_in_arena) (void)dev_free(h->cfg.device_id, h->d_rec[b]);
        if (b == 0 && h->d_live_err) (void)dev_free(h->cfg.device_id, h->d_live_err);
        if (h->d_send[b]) (void)dev_free(h->cfg.device_id, h->d_send[b]);
        if (h->d_recv[b]) (void)dev_free(h->cfg.device_id, h->d_recv[b]);
        if (h->buf_done[b]) (void)hipEventDestroy(h->buf_done[b]);
    }
    if (h->comm_stream) stream_release(h->cfg.device_id, h->comm_stream, !h->comm_dead && hipStreamQuery(h->comm_stream) == hipSuccess);
    if (h->prod_stream) { if (!h->comm_dead) (void)hipStreamSynchronize(h->prod_stream); stream_release(h->cfg.device_id, h->prod_stream, !h->comm_dead); }
    for (int b = 0; b < 2; ++b) if (h->prod_done[b]) (void)hipEventDestroy(h->prod_done[b]);
    if (h->prod_gate) (void)hipEventDestroy(h->prod_gate);
    if (h->d_stage) (void)host_free(h->d_stage);
    for (auto& pr : h->timed) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (h->pinned_rhat) (void)host_free(h->pinned_rhat);
    if (h->pinned_err) (void)host_free(h->pinned_err);
    if (h->d_safe_X) (void)dev_free(h->cfg.device_id, h->d_safe_X);
    if (h->d_safe_lp) (void)dev_free(h->cfg.device_id, h->d_safe_lp);
    if (h->d_spec_X) (void)dev_free(h->cfg.device_id, h->d_spec_X);
    if (h->d_spec_lp) (void)dev_free(h->cfg.device_id, h->d_spec_lp);
    if (h->spec_ev) (void)hipEventDestroy(h->spec_ev);
    if (h->diag_ev) (void)hipEventDestroy(h->diag_ev);
    if (h->rhat_side_ev) (void)hipEventDestroy(h->rhat_side_ev);
    if (h->diag_stream) stream_release(h->cfg.device_id, h->diag_stream, !h->comm_dead && hipStreamQuery(h->diag_stream) == hipSuccess);
    if (h->stall_flag) (void)dev_free(h->cfg.device_id, h->stall_flag);
    if (h->hc_X) (void)hipHostFree(h->hc_X);
    if (h->hc_lp) (void)hipHostFree(h->hc_lp);
    if (h->hc_flag) (void)hipHostFree(const_cast<unsigned int*>(h->hc_flag));
    if (h->hc_count) (void)dev_free(h->cfg.device_id, h->hc_count);
    if (h->xdone) (void)host_free(const_cast<long long*>(h->xdone));
    if (!h->comm_dead) {                 // (a dead handle's communicators were aborted, which frees them)
        if (h->comm_side) (void)ncclCommDestroy(h->comm_side);
        if (h->comm) (void)ncclCommDestroy(h->comm);
    }
    if (h->own_stream && h->stream) stream_release(h->cfg.device_id, h->stream, !h->comm_dead && hipStreamQuery(h->stream) == hipSuccess);
}

extern "C" int32_t demcz_create(demcz_handle** out, const demcz_config* cfg)
{
    if (!out || !cfg) return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_create: null argument");
    *out = nullptr;
    if (cfg->N < 1 || cfg->d < 1 || cfg->d > MAX_D || cfg->K < 1 || cfg->Mcap < 2 || cfg->Gcap < 0 ||
        cfg->Nblocks < 1 || !cfg->block_offsets || !cfg->block_indices || !cfg->eps_scale || cfg->chain_id0 < 0)
        return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT,
                    "demcz_create: need N>=1, 1<=d<=64, K>=1, Mcap>=2, Gcap>=0, Nblocks>=1 and block/eps tables");
    if (cfg->lanes_per_chain != 0 && cfg->lanes_per_chain != 1 && cfg->lanes_per_chain != 8 && cfg->lanes_per_chain != 16 &&
        cfg->lanes_per_chain != DEMCZ_LAYOUT_SPLIT && cfg->lanes_per_chain != DEMCZ_LAYOUT_SPLIT_WAVE)
        return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT,
                    "demcz_create: lanes_per_chain must be 0 (auto), 1, 8, 16, DEMCZ_LAYOUT_SPLIT or DEMCZ_LAYOUT_SPLIT_WAVE");
    const int d = cfg->d;
    // validate blocks: offsets ascending, indices within range and unique inside a block
    if (cfg->block_offsets[0] != 0) return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT, "block_offsets[0] must be 0");
    for (int ib = 0; ib < cfg->Nblocks; ++ib) {
        const int a = cfg->block_offsets[ib], b = cfg->block_offsets[ib + 1];
        if (b <= a || b - a > d) return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT, "empty or oversized block");
        for (int t = a; t < b; ++t) {
            if (cfg->block_indices[t] < 0 || cfg->block_indices[t] >= d)
                return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT, "block index out of range");
            for (int u = a; u < t; ++u)
                if (cfg->block_indices[u] == cfg->block_indices[t])
                    return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT, "duplicate index inside a block");
        }
    }
    switch (cfg->target_kind) {
    case DEMCZ_TARGET_MVNORMAL:
        if (!cfg->mu || !cfg->W) return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT, "MVNORMAL needs mu and W");
        break;
    case DEMCZ_TARGET_ISO_QUAD:
        if (!cfg->mu) return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT, "ISO_QUAD needs mu");
        break;
    case DEMCZ_TARGET_LINREG_SSE:
        if (!cfg->design || !cfg->yobs || cfg->nobs < 1)
            return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT, "LINREG_SSE needs design, yobs, nobs>=1");
        break;
    case DEMCZ_TARGET_HOST_CALLBACK: break;
    default: return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT, "unknown target_kind");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, DEMCZ_ERR_NO_DEVICE, "demcz_create: no HIP device visible (there is no CPU fallback)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev)
        return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_create: device_id out of range");

    demcz_handle* h = new demcz_handle();
    h->cfg = *cfg;
    h->lanes = 1;
    if (const char* re = getenv("DEMCZ_LIVE_REARMS")) h->live_rearms_left = std::max(0, atoi(re));
    auto bail = [&](int32_t code) {
        g_create_error = h->err;
        free_all(h);
        delete h;
        return code;
    };
#define CRCHK(expr)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            h->err = std::string(#expr) + ": " + hipGetErrorString(e_);                          \
            return bail(DEMCZ_ERR_HIP);                                                          \
        }                                                                                        \
    } while (0)
    CRCHK(hipSetDevice(cfg->device_id));
    if (cfg->stream) {
        h->stream = (hipStream_t)cfg->stream;
    } else {
        CRCHK(stream_acquire(cfg->device_id, &h->stream));
        h->own_stream = true;
    }
    h->block_offsets.assign(cfg->block_offsets, cfg->block_offsets + cfg->Nblocks + 1);
    h->block_indices.assign(cfg->block_indices, cfg->block_indices + h->block_offsets.back());
    h->eps.assign(cfg->eps_scale, cfg->eps_scale + d);
    h->slot_of.assign((size_t)cfg->Nblocks * d, -1);
    h->S = 0;
    for (int ib = 0; ib < cfg->Nblocks; ++ib) {
        const int a = h->block_offsets[ib], b = h->block_offsets[ib + 1];
        for (int t = a; t < b; ++t) h->slot_of[(size_t)ib * d + h->block_indices[t]] = t - a;
        h->S += blockstep_nblk(b - a);
    }
    h->full_block = (cfg->Nblocks == 1 && h->block_offsets[1] == d);
    if (h->full_block)
        for (int p = 0; p < d; ++p) h->full_block = h->full_block && (h->block_indices[p] == p);
    if (cfg->target_kind == DEMCZ_TARGET_MVNORMAL && cfg->Nblocks >= 2 && h->block_offsets.back() == d) {
        // the blocks, in order, are consecutive index ranges covering 0..d-1 (any order of the indices inside a block): the sums
        // of the quadratic form are cut at their boundaries (the grouped order)
        bool ok = true;
        uint64_t starts = 0;
        for (int ib = 0; ib < cfg->Nblocks && ok; ++ib) {
            const int lo = h->block_offsets[ib], hi = h->block_offsets[ib + 1];
            uint64_t seen = 0;
            for (int t = lo; t < hi; ++t) {
                const int j = h->block_indices[t];
                if (j < lo || j >= hi || ((seen >> (j - lo)) & 1ull)) { ok = false; break; }
                seen |= 1ull << (j - lo);
            }
            starts |= 1ull << lo;
        }
        if (ok) {
            h->ngrp = cfg->Nblocks;
            h->gstart = starts;
            bool equal = true;
            for (int ib = 0; ib < cfg->Nblocks; ++ib) equal = equal && (h->block_offsets[ib + 1] - h->block_offsets[ib] == h->block_offsets[1]);
            if (equal && d == 20 && h->block_offsets[1] == 5 && !getenv("DEMCZ_NO_MLB_INCREMENTAL")) h->mlb_qb = 5;      // C3: 4 x 5
        }
    }
    {   // layout: L lanes per chain when the chip would otherwise sit idle (small N), else one lane
        int maxb = 0;
        for (int ib = 0; ib < cfg->Nblocks; ++ib) maxb = std::max(maxb, h->block_offsets[ib + 1] - h->block_offsets[ib]);
        const int L = ml_lanes_available(cfg->target_kind, d, h->full_block, cfg->nobs, maxb, cfg->Nblocks);
        // d = 20 in blocks, split form chosen by the library: 32 lanes per chain (two chains to a wave, one parameter a lane)
        const int Lsplit = (L == 16 && d == 20 && !h->full_block && cfg->lanes_per_chain == 0 && cfg->N <= 4096 && mlb32_wanted(cfg->N)) ? 32 : L;
        // 32-bit row indices in the records, 32-bit byte offsets into the archive
        const bool idx32 = cfg->Mcap <= 0xffffffffll && (double)cfg->Mcap * 8.0 * (((d + 7) / 8) * 8) < 4294967296.0;
        const int kind = !idx32 ? 0 : pc_available(cfg->target_kind, d, h->full_block) ? 1
                         : split_ml_available(cfg->target_kind, d, h->full_block, cfg->nobs) ? 2
                         : (!h->full_block && cfg->target_kind == DEMCZ_TARGET_MVNORMAL && L > 1) ? 3 : 0;
        h->split_lanes = (kind == 2) ? 16 : (kind == 3) ? Lsplit : 0;
        {   // four-wave workgroups once there is a chain wave for every SIMD (see demcz_kernels_ml.h, ML_WAVES)
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device_id) != hipSuccess) cus = 0;
            const int lanes_pc = (kind == 2) ? 16 : (kind == 3) ? Lsplit : (L > 1 ? L : 64);
            const int64_t chain_waves = (cfg->N * lanes_pc + 63) / 64;
            h->wpw = (cus > 0 && chain_waves >= 4ll * cus) ? ML_WAVES : 1;
        }
        // chains per consumer workgroup
        h->split_per_wg = (kind == 1) ? PC_CONSUMER_CHAINS : (kind == 2) ? ((cfg->target_kind == DEMCZ_TARGET_LINREG_SSE) ? LR16_CHAINS : 4 * h->wpw)
                          : (kind == 3) ? h->wpw * (64 / Lsplit) : 1;
        const bool split_ok = kind != 0;
        // one wave per chain (demcz_kernels_ps.h): where the replicated consumer is built and a pass's draws fit one DMA
        // (round 5: wherever the archive's rows are reachable by 32-bit offsets -- no longer only where another split consumer is
        //  built for the dimension too)
        const bool ps_ok = idx32 && h->full_block && ps_available(cfg->target_kind, d);
        const int per_wg_default = h->split_per_wg;
        h->split_kind = 0;
        if (cfg->lanes_per_chain == DEMCZ_LAYOUT_SPLIT_WAVE || (cfg->lanes_per_chain == 0 && ps_ok && cfg->N <= PS_MAX_N && cfg->K >= 2 && !getenv("DEMCZ_NO_PS"))) {
            // (us per K-window at d=5, one wave per chain / eight replicated lanes: see DESIGN.md, K1g; K = 1 leaves a
            //  pass one generation: 1.05 against 0.99 us per generation, scripts/k_small.py)
            if (!ps_ok) {
                h->err = "demcz_create: the wave-per-chain split layout is not built for this target / d / block structure";
                return bail(DEMCZ_ERR_INVALID_ARGUMENT);
            }
            h->lanes = DEMCZ_LAYOUT_SPLIT;
            h->split_kind = 4;
            h->split_per_wg = PS_CHAINS;
            // The library's own choice also asks that the consumers of a LIVE launch all fit the chip at once (one launch
            // per K-window is where this layout loses to the replicated consumer): MI355X, d = 5: 1024 chains.
            // Two chains to a wave (window_kernel_ps2d, d <= 5, K a multiple of five): where one chain per wave no longer fits a LIVE
            // launch (N > 1024 on MI355X) but half as many waves do -- up to 2048 chains.  DEMCZ_PS_DUAL=1 forces it at any N (tests).
            const bool dual_env = getenv("DEMCZ_PS_DUAL") != nullptr && atoi(getenv("DEMCZ_PS_DUAL")) != 0;
            const bool dual_ok = d >= 2 && d <= 5 && cfg->K % 5 == 0 && !getenv("DEMCZ_NO_PS2") && !getenv("DEMCZ_NO_PS_DUAL");
            const bool single_fits = (cfg->N + PS_CHAINS - 1) / PS_CHAINS <= live_wg_capacity(h);
            if (dual_ok && (dual_env || (!single_fits && cfg->lanes_per_chain == 0))) {
                h->ps_dual = true;
                h->split_per_wg = 2 * PS_CHAINS;
                h->live_wg_cap = -1;
                if (!dual_env && (cfg->N + 2 * PS_CHAINS - 1) / (2 * PS_CHAINS) > live_wg_capacity(h)) {      // not even that fits
                    h->ps_dual = false;
                    h->split_per_wg = PS_CHAINS;
                    h->live_wg_cap = -1;
                }
            }
            if (!h->ps_dual && cfg->lanes_per_chain == 0 && !single_fits && kind != 0) {
                h->split_kind = kind;
                h->split_per_wg = per_wg_default;
                h->live_wg_cap = -1;
            }
            // (kind == 0 -- a dimension with no other split consumer -- and more chains than a LIVE launch holds, up to PS_MAX_N:
            //  one wave per chain all the same, one launch per K-window; the alternative is the one-lane kernel's 16-32 waves)
        } else if (cfg->lanes_per_chain == DEMCZ_LAYOUT_SPLIT) {
            if (!split_ok) {
                h->err = "demcz_create: the split layout is not built for this target / d / block structure";
                return bail(DEMCZ_ERR_INVALID_ARGUMENT);
            }
            h->lanes = DEMCZ_LAYOUT_SPLIT;
            h->split_kind = kind;
        } else if (cfg->lanes_per_chain == 0 && split_ok && cfg->N <= (kind == 1 ? 8192 : 4096)) {
            // (16 cooperating lanes, d = 20, us per K-window split / fused: N=1024 9.7 / 16.8, N=4096 16.6 / 21.0,
            //  N=8192 35.5 / 29.8)
            h->split_kind = kind;
            // idle CUs do the state-independent three quarters of the work.  Measured at d=5, us per
            // K-window, split / 8 lanes per chain / 1 lane: N=4096 8.6 / 12.4 / 30.0, N=16384 22.4 / 19.0 /
            // 32.6, N=32768 41.4 / 29.7 / 35.7, N=65536 77.7 / 54.5 / 40.9
            h->lanes = DEMCZ_LAYOUT_SPLIT;
        } else if (cfg->lanes_per_chain > 1) {
            if (L != cfg->lanes_per_chain) {
                h->err = "demcz_create: the requested lanes_per_chain layout is not built for this target / d / block structure";
                return bail(DEMCZ_ERR_INVALID_ARGUMENT);
            }
            h->lanes = L;
        } else if (cfg->lanes_per_chain == 0 && L > 1 && cfg->N * L <= 262144) {
            h->lanes = L;       // up to ~4 waves per SIMD; measured crossover with one lane per chain at d=5:
        }                       // N=32768 29.9 vs 35.4 us per window, N=65536 53.8 vs 45.8
    }
    if (h->lanes == DEMCZ_LAYOUT_SPLIT && h->split_kind == 2 && cfg->target_kind == DEMCZ_TARGET_LINREG_SSE && !getenv("DEMCZ_NO_LR_SPEC")) {
        // Eight chains per workgroup, two generations per log-density pass, where that still is at most one workgroup per CU
        // (C5: 2048 chains = 256 workgroups; sixteen chains per workgroup leave half the chip idle there).
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device_id) != hipSuccess) cus = 0;
        if ((cfg->N + LR8_CHAINS - 1) / LR8_CHAINS <= (int64_t)cus && lr8s_dynamic_lds<10>(cfg->nobs) <= ML_MAX_DYNAMIC_LDS) {
            h->lr_spec = true;
            h->split_per_wg = LR8_CHAINS;
        }
    }
    // no pointer of the caller's survives create
    h->cfg.block_offsets = nullptr; h->cfg.block_indices = nullptr; h->cfg.eps_scale = nullptr;
    h->cfg.mu = nullptr; h->cfg.W = nullptr; h->cfg.design = nullptr; h->cfg.yobs = nullptr; h->cfg.stream = nullptr;

    const int64_t N = cfg->N;
    const int64_t N_for_arena = cfg->N;
    h->ZS = (d <= 1) ? 2 : (d <= 2) ? 2 : (d <= 4) ? 4 : ((d + 7) / 8) * 8;     // 16-byte aligned rows; d=5 -> one 64-byte line
    {
        size_t zbytes = (size_t)cfg->Mcap * h->ZS * sizeof(double);
        if (h->lanes == DEMCZ_LAYOUT_SPLIT && h->split_kind == 4 && d <= 5 && !getenv("DEMCZ_NO_PS2")) {
            // (record buffers as demcz_run would size them: a LIVE launch's span, bounded by the history window; a caller that
            //  outruns the arena gets separate buffers and the general kernel)
            const int64_t per_gen = (int64_t)(d + 2) * N_for_arena * (int64_t)sizeof(double);
            // (64 MiB of records per buffer: 1170 generations at C2; a two-chain handle has up to twice the chains: 128 MiB, so that
            //  an autostop slab of 1000 generations is still ONE launch at 2048 chains)
            int64_t ag = std::max<int64_t>(cfg->K, std::min<int64_t>((int64_t)((h->ps_dual ? 128ll : 64ll) << 20) / per_gen, 1 << 20));
            if (cfg->Gcap > 0) ag = std::min<int64_t>(ag, std::max<int64_t>(cfg->Gcap, 1024));      // (no history kept: launches are as long as a call)
            ag = std::max<int64_t>(ag, (int64_t)cfg->K * 4);
            const size_t zb = (zbytes + 255) & ~(size_t)255;
            const size_t rb = (((size_t)ag * (size_t)(d + 2) * (size_t)N_for_arena + REC_PAD) * sizeof(double) + 255) & ~(size_t)255;
            const size_t tb = (((size_t)ag + 64) * sizeof(double) + 255) & ~(size_t)255;
            if (zb + 2 * rb + tb < 0xF0000000ull) {
                h->arena = true;
                h->arena_gens = ag;
                CRCHK(g_dev_pool.acquire((void**)&h->dZ, zb + 2 * rb + tb, cfg->device_id));
                h->dZ_bytes = zb + 2 * rb + tb;
                unsigned char* base = reinterpret_cast<unsigned char*>(h->dZ);
                h->arena_rec[0] = reinterpret_cast<double*>(base + zb);
                h->arena_rec[1] = reinterpret_cast<double*>(base + zb + rb);
                h->arena_temp = reinterpret_cast<double*>(base + zb + 2 * rb);
                CRCHK(hipMemsetAsync(base + zb, 0, 2 * rb + tb, h->stream));       // row 0 / temperature 0: always legal
            }
        }
        if (!h->arena) { CRCHK(g_dev_pool.acquire((void**)&h->dZ, zbytes, cfg->device_id)); h->dZ_bytes = zbytes; }
        if (h->ps_dual && !h->arena) { h->ps_dual = false; h->split_per_wg = PS_CHAINS; h->live_wg_cap = -1; }      // (the two-chain kernel addresses the arena)
        h->pooled_dev = true;
    }
    // (the reference pads with zeros, demcz.jl:11; rows at or beyond M never leave the device, and here they hold
    //  the sentinel LIVE launches recognise an unpublished row by: fill_unwritten_rows() in demcz_set_state)
    CRCHK(dev_malloc(h->cfg.device_id, (void**)&h->d_live_err, 4 * sizeof(unsigned int)));
    CRCHK(hipMemsetAsync(h->d_live_err, 0, 4 * sizeof(unsigned int), h->stream));
    CRCHK(dev_malloc(h->cfg.device_id, (void**)&h->dX, (size_t)N * d * sizeof(double)));
    CRCHK(dev_malloc(h->cfg.device_id, (void**)&h->dlp, (size_t)N * sizeof(double)));
    if (cfg->Gcap > 0) {
        // (one allocation: a window kernel's history store of a pass covers both arrays with one buffer descriptor)
        // (DEMCZ_DEBUG_HIST_SKEW=<bytes>: the history that many bytes into its allocation -- scripts/ps2_stamps.py, address classes)
        const size_t skew = getenv("DEMCZ_DEBUG_HIST_SKEW") ? ((size_t)atol(getenv("DEMCZ_DEBUG_HIST_SKEW")) & ~(size_t)7) : 0;
        CRCHK(g_dev_pool.acquire((void**)&h->dchain_alloc, (size_t)N * (d + 1) * cfg->Gcap * sizeof(double) + skew, cfg->device_id));
        h->dchain = h->dchain_alloc + skew / sizeof(double);
        h->dlogobj = h->dchain + (size_t)N * d * cfg->Gcap;
        h->hist_joint = true;
        CRCHK(hipMemsetAsync(h->dchain, 0, (size_t)N * (d + 1) * cfg->Gcap * sizeof(double), h->stream));   // demcz.jl:24
    }
    CRCHK(dev_malloc(h->cfg.device_id, (void**)&h->dlp_origin, (size_t)N * sizeof(double)));
    CRCHK(dev_alloc_copy(&h->d_block_offsets, h->block_offsets.data(), h->block_offsets.size(), h->stream, cfg->device_id));
    CRCHK(dev_alloc_copy(&h->d_slot_of, h->slot_of.data(), h->slot_of.size(), h->stream, cfg->device_id));
    CRCHK(dev_alloc_copy(&h->d_eps, h->eps.data(), h->eps.size(), h->stream, cfg->device_id));
    {   // what every Philox block of a generation is: rows, a normal pair, or the accept uniform (per block, in order)
        std::vector<int32_t> role;
        for (int ib = 0; ib < cfg->Nblocks; ++ib) {
            const int nblk = (int)blockstep_nblk(h->block_offsets[ib + 1] - h->block_offsets[ib]);
            for (int t = 0; t < nblk; ++t) role.push_back(t == 0 ? 0 : (t == nblk - 1 ? 2 : 1));
        }
        CRCHK(dev_alloc_copy(&h->d_slot_role, role.data(), role.size(), h->stream, cfg->device_id));
    }
    std::vector<double> wp, design_rm;
    if (cfg->target_kind == DEMCZ_TARGET_MVNORMAL || cfg->target_kind == DEMCZ_TARGET_ISO_QUAD)
        CRCHK(dev_alloc_copy(&h->d_mu, cfg->mu, (size_t)d, h->stream, cfg->device_id));
    if (cfg->target_kind == DEMCZ_TARGET_MVNORMAL) {
        wp.resize((size_t)d * (d + 1) / 2);
        for (int i = 0; i < d; ++i)
            for (int j = 0; j <= i; ++j) wp[(size_t)i * (i + 1) / 2 + j] = cfg->W[i + (size_t)d * j];
        CRCHK(dev_alloc_copy(&h->d_Wp, wp.data(), wp.size(), h->stream, cfg->device_id));
    }
    if (cfg->target_kind == DEMCZ_TARGET_LINREG_SSE) {
        design_rm.resize((size_t)cfg->nobs * d);
        for (int64_t o = 0; o < cfg->nobs; ++o)
            for (int j = 0; j < d; ++j) design_rm[(size_t)o * d + j] = cfg->design[o + cfg->nobs * j];
        CRCHK(dev_alloc_copy(&h->d_design, design_rm.data(), design_rm.size(), h->stream, cfg->device_id));
        CRCHK(dev_alloc_copy(&h->d_y, cfg->yobs, (size_t)cfg->nobs, h->stream, cfg->device_id));
    }
    if (cfg->target_kind == DEMCZ_TARGET_HOST_CALLBACK) {
        CRCHK(dev_malloc(h->cfg.device_id, (void**)&h->dXprop, (size_t)N * d * sizeof(double)));
        CRCHK(dev_malloc(h->cfg.device_id, (void**)&h->dlogu, (size_t)N * sizeof(double)));
    }
    if (cfg->target_kind != DEMCZ_TARGET_HOST_CALLBACK) {
        // waves that run chains in one window launch (every one writes its two counters)
        int64_t waves;
        if (h->lanes == DEMCZ_LAYOUT_SPLIT) {
            // (a two-chain handle's irregular launches run one chain per wave: twice the waves)
            const int64_t wgs = (N + (h->ps_dual ? PS_CHAINS : h->split_per_wg) - 1) / (h->ps_dual ? PS_CHAINS : h->split_per_wg);
            waves = wgs * ((h->split_kind == 2 && cfg->target_kind == DEMCZ_TARGET_LINREG_SSE) ? LR16_WAVES : (h->split_kind == 4) ? PS_CHAINS : (h->split_kind == 3 || h->split_kind == 2) ? h->wpw : 1);
        } else if (h->lanes > 1) {
            const int per_wave = 64 / h->lanes;
            const bool lr = h->full_block && cfg->target_kind == DEMCZ_TARGET_LINREG_SSE && d == 10 &&
                            lr16_dynamic_lds<10>(cfg->nobs) <= ML_MAX_DYNAMIC_LDS;      // (uses_lr16: the handle's cfg is not complete yet)
            waves = lr ? ((N + LR16_CHAINS - 1) / LR16_CHAINS) * LR16_WAVES : (N + per_wave - 1) / per_wave;
        } else {
            waves = (N + WINDOW_BS - 1) / WINDOW_BS;
        }
        h->acc_waves = waves;
        h->acc_slots = (int32_t)std::min<int64_t>(4096, std::max<int64_t>(64, (int64_t)(64ll << 20) / (waves * 8)));
        const size_t nb = (size_t)h->acc_slots * (size_t)waves * 2 * sizeof(unsigned int);
        CRCHK(dev_malloc(h->cfg.device_id, (void**)&h->d_acc, nb));
        CRCHK(hipMemsetAsync(h->d_acc, 0, nb, h->stream));
    }
    h->stage_cap = std::max<int64_t>(4096, (int64_t)d * (d + 1) + 64);
    CRCHK(host_malloc((void**)&h->d_stage, (size_t)h->stage_cap * sizeof(double)));
    CRCHK(hipStreamSynchronize(h->stream));   // the host vectors above go out of scope
#undef CRCHK
    *out = h;
    return DEMCZ_OK;
}

// Other replicas publish into this handle's archive from inside their launches: it may only be freed once none of them can
// still be running.  Replica group of this process: every member's streams are drained here and the group is marked dead (its
// other members then only accept demcz_destroy).  Ranks of a communicator: they meet in a reduction behind their launches
// (unless the communicator is dead -- then the archive is NOT freed: a surviving peer's publisher may still be writing it, the
// process is to be restarted anyway), and the IPC mappings of the peers' archives are closed.
static void peer_detach(demcz_handle* h)
{
    if (h->peer_mode == 1 && h->group) {
        PeerGroup* G = h->group;
        if (!G->dead) (void)group_verify(h);          // (what the members have enqueued is completed -- redone if need be -- first)
        for (demcz_handle* m : G->members) {
            if (m->stream) (void)hipStreamSynchronize(m->stream);
            if (m->prod_stream) (void)hipStreamSynchronize(m->prod_stream);
        }
        G->dead = true;
        G->members.erase(std::remove(G->members.begin(), G->members.end(), h), G->members.end());
        for (demcz_handle* m : G->members) { m->n_peers = 0; }
        if (G->members.empty()) delete G;
        h->group = nullptr;
    } else if (h->peer_mode == 3) {
        // (the host has made the ranks meet before this call: demcz_peer_export)
        for (int r = 0; r < DEMCZ_MAX_PEERS; ++r)
            if (h->ipc_mapped[r]) { (void)hipIpcCloseMemHandle(h->ipc_mapped[r]); h->ipc_mapped[r] = nullptr; }
    } else if (h->peer_mode == 2) {
        bool met = false;
        // (every rank destroys its handle at the same point of the program; one that does not -- a handle left to a garbage
        //  collector -- must not hold the others for the full communication deadline: five seconds, then the archive is leaked)
        h->comm_timeout_ms = (h->comm_timeout_ms > 0) ? std::min<int64_t>(h->comm_timeout_ms, 5000) : 5000;
        if (!h->comm_dead && h->comm && h->d_err_all) {
            if (ncclAllReduce(h->d_err_all + 1, h->d_err_all + 1, 1, ncclUint32, ncclMax, h->comm, h->stream) == ncclSuccess)
                met = sync_stream(h, h->stream, "demcz_destroy (peers)") == DEMCZ_OK;
        }
        for (int r = 0; r < DEMCZ_MAX_PEERS; ++r)
            if (h->ipc_mapped[r]) { (void)hipIpcCloseMemHandle(h->ipc_mapped[r]); h->ipc_mapped[r] = nullptr; }
        // ... and once more behind the closing: an exported allocation is only freed when no importer has it mapped any longer
        // (freeing it under a peer's mapping is undefined by the IPC contract, whatever current ROCm makes of it)
        if (met) {
            met = false;
            if (ncclAllReduce(h->d_err_all + 1, h->d_err_all + 1, 1, ncclUint32, ncclMax, h->comm, h->stream) == ncclSuccess)
                met = sync_stream(h, h->stream, "demcz_destroy (peers, mappings closed)") == DEMCZ_OK;
        }
        if (!met && h->archive_fine) h->dZ = nullptr;       // leaked on purpose (see above)
    }
    h->n_peers = 0;
    h->peer_mode = 0;
}

extern "C" int32_t demcz_destroy(demcz_handle* h)
{
    if (!h) return DEMCZ_OK;
    (void)hipSetDevice(h->cfg.device_id);
    // (side streams: a producer kernel for a launch that never came, monitoring checks, a batched exchange.  A sharded handle
    //  waits with its deadline here too -- a peer may have died -- and a dead one has already been given its time to drain)
    // (the copy stream too: a handle destroyed on an error path, or without demcz_get_history_view, may still have D2H copies
    //  of its history in flight into mirrors that are about to go back to the pool)
    for (hipStream_t st : {h->stream, h->prod_stream, h->diag_stream, h->comm_stream, h->hs_stream})
        if (st && !h->comm_dead) (void)sync_stream(h, st, "demcz_destroy");
    peer_detach(h);
    live_release(h);
    free_all(h);
    delete h;
    return DEMCZ_OK;
}

static TargetParams target_params(const demcz_handle* h)
{
    TargetParams tp;
    tp.mu = h->d_mu; tp.Wp = h->d_Wp; tp.c0 = h->cfg.c0;
    tp.design = h->d_design; tp.yobs = h->d_y; tp.nobs = h->cfg.nobs;
    tp.ngrp = h->ngrp; tp.goff = h->d_block_offsets; tp.gstart = h->gstart;
    return tp;
}

static int32_t ensure_scratch(demcz_handle* h, int64_t n)
{
    if (n <= h->scratch_cap) return DEMCZ_OK;
    { int32_t rcq = quiesce_all(h); if (rcq) return rcq; }
    SYNCCHK(h, h->stream);
    if (h->diag_stream) SYNCCHK(h, h->diag_stream);      // a check may still be reading the old buffer
    if (h->prod_stream) SYNCCHK(h, h->prod_stream);      // (the checks of a wave-per-chain handle run there)
    if (h->d_scratch) HIPCHK(h, dev_free(h->cfg.device_id, h->d_scratch));
    h->d_scratch = nullptr; h->scratch_cap = 0;
    HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->d_scratch, (size_t)n * sizeof(double)));
    h->scratch_cap = n;
    return DEMCZ_OK;
}

static int32_t launch_logp(demcz_handle* h, const double* X, int64_t ldX, int64_t n, double* out)
{
    const TargetParams tp = target_params(h);
    const int bs = 64;
    const dim3 grid((unsigned)((n + bs - 1) / bs));
    switch (h->cfg.target_kind) {
    case DEMCZ_TARGET_MVNORMAL:
        hipLaunchKernelGGL(logp_kernel<TARGET_MVNORMAL>, grid, dim3(bs), 0, h->stream, tp, h->cfg.d, X, ldX, n, out); break;
    case DEMCZ_TARGET_ISO_QUAD:
        hipLaunchKernelGGL(logp_kernel<TARGET_ISO_QUAD>, grid, dim3(bs), 0, h->stream, tp, h->cfg.d, X, ldX, n, out); break;
    case DEMCZ_TARGET_LINREG_SSE:
        hipLaunchKernelGGL(logp_kernel<TARGET_LINREG_SSE>, grid, dim3(bs), 0, h->stream, tp, h->cfg.d, X, ldX, n, out); break;
    default: return fail(h, DEMCZ_ERR_STATE, "host-callback target: pass logp explicitly");
    }
    HIPCHK(h, hipGetLastError());
    return DEMCZ_OK;
}

extern "C" int32_t demcz_set_state(demcz_handle* h, const double* X, const double* logp, const double* Z,
                                   int64_t ldZ, int64_t M0)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    DEADCHK(h);
    if (!X || !Z) return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_set_state: X and Z are required");
    if (M0 < 2) return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_set_state: M0 >= 2 needed (two distinct archive rows, demcz.jl:176-179)");
    if (M0 > h->cfg.Mcap || ldZ < M0) return fail(h, DEMCZ_ERR_CAPACITY, "demcz_set_state: M0 exceeds Mcap or ldZ < M0");
    if (!logp && h->cfg.target_kind == DEMCZ_TARGET_HOST_CALLBACK)
        return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_set_state: host-callback target needs logp");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int d = h->cfg.d;
    const int64_t N = h->cfg.N;
    HIPCHK(h, hipMemcpyAsync(h->dX, X, (size_t)N * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
    {   // parameter-major host matrix -> staging -> row-major archive
        int32_t rcz = ensure_scratch(h, M0 * d);
        if (rcz) return rcz;
        HIPCHK(h, hipMemcpy2DAsync(h->d_scratch, (size_t)M0 * sizeof(double), Z, (size_t)ldZ * sizeof(double),
                                   (size_t)M0 * sizeof(double), (size_t)d, hipMemcpyHostToDevice, h->stream));
        const int64_t tot = M0 * d;
        hipLaunchKernelGGL(append_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->dZ, h->ZS,
                           (int64_t)0, (const double*)h->d_scratch, M0, M0, d);
        HIPCHK(h, hipGetLastError());
        const size_t rest = (size_t)(h->cfg.Mcap - M0) * (size_t)h->ZS;
        if (rest > 0) {
            hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, h->stream,
                               reinterpret_cast<unsigned long long*>(h->dZ + (size_t)M0 * h->ZS), rest, LIVE_SENTINEL);
            HIPCHK(h, hipGetLastError());
        }
        HIPCHK(h, hipMemsetAsync(h->d_live_err, 0, 4 * sizeof(unsigned int), h->stream));
    }
    if (logp) {
        HIPCHK(h, hipMemcpyAsync(h->dlp, logp, (size_t)N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    } else {
        int32_t rc = launch_logp(h, h->dX, N, N, h->dlp);
        if (rc) return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->dlp_origin, h->dlp, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    SYNCCHK(h, h->stream);
    h->M = M0;
    h->M_app = M0;
    h->peer_fence = true;
    if (h->no_live && h->rearm_from >= 0) h->rearm_from = 0;      // (a new run numbers its generations from the start again)
    h->live_log.clear();
    h->snap_pending = false;
    h->acc_log.clear();
    { int32_t rcs = rec_scrub(h); if (rcs) return rcs; }
    for (auto& pe : h->pending) if (pe.ev) (void)hipEventDestroy(pe.ev);
    h->pending.clear();
    h->batch_cnt = 0; h->batch_J = -1;
    h->g_done = h->g0;
    h->origin_valid = true;
    h->has_state = true;
    h->proposal_pending = false;
    h->gen_open = false;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_get_state(demcz_handle* h, double* X, double* logp, double* Z, int64_t ldZ, int64_t* M)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    DEADCHK(h);
    if (!h->has_state) return fail(h, DEMCZ_ERR_STATE, "demcz_get_state: no state set");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    {
        int32_t rcv = live_verify(h);
        if (rcv) return rcv;
    }
    const int d = h->cfg.d;
    const int64_t N = h->cfg.N;
    if (X) HIPCHK(h, hipMemcpyAsync(X, h->dX, (size_t)N * d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (logp) HIPCHK(h, hipMemcpyAsync(logp, h->dlp, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (Z) {
        const int64_t Mall = h->M_app;       // every appended row, visible to proposals yet or not
        if (ldZ < Mall) return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_get_state: ldZ < M");
        int32_t rcz = flush_exchanges(h);
        if (rcz) return rcz;
        rcz = ensure_scratch(h, Mall * d);
        if (rcz) return rcz;
        const int64_t tot = Mall * d;
        hipLaunchKernelGGL(export_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->dZ,
                           h->ZS, (int64_t)0, h->d_scratch, Mall, Mall, d);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpy2DAsync(Z, (size_t)ldZ * sizeof(double), h->d_scratch, (size_t)Mall * sizeof(double),
                                   (size_t)Mall * sizeof(double), (size_t)d, hipMemcpyDeviceToHost, h->stream));
    }
    SYNCCHK(h, h->stream);
    if (M) *M = h->M_app;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_get_archive_pinned(demcz_handle* h, double** Z, int64_t* M)
{
    if (!h || !Z || !M) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->has_state) return fail(h, DEMCZ_ERR_STATE, "demcz_get_archive_pinned: no state set");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    int32_t rc = live_verify(h);
    if (rc) return rc;
    const int d = h->cfg.d;
    const int64_t Mall = h->M_app;
    rc = flush_exchanges(h);
    if (rc) return rc;
    rc = ensure_scratch(h, Mall * d);
    if (rc) return rc;
    double* host = nullptr;
    HIPCHK(h, host_malloc((void**)&host, (size_t)Mall * d * sizeof(double)));
    const int64_t tot = Mall * d;
    hipLaunchKernelGGL(export_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->dZ,
                       h->ZS, (int64_t)0, h->d_scratch, Mall, Mall, d);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(host, h->d_scratch, (size_t)tot * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess) {
        (void)host_free(host);
        return fail(h, DEMCZ_ERR_HIP, "demcz_get_archive_pinned: export failed");
    }
    rc = sync_stream(h, h->stream, "demcz_get_archive_pinned");
    if (rc) { (void)host_free(host); return rc; }
    *Z = host;
    *M = Mall;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_set_history_origin(demcz_handle* h, int64_t g0)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (g0 < 0) return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_set_history_origin: g0 >= 0");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (!h->live_log.empty()) {          // unverified LIVE launches wrote their history under the old origin
        int32_t rcv = live_verify(h);
        if (rcv) return rcv;
    }
    // slot 0's predecessor is the current log_obj if exactly g0 generations have been run
    h->origin_valid = h->has_state && (h->g_done == g0 || h->g_done == h->g0);
    if (h->has_state)
        HIPCHK(h, hipMemcpyAsync(h->dlp_origin, h->dlp, (size_t)h->cfg.N * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    if (h->has_state && h->g_done == h->g0) h->g_done = g0;      // nothing run yet: renumbering only
    h->g0 = g0;
    return DEMCZ_OK;
}

// ---- window launch dispatch ------------------------------------------------------------------
template <int TARGET, int D>
static void launch_window_d(const demcz_handle* h, const WindowParams& P, dim3 grid)
{
    if (h->full_block)
        hipLaunchKernelGGL((window_kernel<TARGET, D, true>), grid, dim3(WINDOW_BS), 0, h->stream, P);
    else
        hipLaunchKernelGGL((window_kernel<TARGET, D, false>), grid, dim3(WINDOW_BS), 0, h->stream, P);
}

template <int TARGET>
static void launch_window_generic(const demcz_handle* h, const WindowParams& P, dim3 grid)
{
    const size_t lds = (size_t)(3 * P.d + 1) * WINDOW_BS * sizeof(double);
    hipLaunchKernelGGL(window_kernel_generic<TARGET>, grid, dim3(WINDOW_BS), lds, h->stream, P);
}

template <int TARGET, int D, int L>
static void launch_window_ml(const demcz_handle* h, const WindowParams& P)
{
    constexpr int NG = 64 / L;      // chains per workgroup (one wave)
    const int wpw = h->wpw;
    hipLaunchKernelGGL((window_kernel_ml<TARGET, D, L>), dim3((unsigned)((P.N + NG * wpw - 1) / (NG * wpw))), dim3(64 * wpw), 0, h->stream, P);
}

// which multi-lane layout is compiled for (target, d, full single block): 0 = none
static int ml_lanes_available(int target_kind, int d, bool full_block, int64_t nobs, int max_blocklen = 0, int nblocks = 1)
{
    if (!full_block) {
        // block updates: every block-step must fit one Philox block per lane: 2 + ceil(b/2) <= L
        if (nblocks > MLB_MAX_BLOCKS) return 0;
        if (target_kind == DEMCZ_TARGET_MVNORMAL) {
            if ((d == 5 || d == 6 || d == 10) && 2 + (max_blocklen + 1) / 2 <= 8) return 8;
            if (d == 20 && 2 + (max_blocklen + 1) / 2 <= 16) return 16;
        }
        return 0;
    }
    if (target_kind == DEMCZ_TARGET_LINREG_SSE && d == 10 &&
        lr16_dynamic_lds<10>(nobs) <= ML_MAX_DYNAMIC_LDS)
        return 16;      // design + y resident in LDS
    // (round 5: any other dimension whose draws fit one Philox block per lane, 2 + ceil(d / 2) <= 16 -- window_kernel_ml with the
    //  sixteen lanes as the spec's sixteen partial sums, the design through L2)
    if (target_kind == DEMCZ_TARGET_LINREG_SSE && d >= 2 && d <= 28) return 16;
    if (target_kind == DEMCZ_TARGET_MVNORMAL) {
        if (d >= 2 && d <= 10) return 8;
        if (d == 20) return 16;
    }
    if (target_kind == DEMCZ_TARGET_ISO_QUAD && d == 10) return 8;
    return 0;
}

// Block updates at d = 20 (C3), split form: sixteen lanes per chain put four chains on a wave and ONE wave on a SIMD at C3's 4096
// chains.  Thirty-two lanes per chain (one parameter a lane, twelve lanes of a chain idle) make it two waves per SIMD that could
// fill each other's LDS waits -- measured (scripts/ab_c3.sh, DEMCZ_MLB_L32=1; bit-exact): 55.2 us per K-window against 45.4:
// the block-step is bound by the instructions it issues, not by their latencies, and the idle lanes' share of them is lost.
// Kept as an experiment switch, off by default.
static bool mlb32_wanted(int64_t)
{
    static const char* env = getenv("DEMCZ_MLB_L32");
    return env && atoi(env) != 0;
}

// split layout (demcz_kernels_pc.h)
static bool pc_available(int target_kind, int d, bool full_block)
{
    if (!full_block) return false;
    // measured per K-window at N=1024 (split vs 8 lanes per chain, fused): d=5 7.0 vs 11.4 us, d=8 7.8 vs 11.6,
    // d=10 11.1 vs 13.5
    if (target_kind == DEMCZ_TARGET_ISO_QUAD) return d == 10;
    // (d = 20: the replicated 210-coefficient whitening does not fit registers -- 63 us per window against
    //  16.7 for the 16-lanes-per-chain kernel, which stays the choice there)
    return target_kind == DEMCZ_TARGET_MVNORMAL && d >= 2 && d <= 10;
}

// the split form of the 16-lane layout (window_kernel_ml<.., REC>): where the replicated consumer does not fit
static bool split_ml_available(int target_kind, int d, bool full_block, int64_t nobs)
{
    if (!full_block) return false;
    if (target_kind == DEMCZ_TARGET_MVNORMAL) return d == 20;
    return target_kind == DEMCZ_TARGET_LINREG_SSE && d == 10 && lr16_dynamic_lds<10>(nobs) <= ML_MAX_DYNAMIC_LDS;
}

// one wave per chain: window_kernel_ps (d <= 5: a pass's draws are one DMA) / window_kernel_pw (C4's d = 20)
// (round 5: every dimension; the isotropic quadratic from d = 6 on -- window_kernel_pw evaluates both targets)
static bool ps_available(int target_kind, int d)
{
    if (target_kind == DEMCZ_TARGET_MVNORMAL) return d >= 2 && d <= PW_D_MAX;
    return target_kind == DEMCZ_TARGET_ISO_QUAD && d >= PW_D_MIN && d <= PW_D_MAX;
}

static int pc_roles(int d) { return ((d == 1) ? 1 : (d + 1) / 2) + 2; }
// doubles of draw record per (generation, chain), and producer lanes per (generation, chain)
static int64_t rec_fields(const demcz_handle* h) { return (h->split_kind == 3) ? 2 * h->S : (int64_t)h->cfg.d + 2; }
static int64_t rec_roles(const demcz_handle* h) { return (h->split_kind == 3) ? h->S : pc_roles(h->cfg.d); }

#ifdef DEMCZ_STAMPS
constexpr int64_t DEMCZ_STAMP_WGS = 1 << 16;
#endif

template <int TARGET, int D>
static void launch_ps(const demcz_handle* h, const WindowParams& P, int64_t blocks, bool live)
{
    const dim3 grid((unsigned)blocks), wg(64 * PS_CHAINS), wgl(64 * (PS_CHAINS + 1));     // LIVE: chain waves + publisher wave
    if (h->dual_now && ps2_applicable(h, P)) {      // ... with two chains to a wave
        ++h->kernel_counts[0];
        if (P.temperature) {
            if (live) hipLaunchKernelGGL((window_kernel_ps2d<TARGET, D, true, true>), grid, wgl, 0, h->stream, P);
            else hipLaunchKernelGGL((window_kernel_ps2d<TARGET, D, false, true>), grid, wg, 0, h->stream, P);
        } else {
            if (live) hipLaunchKernelGGL((window_kernel_ps2d<TARGET, D, true, false>), grid, wgl, 0, h->stream, P);
            else hipLaunchKernelGGL((window_kernel_ps2d<TARGET, D, false, false>), grid, wg, 0, h->stream, P);
        }
        return;
    }
    if (!h->ps_dual && ps2_applicable(h, P)) {               // the regular launch: the steady-state kernel
        ++h->kernel_counts[0];
        if (P.temperature) {
            if (live) hipLaunchKernelGGL((window_kernel_ps2<TARGET, D, true, true>), grid, wgl, 0, h->stream, P);
            else hipLaunchKernelGGL((window_kernel_ps2<TARGET, D, false, true>), grid, wg, 0, h->stream, P);
            return;
        }
        if (live) hipLaunchKernelGGL((window_kernel_ps2<TARGET, D, true, false>), grid, wgl, 0, h->stream, P);
        else hipLaunchKernelGGL((window_kernel_ps2<TARGET, D, false, false>), grid, wg, 0, h->stream, P);
        return;
    }
    if (P.temperature) {
        if (live) hipLaunchKernelGGL((window_kernel_ps<TARGET, D, true, true>), grid, wgl, 0, h->stream, P);
        else hipLaunchKernelGGL((window_kernel_ps<TARGET, D, false, true>), grid, wg, 0, h->stream, P);
    } else {
        if (live) hipLaunchKernelGGL((window_kernel_ps<TARGET, D, true, false>), grid, wgl, 0, h->stream, P);
        else hipLaunchKernelGGL((window_kernel_ps<TARGET, D, false, false>), grid, wg, 0, h->stream, P);
    }
}

// d = 20, MvNormal, LIVE: candidates and log-densities on the FP64 matrix instruction (demcz_kernels_pw.h, MF).  Built, bit-exact
// (tests/test_gpu_long_oracle.py) -- and measured 4 % SLOWER than the scalar form at C4's shard (6.55-6.60 against 6.2-6.4 us
// per K-window, interleaved runs, profiles/r04i_pw_mfma.txt): 36 matrix instructions at 65 clocks each are the vector rate,
// as DESIGN.md section 9 had costed.  Kept behind DEMCZ_PW_MFMA=1 as the measured experiment; off by default.
static bool pw_matrix_form(const demcz_handle* h)
{
    const char* on = getenv("DEMCZ_PW_MFMA");
    return on && atoi(on) != 0 && h->cfg.d == 20 && h->cfg.target_kind == DEMCZ_TARGET_MVNORMAL;
}

// regular LIVE launches (demcz_kernels_pw.h, REG): K, the distance to the first boundary and the length multiples of five
static bool pw_regular(const WindowParams& P, bool live)
{
    static const bool no_reg = getenv("DEMCZ_NO_PW_REG") != nullptr;
    return live && !no_reg && P.K % PS_R == 0 && P.to_boundary % PS_R == 0 && P.ngen % PS_R == 0 && P.ngen >= PS_R;
}

// window_kernel_pw (d = 6..32, MvNormal / isotropic quadratic): instantiated in translation units of its own, demcz_pw_dispatch.h
static int32_t launch_pw(demcz_handle* h, const WindowParams& P, int64_t blocks, bool live)
{
    const int target = (h->cfg.target_kind == DEMCZ_TARGET_ISO_QUAD) ? TARGET_ISO_QUAD : TARGET_MVNORMAL;
    const int form = (live && pw_matrix_form(h)) ? PW_FORM_MATRIX : pw_regular(P, live) ? PW_FORM_REGULAR : PW_FORM_GENERAL;
    if (pw_launch(target, P.d, live, P.temperature != nullptr, form, (unsigned)blocks, h->stream, P) != 0)
        return fail(h, DEMCZ_ERR_STATE, "split layout: dimension / target not built for the wave-per-chain consumer");
    return DEMCZ_OK;
}

template <int TARGET, int D>
static void launch_pc(const demcz_handle* h, const WindowParams& P, int64_t blocks, bool live)
{
    const dim3 grid((unsigned)blocks), wg(64), wgl(64 * PC8_LIVE_WAVES);     // LIVE: chains wave + publisher wave
    if (P.temperature) {      // tempered accept (demcz_anneal.jl:172-178): its own instantiation, no branch per generation
        if (live) hipLaunchKernelGGL((window_kernel_pc8<TARGET, D, true, true>), grid, wgl, 0, h->stream, P);
        else hipLaunchKernelGGL((window_kernel_pc8<TARGET, D, false, true>), grid, wg, 0, h->stream, P);
    } else {
        if (live) hipLaunchKernelGGL((window_kernel_pc8<TARGET, D, true, false>), grid, wgl, 0, h->stream, P);
        else hipLaunchKernelGGL((window_kernel_pc8<TARGET, D, false, false>), grid, wg, 0, h->stream, P);
    }
}

// live: the launch runs through K boundaries whose rows later generations of the SAME launch draw from
static int32_t launch_window_pc(demcz_handle* h, const WindowParams& P, bool live = false)
{
    const int64_t nbc = (P.N + 63) / 64;
    const bool lr_split = h->split_kind == 2 && h->cfg.target_kind == DEMCZ_TARGET_LINREG_SSE;
    // 64-lane producer units (the wave-per-chain and replicated consumers' producer, pc_produce, has two lane mappings)
    const bool pc_records = !(h->split_kind == 3);
    const int64_t units = pc_records ? produce_units(P.N, (int)rec_roles(h), P.rec_fields, P.next_ngen) : nbc * rec_roles(h) * P.next_ngen;
    // producer units per workgroup = waves per workgroup of the instantiation that is launched
    const int upw = lr_split ? LR16_WAVES : (h->split_kind == 4) ? PS_CHAINS + (live ? 1 : 0) : (h->split_kind == 3 || h->split_kind == 2) ? h->wpw : (h->split_kind == 1 && live) ? PC8_LIVE_WAVES : 1;
    const int64_t blocks = P.consumer_blocks + (units + upw - 1) / upw;
    if (blocks <= 0) return DEMCZ_OK;
    if (h->split_kind == 4 || h->lr_spec) {
        // consumers whose LIVE launches have the producer half as a kernel of its own beside them: one wave per chain (ps / pw),
        // and the regression target's eight-chains-per-workgroup kernel (its workgroups take a CU's LDS each: producer
        // workgroups of the same grid could only follow them)
        if (h->split_kind == 4 && P.ZS != ((P.d <= 2) ? 2 : (P.d <= 4) ? 4 : ((P.d + 7) / 8) * 8)) return fail(h, DEMCZ_ERR_STATE, "split layout: archive row stride");
        const int bin = (P.rec_in == h->d_rec[0]) ? 0 : 1, bout = (P.rec_out == h->d_rec[0]) ? 0 : 1;
        const bool one_launch = !live && P.consumer_blocks > 0;     // a short launch: producer workgroups ride in the consumer's grid
        if (units > 0 && one_launch && h->prod_pending[bout]) {
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->prod_done[bout], 0));      // (never two writers of one buffer)
            h->prod_pending[bout] = false;
        }
        if (units > 0 && !one_launch) {
            // the producer half.  Beside a LIVE consumer: on the side stream, behind the consumer that last read
            // the buffer it refills (= everything enqueued on the main stream so far).  Alone (records for THIS window,
            // needed at once): on the main stream.
            hipStream_t ps = h->stream;
            static const bool serial_env = getenv("DEMCZ_PRODUCE_SERIAL") != nullptr;     // diagnosis: producer in front of its consumer's successor, on the main stream
            // (members of a replica group of one process keep to ONE stream each: HIP multiplexes streams over a few hardware
            //  queues, demcz_peer_group has made sure the members' main streams can run at the same time, and a side stream of
            //  one member that shares a hardware queue with another member's main stream was seen to stall the group until the
            //  poll limit -- tests/test_gpu_peer.py after a session that had pooled 30 streams.  Ranks on GPUs of their own
            //  have no such neighbours.)
            if (P.consumer_blocks > 0 && !serial_env && h->peer_mode != 1) {
                if (!h->prod_stream) {
                    HIPCHK(h, stream_acquire(h->cfg.device_id, &h->prod_stream));
                    HIPCHK(h, hipEventCreateWithFlags(&h->prod_gate, hipEventDisableTiming));
                    for (int b = 0; b < 2; ++b) HIPCHK(h, hipEventCreateWithFlags(&h->prod_done[b], hipEventDisableTiming));
                }
                if (h->after_launch_ev) {
                    HIPCHK(h, hipStreamWaitEvent(h->prod_stream, h->after_launch_ev, 0));
                } else {
                    HIPCHK(h, hipEventRecord(h->prod_gate, h->stream));
                    HIPCHK(h, hipStreamWaitEvent(h->prod_stream, h->prod_gate, 0));
                }
                ps = h->prod_stream;
            } else if (h->prod_pending[bout]) {
                HIPCHK(h, hipStreamWaitEvent(h->stream, h->prod_done[bout], 0));      // (never two writers of one buffer)
                h->prod_pending[bout] = false;
            }
            const dim3 pg((unsigned)((units + PRODUCE_WAVES - 1) / PRODUCE_WAVES)), pw(64 * PRODUCE_WAVES);
            // Beside consumers the producers are throttled to PRODUCE_WGS_PER_CU workgroups a CU by an LDS allocation they
            // never touch: more of them only take issue slots from the chain waves (measured); alone they
            // take the whole chip.
            static const size_t throttle_env = getenv("DEMCZ_PRODUCE_LDS") ? (size_t)atol(getenv("DEMCZ_PRODUCE_LDS")) : PRODUCE_THROTTLE_LDS;
            // (regression target: what a consumer workgroup leaves of its CU's LDS holds ONE 8 KB producer workgroup)
            // (two chains to a wave: one 40 KB consumer workgroup per CU at 2048 chains leaves room for three 32 KB producer
            //  workgroups, and the producer has twice the draws to make per launch: 131 against 144 us per launch, profiles/r04j_dual.txt)
            // (wave-per-chain consumers at d > 20 take 70-110 KB of a CU's LDS themselves: the producers' allocation is what is left)
            size_t big_d = throttle_env;
            if (h->split_kind == 4 && P.d > 20) {
                const size_t clds = (size_t)pw_query(h->cfg.target_kind == DEMCZ_TARGET_ISO_QUAD ? TARGET_ISO_QUAD : TARGET_MVNORMAL, P.d, PW_QUERY_LIVE_LDS_BYTES);
                const size_t room = (clds + 4096 < ML_MAX_DYNAMIC_LDS) ? ML_MAX_DYNAMIC_LDS - clds - 4096 : 8192;
                big_d = std::max<size_t>(8192, std::min<size_t>(throttle_env, room));
            }
            const size_t dyn = (ps != h->stream) ? (h->lr_spec ? (size_t)8192 : (h->ps_dual && !getenv("DEMCZ_PRODUCE_LDS")) ? (size_t)32768 : big_d) : 0;
            switch (P.d) {
#define DEMCZ_PRODUCE_CASE(DD) case DD: hipLaunchKernelGGL((produce_kernel<DD>), pg, pw, dyn, ps, P); break;
            DEMCZ_PRODUCE_CASE(2) DEMCZ_PRODUCE_CASE(3) DEMCZ_PRODUCE_CASE(4) DEMCZ_PRODUCE_CASE(5) DEMCZ_PRODUCE_CASE(6) DEMCZ_PRODUCE_CASE(7)
            DEMCZ_PRODUCE_CASE(8) DEMCZ_PRODUCE_CASE(9) DEMCZ_PRODUCE_CASE(10) DEMCZ_PRODUCE_CASE(11) DEMCZ_PRODUCE_CASE(12) DEMCZ_PRODUCE_CASE(13)
            DEMCZ_PRODUCE_CASE(14) DEMCZ_PRODUCE_CASE(15) DEMCZ_PRODUCE_CASE(16) DEMCZ_PRODUCE_CASE(17) DEMCZ_PRODUCE_CASE(18) DEMCZ_PRODUCE_CASE(19)
            DEMCZ_PRODUCE_CASE(20) DEMCZ_PRODUCE_CASE(21) DEMCZ_PRODUCE_CASE(22) DEMCZ_PRODUCE_CASE(23) DEMCZ_PRODUCE_CASE(24) DEMCZ_PRODUCE_CASE(25)
            DEMCZ_PRODUCE_CASE(26) DEMCZ_PRODUCE_CASE(27) DEMCZ_PRODUCE_CASE(28) DEMCZ_PRODUCE_CASE(29) DEMCZ_PRODUCE_CASE(30) DEMCZ_PRODUCE_CASE(31)
            DEMCZ_PRODUCE_CASE(32)
#undef DEMCZ_PRODUCE_CASE
            default: return fail(h, DEMCZ_ERR_STATE, "split layout: dimension not built");
            }
            HIPCHK(h, hipGetLastError());
            if (ps != h->stream) {
                HIPCHK(h, hipEventRecord(h->prod_done[bout], ps));
                h->prod_pending[bout] = true;
            }
        }
        if (P.consumer_blocks > 0) {
            if (h->prod_pending[bin]) {            // this launch's records were made on the side stream
                // A blocking caller (demcz_run_checked) waits for them HERE, on the host: the producer finishes half a launch
                // before the consumer that is running does, so the host is still ahead of the GPU -- and the window launch goes
                // into the queue with nothing in front of it.  A wait enqueued while its event is still pending is a barrier
                // packet between two window kernels (~ 10 us of an idle queue per 1000-generation step, scripts/probes/fixed_cost.py).
                if (h->host_paced) { int32_t rcw = sync_event(h, h->prod_done[bin], "launch_window_pc"); if (rcw) return rcw; }
                else HIPCHK(h, hipStreamWaitEvent(h->stream, h->prod_done[bin], 0));
                h->prod_pending[bin] = false;
            }
            static_assert(PS_CHAINS == LR16_WAVES, "producer units per workgroup of a one-launch grid");
            const int64_t grid = P.consumer_blocks + (one_launch ? (units + PS_CHAINS - 1) / PS_CHAINS : 0);
            if (h->lr_spec) {
                const size_t dynl = lr8s_dynamic_lds<10>(P.tp.nobs);
                if (!h->lds_raised_spec) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_kernel_lr8s<10, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ML_MAX_DYNAMIC_LDS);
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_kernel_lr8s<10, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ML_MAX_DYNAMIC_LDS);
                    h->lds_raised_spec = true;
                }
                const dim3 g((unsigned)grid), wgl(64 * LR16_WAVES);
                if (live) hipLaunchKernelGGL((window_kernel_lr8s<10, true>), g, wgl, dynl, h->stream, P);
                else hipLaunchKernelGGL((window_kernel_lr8s<10, false>), g, wgl, dynl, h->stream, P);
            } else
            switch (P.d) {
            case 2: launch_ps<TARGET_MVNORMAL, 2>(h, P, grid, live); break;
            case 3: launch_ps<TARGET_MVNORMAL, 3>(h, P, grid, live); break;
            case 4: launch_ps<TARGET_MVNORMAL, 4>(h, P, grid, live); break;
            case 5: launch_ps<TARGET_MVNORMAL, 5>(h, P, grid, live); break;
            default: { int32_t rcw = launch_pw(h, P, grid, live); if (rcw) return rcw; } break;
            }
        }
    } else if (h->split_kind == 3) {
        const dim3 grid((unsigned)blocks), wg(64 * h->wpw);
#define DEMCZ_LAUNCH_MLB_REC(DD, LL)                                                                                         \
        do {                                                                                                                 \
            if (h->ngrp > 1) {      /* sums cut at the block boundaries (full evaluation: the group-start mask) */              \
                if (live) hipLaunchKernelGGL((window_kernel_mlb<TARGET_MVNORMAL, DD, LL, true, true, 0, true>), grid, wg, 0, h->stream, P);   \
                else hipLaunchKernelGGL((window_kernel_mlb<TARGET_MVNORMAL, DD, LL, true, false, 0, true>), grid, wg, 0, h->stream, P);       \
            }                                                                                                                \
            else if (live) hipLaunchKernelGGL((window_kernel_mlb<TARGET_MVNORMAL, DD, LL, true, true>), grid, wg, 0, h->stream, P);   \
            else hipLaunchKernelGGL((window_kernel_mlb<TARGET_MVNORMAL, DD, LL, true, false>), grid, wg, 0, h->stream, P);       \
        } while (0)
        switch (P.d) {
        case 5: DEMCZ_LAUNCH_MLB_REC(5, 8); break;
        case 6: DEMCZ_LAUNCH_MLB_REC(6, 8); break;
        case 10: DEMCZ_LAUNCH_MLB_REC(10, 8); break;
        case 20:
            if (h->split_lanes == 32) DEMCZ_LAUNCH_MLB_REC(20, 32);
            else if (h->mlb_qb == 5) {      // four blocks of five: the incremental form
                if (live) hipLaunchKernelGGL((window_kernel_mlb<TARGET_MVNORMAL, 20, 16, true, true, 5>), grid, wg, 0, h->stream, P);
                else hipLaunchKernelGGL((window_kernel_mlb<TARGET_MVNORMAL, 20, 16, true, false, 5>), grid, wg, 0, h->stream, P);
            }
            else DEMCZ_LAUNCH_MLB_REC(20, 16);
            break;
        default: return fail(h, DEMCZ_ERR_STATE, "split layout: dimension not built");
        }
#undef DEMCZ_LAUNCH_MLB_REC
    } else if (lr_split) {
        const dim3 grid((unsigned)blocks), wg(64 * LR16_WAVES);
        const size_t dyn = lr16_dynamic_lds<10>(P.tp.nobs);
        if (!h->lds_raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_kernel_lr16<10, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ML_MAX_DYNAMIC_LDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_kernel_lr16<10, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ML_MAX_DYNAMIC_LDS);
            h->lds_raised = true;
        }
        if (live) hipLaunchKernelGGL((window_kernel_lr16<10, true, true>), grid, wg, dyn, h->stream, P);
        else hipLaunchKernelGGL((window_kernel_lr16<10, true, false>), grid, wg, dyn, h->stream, P);
    } else if (h->split_kind == 2) {
        const dim3 grid((unsigned)blocks), wg(64 * h->wpw);
        if (live) hipLaunchKernelGGL((window_kernel_ml<TARGET_MVNORMAL, 20, 16, true, true>), grid, wg, 0, h->stream, P);
        else hipLaunchKernelGGL((window_kernel_ml<TARGET_MVNORMAL, 20, 16, true, false>), grid, wg, 0, h->stream, P);
    } else if (h->cfg.target_kind == DEMCZ_TARGET_MVNORMAL) {
        switch (P.d) {
        case 2: launch_pc<TARGET_MVNORMAL, 2>(h, P, blocks, live); break;
        case 3: launch_pc<TARGET_MVNORMAL, 3>(h, P, blocks, live); break;
        case 4: launch_pc<TARGET_MVNORMAL, 4>(h, P, blocks, live); break;
        case 5: launch_pc<TARGET_MVNORMAL, 5>(h, P, blocks, live); break;
        case 6: launch_pc<TARGET_MVNORMAL, 6>(h, P, blocks, live); break;
        case 7: launch_pc<TARGET_MVNORMAL, 7>(h, P, blocks, live); break;
        case 8: launch_pc<TARGET_MVNORMAL, 8>(h, P, blocks, live); break;
        case 9: launch_pc<TARGET_MVNORMAL, 9>(h, P, blocks, live); break;
        case 10: launch_pc<TARGET_MVNORMAL, 10>(h, P, blocks, live); break;
        default: return fail(h, DEMCZ_ERR_STATE, "split layout: dimension not built");
        }
    } else {
        launch_pc<TARGET_ISO_QUAD, 10>(h, P, blocks, live);
    }
    HIPCHK(h, hipGetLastError());
    return DEMCZ_OK;
}

// Records of the launch (g .. g+ngen-1 against M rows) are in d_rec[rec_cur] when this returns: either
// the previous launch's producer half made them, or a producer-only launch is enqueued now.  Then
// `P` is completed so that this launch's producer half prepares (next_g, next_ngen, next_M).
// Both record buffers hold `gens` generations per (field, chain) afterwards.  demcz_run reserves the most a
// launch of this handle can need before it starts timing or launching (so no allocation, synchronisation or
// memset ever sits between two window launches); pc_prepare only grows them if a caller outruns that.
static int32_t rec_reserve(demcz_handle* h, int64_t gens)
{
    if (gens <= h->rec_cap) return DEMCZ_OK;
    { int32_t rcq = quiesce_all(h); if (rcq) return rcq; }
    SYNCCHK(h, h->stream);
    if (h->prod_stream) SYNCCHK(h, h->prod_stream);
    h->prod_pending[0] = h->prod_pending[1] = false;
    if (h->arena && gens <= h->arena_gens && h->rec_cap == 0) {
        // the arena's buffers (zeroed at create): next to the archive, so that window_kernel_ps2 reaches them by 32-bit offsets
        for (int b = 0; b < 2; ++b) { h->d_rec[b] = h->arena_rec[b]; h->rec_desc[b].valid = false; }
        h->rec_in_arena = true;
        h->rec_cap = h->arena_gens;
        return DEMCZ_OK;
    }
    for (int b = 0; b < 2; ++b) {
        if (h->d_rec[b] && !h->rec_in_arena) HIPCHK(h, dev_free(h->cfg.device_id, h->d_rec[b]));
        h->d_rec[b] = nullptr;
        // (+ REC_PAD doubles: a consumer's last 16-byte chunk fetch may run past its row's last generation)
        const size_t nd = (size_t)gens * (size_t)rec_fields(h) * h->cfg.N + REC_PAD;
        HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->d_rec[b], nd * sizeof(double)));
        HIPCHK(h, hipMemsetAsync(h->d_rec[b], 0, nd * sizeof(double), h->stream));   // row 0: always a legal index
        h->rec_desc[b].valid = false;
    }
    h->rec_in_arena = false;
    h->rec_cap = gens;
    return DEMCZ_OK;
}

// A launch window_kernel_ps2 can take (demcz_kernels_ps2.h): regular passes, everything reachable by 32-bit offsets.
static bool ps2_applicable(const demcz_handle* h, const WindowParams& P)
{
    static const bool off = getenv("DEMCZ_NO_PS2") != nullptr;
    if (off || h->split_kind != 4 || P.d < 2 || P.d > 5 || !h->arena || !h->rec_in_arena) return false;
    if (P.temperature && (P.temperature < h->arena_temp || P.temperature >= h->arena_temp + h->arena_gens)) return false;
    if (P.K % PS2_R != 0 || P.to_boundary % PS2_R != 0 || P.ngen % PS2_R != 0 || P.ngen < PS2_R) return false;
    if (P.chain && (!h->hist_joint || (double)h->cfg.N * (h->cfg.d + 1) * (double)h->cfg.Gcap * 8.0 >= 4293918720.0)) return false;
    return true;
}

static int32_t pc_prepare(demcz_handle* h, WindowParams& P, int64_t cur_rows, int64_t next_g, int64_t next_ngen, int64_t next_M,
                          int64_t next_rows, int32_t next_boff)
{
    {
        int32_t rcr = rec_reserve(h, std::max<int64_t>(P.ngen, next_ngen));
        if (rcr) return rcr;
    }
    P.rec_stride = h->rec_cap;
    const int cur = h->rec_cur;
    const int32_t cur_boff = P.K - P.to_boundary;
    auto& dc = h->rec_desc[cur];
    // the row indices in a record were drawn against M + (boundaries passed) * rows: all of that must agree
    if (!(dc.valid && dc.g_first == P.g_first && dc.ngen >= P.ngen && dc.M == P.M && dc.rows == cur_rows && dc.boff == cur_boff)) {
        WindowParams Q = P;                 // producer-only launch for THIS window
        Q.consumer_blocks = 0;
        Q.rec_out = h->d_rec[cur];
        Q.next_g_first = P.g_first; Q.next_ngen = P.ngen; Q.next_M = P.M; Q.next_rows = cur_rows; Q.next_boff = cur_boff;
        int32_t rc = launch_window_pc(h, Q);
        if (rc) return rc;
        dc.valid = true; dc.g_first = P.g_first; dc.M = P.M; dc.ngen = P.ngen; dc.rows = cur_rows; dc.boff = cur_boff;
    }
    P.rec_in = h->d_rec[cur];
    const int per_wg = (h->ps_dual && !h->dual_now) ? PS_CHAINS : h->split_per_wg;      // (an irregular launch of a two-chain handle: one chain per wave)
    P.consumer_blocks = (int32_t)((P.N + per_wg - 1) / per_wg);
    P.rec_out = h->d_rec[cur ^ 1];
    P.next_g_first = next_g; P.next_ngen = (int32_t)std::max<int64_t>(next_ngen, 0); P.next_M = next_M;
    P.next_rows = next_rows; P.next_boff = next_boff;
    auto& dn = h->rec_desc[cur ^ 1];
    dn.valid = next_ngen > 0; dn.g_first = next_g; dn.M = next_M; dn.ngen = (int32_t)std::max<int64_t>(next_ngen, 0);
    dn.rows = next_rows; dn.boff = next_boff;
    h->rec_cur = cur ^ 1;
    return DEMCZ_OK;
}

template <int TARGET, int D, int L>
static void launch_window_mlb(const demcz_handle* h, const WindowParams& P)
{
    constexpr int G = 64 / L;
    if (h->ngrp > 1) hipLaunchKernelGGL((window_kernel_mlb<TARGET, D, L, false, false, 0, true>), dim3((unsigned)((P.N + G - 1) / G)), dim3(64), 0, h->stream, P);
    else hipLaunchKernelGGL((window_kernel_mlb<TARGET, D, L>), dim3((unsigned)((P.N + G - 1) / G)), dim3(64), 0, h->stream, P);
}

// the regression target on the matrix-instruction kernels (demcz_kernels_lr.h): d = 10 with design + y resident in LDS; any other
// regression shape with sixteen lanes per chain runs window_kernel_ml<LINREG_SSE, d, 16>
static bool uses_lr16(const demcz_handle* h)
{
    return h->cfg.target_kind == DEMCZ_TARGET_LINREG_SSE && h->full_block && h->cfg.d == 10 &&
           lr16_dynamic_lds<10>(h->cfg.nobs) <= ML_MAX_DYNAMIC_LDS;
}

// the regression target on sixteen lanes per chain: helper waves share the chain wave's log-density (demcz_kernels_ml.h, COOP) while
// the residuals of a workgroup's four chains fit its LDS
static bool ml_coop(const demcz_handle* h)
{
    static const bool off = getenv("DEMCZ_NO_ML_COOP") != nullptr;
    return ML_LRDPP && !off && h->cfg.target_kind == DEMCZ_TARGET_LINREG_SSE && h->full_block && h->lanes == 16 && !uses_lr16(h) &&
           h->cfg.nobs <= ML_COOP_MAX_OBS;
}

static bool try_launch_ml(const demcz_handle* h, const WindowParams& P)
{
    if (h->lanes <= 1 || h->lanes == DEMCZ_LAYOUT_SPLIT) return false;
    const int d = P.d;
    if (!h->full_block) {
        if (h->cfg.target_kind != DEMCZ_TARGET_MVNORMAL) return false;
        switch (d) {
        case 5: launch_window_mlb<TARGET_MVNORMAL, 5, 8>(h, P); return true;
        case 6: launch_window_mlb<TARGET_MVNORMAL, 6, 8>(h, P); return true;
        case 10: launch_window_mlb<TARGET_MVNORMAL, 10, 8>(h, P); return true;
        case 20:
            if (h->mlb_qb == 5) hipLaunchKernelGGL((window_kernel_mlb<TARGET_MVNORMAL, 20, 16, false, false, 5>), dim3((unsigned)((P.N + 3) / 4)), dim3(64), 0, h->stream, P);
            else launch_window_mlb<TARGET_MVNORMAL, 20, 16>(h, P);
            return true;
        }
        return false;
    }
    if (h->cfg.target_kind == DEMCZ_TARGET_MVNORMAL) {
        switch (d) {
        case 2: launch_window_ml<TARGET_MVNORMAL, 2, 8>(h, P); return true;
        case 3: launch_window_ml<TARGET_MVNORMAL, 3, 8>(h, P); return true;
        case 4: launch_window_ml<TARGET_MVNORMAL, 4, 8>(h, P); return true;
        case 5: launch_window_ml<TARGET_MVNORMAL, 5, 8>(h, P); return true;
        case 6: launch_window_ml<TARGET_MVNORMAL, 6, 8>(h, P); return true;
        case 7: launch_window_ml<TARGET_MVNORMAL, 7, 8>(h, P); return true;
        case 8: launch_window_ml<TARGET_MVNORMAL, 8, 8>(h, P); return true;
        case 9: launch_window_ml<TARGET_MVNORMAL, 9, 8>(h, P); return true;
        case 10: launch_window_ml<TARGET_MVNORMAL, 10, 8>(h, P); return true;
        case 20: launch_window_ml<TARGET_MVNORMAL, 20, 16>(h, P); return true;
        }
    } else if (h->cfg.target_kind == DEMCZ_TARGET_ISO_QUAD && d == 10) {
        launch_window_ml<TARGET_ISO_QUAD, 10, 8>(h, P);
        return true;
    } else if (h->cfg.target_kind == DEMCZ_TARGET_LINREG_SSE && !uses_lr16(h)) {
        // sixteen lanes per chain at any dimension 2..28 (demcz_mlr_dispatch.h: the kernels live in translation units of their own)
        const bool coop = ml_coop(h);
        const int waves = h->wpw;
        const unsigned blocks = coop ? (unsigned)((P.N + 3) / 4) : (unsigned)((P.N + 4 * waves - 1) / (4 * waves));
        return mlr_launch(d, coop, blocks, waves, h->stream, P) == 0;
    } else if (h->cfg.target_kind == DEMCZ_TARGET_LINREG_SSE && d == 10) {
        {   // the regression target: 16 chains per workgroup of four waves on the 16x16x4 FP64 matrix instruction
            const size_t dyn = lr16_dynamic_lds<10>(P.tp.nobs);
            if (!h->lds_raised) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_kernel_lr16<10, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ML_MAX_DYNAMIC_LDS);
                h->lds_raised = true;
            }
            hipLaunchKernelGGL((window_kernel_lr16<10, false, false>), dim3((unsigned)((P.N + LR16_CHAINS - 1) / LR16_CHAINS)), dim3(64 * LR16_WAVES),
                               dyn, h->stream, P);
        }
        return true;
    }
    return false;
}

static int32_t launch_window(demcz_handle* h, const WindowParams& P, bool live = false)
{
    if (P.consumer_blocks > 0 || h->lanes != DEMCZ_LAYOUT_SPLIT) {      // (not the producer-only launches)
        h->last_live = live ? 1 : 0;
        h->last_temper = P.temperature ? 1 : 0;
        h->last_ps2 = (h->lanes == DEMCZ_LAYOUT_SPLIT && h->split_kind == 4 && !h->lr_spec && ps2_applicable(h, P) && (!h->ps_dual || h->dual_now)) ? 1 : 0;
        h->last_dual = h->dual_now ? 1 : 0;
        h->last_pw_reg = (h->lanes == DEMCZ_LAYOUT_SPLIT && h->split_kind == 4 && P.d > 5 && !pw_matrix_form(h) && pw_regular(P, live)) ? 1 : 0;
    }
    if (h->snap_pending) {
        // the redo snapshot of the state (demcz_run): window_kernel_ps2 writes it as it loads the state -- two 5 us copy launches
        // less in front of every autostop slab -- any other kernel gets the copies
        h->snap_pending = false;
        if (h->lanes == DEMCZ_LAYOUT_SPLIT && !h->lr_spec && ps2_applicable(h, P) && (!h->ps_dual || h->dual_now)) {
            WindowParams Q = P;
            Q.safe_X = h->d_safe_X;
            Q.safe_lp = h->d_safe_lp;
            return launch_window(h, Q, live);
        }
        HIPCHK(h, hipMemcpyAsync(h->d_safe_X, h->dX, (size_t)h->cfg.N * h->cfg.d * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_safe_lp, h->dlp, (size_t)h->cfg.N * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    const dim3 grid((unsigned)((P.N + WINDOW_BS - 1) / WINDOW_BS));
    const int d = P.d;
    if (h->lanes == DEMCZ_LAYOUT_SPLIT) {
        int32_t rc = launch_window_pc(h, P, live);
        if (rc) return rc;
        ++h->launches;
        return DEMCZ_OK;
    }
    if (try_launch_ml(h, P)) {
        HIPCHK(h, hipGetLastError());
        ++h->launches;
        return DEMCZ_OK;
    }
    switch (h->cfg.target_kind) {
    case DEMCZ_TARGET_MVNORMAL:
        switch (d) {
        case 2: launch_window_d<TARGET_MVNORMAL, 2>(h, P, grid); break;
        case 3: launch_window_d<TARGET_MVNORMAL, 3>(h, P, grid); break;
        case 4: launch_window_d<TARGET_MVNORMAL, 4>(h, P, grid); break;
        case 5: launch_window_d<TARGET_MVNORMAL, 5>(h, P, grid); break;
        case 8: launch_window_d<TARGET_MVNORMAL, 8>(h, P, grid); break;
        case 10: launch_window_d<TARGET_MVNORMAL, 10>(h, P, grid); break;
        case 20: launch_window_d<TARGET_MVNORMAL, 20>(h, P, grid); break;
        default: launch_window_generic<TARGET_MVNORMAL>(h, P, grid); break;
        }
        break;
    case DEMCZ_TARGET_ISO_QUAD:
        switch (d) {
        case 10: launch_window_d<TARGET_ISO_QUAD, 10>(h, P, grid); break;
        default: launch_window_generic<TARGET_ISO_QUAD>(h, P, grid); break;
        }
        break;
    case DEMCZ_TARGET_LINREG_SSE:
        switch (d) {
        case 10: launch_window_d<TARGET_LINREG_SSE, 10>(h, P, grid); break;
        case 26: launch_window_d<TARGET_LINREG_SSE, 26>(h, P, grid); break;
        default: launch_window_generic<TARGET_LINREG_SSE>(h, P, grid); break;
        }
        break;
    default: return fail(h, DEMCZ_ERR_STATE, "demcz_run: host-callback target uses demcz_propose/accept_commit");
    }
    HIPCHK(h, hipGetLastError());
    ++h->launches;
    return DEMCZ_OK;
}

// Synchronous exchange (lag == 0) of a sharded run: all-gather the end-of-window states and
// scatter them into every replica of the archive; the next window sees them.
static int32_t append_after_window(demcz_handle* h)
{
    const int d = h->cfg.d;
    const int64_t N = h->cfg.N;
    if (h->comm) {   // also at nranks == 1, so the collective path is exercised on a one-GPU box
        const int64_t total = N * h->nranks;
        if (h->M_app + total > h->cfg.Mcap) return fail(h, DEMCZ_ERR_CAPACITY, "demcz_run: Z capacity exceeded");
        { int32_t rcs = maybe_stall(h, h->stream); if (rcs) return rcs; }
        NCCLCHK(h, ncclAllGather(h->dX, h->d_gather, (size_t)N * d, ncclDouble, h->comm, h->stream));
        const int64_t tot = total * d;
        hipLaunchKernelGGL(append_gathered_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->dZ,
                           h->ZS, h->M_app, h->d_gather, N, h->nranks, d);
        HIPCHK(h, hipGetLastError());
        h->M_app += total;
        h->M = h->M_app;
    }
    return DEMCZ_OK;
}

// Deferred exchange (lag E >= 1) of a sharded run: the snapshots of up to E boundaries travel in one
// all-gather on a side stream while the next windows compute; the rows become visible E windows after
// the batch closes (see demcz_set_append_lag), by which time the compute stream waits on the event.
static int32_t exchange_batch(demcz_handle* h)
{
    if (h->batch_cnt == 0) return DEMCZ_OK;
    const int d = h->cfg.d, cnt = h->batch_cnt, buf = h->batch_buf;
    const int64_t N = h->cfg.N;
    hipEvent_t ready;
    HIPCHK(h, hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(ready, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->comm_stream, ready, 0));
    HIPCHK(h, hipEventDestroy(ready));
    { int32_t rcs = maybe_stall(h, h->comm_stream); if (rcs) return rcs; }
    NCCLCHK(h, ncclAllGather(h->d_send[buf], h->d_recv[buf], (size_t)N * d * cnt, ncclDouble, h->comm_side, h->comm_stream));
    const int64_t tot = N * d * cnt * h->nranks;
    hipLaunchKernelGGL(append_batch_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->comm_stream, h->dZ, h->ZS,
                       h->batch_base, (const double*)h->d_recv[buf], N, h->nranks, cnt, d);
    HIPCHK(h, hipGetLastError());
    hipEvent_t done;
    HIPCHK(h, hipEventCreateWithFlags(&done, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(done, h->comm_stream));
    h->pending.back().ev = done;                       // the entry that covers this batch
    h->pending.back().xseq = ++h->xseq;
    if (!h->xdone) {
        HIPCHK(h, host_malloc((void**)&h->xdone, sizeof(long long)));
        *h->xdone = 0;
    }
    hipLaunchKernelGGL(mark_kernel, dim3(1), dim3(1), 0, h->comm_stream, h->xdone, (long long)h->xseq);
    HIPCHK(h, hipGetLastError());
    h->buf_xseq[buf] = h->xseq;
    HIPCHK(h, hipEventRecord(h->buf_done[buf], h->comm_stream));
    h->batch_cnt = 0;
    h->batch_buf ^= 1;
    h->batch_J = -1;                                   // later boundaries of the same batch open a new entry
    return DEMCZ_OK;
}

// Rows whose visibility generation has been reached join the part of the archive proposals draw from.
static int32_t admit_pending(demcz_handle* h, int64_t g)
{
    while (!h->pending.empty() && h->pending.front().visible_from <= g) {
        auto pe = h->pending.front();
        if (pe.ev) {
            // (a blocking caller waits on the host: see launch_window_pc -- the exchange of a batch has the whole next batch to
            //  finish, and the batch after that is only enqueued now)
            if (h->host_paced && h->xdone) {
                volatile long long* xd = h->xdone;
                const long long need = (long long)pe.xseq;
                int32_t rcw = wait_deadline(h, [xd, need]() { return (*xd >= need) ? hipSuccess : hipErrorNotReady; }, "admit_pending (exchange of a batch)");
                if (rcw) return rcw;
            }
            else HIPCHK(h, hipStreamWaitEvent(h->stream, pe.ev, 0));
            HIPCHK(h, hipEventDestroy(pe.ev));
            h->xseq_waited = std::max(h->xseq_waited, pe.xseq);      // (the side stream runs its exchanges in order)
        }
        h->M = pe.M_after;
        h->pending.pop_front();
    }
    return DEMCZ_OK;
}

// Everything appended so far is in the archive when this returns (used before reading Z back).
static int32_t flush_exchanges(demcz_handle* h)
{
    if (h->comm && h->lag > 0) {
        int32_t rc = exchange_batch(h);
        if (rc) return rc;
        for (auto& pe : h->pending)
            if (pe.ev) HIPCHK(h, hipStreamWaitEvent(h->stream, pe.ev, 0));
    }
    return DEMCZ_OK;
}

// A LIVE launch that gave up waiting for a row leaves a word behind; results after it are void.
static int32_t check_live_err(demcz_handle* h)
{
    unsigned int e[4] = {0, 0, 0, 0};
    if (!h->d_live_err) return DEMCZ_OK;         // (the scratch handles of the *_array diagnostics run no chains)
    if (h->err_clean) return DEMCZ_OK;           // (read as zero since the last LIVE launch: a blocking 16-byte copy less per call)
    HIPCHK(h, hipMemcpy(e, h->d_live_err, sizeof(e), hipMemcpyDeviceToHost));
    if (!e[0]) h->err_clean = true;
    if (e[0])
        return fail(h, DEMCZ_ERR_HIP, "demcz_run: an archive row appended inside the launch never became visible (LIVE hand-off): "
                                      "generation " + std::to_string(e[1]) + " of the launch, row " + std::to_string(e[2]) +
                                      ", workgroup " + std::to_string(e[3]) + ", rows appended " + std::to_string(h->M_app));
    return DEMCZ_OK;
}

// What a failed hand-off leaves behind is undone: X, log_obj, M go back to the state before the first unverified
// demcz_run call, the rows written since read as unwritten again, the error word is cleared, and the handle stops
// using LIVE launches.  Returns the calls made since (for live_verify to redo).
static int32_t live_rollback(demcz_handle* h, std::vector<demcz_handle::RunCall>& calls)
{
    calls.swap(h->live_log);
    h->live_log.clear();
    const int64_t N = h->cfg.N;
    const int d = h->cfg.d;
    SYNCCHK(h, h->stream);
    if (h->diag_stream) SYNCCHK(h, h->diag_stream);
    if (h->prod_stream) SYNCCHK(h, h->prod_stream);
    if (!h->snap_pending) {              // (still pending: nothing was launched since, the state is the snapshot)
        HIPCHK(h, hipMemcpyAsync(h->dX, h->d_safe_X, (size_t)N * d * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->dlp, h->d_safe_lp, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    h->snap_pending = false;
    if (h->M_app > h->safe_M_app) {
        const size_t rest = (size_t)(h->M_app - h->safe_M_app) * (size_t)h->ZS;
        hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, h->stream,
                           reinterpret_cast<unsigned long long*>(h->dZ + (size_t)h->safe_M_app * h->ZS), rest, LIVE_SENTINEL);
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipMemsetAsync(h->d_live_err, 0, 4 * sizeof(unsigned int), h->stream));
    SYNCCHK(h, h->stream);
    h->peer_fence = true;
    // Where the redo may go LIVE again: behind the demcz_run call that holds the generation whose row never arrived.  The row
    // says which boundary it belongs to -- rows [safe_M_app + j * rows, + rows) are boundary j since the verified point, appended
    // after generation (safe_g_done / K + 1 + j) * K -- and the wave that gave up was working on a generation behind it.  No row
    // on record (the launch found the word already set; a kernel that does not note it): behind everything enqueued so far.
    int64_t from = h->g_done;
    if (h->fail_row_hint != 0xffffffffu && (int64_t)h->fail_row_hint >= h->safe_M_app && (int64_t)h->fail_row_hint < h->M_app) {
        const int64_t rows = h->cfg.N * ((h->comm || h->peer_mode != 0) ? h->nranks : 1);
        const int64_t j = ((int64_t)h->fail_row_hint - h->safe_M_app) / rows;
        from = std::min<int64_t>(from, (h->safe_g_done / h->cfg.K + 1 + j) * (int64_t)h->cfg.K + 1);
    }
    h->fail_row_hint = 0xffffffffu;
    h->M = h->safe_M;
    h->M_app = h->safe_M_app;
    h->g_done = h->safe_g_done;
    while (!h->acc_log.empty() && h->acc_log.back().g_last > h->safe_g_done) h->acc_log.pop_back();
    { int32_t rcs = rec_scrub(h); if (rcs) return rcs; }
    h->no_live = true;
    h->rearm_from = -1;
    if (h->peer_mode != 1 && h->peer_mode != 3 && h->live_rearms_left > 0) {      // (a replica group re-arms as a whole: group_verify)
        --h->live_rearms_left;
        h->rearm_from = from;
    }
    // (a fault a test injected has fired: the redo's LIVE launches run with the handle's own poll limit)
    h->live_fault_polls = 0;
    live_release(h);
    ++h->live_redos;
    return DEMCZ_OK;
}

// The handle tries LIVE launches again (see demcz_handle::live_rearms_left).  Called at a point where everything enqueued so far
// ran one launch per K-window -- nothing of it can have failed -- and every rank of a sharded run is at the same call with the
// same decision to try (it follows from the max-reduced error word and the min-reduced row of live_failed).  Whether a rank CAN
// -- its share of the device's LIVE budget may have gone to another handle of its process meanwhile -- is agreed by a
// min-reduction: either all ranks hand rows over inside their launches again or none does (a rank exchanging through
// ncclAllGather while its peers publish and poll would stall all of them until the communicator's deadline).  The ranks meet
// once more, stream-ordered, before the first launch that publishes (peer_fence, set by the rollback).
static int32_t live_try_rearm(demcz_handle* h)
{
    h->rearm_from = -1;
    h->no_live = false;
    unsigned int ok = live_span(h) > 0 ? 1u : 0u;
    if (h->peer_mode == 2 && h->comm && h->d_err_all) {
        HIPCHK(h, hipMemcpyAsync(h->d_err_all + 2, &ok, sizeof(ok), hipMemcpyHostToDevice, h->stream));
        NCCLCHK(h, ncclAllReduce(h->d_err_all + 2, h->d_err_all + 2, 1, ncclUint32, ncclMin, h->comm, h->stream));
        SYNCCHK(h, h->stream);
        HIPCHK(h, hipMemcpy(&ok, h->d_err_all + 2, sizeof(ok), hipMemcpyDeviceToHost));
    }
    if (!ok) {
        h->no_live = true;
        live_release(h);
        return DEMCZ_OK;
    }
    ++h->live_rearms;
    if (getenv("DEMCZ_DEBUG_LIVE")) fprintf(stderr, "[demcz] rank %d: LIVE launches re-armed (%d re-arms left)\n", h->rank, (int)h->live_rearms_left);
    return DEMCZ_OK;
}

// Did a LIVE launch since the last verified point give up waiting for a row?  The stream is drained first.  With the ranks of a
// communicator publishing into each other's replicas (peer_mode 2) the answer must be the SAME on every rank -- a rank whose own
// waits all succeeded still has to join the others' redo, which exchanges rows through RCCL -- so the error words are
// max-reduced over the communicator (every rank makes the same calls in the same order, so every rank is here together).
static int32_t live_failed(demcz_handle* h, bool& failed)
{
    failed = false;
    SYNCCHK(h, h->stream);
    unsigned int e[4] = {0, 0, 0, 0};
    if (h->peer_mode == 2 && !h->no_live && h->comm) {
        NCCLCHK(h, ncclAllReduce(h->d_live_err, h->d_err_all, 1, ncclUint32, ncclMax, h->comm, h->stream));
        SYNCCHK(h, h->stream);
        HIPCHK(h, hipMemcpy(e, h->d_err_all, sizeof(unsigned int), hipMemcpyDeviceToHost));
    } else if (h->pinned_err && h->pinned_err_launches == h->launches) {
        // (copied behind the last window launch, and the stream has been drained since: demcz_run_checked)
        std::memcpy(e, h->pinned_err, sizeof(e));
    } else {
        HIPCHK(h, hipMemcpy(e, h->d_live_err, sizeof(e), hipMemcpyDeviceToHost));
    }
    h->pinned_err_launches = -1;
    failed = e[0] != 0u;
    if (!failed && h->peer_mode != 2) h->err_clean = true;
    if (failed) {
        // which row: what live_rollback places the re-arming point by.  Ranks of a communicator need the SAME point: the smallest
        // row any of them has on record (0xffffffff: none on this rank -- its own waits all succeeded, or the wave that gave up
        // could not say).  Every rank knows `failed` from the reduction above, so every rank is in this one too.
        unsigned int row = 0xffffffffu;
        if (h->peer_mode == 2 && !h->no_live && h->comm) {
            unsigned int le[4] = {0, 0, 0, 0};
            HIPCHK(h, hipMemcpy(le, h->d_live_err, sizeof(le), hipMemcpyDeviceToHost));
            if (le[0] && le[0] == 1u) row = le[2];
            HIPCHK(h, hipMemcpyAsync(h->d_err_all + 2, &row, sizeof(row), hipMemcpyHostToDevice, h->stream));
            NCCLCHK(h, ncclAllReduce(h->d_err_all + 2, h->d_err_all + 2, 1, ncclUint32, ncclMin, h->comm, h->stream));
            SYNCCHK(h, h->stream);
            HIPCHK(h, hipMemcpy(&row, h->d_err_all + 2, sizeof(row), hipMemcpyDeviceToHost));
        } else if (e[0] == 1u) {           // (the word a wave's compare-and-swap wrote; a test's pre-set word has no row behind it)
            row = e[2];
        }
        h->fail_row_hint = row;
        if (getenv("DEMCZ_DEBUG_LIVE")) {
            unsigned int le[4] = {0, 0, 0, 0};
            (void)hipMemcpy(le, h->d_live_err, sizeof(le), hipMemcpyDeviceToHost);
            fprintf(stderr, "[demcz] rank %d: hand-off failed: word %#x, generation %u of its launch, row %u, workgroup %u; verified point: g %lld, M %lld; enqueued: g %lld, M %lld; launches %lld\n",
                    h->rank, le[0], le[1], le[2], le[3], (long long)h->safe_g_done, (long long)h->safe_M_app, (long long)h->g_done, (long long)h->M_app, (long long)h->launches);
        }
    }
    return DEMCZ_OK;
}

// Every entry point that hands results to the caller goes through here: the stream is drained, and if a LIVE
// launch since the last verification gave up waiting for a row, everything since then is redone with one launch
// per K-window (bit-identical results; the handle stays in that mode).  DEMCZ_OK afterwards means the results are valid.
static int32_t live_verify(demcz_handle* h)
{
    DEADCHK(h);
    if (h->peer_mode == 1) return group_verify(h);
    if (h->live_log.empty() || h->replaying) return check_live_err(h);
    bool failed = false;
    int32_t rc = live_failed(h, failed);
    if (rc) return rc;
    if (!failed) { h->live_log.clear(); return DEMCZ_OK; }
    if (h->peer_mode == 3) {
        h->no_live = true;
        return fail(h, DEMCZ_ERR_STATE, "a row another rank should have published never became visible (in-launch hand-off, host-mediated IPC peers): "
                                        "results since the last verified point are void on every rank");
    }
    // The redo: back to the verified point, the logged calls again.  They run one launch per K-window up to the call that holds
    // the generation whose row never came; a call that starts behind it may go LIVE again (live_try_rearm, in demcz_run) -- its
    // launches are logged like any others, against the SAME verified point (`replaying`: no new snapshot), and should one of them
    // fail too the whole list is redone once more.  Bounded: every failure takes one of the handle's re-arms, and without one
    // left the redo cannot fail.
    std::vector<demcz_handle::RunCall> calls, dropped;
    rc = live_rollback(h, calls);
    if (rc) return rc;
    for (;;) {
        h->replaying = true;
        for (const auto& c : calls) {
            rc = demcz_run(h, c.g_from, c.g_to, c.gamma, c.tempered ? c.temperature.data() : nullptr);
            if (rc) break;
        }
        h->replaying = false;
        if (rc) return rc;
        if (h->live_log.empty()) break;
        rc = live_failed(h, failed);
        if (rc) return rc;
        if (!failed) { h->live_log.clear(); return DEMCZ_OK; }
        rc = live_rollback(h, dropped);
        if (rc) return rc;
    }
    SYNCCHK(h, h->stream);
    return check_live_err(h);
}

// ---- replica groups of one process (demcz_peer_group) --------------------------------------------------------------------------
// The calls of `calls[m]` (the same list for every member) executed for all members in lockstep, one launch per K-window, the
// boundary rows of all members appended to every replica in rank order from the members' own state buffers: what the
// in-launch hand-off does, without it.  The host waits for every phase -- this is the fall-back, not the product path.
static int32_t group_execute(PeerGroup* G, const std::vector<std::vector<demcz_handle::RunCall>>& calls)
{
    const size_t ncall = calls.empty() ? 0 : calls[0].size();
    for (const auto& cm : calls)
        if (cm.size() != ncall) return DEMCZ_ERR_STATE;
    auto sync_all = [&]() -> int32_t {
        for (demcz_handle* m : G->members) {
            if (hipStreamSynchronize(m->stream) != hipSuccess) return fail(m, DEMCZ_ERR_HIP, "replica group: stream synchronisation failed");
            if (m->prod_stream && hipStreamSynchronize(m->prod_stream) != hipSuccess) return fail(m, DEMCZ_ERR_HIP, "replica group: stream synchronisation failed");
        }
        return DEMCZ_OK;
    };
    int32_t rc = DEMCZ_OK;
    for (demcz_handle* m : G->members) { m->replaying = true; m->external_append = true; }
    for (size_t ic = 0; ic < ncall && rc == DEMCZ_OK; ++ic) {
        const auto& c0 = calls[0][ic];
        for (size_t mi = 0; mi < G->members.size(); ++mi)
            if (calls[mi][ic].g_from != c0.g_from || calls[mi][ic].g_to != c0.g_to) rc = DEMCZ_ERR_STATE;
        if (rc) break;
        const int K = G->members[0]->cfg.K;
        for (int64_t g = c0.g_from; g <= c0.g_to && rc == DEMCZ_OK;) {
            const int64_t w_end = std::min<int64_t>(((g - 1) / K + 1) * (int64_t)K, c0.g_to);
            for (size_t mi = 0; mi < G->members.size() && rc == DEMCZ_OK; ++mi) {
                const auto& c = calls[mi][ic];
                rc = demcz_run(G->members[mi], g, w_end, c.gamma, c.tempered ? c.temperature.data() + (g - c.g_from) : nullptr);
            }
            if (rc == DEMCZ_OK) rc = sync_all();
            if (rc == DEMCZ_OK && w_end % K == 0) {
                for (demcz_handle* m : G->members)
                    for (demcz_handle* src : G->members) {
                        if (rc == DEMCZ_OK) rc = demcz_append_rows_device(m, src->dX, src->cfg.N, src->cfg.N);
                        if (rc != DEMCZ_OK && m != src && m->err.empty()) m->err = src->err;
                    }
                if (rc == DEMCZ_OK) rc = sync_all();
            }
            g = w_end + 1;
        }
    }
    for (demcz_handle* m : G->members) { m->replaying = false; m->external_append = false; m->rec_desc[0].valid = m->rec_desc[1].valid = false; }
    return rc;
}

// Verification for a member of a replica group = for the whole group (one host thread drives all members, and a member's
// results depend on every other member's launches having run): all streams are drained, all error words looked at.
static int32_t group_verify(demcz_handle* h)
{
    PeerGroup* G = h->group;
    if (!G || G->busy || h->replaying) return check_live_err(h);
    if (G->dead) {      // (a member is gone; it verified the whole group on its way out, so what is there is valid)
        if (h->live_log.empty()) return check_live_err(h);
        return fail(h, DEMCZ_ERR_STATE, "a member of this handle's replica group has been destroyed");
    }
    bool any_log = false;
    for (demcz_handle* m : G->members) any_log = any_log || !m->live_log.empty();
    if (!any_log) return check_live_err(h);
    for (demcz_handle* m : G->members)
        if (m->live_log.size() != h->live_log.size())
            return fail(h, DEMCZ_ERR_STATE, "replica group: every member must be given the same demcz_run calls before any member's results are asked for");
    G->busy = true;
    struct Unbusy { PeerGroup* g; ~Unbusy() { g->busy = false; } } unbusy{G};
    bool failed = false;
    for (demcz_handle* m : G->members) {
        HIPCHK(m, hipStreamSynchronize(m->stream));
        if (m->prod_stream) HIPCHK(m, hipStreamSynchronize(m->prod_stream));
    }
    for (demcz_handle* m : G->members) {
        unsigned int e[4] = {0, 0, 0, 0};
        HIPCHK(m, hipMemcpy(e, m->d_live_err, sizeof(e), hipMemcpyDeviceToHost));
        failed = failed || e[0] != 0u;
        if (e[0] && getenv("DEMCZ_DEBUG_LIVE"))
            fprintf(stderr, "[demcz] replica %d of %d: hand-off timed out at generation %u of its launch, row %u, workgroup %u (M_app %lld, M %lld, launches %lld, live_claimed %d)\n",
                    m->rank, m->nranks, e[1], e[2], e[3], (long long)m->M_app, (long long)m->M, (long long)m->launches, (int)m->live_claimed);
    }
    if (!G->failed && !failed) {
        for (demcz_handle* m : G->members) m->live_log.clear();
        return DEMCZ_OK;
    }
    std::vector<std::vector<demcz_handle::RunCall>> calls(G->members.size());
    if (!G->failed) {
        // first failure: every member goes back to the state before its first unverified call
        for (size_t mi = 0; mi < G->members.size(); ++mi) {
            int32_t rc = live_rollback(G->members[mi], calls[mi]);
            if (rc) { if (G->members[mi] != h) h->err = G->members[mi]->err; return rc; }
        }
        G->failed = true;
    } else {
        // the group already runs deferred: the logged calls have not been executed at all
        for (size_t mi = 0; mi < G->members.size(); ++mi) { calls[mi].swap(G->members[mi]->live_log); G->members[mi]->live_log.clear(); }
    }
    int32_t rc = group_execute(G, calls);
    if (rc) {
        if (h->err.empty()) for (demcz_handle* m : G->members) if (!m->err.empty()) { h->err = m->err; break; }
        if (rc == DEMCZ_ERR_STATE && h->err.empty()) h->err = "replica group: the members were not given the same calls";
        return rc;
    }
    for (demcz_handle* m : G->members) m->g_done = calls[0].empty() ? m->g_done : calls[0].back().g_to;
    if (G->failed && !G->lockstep_only && G->rearms_left > 0) {
        // everything logged has been executed in lockstep, every member is at the same verified point: the group hands its rows
        // over inside the launches again from the next call on (bounded: see demcz_handle::live_rearms_left)
        --G->rearms_left;
        G->failed = false;
        for (demcz_handle* m : G->members) { m->no_live = false; m->rearm_from = -1; m->live_rearms_left = G->rearms_left; ++m->live_rearms; }
    }
    return check_live_err(h);
}

// The consumer workgroups of a LIVE launch wait for each other's rows, so all of them must be resident at
// once.  Capacity = what the occupancy query gives for the LIVE instantiation of this handle's kernel
// (both accept variants) x CUs; half of it is used, which leaves room for the producer half, for other
// streams' kernels and for the query being one block per CU optimistic (MI355X_MICROARCH.md, residency).
// Beyond that the split layout falls back to one launch per K-window.
template <int TARGET, int D>
static int pc_live_blocks_per_cu()
{
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void*>(&window_kernel_pc8<TARGET, D, true, false>), 64 * PC8_LIVE_WAVES, 0) != hipSuccess) a = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void*>(&window_kernel_pc8<TARGET, D, true, true>), 64 * PC8_LIVE_WAVES, 0) != hipSuccess) b = 0;
    return std::min(a, b);
}

template <int D>
static int ps2_live_blocks_per_cu()
{
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void*>(&window_kernel_ps2<TARGET_MVNORMAL, D, true, false>), 64 * (PS_CHAINS + 1), 0) != hipSuccess) a = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void*>(&window_kernel_ps2<TARGET_MVNORMAL, D, true, true>), 64 * (PS_CHAINS + 1), 0) != hipSuccess) b = 0;
    return std::min(a, b);
}

template <int D>
static int ps_live_blocks_per_cu()
{
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void*>(&window_kernel_ps<TARGET_MVNORMAL, D, true, false>), 64 * (PS_CHAINS + 1), 0) != hipSuccess) a = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void*>(&window_kernel_ps<TARGET_MVNORMAL, D, true, true>), 64 * (PS_CHAINS + 1), 0) != hipSuccess) b = 0;
    return std::min(a, b);
}

template <int D>
static int ps2d_live_blocks_per_cu()
{
    int a = 0, b = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void*>(&window_kernel_ps2d<TARGET_MVNORMAL, D, true, false>), 64 * (PS_CHAINS + 1), 0) != hipSuccess) a = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void*>(&window_kernel_ps2d<TARGET_MVNORMAL, D, true, true>), 64 * (PS_CHAINS + 1), 0) != hipSuccess) b = 0;
    return std::min(a, b);
}

static int64_t live_wg_capacity(demcz_handle* h)
{
    if (h->live_wg_cap >= 0) return h->live_wg_cap;
    int per_cu = 0;
    if (h->split_kind == 4 && h->ps_dual) {      // (its LIVE launches are all of the two-chain kernel)
        switch (h->cfg.d) {
        case 2: per_cu = ps2d_live_blocks_per_cu<2>(); break;
        case 3: per_cu = ps2d_live_blocks_per_cu<3>(); break;
        case 4: per_cu = ps2d_live_blocks_per_cu<4>(); break;
        case 5: per_cu = ps2d_live_blocks_per_cu<5>(); break;
        default: per_cu = 0;
        }
    } else if (h->split_kind == 4) {
        switch (h->cfg.d) {
        case 2: per_cu = std::min(ps_live_blocks_per_cu<2>(), ps2_live_blocks_per_cu<2>()); break;
        case 3: per_cu = std::min(ps_live_blocks_per_cu<3>(), ps2_live_blocks_per_cu<3>()); break;
        case 4: per_cu = std::min(ps_live_blocks_per_cu<4>(), ps2_live_blocks_per_cu<4>()); break;
        case 5: per_cu = std::min(ps_live_blocks_per_cu<5>(), ps2_live_blocks_per_cu<5>()); break;
        default: per_cu = pw_query(h->cfg.target_kind == DEMCZ_TARGET_ISO_QUAD ? TARGET_ISO_QUAD : TARGET_MVNORMAL, h->cfg.d, PW_QUERY_LIVE_BLOCKS_PER_CU);
        }
        // window_kernel_pw at d > 20: ONE workgroup per CU (its registers, from d = 23 on its LDS too: 86-111 KB).  A launch of
        // 1024 chains is then a workgroup on every CU, and what else runs on the chip beside it (producer workgroups, the R-hat
        // kernels: finite, they wait for nothing) can delay a consumer workgroup's start but not prevent it -- the same argument
        // as for the regression kernel's full-LDS workgroups below.  No halving there (undone by the doubling).
        if (h->cfg.d > 20 && per_cu == 1) per_cu = 2;
    } else if (h->split_kind == 3) {
        const void* f = nullptr;
        switch (h->cfg.d) {
#define DEMCZ_MLB_FN(DD, LL) ((h->ngrp > 1) ? reinterpret_cast<const void*>(&window_kernel_mlb<TARGET_MVNORMAL, DD, LL, true, true, 0, true>) \
                                            : reinterpret_cast<const void*>(&window_kernel_mlb<TARGET_MVNORMAL, DD, LL, true, true>))
        case 5: f = DEMCZ_MLB_FN(5, 8); break;
        case 6: f = DEMCZ_MLB_FN(6, 8); break;
        case 10: f = DEMCZ_MLB_FN(10, 8); break;
        case 20: f = (h->split_lanes == 32) ? DEMCZ_MLB_FN(20, 32)
                     : (h->mlb_qb == 5) ? reinterpret_cast<const void*>(&window_kernel_mlb<TARGET_MVNORMAL, 20, 16, true, true, 5>)
                                        : DEMCZ_MLB_FN(20, 16); break;
#undef DEMCZ_MLB_FN
        }
        if (!f || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f, 64 * h->wpw, 0) != hipSuccess) per_cu = 0;
    } else if (h->split_kind == 2 && h->lr_spec) {
        // Every consumer workgroup takes a CU's LDS for itself (the design matrix): the launch's 256 workgroups are resident
        // together iff every CU is there for them, and a CU's other tenants (producer workgroups of the same grid come behind
        // the consumers and need the same LDS; the R-hat kernels need none) cannot take it from them.  No halving here.
        const size_t dyn = lr8s_dynamic_lds<10>(h->cfg.nobs);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_kernel_lr8s<10, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ML_MAX_DYNAMIC_LDS);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&window_kernel_lr8s<10, true>), 64 * LR16_WAVES, dyn) != hipSuccess) per_cu = 0;
        per_cu *= 2;        // (undoes the halving below)
    } else if (h->split_kind == 2 && h->cfg.target_kind == DEMCZ_TARGET_LINREG_SSE) {
        const size_t dyn = lr16_dynamic_lds<10>(h->cfg.nobs);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_kernel_lr16<10, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ML_MAX_DYNAMIC_LDS);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&window_kernel_lr16<10, true, true>), 64 * LR16_WAVES, dyn) != hipSuccess) per_cu = 0;
    } else if (h->split_kind == 2) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&window_kernel_ml<TARGET_MVNORMAL, 20, 16, true, true>), 64 * h->wpw, 0) != hipSuccess) per_cu = 0;
    } else if (h->cfg.target_kind == DEMCZ_TARGET_ISO_QUAD) {
        per_cu = pc_live_blocks_per_cu<TARGET_ISO_QUAD, 10>();
    } else {
        switch (h->cfg.d) {
        case 2: per_cu = pc_live_blocks_per_cu<TARGET_MVNORMAL, 2>(); break;
        case 3: per_cu = pc_live_blocks_per_cu<TARGET_MVNORMAL, 3>(); break;
        case 4: per_cu = pc_live_blocks_per_cu<TARGET_MVNORMAL, 4>(); break;
        case 5: per_cu = pc_live_blocks_per_cu<TARGET_MVNORMAL, 5>(); break;
        case 6: per_cu = pc_live_blocks_per_cu<TARGET_MVNORMAL, 6>(); break;
        case 7: per_cu = pc_live_blocks_per_cu<TARGET_MVNORMAL, 7>(); break;
        case 8: per_cu = pc_live_blocks_per_cu<TARGET_MVNORMAL, 8>(); break;
        case 9: per_cu = pc_live_blocks_per_cu<TARGET_MVNORMAL, 9>(); break;
        case 10: per_cu = pc_live_blocks_per_cu<TARGET_MVNORMAL, 10>(); break;
        default: per_cu = 0;
        }
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device_id) != hipSuccess) cus = 0;
    h->live_wg_cap = (int64_t)per_cu * cus / 2;
    return h->live_wg_cap;
}

// Generations one LIVE launch of the split layout may span (0: not applicable -- other layouts, sharded
// runs, deferred visibility, appends owned by the caller, more consumer workgroups than may wait for each
// other).  Bounded by the draw records it needs.
// The waves of a LIVE launch wait for each other, so the residency rule above must hold for everything the
// process has in flight on the device: ONE handle per device uses LIVE launches at a time (the first to ask keeps
// the slot until it is destroyed); other handles on that device run one launch per K-window -- same results.
// (Another PROCESS on the GPU is outside this rule; the bounded poll + the automatic redo in demcz_run_checked /
//  check_live_err cover it.)
static std::mutex g_live_mutex;
static int64_t g_live_used[64] = {0};      // per-mille of each device's LIVE capacity held by handles of this process

// Round 4: a budget instead of one owner per device.  A handle claims the share of ITS kernel's capacity that its consumer
// workgroups are (capacities differ by kernel: registers, LDS); claims of a device add up to at most the whole.  That is what
// lets the R replicas of a sharded run share one GPU (demcz_peer_group), each with 1/R of the chains, and lets two small
// independent samplers both keep their LIVE launches.
static bool live_claim(demcz_handle* h)
{
    if (h->live_claimed) return true;
    const int dev = h->cfg.device_id;
    if (dev < 0 || dev >= 64) return false;
    const int64_t cap = live_wg_capacity(h);
    const int64_t wgs = (h->cfg.N + h->split_per_wg - 1) / h->split_per_wg;
    if (cap <= 0 || wgs > cap) return false;
    const int64_t share = std::max<int64_t>(1, (wgs * 1000 + cap - 1) / cap);
    std::lock_guard<std::mutex> lk(g_live_mutex);
    if (g_live_used[dev] + share > 1000) return false;
    g_live_used[dev] += share;
    h->live_share = share;
    h->live_claimed = true;
    return true;
}

static void live_release(demcz_handle* h)
{
    if (!h->live_claimed) return;
    std::lock_guard<std::mutex> lk(g_live_mutex);
    const int dev = h->cfg.device_id;
    if (dev >= 0 && dev < 64) g_live_used[dev] = std::max<int64_t>(0, g_live_used[dev] - h->live_share);
    h->live_share = 0;
    h->live_claimed = false;
}

// A handle that becomes a peer gives up the two-chains-to-a-wave form (window_kernel_ps2d).  That form only takes REGULAR
// launches; its irregular ones (a start or a tail that is not a multiple of five, temperatures outside the arena, a history of
// 4 GiB or more) run the general kernel with one chain per wave -- twice the waves, which are not all resident, so never LIVE --
// and a non-LIVE launch appends N rows per boundary at M_append + b * N: it knows nothing of `brows` / `row_off`, publishes
// nothing to the peers and waits for nothing of theirs, while the host advances M by N * shards (the replicas'
// archives would silently hold sentinel rows read as data).  With one chain per wave every launch of a peer is either LIVE or --
// when the shard's chains are more than a LIVE launch holds (N > 1024 per shard at d <= 5) -- the handle runs the exchange path
// (RCCL all-gather; a replica group: lockstep), which appends all shards' rows itself.  Same results either way.
static void peer_no_dual(demcz_handle* h)
{
    if (!h->ps_dual) return;
    live_release(h);
    h->ps_dual = false;
    h->dual_now = false;
    h->split_per_wg = PS_CHAINS;
    h->live_wg_cap = -1;
    rec_invalidate(h);
}

// layouts whose LIVE consumers re-read a missing row through live_reload (system scope when there are peers)
static bool peer_capable(const demcz_handle* h)
{
    return h->lanes == DEMCZ_LAYOUT_SPLIT && h->split_kind != 1 && h->split_kind != 0;
}

static int64_t live_span(demcz_handle* h)
{
    // (sharded: only with the rows handed over inside the launches -- peer_mode 2; otherwise a launch ends at the exchange)
    if (h->lanes != DEMCZ_LAYOUT_SPLIT || (h->comm && h->peer_mode != 2) || h->lag > 0 || h->external_append || h->no_live) return 0;
    if (h->peer_mode != 0 && !peer_capable(h)) return 0;
    if (h->peer_mode == 1 && (!h->group || h->group->failed || h->group->dead)) return 0;
    static const bool disabled = (getenv("DEMCZ_NO_LIVE") != nullptr);     // safety valve: one launch per K-window
    if (disabled) return 0;
    const int per_wg = h->split_per_wg;
    if ((h->cfg.N + per_wg - 1) / per_wg > live_wg_capacity(h)) return 0;
    if (!live_claim(h)) return 0;
    const int64_t per_gen = rec_fields(h) * h->cfg.N * (int64_t)sizeof(double);
    // 64 MiB of records per buffer inside the arena (C2: 1170 generations, more than an autostop slab), 1 GiB outside it (C4's
    // shard 5960 generations instead of 372, C5 5450 instead of 340, C3 820 instead of 51; allocated as launches need it): every
    // launch boundary costs 15-25 us of gap, first-pass latency and tail imbalance -- C4's shard 7.4 -> 6.1 us per K-window,
    // C5 22.3 -> 21.5 (profiles/r03f_launch_span.txt)
    static const int64_t env_mib = getenv("DEMCZ_REC_MIB") ? atol(getenv("DEMCZ_REC_MIB")) : 0;
    // (block updates, split kind 3: flat between 64 and 256 MiB, 4 % slower at 1 GiB -- 128)
    const int64_t mib = env_mib > 0 ? env_mib : (h->arena ? (h->ps_dual ? 128 : 64) : (h->split_kind == 3) ? 128 : 1024);
    const int64_t span = (int64_t)(mib << 20) / per_gen;
    return std::max<int64_t>(h->cfg.K, std::min<int64_t>(span, 1 << 20));
}

// Streamed history: generations g_from..g_to (already enqueued on the compute stream) are copied to the pinned host mirrors on
// the copy stream, behind an event -- while the compute stream goes on with the next slab.
static int32_t stream_history(demcz_handle* h, int64_t g_from, int64_t g_to)
{
    if (!h->hs_on || h->cfg.Gcap <= 0 || g_to < g_from) return DEMCZ_OK;
    const int64_t N = h->cfg.N, s0 = g_from - h->g0 - 1, G = g_to - g_from + 1;
    const int d = h->cfg.d;
    if (s0 < 0 || s0 + G > h->cfg.Gcap) return DEMCZ_OK;
    hipEvent_t ev = h->after_launch_ev;
    if (!ev) {
        if (!h->diag_ev) HIPCHK(h, hipEventCreateWithFlags(&h->diag_ev, hipEventDisableTiming));
        HIPCHK(h, hipEventRecord(h->diag_ev, h->stream));
        h->after_launch_ev = ev = h->diag_ev;
    }
    HIPCHK(h, hipStreamWaitEvent(h->hs_stream, ev, 0));
    HIPCHK(h, hipMemcpyAsync(h->hs_chain + (size_t)N * d * s0, h->dchain + (size_t)N * d * s0, (size_t)N * d * G * sizeof(double), hipMemcpyDeviceToHost, h->hs_stream));
    HIPCHK(h, hipMemcpyAsync(h->hs_logobj + (size_t)N * s0, h->dlogobj + (size_t)N * s0, (size_t)N * G * sizeof(double), hipMemcpyDeviceToHost, h->hs_stream));
    return DEMCZ_OK;
}

extern "C" int32_t demcz_history_stream(demcz_handle* h, int32_t enabled)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    DEADCHK(h);
    if (h->cfg.Gcap <= 0) return fail(h, DEMCZ_ERR_STATE, "demcz_history_stream: handle keeps no history (Gcap = 0)");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (enabled && !h->hs_on) {
        const size_t nc = (size_t)h->cfg.N * h->cfg.d * h->cfg.Gcap * sizeof(double), nl = (size_t)h->cfg.N * h->cfg.Gcap * sizeof(double);
        if (!h->hs_chain) HIPCHK(h, g_host_pool.acquire((void**)&h->hs_chain, nc, -1));
        if (!h->hs_logobj) HIPCHK(h, g_host_pool.acquire((void**)&h->hs_logobj, nl, -1));
        if (!h->hs_stream) HIPCHK(h, stream_acquire(h->cfg.device_id, &h->hs_stream));
    }
    h->hs_on = enabled != 0;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_get_history_view(demcz_handle* h, int64_t g_from, int64_t g_to, double** chain, double** log_obj)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    int32_t rc = check_hist_range(h, g_from, g_to, "demcz_get_history_view");
    if (rc) return rc;
    if (!h->hs_on || !h->hs_chain) return fail(h, DEMCZ_ERR_STATE, "demcz_get_history_view: call demcz_history_stream(h, 1) before the run");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    rc = live_verify(h);                       // (a voided LIVE slab is redone here, and the redo streams its generations again)
    if (rc) return rc;
    rc = sync_stream(h, h->hs_stream, "demcz_get_history_view");
    if (rc) return rc;
    const int64_t s0 = g_from - h->g0 - 1;
    if (chain) *chain = h->hs_chain + (size_t)h->cfg.N * h->cfg.d * s0;
    if (log_obj) *log_obj = h->hs_logobj + (size_t)h->cfg.N * s0;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_detach_history(demcz_handle* h, void** chain_base, void** logobj_base)
{
    if (!h || !chain_base || !logobj_base) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->hs_chain) return fail(h, DEMCZ_ERR_STATE, "demcz_detach_history: no streamed history on this handle");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (h->hs_stream) { int32_t rc = sync_stream(h, h->hs_stream, "demcz_detach_history"); if (rc) return rc; }
    *chain_base = h->hs_chain; *logobj_base = h->hs_logobj;
    h->hs_chain = h->hs_logobj = nullptr;      // the caller owns them now: demcz_release_host_buffer
    h->hs_on = false;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_release_host_buffer(void* base)
{
    if (base) g_host_pool.release(base, -1);
    return DEMCZ_OK;
}

extern "C" int32_t demcz_pool_trim(int64_t* device_bytes, int64_t* pinned_bytes)
{
    const size_t a = g_dev_pool.trim(), b = g_host_pool.trim();
    if (device_bytes) *device_bytes = (int64_t)a;
    if (pinned_bytes) *pinned_bytes = (int64_t)b;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_run(demcz_handle* h, int64_t g_from, int64_t g_to, double gamma, const double* temperature)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    DEADCHK(h);
    if (!h->has_state) return fail(h, DEMCZ_ERR_STATE, "demcz_run: call demcz_set_state first");
    if (g_from < 1 || g_to < g_from) return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_run: need 1 <= g_from <= g_to");
    if (h->cfg.target_kind == DEMCZ_TARGET_HOST_CALLBACK)
        return fail(h, DEMCZ_ERR_STATE, "demcz_run: host-callback target uses demcz_propose/accept_commit");
    const bool hist = h->cfg.Gcap > 0;
    if (hist && (g_from - h->g0 - 1 < 0 || g_to - h->g0 > h->cfg.Gcap))
        return fail(h, DEMCZ_ERR_CAPACITY, "demcz_run: generations outside the history window (demcz_set_history_origin)");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int64_t G = g_to - g_from + 1;
    const int K = h->cfg.K;
    const bool sharded = (h->comm != nullptr);
    if (h->peer_mode == 1 && (!h->group || h->group->dead))
        return fail(h, DEMCZ_ERR_STATE, "demcz_run: a member of this handle's replica group has been destroyed");
    if (h->no_live && h->rearm_from >= 0 && g_from > h->rearm_from) {
        // behind the generation a hand-off failed at: LIVE launches again (the calls up to here ran one launch per K-window)
        int32_t rcr = live_try_rearm(h);
        if (rcr) return rcr;
    }
    // `peer`: the boundary rows of ALL shards reach this replica from inside the launches (live_publish) -- no exchange step
    const bool peer = h->peer_mode != 0 && live_span(h) > 0;
    const bool rccl_exchange = sharded && !peer;
    const bool kernel_appends = !rccl_exchange && !h->external_append;
    const int64_t shards = (sharded || h->peer_mode != 0) ? h->nranks : 1;
    // capacity check for all appends of this call
    {
        const int64_t nb = g_to / K - (g_from - 1) / K;
        const int64_t rows = h->cfg.N * shards;
        if (!h->external_append && h->M_app + nb * rows > h->cfg.Mcap)
            return fail(h, DEMCZ_ERR_CAPACITY, "demcz_run: Z row capacity (Mcap) would be exceeded");
        if (h->external_append && nb > 1)
            return fail(h, DEMCZ_ERR_STATE, "demcz_run: with external append a call may cross at most one K boundary, at its end");
        if (h->external_append && nb == 1 && (g_to % K) != 0)
            return fail(h, DEMCZ_ERR_STATE, "demcz_run: with external append the K boundary must be the last generation of the call");
    }
    if (h->peer_mode == 3 && h->peers_closed)
        return fail(h, DEMCZ_ERR_STATE, "demcz_run: the peers' archives have been closed (demcz_peer_detach): this handle runs no further generations");
    if (h->peer_mode == 3 && !peer)
        return fail(h, DEMCZ_ERR_STATE, "demcz_run: host-mediated IPC peers run with the in-launch hand-off only (it is not available: a timed-out "
                                        "hand-off, an append lag, caller-owned appends, or more chains than a LIVE launch holds)");
    if (h->peer_mode == 1 && !peer && !h->replaying) {
        // A replica group that lost (or never had) its in-launch hand-off: its members cannot exchange rows by themselves, so
        // the call is only logged here and executed for ALL members in lockstep, one launch per K-window, at the next
        // verification (group_verify) -- by which time every member has been given the same call.
        if (h->lag != 0 || h->external_append) return fail(h, DEMCZ_ERR_STATE, "demcz_run: replica groups run with append lag 0 and library-owned appends");
        demcz_handle::RunCall rcall{g_from, g_to, gamma, temperature != nullptr, {}};
        if (temperature) rcall.temperature.assign(temperature, temperature + G);
        h->live_log.push_back(std::move(rcall));
        return DEMCZ_OK;
    }
    // (a handle with the arena keeps a call's temperatures inside it when they fit: window_kernel_ps2 then reaches them with the
    //  32-bit offsets it reaches everything else with)
    const bool temp_in_arena = temperature && h->arena && h->arena_temp && G <= h->arena_gens;
    if (temperature && temp_in_arena) {
        SYNCCHK(h, h->stream);              // (earlier windows may still read the region)
        HIPCHK(h, hipMemcpyAsync(h->arena_temp, temperature, (size_t)G * sizeof(double), hipMemcpyHostToDevice, h->stream));
        SYNCCHK(h, h->stream);              // the caller may reuse its buffer on return
    } else if (temperature) {
        if (G > h->temp_cap) {
            { int32_t rcq = quiesce_all(h); if (rcq) return rcq; }
            SYNCCHK(h, h->stream);
            if (h->dtemp) HIPCHK(h, dev_free(h->cfg.device_id, h->dtemp));
            h->dtemp = nullptr; h->temp_cap = 0;
            // (+ 8: the wave-per-chain consumer fetches a pass's temperatures as whole 16-byte pieces, demcz_kernels_ps.h)
            HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->dtemp, (size_t)(G + 8) * sizeof(double)));
            HIPCHK(h, hipMemsetAsync(h->dtemp, 0, (size_t)(G + 8) * sizeof(double), h->stream));
            h->temp_cap = G;
        }
        // stream-ordered behind earlier windows that still read dtemp
        HIPCHK(h, hipMemcpyAsync(h->dtemp, temperature, (size_t)G * sizeof(double), hipMemcpyHostToDevice, h->stream));
        SYNCCHK(h, h->stream);   // the caller may reuse its buffer on return
    }
    WindowParams P;
    P.Z = h->dZ; P.Zw = h->dZ; P.ZS = h->ZS;
    P.Xcur = h->dX; P.lpcur = h->dlp;
    P.chain = hist ? h->dchain : nullptr; P.logobj = hist ? h->dlogobj : nullptr;
#ifdef DEMCZ_EXP_NOHIST           // (timing experiment, never the shipped library: no window kernel writes any history --
    P.chain = nullptr; P.logobj = nullptr;      //  scripts/build_variant.py nohist -DDEMCZ_EXP_NOHIST; profiles/r04o_history_store.txt)
#endif
    P.N = h->cfg.N; P.chain_id0 = h->cfg.chain_id0; P.d = h->cfg.d;
    P.gamma = gamma; P.seed = h->cfg.seed; P.S = h->S; P.Nblocks = h->cfg.Nblocks;
    P.block_offsets = h->d_block_offsets; P.slot_of = h->d_slot_of; P.eps = h->d_eps; P.slot_role = h->d_slot_role;
    P.tp = target_params(h);
    P.snap = nullptr;
    P.K = K;
    P.acc_out = nullptr; P.live_err = h->d_live_err; P.live_spin_limit = LIVE_SPIN_LIMIT;
    P.safe_X = nullptr; P.safe_lp = nullptr;
    P.rec_in = nullptr; P.rec_out = nullptr; P.next_g_first = 0; P.next_M = 0; P.next_ngen = 0; P.consumer_blocks = 0;
    P.next_rows = 0; P.next_boff = 0; P.rec_stride = 0;
    P.rec_fields = (h->lanes == DEMCZ_LAYOUT_SPLIT && h->split_kind == 2) ? h->cfg.d + 2 : 0;    // lane-per-parameter consumers: record-major
    P.z_bytes = (uint32_t)std::min<size_t>(h->dZ_bytes, 0xffffffffull);
    P.hist_bytes = (hist && h->hist_joint) ? (uint32_t)std::min<double>((double)h->cfg.N * (h->cfg.d + 1) * (double)h->cfg.Gcap * 8.0, 4294967295.0) : 0u;
    P.brows = h->cfg.N * (peer ? shards : 1);
    P.row_off = peer ? (int64_t)h->rank * h->cfg.N : 0;
    P.n_peers = peer ? h->n_peers : 0;
    for (int r = 0; r < DEMCZ_MAX_PEERS; ++r) P.peer_Z[r] = (peer && r < h->n_peers) ? h->peer_Z[r] : nullptr;
    if (peer && h->peer_mode == 2 && h->peer_fence) {
        // this replica's unwritten rows were re-filled with the sentinel (set_state / a rollback) on this stream: no peer may
        // publish into it before that is done, so the ranks meet here, stream-ordered, before the first launch that publishes
        NCCLCHK(h, ncclAllReduce(h->d_err_all + 1, h->d_err_all + 1, 1, ncclUint32, ncclMax, h->comm, h->stream));
        h->peer_fence = false;
    }
#ifdef DEMCZ_STAMPS
    if (!h->d_stamps) {
        HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->d_stamps, (size_t)DEMCZ_STAMP_WGS * 16 * sizeof(unsigned long long)));
        HIPCHK(h, hipMemset(h->d_stamps, 0, (size_t)DEMCZ_STAMP_WGS * 16 * sizeof(unsigned long long)));
    }
    P.stamps = h->d_stamps;
#endif
    const int E = h->lag;
    int64_t g = g_from;
    if (h->lanes == DEMCZ_LAYOUT_SPLIT) {
        // draw-record buffers: sized ONCE for the longest launch this handle can make -- a LIVE launch spans up
        // to live_span() generations (never more than the run is long: Gcap, when a history is kept), any other
        // at most a K-window / a batch of E windows -- before anything of this call is timed or enqueued
        const int64_t lm = live_span(h);
        int64_t cap = (lm > 0) ? lm : (int64_t)K * std::max(E, 1);
        if (lm > 0 && hist) cap = std::max<int64_t>(std::min<int64_t>(cap, h->cfg.Gcap), std::min<int64_t>(cap, G));
        int32_t rcr = rec_reserve(h, cap);
        if (rcr) return rcr;
    }
    if (live_span(h) > 0) {
        // this call may issue LIVE launches: they are verified at the next synchronising entry point, and redone
        // from here (live_verify) should a row hand-off inside one of them fail
        // bound the redo: verify now (a synchronisation every 256 calls).  Not inside demcz_run_checked: that call rolls back
        // to ITS entry and redoes itself from there (statistics and stop decisions included), so the snapshot must not move
        // into the middle of it.  Not inside a redo either (`replaying`): a call of a redo that has re-armed is logged against
        // the verified point the redo started from -- the snapshot taken there is the one a second failure goes back to.
        if (h->live_log.size() >= 256 && !h->in_checked && !h->replaying && h->peer_mode != 1) {      // (a replica group verifies as a whole: group_verify)
            int32_t rcv = live_verify(h);
            if (rcv) return rcv;
        }
        if (live_span(h) > 0) {
            if (h->live_log.empty() && !h->replaying) {
                const int64_t N = h->cfg.N;
                const int d = h->cfg.d;
                if (!h->d_safe_X) HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->d_safe_X, (size_t)N * d * sizeof(double)));
                if (!h->d_safe_lp) HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->d_safe_lp, (size_t)N * sizeof(double)));
                if (h->live_fault_polls < 0) {   // fault injection: whatever the snapshot buffers held must never reach the state
                    HIPCHK(h, hipMemsetAsync(h->d_safe_X, 0xff, (size_t)N * d * sizeof(double), h->stream));
                    HIPCHK(h, hipMemsetAsync(h->d_safe_lp, 0xff, (size_t)N * sizeof(double), h->stream));
                }
                h->snap_pending = true;          // (made by the first launch: launch_window)
                h->safe_M = h->M; h->safe_M_app = h->M_app; h->safe_g_done = h->g_done;
            }
            demcz_handle::RunCall rcall{g_from, g_to, gamma, temperature != nullptr, {}};
            if (temperature) rcall.temperature.assign(temperature, temperature + G);
            h->live_log.push_back(std::move(rcall));
        }
    }
    // demcz_set_kernel_timing: ONE event pair around the back-to-back window launches of this call (an event
    // between two launches would put a bubble into the stream it is meant to measure)
    struct EventPair {          // destroyed on every error return; handed to h->timed on success
        hipEvent_t a = nullptr, b = nullptr;
        ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } tev;
    int64_t timed_launches = 0;
    const bool timed = h->timing && h->timed.size() < 4096;
    if (timed) {
        HIPCHK(h, hipEventCreate(&tev.a));
        HIPCHK(h, hipEventCreate(&tev.b));
        HIPCHK(h, hipEventRecord(tev.a, h->stream));
    }
    while (g <= g_to) {
        const int64_t next_boundary = ((g - 1) / K + 1) * K;      // first multiple of K that is >= g
        // Synchronous schedule: a launch ends at the next K boundary, the kernel boundary is the
        // grid-wide barrier before the new rows are drawn from.  Deferred schedule: nothing appended
        // inside a batch becomes visible inside it, so one launch covers the rest of the batch.
        int64_t w_end = std::min(next_boundary, g_to);
        if (E > 0 && !h->external_append) {
            const int64_t jn = next_boundary / K, J = ((jn + E - 1) / E) * E;
            w_end = std::min(J * (int64_t)K, g_to);
        }
        // Split layout on one GPU: the launch runs on through the boundaries; waves hand the appended rows
        // to each other inside it (LIVE, demcz_kernels_pc.h), so the schedule is still the synchronous one.
        int64_t live_max = live_span(h);
        h->dual_now = false;
        if (h->ps_dual) {
            // A two-chain handle runs its REGULAR launches (start, boundary distance and length multiples of five; everything
            // reachable through the arena) on window_kernel_ps2d; anything else takes the general kernel, one chain per wave, and
            // -- twice the waves: they would not all be resident -- never across a K boundary.
            const int64_t tb0 = next_boundary - g + 1, rest = g_to - g + 1;
            const bool temp_ok = !temperature || temp_in_arena;
            const bool hist_ok = !hist || (double)h->cfg.N * (h->cfg.d + 1) * (double)h->cfg.Gcap * 8.0 < 4293918720.0;
            const bool reg = K % PS2_R == 0 && tb0 % PS2_R == 0 && rest >= PS2_R && temp_ok && hist_ok && h->arena && h->rec_in_arena;
            if (reg) {
                h->dual_now = true;
                if (live_max > 0) live_max = std::max<int64_t>((std::min(live_max, rest) / PS2_R) * PS2_R, PS2_R);
                else w_end = g + ((w_end - g + 1) / PS2_R) * PS2_R - 1;       // (a K-window cut short by the call's end: whole passes only)
            } else {
                live_max = 0;
            }
        }
        if (live_max > 0) {
            // How far this launch goes: as far as the records allow -- but no further than the draws that ARE there, where the
            // launch before prepared fewer than that (a call longer than the last one); and a launch whose draws have to be made
            // first, with nothing to run beside (the first of a run), stays short: its producer is serial time
            // (64 MiB of records, the span of every launch until round 3f; profiles/r03f_launch_span.txt).
            int64_t n = std::min(live_max, g_to - g + 1);
            const auto& dc = h->rec_desc[h->rec_cur];
            const int64_t rows_v = h->cfg.N * (peer ? shards : 1);      // (live: immediate visibility; all shards' rows with peers)
            const int32_t boff = (int32_t)(K - (next_boundary - g + 1));
            const bool ready = dc.valid && dc.g_first == g + h->rng_offset && dc.M == h->M && dc.rows == rows_v && dc.boff == boff;
            if (ready && dc.ngen >= K) {
                if (dc.ngen < n) n = std::max<int64_t>((dc.ngen / K) * K, K);
            } else {
                const int64_t per_gen = rec_fields(h) * h->cfg.N * (int64_t)sizeof(double);
                const int64_t cold = std::max<int64_t>(K, ((int64_t)((h->ps_dual ? 128ll : 64ll) << 20) / per_gen / K) * K);
                n = std::min(n, cold);
            }
            if (h->dual_now) n = std::max<int64_t>((n / PS2_R) * PS2_R, PS2_R);
            w_end = g + n - 1;
        }
        int32_t rc = admit_pending(h, g);
        if (rc) return rc;
        const int64_t nbound = w_end / K - (g - 1) / K;           // boundaries inside this launch
        P.M = h->M;
        P.M_append = h->M_app;
        P.g_first = g + h->rng_offset;      // only positions the Philox streams
        P.ngen = (int32_t)(w_end - g + 1);
        P.to_boundary = (int32_t)(next_boundary - g + 1);
        P.slot_first = hist ? (g - h->g0 - 1) : 0;
        P.temperature = temperature ? (temp_in_arena ? h->arena_temp : h->dtemp) + (g - g_from) : nullptr;
        P.do_append = (nbound > 0 && kernel_appends) ? 1 : 0;
        P.snap = nullptr;
        if (nbound > 0 && rccl_exchange && E > 0) {
            if (h->batch_cnt == 0) {
                // the buffer was last read by the exchange two batches ago
                // (every wait is a barrier packet between two window kernels: not asked for twice)
                if (h->buf_xseq[h->batch_buf] > h->xseq_waited) HIPCHK(h, hipStreamWaitEvent(h->stream, h->buf_done[h->batch_buf], 0));
                h->batch_base = h->M_app;
            }
            P.snap = h->d_send[h->batch_buf] + (size_t)h->batch_cnt * h->cfg.N * h->cfg.d;
        }
        if (h->lanes == DEMCZ_LAYOUT_SPLIT) {
            // what the launch after this one will be, so that this launch's producer half can prepare
            // its draws: it starts at w_end + 1, runs to its own boundary / batch end / span, and sees ...
            const int64_t ng = w_end + 1;
            const int64_t nb1 = ((ng - 1) / K + 1) * (int64_t)K;       // its first boundary
            int64_t nend = nb1;
            if (E > 0 && !h->external_append) nend = ((nend / K + E - 1) / E) * E * (int64_t)K;
            int64_t nM = h->M, nn = nend - ng + 1;
            const int64_t rows_n = h->cfg.N * shards;
            if (live_max > 0) nn = std::min(live_max, (w_end < g_to) ? g_to - w_end : G);   // a next call is taken to be as long as this one
            if (h->external_append) nn = 0;                      // the caller appends: M is not ours to predict
            else if (E == 0) nM = h->M_app + nbound * rows_n;                    // ... the rows appended now
            else for (const auto& pe : h->pending) if (pe.visible_from <= ng) nM = pe.M_after;   // ... or admitted by then
            const int64_t vis_rows = (E == 0) ? rows_n : 0;      // rows that join per boundary passed inside a launch
            const int32_t nboff = (int32_t)(K - (nb1 - ng + 1));
            rc = pc_prepare(h, P, vis_rows, ng + h->rng_offset, nn, nM, vis_rows, nboff);
            if (rc) return rc;
        }
        // boundaries whose rows generations of this same launch draw from.  With peers EVERY launch is of the LIVE kind: the rows
        // of the boundary that ended the launch before may still be on their way from another replica's publisher
        // (DEMCZ_FORCE_LIVE_KERNEL=1, diagnosis only: the LIVE instantiation also for launches with no boundary inside -- what the
        //  instantiation itself costs, scripts/live_fixed_cost.py)
        static const bool force_live = getenv("DEMCZ_FORCE_LIVE_KERNEL") != nullptr;
        const bool live = live_max > 0 && (((w_end - 1) / K - (g - 1) / K) > 0 || peer || force_live);
        P.live_err = h->d_live_err;
        P.live_spin_limit = h->live_spin_limit ? (int32_t)h->live_spin_limit : LIVE_SPIN_LIMIT;
        if (h->live_fault_polls > 0 && g >= h->live_fault_g) P.live_spin_limit = h->live_fault_polls;
        if (h->live_fault_polls < 0 && live && g >= h->live_fault_g)      // "a wave of this launch has already given up": every wave leaves at once
            HIPCHK(h, hipMemsetAsync(h->d_live_err, 0x01, sizeof(unsigned int), h->stream));
        P.acc_out = h->d_acc ? h->d_acc + (size_t)h->acc_next * (size_t)h->acc_waves * 2 : nullptr;
        if (h->ps_dual && P.acc_out)     // (its two kernels write different numbers of waves' counters into a slot of the ring)
            HIPCHK(h, hipMemsetAsync(P.acc_out, 0, (size_t)h->acc_waves * 2 * sizeof(unsigned int), h->stream));
        if (live) h->err_clean = false;
        rc = launch_window(h, P, live);
        h->after_launch_ev = nullptr;         // (whatever was recorded before this launch says nothing about it)
        if (rc) return rc;
        if (h->d_acc) {
            h->acc_log.push_back({g, w_end, h->acc_next});
            h->acc_next = (h->acc_next + 1) % h->acc_slots;
            if ((int64_t)h->acc_log.size() > h->acc_slots) h->acc_log.pop_front();
        }
        ++timed_launches;
        // streamed history: this launch's generations leave for the host mirrors behind it, while the next launch computes
        // (one marker per launch on the compute stream: launches that stream are long ones)
        if (h->hs_on && hist) { int32_t rch = stream_history(h, g, w_end); if (rch) return rch; }
        if (nbound > 0 && !h->external_append) {
            const int64_t rows = h->cfg.N * shards;
            if (E == 0) {
                if (kernel_appends) { h->M_app += nbound * rows; h->M = h->M_app; }
                else { rc = append_after_window(h); if (rc) return rc; }
            } else {
                // boundary j belongs to the batch closing at J = ceil(j/E)*E; its rows are drawn from
                // generation (J + E)*K + 1 on -- the same rule on every rank and for any sharding
                for (int64_t j = (g - 1) / K + 1; j <= w_end / K; ++j) {
                    const int64_t J = ((j + E - 1) / E) * E;
                    h->M_app += rows;
                    if (h->batch_J != J) {
                        h->pending.push_back({(J + E) * (int64_t)K + 1, h->M_app, nullptr});
                        h->batch_J = J;
                    } else {
                        h->pending.back().M_after = h->M_app;
                    }
                    if (rccl_exchange) {
                        ++h->batch_cnt;
                        if (j == J) { rc = exchange_batch(h); if (rc) return rc; }
                    }
                }
            }
        }
        g = w_end + 1;
    }
    if (timed) {
        HIPCHK(h, hipEventRecord(tev.b, h->stream));
        h->after_launch_ev = tev.b;
        h->timed.emplace_back(tev.a, tev.b);
        tev.a = tev.b = nullptr;
        h->timed_launches += timed_launches;
    }
    if (rccl_exchange && E > 0) {           // nothing stays un-exchanged across calls
        int32_t rc = exchange_batch(h);
        if (rc) return rc;
    }
    h->g_done = g_to;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_synchronize(demcz_handle* h)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    DEADCHK(h);
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    SYNCCHK(h, h->stream);
    if (h->comm_stream) SYNCCHK(h, h->comm_stream);
    if (h->prod_stream) SYNCCHK(h, h->prod_stream);      // (the next launch's draws, monitoring checks)
    return live_verify(h);
}

static int32_t check_hist_range(demcz_handle* h, int64_t g_from, int64_t g_to, const char* who)
{
    if (h->cfg.Gcap <= 0) return fail(h, DEMCZ_ERR_STATE, std::string(who) + ": handle keeps no history (Gcap = 0)");
    if (g_from < 1 || g_to < g_from || g_from - h->g0 - 1 < 0 || g_to - h->g0 > h->cfg.Gcap)
        return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, std::string(who) + ": generations outside the history window");
    return DEMCZ_OK;
}

extern "C" int32_t demcz_get_history(demcz_handle* h, int64_t g_from, int64_t g_to, double* chain, double* log_obj)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    DEADCHK(h);
    int32_t rc = check_hist_range(h, g_from, g_to, "demcz_get_history");
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    rc = live_verify(h);
    if (rc) return rc;
    const int64_t N = h->cfg.N, G = g_to - g_from + 1, s0 = g_from - h->g0 - 1;
    const int d = h->cfg.d;
    if (chain)
        HIPCHK(h, hipMemcpyAsync(chain, h->dchain + (size_t)N * d * s0, (size_t)N * d * G * sizeof(double),
                                 hipMemcpyDeviceToHost, h->stream));
    if (log_obj)
        HIPCHK(h, hipMemcpyAsync(log_obj, h->dlogobj + (size_t)N * s0, (size_t)N * G * sizeof(double),
                                 hipMemcpyDeviceToHost, h->stream));
    SYNCCHK(h, h->stream);
    return DEMCZ_OK;
}

extern "C" int32_t demcz_get_changed(demcz_handle* h, int64_t g_from, int64_t g_to, int64_t* changed)
{
    if (!h || !changed) return DEMCZ_ERR_INVALID_ARGUMENT;
    int32_t rc = check_hist_range(h, g_from, g_to, "demcz_get_changed");
    if (rc) return rc;
    const int64_t G = g_to - g_from + 1, s0 = g_from - h->g0 - 1;
    if (s0 == 0 && !h->origin_valid)
        return fail(h, DEMCZ_ERR_STATE, "demcz_get_changed: the log_obj before the first history slot is not known "
                                        "(set the history origin when exactly g0 generations have been run)");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    rc = live_verify(h);
    if (rc) return rc;
    rc = ensure_scratch(h, G);
    if (rc) return rc;
    static_assert(sizeof(long long) == sizeof(double), "scratch reuse");
    hipLaunchKernelGGL(changed_from_history_kernel, dim3((unsigned)G), dim3(256), 0, h->stream, (const double*)h->dlogobj,
                       (const double*)h->dlp_origin, h->cfg.N, s0, (long long*)h->d_scratch);
    HIPCHK(h, hipGetLastError());
    static_assert(sizeof(int64_t) == sizeof(long long), "int64 layout");
    HIPCHK(h, hipMemcpyAsync(changed, h->d_scratch, (size_t)G * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
    SYNCCHK(h, h->stream);
    return DEMCZ_OK;
}

namespace demcz {
// sum over `nslots` consecutive ring slots (from slot0, modulo `ring`) of the per-wave launch totals, minus the
// first-generation counts of slot0 when `skip_first`
__global__ void __launch_bounds__(256) sum_counts_kernel(const unsigned int* acc, int64_t waves, int slot0, int nslots, int ring,
                                                         int skip_first, long long* out)
{
    __shared__ long long red[256];
    long long a = 0;
    for (int64_t i = threadIdx.x; i < (int64_t)nslots * waves; i += 256) {
        const int sl = (int)(i / waves);
        const int64_t w = i % waves;
        const unsigned int* e = acc + ((size_t)((slot0 + sl) % ring) * (size_t)waves + (size_t)w) * 2;
        a += (long long)e[0];
        if (skip_first && sl == 0) a -= (long long)e[1];
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}
}  // namespace demcz

extern "C" int32_t demcz_get_changed_total(demcz_handle* h, int64_t g_from, int64_t g_to, int64_t* total, int32_t* from_ballots)
{
    if (!h || !total || g_from < 1 || g_to < g_from) return DEMCZ_ERR_INVALID_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    int32_t rc = live_verify(h);
    if (rc) return rc;
    // whole launches whose generations are exactly g_from..g_to, or g_from-1..g_to with the first generation's own
    // count taken off (the window sum of demcz_anneal.jl:50 has no predecessor for its first column)
    int first = -1, last = -1, skip = 0;
    for (int i = 0; i < (int)h->acc_log.size(); ++i) {
        const auto& r = h->acc_log[i];
        if (first < 0 && (r.g_first == g_from || r.g_first == g_from - 1) && r.g_last >= g_from) { first = i; skip = (r.g_first == g_from - 1); }
        if (first >= 0 && r.g_last == g_to) { last = i; break; }
    }
    bool ok = first >= 0 && last >= first;
    for (int i = first; ok && i < last; ++i)
        ok = h->acc_log[i].g_last + 1 == h->acc_log[i + 1].g_first && (h->acc_log[i].slot + 1) % h->acc_slots == h->acc_log[i + 1].slot;
    if (ok && h->d_acc) {
        rc = ensure_scratch(h, 1);
        if (rc) return rc;
        static_assert(sizeof(long long) == sizeof(double), "scratch reuse");
        hipLaunchKernelGGL(sum_counts_kernel, dim3(1), dim3(256), 0, h->stream, (const unsigned int*)h->d_acc, h->acc_waves,
                           (int)h->acc_log[first].slot, last - first + 1, (int)h->acc_slots, skip, (long long*)h->d_scratch);
        HIPCHK(h, hipGetLastError());
        long long v = 0;
        HIPCHK(h, hipMemcpyAsync(&v, h->d_scratch, sizeof(v), hipMemcpyDeviceToHost, h->stream));
        SYNCCHK(h, h->stream);
        *total = (int64_t)v;
        if (from_ballots) *from_ballots = 1;
        return DEMCZ_OK;
    }
    // not a union of whole launches (or a host-closure handle): count from the history, as the checker does
    std::vector<int64_t> ch((size_t)(g_to - g_from + 1));
    rc = demcz_get_changed(h, g_from, g_to, ch.data());
    if (rc) return rc;
    int64_t t = 0;
    for (int64_t v : ch) t += v;
    *total = t;
    if (from_ballots) *from_ballots = 0;
    return DEMCZ_OK;
}

// ---- R-hat ---------------------------------------------------------------------------------------
struct RhatPlan { int64_t N, w, s0, n, nd; int d, nchunk; double *S1, *S2, *mean_j, *s2_j, *sums; };

static int32_t rhat_prepare(demcz_handle* h, int64_t g_from, int64_t g_to, RhatPlan& r, bool compute, hipStream_t qs = nullptr)
{
    int32_t rc = check_hist_range(h, g_from, g_to, "demcz_rhat");
    if (rc) return rc;
    r.N = h->cfg.N; r.w = g_to - g_from + 1; r.s0 = g_from - h->g0 - 1; r.d = h->cfg.d;
    r.n = r.w / 2;                                            // utils.jl:4
    if (r.n < 2) return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_rhat: window needs at least 4 generations");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    r.nchunk = (int)std::min<int64_t>(std::max<int64_t>(1, r.n / 32), 32);
    r.nd = r.N * r.d;
    const int64_t need = 2 * (2 * r.nchunk * r.nd) + 2 * (2 * r.nd) + 4 * r.d;
    rc = ensure_scratch(h, need);
    if (rc) return rc;
    r.S1 = h->d_scratch;
    r.S2 = r.S1 + 2 * r.nchunk * r.nd;
    r.mean_j = r.S2 + 2 * r.nchunk * r.nd;
    r.s2_j = r.mean_j + 2 * r.nd;
    r.sums = r.s2_j + 2 * r.nd;          // [0,d): stage 0 / grand mean; [d,3d): stage 1
    if (compute) {
        const int bs = 256;
        const unsigned gx = (unsigned)((r.nd + bs - 1) / bs);
        if (!qs) qs = h->stream;
        hipLaunchKernelGGL(rhat_moments_kernel, dim3(gx, 2 * r.nchunk), dim3(bs), 0, qs, h->dchain, r.N, r.d, r.s0, r.n, r.nchunk, r.S1, r.S2);
        hipLaunchKernelGGL(rhat_chainstats_kernel, dim3(gx, 2), dim3(bs), 0, qs, h->dchain, r.N, r.d, r.s0, r.n, r.nchunk, r.S1, r.S2, r.mean_j, r.s2_j);
        HIPCHK(h, hipGetLastError());
    }
    return DEMCZ_OK;
}

extern "C" int32_t demcz_rhat_partial(demcz_handle* h, int64_t g_from, int64_t g_to, int32_t stage, const double* grand, double* out)
{
    if (!h || !out || (stage != 0 && stage != 1) || (stage == 1 && !grand)) return DEMCZ_ERR_INVALID_ARGUMENT;
    RhatPlan r;
    int32_t rc = live_verify(h);
    if (rc) return rc;
    rc = rhat_prepare(h, g_from, g_to, r, true);
    if (rc) return rc;
    const int d = r.d;
    if (stage == 0) {
        hipLaunchKernelGGL(rhat_reduce_kernel, dim3(d), dim3(256), 0, h->stream, r.mean_j, r.s2_j, r.N, d, 0, (const double*)nullptr, 1.0, r.sums);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(out, r.sums, (size_t)d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    } else {
        HIPCHK(h, hipMemcpyAsync(r.sums, grand, (size_t)d * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(rhat_reduce_kernel, dim3(d), dim3(256), 0, h->stream, r.mean_j, r.s2_j, r.N, d, 1, (const double*)r.sums, 1.0, r.sums + d);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(out, r.sums + d, (size_t)2 * d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    }
    SYNCCHK(h, h->stream);
    return DEMCZ_OK;
}

// Enqueues the R-hat of generations g_from..g_to and the copy of its d values to `out` (host memory; pinned
// if the caller wants the copy to be asynchronous); does not wait.
// `side`: unsharded runs only -- the statistic of a finished slab is computed on a second stream, behind an event,
// beside the next slab's window kernel (which leaves most of the chip idle at small N).
static int32_t rhat_enqueue(demcz_handle* h, int64_t g_from, int64_t g_to, double* out, bool side = false)
{
    RhatPlan r;
    hipStream_t qs = h->stream;
    // Monitoring checks run beside the next slab.  Where the handle already has a side stream -- the producer stream of the
    // wave-per-chain layout -- they go there, behind the producer kernel: a third active queue cost 10-15 us per
    // 1000-generation slab (the consumer of the next slab started that much later; scripts/step_overhead.py).
    static const bool own_env = getenv("DEMCZ_RHAT_OWN_STREAM") != nullptr;
    if (side && !h->comm && !own_env && h->prod_stream) {
        if (!h->diag_ev) HIPCHK(h, hipEventCreateWithFlags(&h->diag_ev, hipEventDisableTiming));
        if (h->after_launch_ev) {
            HIPCHK(h, hipStreamWaitEvent(h->prod_stream, h->after_launch_ev, 0));
        } else {
            HIPCHK(h, hipEventRecord(h->diag_ev, h->stream));
            HIPCHK(h, hipStreamWaitEvent(h->prod_stream, h->diag_ev, 0));
            h->after_launch_ev = h->diag_ev;
        }
        qs = h->prod_stream;
    } else if (side && h->comm && h->peer_mode == 2 && !h->no_live && h->comm_side && h->comm_stream && h->lag == 0) {
        // sharded with the rows handed over inside the launches: nothing else of this handle uses the side communicator, and the
        // compute stream carries no collective between two slabs -- the statistic goes beside the next slab
        if (!h->diag_ev) HIPCHK(h, hipEventCreateWithFlags(&h->diag_ev, hipEventDisableTiming));
        if (h->after_launch_ev) {
            HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->after_launch_ev, 0));
        } else {
            HIPCHK(h, hipEventRecord(h->diag_ev, h->stream));
            HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->diag_ev, 0));
            h->after_launch_ev = h->diag_ev;
        }
        qs = h->comm_stream;
    } else if (side && !h->comm) {
        if (!h->diag_stream) HIPCHK(h, stream_acquire(h->cfg.device_id, &h->diag_stream));
        if (!h->diag_ev) HIPCHK(h, hipEventCreateWithFlags(&h->diag_ev, hipEventDisableTiming));
        if (h->after_launch_ev) {
            HIPCHK(h, hipStreamWaitEvent(h->diag_stream, h->after_launch_ev, 0));
        } else {
            HIPCHK(h, hipEventRecord(h->diag_ev, h->stream));
            HIPCHK(h, hipStreamWaitEvent(h->diag_stream, h->diag_ev, 0));
            h->after_launch_ev = h->diag_ev;
        }
        qs = h->diag_stream;
    }
    // Every check works in the handle's one scratch buffer.  Checks on a side stream follow each other there; a check on the
    // compute stream (the call's last slab; demcz_rhat) waits for the latest of them.  (Round 5: a 300-slab demcz_run_checked of
    // four-generation slabs showed the second-to-last check's statistic computed from a scratch the last check was already
    // writing -- one run in four; with 1000-generation slabs the side check is long done when the last slab ends.)
    if (qs == h->stream && h->rhat_side_pending) {
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->rhat_side_ev, 0));
        h->rhat_side_pending = false;
    }
    struct SideMark {
        demcz_handle* h; hipStream_t qs;
        ~SideMark() {
            if (qs == h->stream) return;
            if (!h->rhat_side_ev && hipEventCreateWithFlags(&h->rhat_side_ev, hipEventDisableTiming) != hipSuccess) { h->rhat_side_ev = nullptr; return; }
            if (hipEventRecord(h->rhat_side_ev, qs) == hipSuccess) h->rhat_side_pending = true;
        }
    } side_mark{h, qs};
    int32_t rc = rhat_prepare(h, g_from, g_to, r, true, qs);
    if (rc) return rc;
    const int d = r.d;
    const int64_t n = r.n;
    const int64_t m = 2 * r.N * h->nranks;                   // utils.jl:5
    const bool sharded = (h->comm != nullptr);
    double* sums = r.sums;               // [0,d): sum_j mean_j; [d,3d): stage-1 sums; [3d,4d): R-hat
    if (!sharded) {
        hipLaunchKernelGGL(rhat_tail_kernel, dim3(d), dim3(256), 0, qs, r.mean_j, r.s2_j, r.N, d, (double)n, (double)m, sums + 3 * d);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(out, sums + 3 * d, (size_t)d * sizeof(double), hipMemcpyDeviceToHost, qs));
        return DEMCZ_OK;
    }
    // stage 0 -> all-reduce -> stage 1 with the grand mean formed on the device -> all-reduce ->
    // utils.jl:13-18 on the device: one copy and one synchronisation per check
    const ncclComm_t cq = (qs == h->comm_stream && qs != h->stream) ? h->comm_side : h->comm;
    hipLaunchKernelGGL(rhat_reduce_kernel, dim3(d), dim3(256), 0, qs, r.mean_j, r.s2_j, r.N, d, 0, (const double*)nullptr, 1.0, sums);
    HIPCHK(h, hipGetLastError());
    NCCLCHK(h, ncclAllReduce(sums, sums, (size_t)d, ncclDouble, ncclSum, cq, qs));
    hipLaunchKernelGGL(rhat_reduce_kernel, dim3(d), dim3(256), 0, qs, r.mean_j, r.s2_j, r.N, d, 1, (const double*)sums, (double)m, sums + d);
    HIPCHK(h, hipGetLastError());
    NCCLCHK(h, ncclAllReduce(sums + d, sums + d, (size_t)2 * d, ncclDouble, ncclSum, cq, qs));
    hipLaunchKernelGGL(rhat_final_kernel, dim3((unsigned)((d + 63) / 64)), dim3(64), 0, qs, (const double*)(sums + d), d, (double)n, (double)m, sums + 3 * d);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, sums + 3 * d, (size_t)d * sizeof(double), hipMemcpyDeviceToHost, qs));
    return DEMCZ_OK;
}

extern "C" int32_t demcz_rhat(demcz_handle* h, int64_t g_from, int64_t g_to, double* rhat)
{
    if (!h || !rhat) return DEMCZ_ERR_INVALID_ARGUMENT;
    int32_t rc = live_verify(h);
    if (rc) return rc;
    rc = rhat_enqueue(h, g_from, g_to, rhat);
    if (rc) return rc;
    SYNCCHK(h, h->stream);
    return DEMCZ_OK;
}

extern "C" int32_t demcz_accept_ratio(demcz_handle* h, int64_t g_from, int64_t g_to, double* ratio)
{
    if (!h || !ratio) return DEMCZ_ERR_INVALID_ARGUMENT;
    int32_t rc = check_hist_range(h, g_from, g_to, "demcz_accept_ratio");
    if (rc) return rc;
    const int64_t N = h->cfg.N, w = g_to - g_from + 1, s0 = g_from - h->g0 - 1;
    if (w < 2) return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_accept_ratio: need at least 2 generations");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    rc = live_verify(h);
    if (rc) return rc;
    rc = ensure_scratch(h, 2 * N);
    if (rc) return rc;
    {   // chains x time chunks: enough workgroups to fill the chip whatever the shape of the window
        const int64_t cb = (N + 63) / 64;
        const int nchunk = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((w - 1 + 31) / 32, 1024), (4096 + cb - 1) / cb));
        unsigned int* cnt = reinterpret_cast<unsigned int*>(h->d_scratch + N);
        HIPCHK(h, hipMemsetAsync(cnt, 0, (size_t)N * sizeof(unsigned int), h->stream));
        hipLaunchKernelGGL(changed_per_chain_kernel, dim3((unsigned)cb, (unsigned)nchunk), dim3(64), 0, h->stream, h->dlogobj, N, s0, w, nchunk, cnt);
        hipLaunchKernelGGL(changed_ratio_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, (const unsigned int*)cnt, N, w, h->d_scratch);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(ratio, h->d_scratch, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    SYNCCHK(h, h->stream);
    return DEMCZ_OK;
}

extern "C" int32_t demcz_mean_cov(demcz_handle* h, int64_t g_from, int64_t g_to, double* mean, double* cov)
{
    if (!h || !mean || !cov) return DEMCZ_ERR_INVALID_ARGUMENT;
    int32_t rc = check_hist_range(h, g_from, g_to, "demcz_mean_cov");
    if (rc) return rc;
    const int64_t N = h->cfg.N, w = g_to - g_from + 1, s0 = g_from - h->g0 - 1;
    const int d = h->cfg.d;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    rc = live_verify(h);
    if (rc) return rc;
    const int T = (d + MC_TS - 1) / MC_TS, npair = T * (T + 1) / 2;
    const int64_t cb = (N + 255) / 256;
    const int nchunk = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>((w + 15) / 16, 512), (2048 + cb * npair - 1) / (cb * npair)));
    const int64_t nblk_xy = cb * nchunk;
    rc = ensure_scratch(h, (int64_t)d * (d + 1) + d + nblk_xy * npair * MC_VALS);
    if (rc) return rc;
    double* sums = h->d_scratch;
    double* partial = sums + (int64_t)d * (d + 1) + d;
    hipLaunchKernelGGL(meancov_partial_kernel, dim3((unsigned)cb, (unsigned)nchunk, (unsigned)npair), dim3(256), 0, h->stream,
                       (const double*)h->dchain, N, d, s0, w, nchunk, partial);
    hipLaunchKernelGGL(meancov_final_kernel, dim3((unsigned)((d * (d + 1) + 63) / 64)), dim3(64), 0, h->stream, (const double*)partial, d,
                       (int)nblk_xy, sums);
    HIPCHK(h, hipGetLastError());
    std::vector<double> hs((size_t)d * (d + 1)), ref((size_t)d);
    HIPCHK(h, hipMemcpyAsync(hs.data(), sums, hs.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpy2DAsync(ref.data(), sizeof(double), h->dchain + (size_t)N * d * s0, (size_t)N * sizeof(double),
                               sizeof(double), (size_t)d, hipMemcpyDeviceToHost, h->stream));
    SYNCCHK(h, h->stream);
    const double cnt = (double)(N * w);
    std::vector<double> dm((size_t)d);      // mean - ref
    for (int p = 0; p < d; ++p) { dm[p] = hs[(size_t)p + (size_t)d * d] / cnt; mean[p] = ref[p] + dm[p]; }
    for (int p = 0; p < d; ++p)
        for (int q = 0; q < d; ++q) cov[p + d * q] = hs[(size_t)p + (size_t)d * q] / cnt - dm[p] * dm[q];   // utils.jl:104
    return DEMCZ_OK;
}

// ---- host-closure mode -------------------------------------------------------------------------
namespace demcz {
// (host_X / done_count / host_flag: the pipelined closure mode, demcz_closure_buffers -- the proposals also go to pinned host memory,
//  and the last workgroup to finish publishes `seq` in the flag word the host is spinning on)
__device__ __forceinline__ void propose_done(unsigned int* done_count, unsigned int* host_flag, unsigned int seq)
{
    if (!host_flag) return;
    __threadfence_system();                       // this thread's stores to host memory, before the workgroup reports
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int prev = __hip_atomic_fetch_add(done_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(done_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (ready for the next launch)
            __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void propose_kernel(const WindowParams P, int ib, uint64_t blk0, double* Xprop, double* logu, double* host_X,
                               unsigned int* done_count, unsigned int* host_flag, unsigned int seq)
{
    extern __shared__ double lds[];
    const int64_t c = (int64_t)blockIdx.x * WINDOW_BS + threadIdx.x;
    if (c >= P.N) { propose_done(done_count, host_flag, seq); return; }
    const int tid = threadIdx.x;
    const int d = P.d;
    rng_state st;
    rng_seek(st, P.seed, (uint64_t)(P.chain_id0 + c), blk0);
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
        lds[(2 * pr) * WINDOW_BS + tid] = z0;
        lds[(2 * pr + 1) * WINDOW_BS + tid] = z1;
    }
    const double scale = (b == 1) ? P.gamma : P.gamma / sqrt((double)(2 * b));
    const int32_t* so = P.slot_of + ib * d;
    const double* za = P.Z + (int64_t)i1 * P.ZS;
    const double* zb = P.Z + (int64_t)i2 * P.ZS;
    for (int p = 0; p < d; ++p) {
        const int t = so[p];
        double xv = P.Xcur[c + P.N * p];
        if (t >= 0) {
            double diff = za[p] - zb[p];
            double zt = lds[((b == 1) ? 0 : t) * WINDOW_BS + tid];
            double t1 = scale * diff;
            double t2 = P.eps[p] * zt;
            double delta = t1 + t2;
            xv = xv + delta;
        }
        Xprop[c + P.N * p] = xv;
        if (host_X) host_X[c + P.N * p] = xv;
    }
    rng_next(st, r1, r2);
    logu[c] = dm_log(u_open(r1));
    propose_done(done_count, host_flag, seq);
}

__global__ void accept_commit_kernel(int64_t N, int d, double* Xcur, double* lpcur, const double* Xprop,
                                     const double* lpprop, const double* logu, int has_T, double T)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const double lp = lpcur[c], lpp = lpprop[c];
    double dlt = lpp - lp;
    if (has_T) dlt = dlt / T;
    if (logu[c] < dlt) {
        for (int p = 0; p < d; ++p) Xcur[c + N * p] = Xprop[c + N * p];
        lpcur[c] = lpp;
    }
}

__global__ void end_generation_kernel(int64_t N, int d, const double* Xcur, const double* lpcur,
                                      double* chain, double* logobj, int64_t slot,
                                      double* Z, int64_t ZS, int64_t M, int do_append)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    for (int p = 0; p < d; ++p) {
        const double xv = Xcur[c + N * p];
        if (chain) chain[c + N * (p + (int64_t)d * slot)] = xv;
        if (do_append) Z[(M + c) * ZS + p] = xv;
    }
    if (logobj) logobj[c + N * slot] = lpcur[c];
}
}  // namespace demcz

extern "C" int32_t demcz_propose(demcz_handle* h, int64_t g, int32_t ib, double gamma, double* Xprop)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (h->cfg.target_kind != DEMCZ_TARGET_HOST_CALLBACK) return fail(h, DEMCZ_ERR_STATE, "demcz_propose: handle was not created with DEMCZ_TARGET_HOST_CALLBACK");
    if (!h->has_state) return fail(h, DEMCZ_ERR_STATE, "demcz_propose: call demcz_set_state first");
    if (g < 1 || ib < 0 || ib >= h->cfg.Nblocks) return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_propose: bad generation or block");
    if (h->proposal_pending) return fail(h, DEMCZ_ERR_STATE, "demcz_propose: previous proposal not committed");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int64_t N = h->cfg.N;
    const int d = h->cfg.d;
    h->gen_open = true;
    int64_t off = 0;
    for (int t = 0; t < ib; ++t) off += blockstep_nblk(h->block_offsets[t + 1] - h->block_offsets[t]);
    WindowParams P{};
    P.Z = h->dZ; P.ZS = h->ZS; P.M = h->M; P.Xcur = h->dX; P.N = N; P.chain_id0 = h->cfg.chain_id0; P.d = d;
    P.gamma = gamma; P.seed = h->cfg.seed; P.block_offsets = h->d_block_offsets; P.slot_of = h->d_slot_of; P.eps = h->d_eps;
    const uint64_t blk0 = (uint64_t)(g + h->rng_offset - 1) * (uint64_t)h->S + (uint64_t)off;
    if (h->hc_X) {
        // pipelined: the kernel writes the proposals into pinned host memory itself and raises the flag; the host spins on it
        const unsigned int seq = ++h->hc_seq;
        double* hX = nullptr; unsigned int* hF = nullptr;
        HIPCHK(h, hipHostGetDevicePointer((void**)&hX, h->hc_X, 0));
        HIPCHK(h, hipHostGetDevicePointer((void**)&hF, const_cast<unsigned int*>(h->hc_flag), 0));
        hipLaunchKernelGGL(propose_kernel, dim3((unsigned)((N + WINDOW_BS - 1) / WINDOW_BS)), dim3(WINDOW_BS),
                           (size_t)(d + 1) * WINDOW_BS * sizeof(double), h->stream, P, (int)ib, blk0, h->dXprop, h->dlogu, hX, h->hc_count, hF, seq);
        HIPCHK(h, hipGetLastError());
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned long long polls = 0; __atomic_load_n(const_cast<unsigned int*>(h->hc_flag), __ATOMIC_ACQUIRE) != seq; ++polls) {
            if ((polls & 0xffffull) == 0xffffull) {       // (now and then: did the launch fail?  is a sharded peer stalling the stream?)
                const hipError_t qe = hipStreamQuery(h->stream);
                if (qe != hipSuccess && qe != hipErrorNotReady) return fail(h, DEMCZ_ERR_HIP, std::string("demcz_propose: ") + hipGetErrorString(qe));
                if (qe == hipSuccess && __atomic_load_n(const_cast<unsigned int*>(h->hc_flag), __ATOMIC_ACQUIRE) != seq)
                    return fail(h, DEMCZ_ERR_HIP, "demcz_propose: the propose kernel finished without raising its flag");
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return fail(h, DEMCZ_ERR_HIP, "demcz_propose: no proposal within 60 s");
            }
        }
        if (Xprop && Xprop != h->hc_X) std::memcpy(Xprop, h->hc_X, (size_t)N * d * sizeof(double));
    } else {
        if (!Xprop) return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_propose: Xprop is required (or call demcz_closure_buffers first)");
        hipLaunchKernelGGL(propose_kernel, dim3((unsigned)((N + WINDOW_BS - 1) / WINDOW_BS)), dim3(WINDOW_BS),
                           (size_t)(d + 1) * WINDOW_BS * sizeof(double), h->stream, P, (int)ib, blk0, h->dXprop, h->dlogu,
                           (double*)nullptr, (unsigned int*)nullptr, (unsigned int*)nullptr, 0u);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(Xprop, h->dXprop, (size_t)N * d * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        SYNCCHK(h, h->stream);
    }
    h->proposal_pending = true;
    return DEMCZ_OK;
}

// Pinned host buffers for the closure mode: *Xprop (N x d column-major, ld N) receives every proposal of demcz_propose, *logp (N) is
// where the caller leaves the closure's values for demcz_accept_commit -- the kernels read and write them in place (see
// demcz_handle::hc_X).  The buffers belong to the handle and live as long as it does.
extern "C" int32_t demcz_closure_buffers(demcz_handle* h, double** Xprop, double** logp)
{
    if (!h || !Xprop || !logp) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (h->cfg.target_kind != DEMCZ_TARGET_HOST_CALLBACK) return fail(h, DEMCZ_ERR_STATE, "demcz_closure_buffers: handle was not created with DEMCZ_TARGET_HOST_CALLBACK");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (!h->hc_X) {
        const size_t N = (size_t)h->cfg.N, d = (size_t)h->cfg.d;
        double* x = nullptr; double* l = nullptr; unsigned int* f = nullptr;
        if (hipHostMalloc((void**)&x, N * d * sizeof(double), hipHostMallocMapped) != hipSuccess ||
            hipHostMalloc((void**)&l, N * sizeof(double), hipHostMallocMapped) != hipSuccess ||
            hipHostMalloc((void**)&f, 64, hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            if (x) (void)hipHostFree(x);
            if (l) (void)hipHostFree(l);
            if (f) (void)hipHostFree(f);
            return fail(h, DEMCZ_ERR_HIP, "demcz_closure_buffers: pinned allocation failed");
        }
        std::memset(x, 0, N * d * sizeof(double)); std::memset(l, 0, N * sizeof(double)); std::memset(f, 0, 64);
        HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->hc_count, sizeof(unsigned int)));
        HIPCHK(h, hipMemsetAsync(h->hc_count, 0, sizeof(unsigned int), h->stream));
        h->hc_X = x; h->hc_lp = l; h->hc_flag = f;
    }
    *Xprop = h->hc_X;
    *logp = h->hc_lp;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_accept_commit(demcz_handle* h, const double* logp_prop, const double* temperature)
{
    if (!h || (!logp_prop && !h->hc_lp)) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->proposal_pending) return fail(h, DEMCZ_ERR_STATE, "demcz_accept_commit: no pending proposal");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int64_t N = h->cfg.N;
    if (h->hc_lp) {
        // pipelined: the kernel reads the log-densities from pinned host memory; nothing is waited for -- the next demcz_propose is
        // enqueued behind this kernel, and the caller does not write the buffer again before that proposal has come back
        if (logp_prop && logp_prop != h->hc_lp) std::memcpy(h->hc_lp, logp_prop, (size_t)N * sizeof(double));
        double* lpd = nullptr;
        HIPCHK(h, hipHostGetDevicePointer((void**)&lpd, h->hc_lp, 0));
        hipLaunchKernelGGL(accept_commit_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, N, h->cfg.d, h->dX,
                           h->dlp, h->dXprop, (const double*)lpd, h->dlogu, temperature ? 1 : 0, temperature ? *temperature : 1.0);
        HIPCHK(h, hipGetLastError());
        h->proposal_pending = false;
        return DEMCZ_OK;
    }
    int32_t rc = ensure_scratch(h, N);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_scratch, logp_prop, (size_t)N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(accept_commit_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, N, h->cfg.d, h->dX,
                       h->dlp, h->dXprop, h->d_scratch, h->dlogu, temperature ? 1 : 0, temperature ? *temperature : 1.0);
    HIPCHK(h, hipGetLastError());
    SYNCCHK(h, h->stream);
    h->proposal_pending = false;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_end_generation(demcz_handle* h, int64_t g)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (h->cfg.target_kind != DEMCZ_TARGET_HOST_CALLBACK) return fail(h, DEMCZ_ERR_STATE, "demcz_end_generation: host-callback handles only");
    if (h->proposal_pending || !h->gen_open) return fail(h, DEMCZ_ERR_STATE, "demcz_end_generation: no open generation or uncommitted proposal");
    const bool hist = h->cfg.Gcap > 0;
    if (hist && (g - h->g0 - 1 < 0 || g - h->g0 > h->cfg.Gcap)) return fail(h, DEMCZ_ERR_CAPACITY, "demcz_end_generation: outside the history window");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int64_t N = h->cfg.N;
    const bool boundary = (g % h->cfg.K) == 0;
    const bool sharded = (h->comm != nullptr);
    const bool kappend = boundary && !sharded && !h->external_append;
    if (h->lag != 0) return fail(h, DEMCZ_ERR_STATE, "demcz_end_generation: the host-closure path runs with append lag 0");
    if (kappend && h->M_app + N > h->cfg.Mcap) return fail(h, DEMCZ_ERR_CAPACITY, "demcz_end_generation: Z capacity exceeded");
    hipLaunchKernelGGL(end_generation_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, h->stream, N, h->cfg.d, h->dX, h->dlp,
                       hist ? h->dchain : nullptr, hist ? h->dlogobj : nullptr,
                       hist ? (g - h->g0 - 1) : 0, h->dZ, h->ZS, h->M, kappend ? 1 : 0);
    HIPCHK(h, hipGetLastError());
    if (kappend) { h->M_app += N; h->M = h->M_app; }
    else if (boundary && sharded) { int32_t rc = append_after_window(h); if (rc) return rc; }
    h->gen_open = false;
    h->g_done = g;
    return DEMCZ_OK;
}

// ---- multi-GPU -------------------------------------------------------------------------------------
extern "C" int32_t demcz_comm_unique_id(void* unique_id_128B)
{
    if (!unique_id_128B) return DEMCZ_ERR_INVALID_ARGUMENT;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return fail(nullptr, DEMCZ_ERR_HIP, "ncclGetUniqueId failed");
    std::memcpy(unique_id_128B, &id, sizeof(id));
    return DEMCZ_OK;
}

extern "C" int32_t demcz_comm_init(demcz_handle* h, const void* unique_id_128B, int32_t nranks, int32_t rank)
{
    if (!h || !unique_id_128B || nranks < 1 || rank < 0 || rank >= nranks) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (h->comm) return fail(h, DEMCZ_ERR_STATE, "demcz_comm_init: communicator already initialised");
    if (!h->live_log.empty()) { int32_t rcv = live_verify(h); if (rcv) return rcv; }
    if (h->cfg.chain_id0 != (int64_t)rank * h->cfg.N)
        return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_comm_init: chain_id0 must be rank * N (equal shards in rank order)");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    ncclUniqueId id;
    std::memcpy(&id, unique_id_128B, sizeof(id));
    NCCLCHK(h, ncclCommInitRank(&h->comm, nranks, id, rank));
    h->nranks = nranks;
    h->rank = rank;
    if (const char* tenv = getenv("DEMCZ_COMM_TIMEOUT_MS")) h->comm_timeout_ms = std::max<long long>(0, atoll(tenv));
    HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->d_gather, (size_t)h->cfg.N * h->cfg.d * nranks * sizeof(double)));
    return peer_setup_ipc(h);
}

// ---- replicas that publish into each other (round 4) ----------------------------------------------------------------------------
extern "C" int32_t demcz_peer_group(demcz_handle** handles, int32_t R)
{
    if (!handles || R < 2 || R > DEMCZ_MAX_PEERS + 1) return DEMCZ_ERR_INVALID_ARGUMENT;
    demcz_handle* h0 = handles[0];
    if (!h0) return DEMCZ_ERR_INVALID_ARGUMENT;
    for (int r = 0; r < R; ++r) {
        demcz_handle* m = handles[r];
        if (!m) return DEMCZ_ERR_INVALID_ARGUMENT;
        for (int q = 0; q < r; ++q) if (handles[q] == m) return fail(m, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_peer_group: a handle appears twice");
        if (m->comm || m->peer_mode != 0) return fail(m, DEMCZ_ERR_STATE, "demcz_peer_group: handle is already sharded");
        if (m->lag != 0 || m->external_append) return fail(m, DEMCZ_ERR_STATE, "demcz_peer_group: append lag 0 and library-owned appends only");
        if (!m->live_log.empty()) { int32_t rcv = live_verify(m); if (rcv) return rcv; }
        if (m->cfg.device_id != h0->cfg.device_id || m->cfg.N != h0->cfg.N || m->cfg.d != h0->cfg.d || m->cfg.K != h0->cfg.K ||
            m->cfg.Mcap != h0->cfg.Mcap || m->cfg.seed != h0->cfg.seed || m->ZS != h0->ZS || m->lanes != h0->lanes ||
            m->split_kind != h0->split_kind || m->cfg.target_kind != h0->cfg.target_kind || m->cfg.Nblocks != h0->cfg.Nblocks)
            return fail(m, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_peer_group: the members must be shards of ONE run: same device, N, d, K, Mcap, seed, target, layout");
        if (m->cfg.chain_id0 != (int64_t)r * m->cfg.N)
            return fail(m, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_peer_group: chain_id0 of member r must be r * N (equal shards in rank order)");
        if (m->M != h0->M || m->M_app != h0->M_app || m->g_done != h0->g_done)
            return fail(m, DEMCZ_ERR_STATE, "demcz_peer_group: the members' archives / generation counters differ");
    }
    // The members' launches wait for each other's rows, so they must be able to RUN at the same time.  HIP multiplexes its streams
    // over a few hardware queues (GPU_MAX_HW_QUEUES, four by default), and two kernels on one hardware queue run one after the
    // other whatever streams they came from: two members whose streams share a queue would wait for each other until the poll
    // limit.  So: a rendezvous kernel on every member's stream (each waits, at most 2 ms, until all R have started); if they do
    // not all meet, the members that own their stream get fresh ones (a new stream goes to the least used queue) and the
    // rendezvous is tried again.  No luck after a few rounds: the group runs in lockstep from the start (G->failed).
    bool concurrent = false;
    {
        unsigned int* d_rv = nullptr;
        HIPCHK(h0, hipSetDevice(h0->cfg.device_id));
        HIPCHK(h0, dev_malloc(h0->cfg.device_id, (void**)&d_rv, 2 * sizeof(unsigned int)));
        std::vector<hipStream_t> spare;
        for (int round = 0; round < 6 && !concurrent; ++round) {
            if (hipMemset(d_rv, 0, 2 * sizeof(unsigned int)) != hipSuccess) break;
            for (int r = 0; r < R; ++r)
                hipLaunchKernelGGL(rendezvous_kernel, dim3(1), dim3(1), 0, handles[r]->stream, d_rv, (unsigned int)R, d_rv + 1, 200000ull);
            bool okl = hipGetLastError() == hipSuccess;
            for (int r = 0; r < R; ++r) okl = (hipStreamSynchronize(handles[r]->stream) == hipSuccess) && okl;
            unsigned int rv[2] = {0, 0};
            if (!okl || hipMemcpy(rv, d_rv, sizeof(rv), hipMemcpyDeviceToHost) != hipSuccess) break;
            if (rv[1] == (unsigned int)R) { concurrent = true; break; }
            for (int r = 0; r < R; ++r) {
                demcz_handle* m = handles[r];
                if (!m->own_stream) continue;
                hipStream_t ns = nullptr;
                if (hipStreamCreateWithFlags(&ns, hipStreamNonBlocking) != hipSuccess) continue;
                spare.push_back(m->stream);             // (kept alive until the group's streams are settled: the queue
                m->stream = ns;                          //  assignment of the next new stream depends on what exists)
                m->after_launch_ev = nullptr;
            }
        }
        for (hipStream_t st : spare) stream_release(h0->cfg.device_id, st, true);
        (void)dev_free(h0->cfg.device_id, d_rv);
    }
    PeerGroup* G = new PeerGroup();
    for (int r = 0; r < R; ++r) G->members.push_back(handles[r]);
    G->rearms_left = h0->live_rearms_left;
    for (int r = 0; r < R; ++r) {
        demcz_handle* m = handles[r];
        peer_no_dual(m);
        m->group = G;
        m->peer_mode = 1;
        m->nranks = R;
        m->rank = r;
        m->n_peers = 0;
        for (int q = 0; q < R; ++q) if (q != r) m->peer_Z[m->n_peers++] = handles[q]->dZ;
        rec_invalidate(m);                  // (draws made against N rows per boundary no longer apply)
    }
    // Hand-off inside the launches: a layout that has it, streams that can overlap -- and EVERY member able to issue LIVE launches
    // (its shard within what a launch may hold, its share of the device's LIVE budget granted).  All of them or none: a member
    // without LIVE launches only logs its calls (demcz_run), and they are executed for the whole group, in lockstep, at the next
    // verification -- which group_verify does when the group is marked `failed`.
    bool all_live = peer_capable(h0) && concurrent;
    for (int r = 0; r < R && all_live; ++r) all_live = live_span(handles[r]) > 0;
    if (!all_live) for (int r = 0; r < R; ++r) live_release(handles[r]);
    G->lockstep_only = !all_live;
    G->failed = G->lockstep_only;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_get_peer_status(const demcz_handle* h, int32_t* mode, int32_t* peers)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (mode) *mode = h->peer_mode;
    if (peers) *peers = h->n_peers;
    return DEMCZ_OK;
}

// The archive moves into a fine-grained allocation of its own (coherent for writers on other GPUs while kernels run; the pool's
// buffers are ordinary coarse-grained device memory), with whatever it -- and, in the arena, the record buffers behind it --
// already holds (set_state may have run), and is exported with hipIpcGetMemHandle.  `ok` = false: an allocation flag or IPC
// refused; the handle is unchanged then.
static int32_t archive_make_fine(demcz_handle* h, hipIpcMemHandle_t* mh, bool* ok)
{
    *ok = false;
    if (h->archive_fine) {
        *ok = hipIpcGetMemHandle(mh, h->dZ) == hipSuccess;
        if (!*ok) (void)hipGetLastError();
        return DEMCZ_OK;
    }
    double* fine = nullptr;
    const size_t box_off = (h->dZ_bytes + 255) & ~(size_t)255;
    if (hipExtMallocWithFlags((void**)&fine, box_off + PEER_MAILBOX_BYTES, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return DEMCZ_OK; }
    if (hipIpcGetMemHandle(mh, fine) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(fine); return DEMCZ_OK; }
    { int32_t rcq = quiesce_all(h); if (rcq) { (void)hipFree(fine); return rcq; } }
    if (h->prod_stream) (void)hipStreamSynchronize(h->prod_stream);
    if (hipMemcpyAsync(fine, h->dZ, h->dZ_bytes, hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
        hipMemsetAsync(reinterpret_cast<unsigned char*>(fine) + box_off, 0, PEER_MAILBOX_BYTES, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) {
        (void)hipFree(fine);
        return fail(h, DEMCZ_ERR_HIP, "copy into the fine-grained archive failed");
    }
    rec_invalidate(h);
    const ptrdiff_t shift = reinterpret_cast<unsigned char*>(fine) - reinterpret_cast<unsigned char*>(h->dZ);
    auto rebase = [&](double*& q) { if (q) q = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(q) + shift); };
    if (h->arena) {
        rebase(h->arena_rec[0]); rebase(h->arena_rec[1]); rebase(h->arena_temp);
        if (h->rec_in_arena) { rebase(h->d_rec[0]); rebase(h->d_rec[1]); }
    }
    g_dev_pool.release(h->dZ, h->cfg.device_id);
    h->dZ = fine;
    h->archive_fine = true;
    h->mailbox_off = box_off;
    *ok = true;
    return DEMCZ_OK;
}

// opens the other ranks' archives (handles in rank order, this rank's own skipped); all or nothing
static bool peers_open(demcz_handle* h, const hipIpcMemHandle_t* handles, int R, int rank)
{
    void* mapped[DEMCZ_MAX_PEERS] = {nullptr};
    int nmap = 0;
    for (int r = 0; r < R; ++r) {
        if (r == rank) continue;
        void* ptr = nullptr;
        if (hipIpcOpenMemHandle(&ptr, handles[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            for (int i = 0; i < nmap; ++i) (void)hipIpcCloseMemHandle(mapped[i]);
            return false;
        }
        mapped[nmap++] = ptr;
    }
    for (int i = 0; i < nmap; ++i) { h->ipc_mapped[i] = mapped[i]; h->peer_Z[i] = reinterpret_cast<double*>(mapped[i]); }
    h->n_peers = nmap;
    return true;
}

// demcz_comm_init, second half: every rank's archive becomes a fine-grained allocation of its own, its IPC handle travels in one
// ncclAllGather, and every rank opens the other ranks' archives.  Any refusal anywhere (an allocation flag, IPC, peer access)
// switches the mode off on ALL ranks (min-reduced), and the run exchanges its rows through ncclAllGather as before.  Returns
// DEMCZ_OK either way unless the communicator itself fails.
static int32_t peer_setup_ipc(demcz_handle* h)
{
    const bool off = getenv("DEMCZ_NO_PEER") != nullptr;
    const bool self = getenv("DEMCZ_PEER_SELF") != nullptr;      // a one-rank communicator walks the path too (tests)
    const int R = h->nranks;
    if (off || R > DEMCZ_MAX_PEERS + 1 || (R < 2 && !self)) return DEMCZ_OK;
    HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->d_err_all, 4 * sizeof(unsigned int)));
    HIPCHK(h, hipMemsetAsync(h->d_err_all, 0, 4 * sizeof(unsigned int), h->stream));
    struct Rec { hipIpcMemHandle_t mh; int32_t ok; int32_t pad; };
    static_assert(sizeof(Rec) % 8 == 0, "all-gather record");
    Rec mine{};
    if (peer_capable(h)) {
        bool okf = false;
        int32_t rcf = archive_make_fine(h, &mine.mh, &okf);
        if (rcf) return rcf;
        mine.ok = okf ? 1 : 0;
    }
    Rec* d_rec = nullptr;
    HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&d_rec, sizeof(Rec) * (size_t)(R + 1)));
    std::vector<Rec> all((size_t)R);
    int32_t rc = DEMCZ_OK;
    auto cleanup = [&]() { if (d_rec) (void)dev_free(h->cfg.device_id, d_rec); };
    if (hipMemcpyAsync(d_rec + R, &mine, sizeof(Rec), hipMemcpyHostToDevice, h->stream) != hipSuccess) { cleanup(); return fail(h, DEMCZ_ERR_HIP, "demcz_comm_init: upload failed"); }
    if (ncclAllGather(d_rec + R, d_rec, sizeof(Rec), ncclChar, h->comm, h->stream) != ncclSuccess) { cleanup(); return fail(h, DEMCZ_ERR_COMM, "demcz_comm_init: all-gather of the IPC handles failed"); }
    rc = sync_stream(h, h->stream, "demcz_comm_init (IPC handles)");
    if (rc == DEMCZ_OK && hipMemcpy(all.data(), d_rec, sizeof(Rec) * (size_t)R, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(h, DEMCZ_ERR_HIP, "demcz_comm_init: download failed");
    if (rc) { cleanup(); return rc; }
    unsigned int ok = 1u;
    for (int r = 0; r < R; ++r) ok &= all[(size_t)r].ok ? 1u : 0u;
    if (ok) {
        std::vector<hipIpcMemHandle_t> hs((size_t)R);
        for (int r = 0; r < R; ++r) hs[(size_t)r] = all[(size_t)r].mh;
        ok = peers_open(h, hs.data(), R, h->rank) ? 1u : 0u;
    }
    // every rank must have opened every archive, or nobody uses any
    unsigned int* d_ok = reinterpret_cast<unsigned int*>(d_rec);
    if (hipMemcpyAsync(d_ok, &ok, sizeof(ok), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        ncclAllReduce(d_ok, d_ok, 1, ncclUint32, ncclMin, h->comm, h->stream) != ncclSuccess) rc = fail(h, DEMCZ_ERR_COMM, "demcz_comm_init: reduction failed");
    if (rc == DEMCZ_OK) rc = sync_stream(h, h->stream, "demcz_comm_init (IPC agreement)");
    unsigned int all_ok = 0u;
    if (rc == DEMCZ_OK && hipMemcpy(&all_ok, d_ok, sizeof(all_ok), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(h, DEMCZ_ERR_HIP, "demcz_comm_init: download failed");
    auto close_all = [&]() {
        for (int i = 0; i < DEMCZ_MAX_PEERS; ++i) if (h->ipc_mapped[i]) { (void)hipIpcCloseMemHandle(h->ipc_mapped[i]); h->ipc_mapped[i] = nullptr; }
        h->n_peers = 0;
    };
    if (rc || !all_ok) {
        cleanup();
        close_all();
        return rc;
    }
    // Second agreement (round 5), two questions in one reduction:
    //  * first contact (peer_ping_kernel): does a token stored from every peer's running kernel reach this rank's polling kernel
    //    through the mappings just opened, within 200 ms?  The hand-off has only ever run between processes on ONE GPU before
    //    a multi-GPU node sees it; this asks the links themselves, before any row depends on the answer;
    //  * can THIS rank issue LIVE launches at all -- live_span() depends on the process's LIVE budget on its device (another handle
    //    of the process may hold it) and on the occupancy query?  A rank that exchanged through ncclAllGather
    //    while its peers publish and poll would stall all of them until the communicator's deadline.
    // Either every rank hands its rows over inside the launches, or none does.
    h->peer_mode = 2;
    peer_no_dual(h);
    unsigned int* d_res = reinterpret_cast<unsigned int*>(d_rec) + 4;
    unsigned int agree = 1u;
    {
        PingBoxes pb{};
        for (int i = 0; i < h->n_peers; ++i) pb.box[i] = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(h->peer_Z[i]) + h->mailbox_off);
        unsigned long long* own = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(h->dZ) + h->mailbox_off);
        static const unsigned long long ping_ms = getenv("DEMCZ_PING_MS") ? (unsigned long long)atol(getenv("DEMCZ_PING_MS")) : 200ull;
        hipLaunchKernelGGL(peer_ping_kernel, dim3(1), dim3(64), 0, h->stream, own, pb, h->n_peers, h->rank, R, 0xC0FFEE0000000001ull,
                           ping_ms * 100000ull, d_res);
        unsigned int res[2] = {0, 0};
        if (hipGetLastError() != hipSuccess) rc = fail(h, DEMCZ_ERR_HIP, "demcz_comm_init: ping launch failed");
        if (rc == DEMCZ_OK) rc = sync_stream(h, h->stream, "demcz_comm_init (first contact)");
        if (rc == DEMCZ_OK && hipMemcpy(res, d_res, sizeof(res), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(h, DEMCZ_ERR_HIP, "demcz_comm_init: download failed");
        h->ping_wait_us = (double)res[1] / 100.0;
        const bool can_live = rc == DEMCZ_OK && live_span(h) > 0;
        agree = (res[0] && can_live) ? 1u : 0u;
        if (getenv("DEMCZ_DEBUG_LIVE")) fprintf(stderr, "[demcz] rank %d: first contact %s after %.1f us, LIVE launches %s\n", h->rank, res[0] ? "ok" : "FAILED", h->ping_wait_us, can_live ? "possible" : "not possible");
    }
    if (rc == DEMCZ_OK && (hipMemcpyAsync(d_ok, &agree, sizeof(agree), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
                           ncclAllReduce(d_ok, d_ok, 1, ncclUint32, ncclMin, h->comm, h->stream) != ncclSuccess)) rc = fail(h, DEMCZ_ERR_COMM, "demcz_comm_init: reduction failed");
    if (rc == DEMCZ_OK) rc = sync_stream(h, h->stream, "demcz_comm_init (hand-off agreement)");
    unsigned int all_agree = 0u;
    if (rc == DEMCZ_OK && hipMemcpy(&all_agree, d_ok, sizeof(all_agree), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(h, DEMCZ_ERR_HIP, "demcz_comm_init: download failed");
    cleanup();
    h->ping_ok = (rc == DEMCZ_OK && all_agree) ? 1 : 0;
    if (rc || !all_agree) {
        h->peer_mode = 0;
        live_release(h);
        close_all();
        return rc;
    }
    h->peer_fence = true;
    // a second communicator and stream for the monitoring R-hat of a finished slab (rhat_enqueue): its two small all-reduces then
    // run beside the next slab's launch instead of holding every rank's compute stream until all ranks have joined
    // (collective: every rank is here -- the agreement above was unanimous)
    if (!h->comm_stream) HIPCHK(h, stream_acquire(h->cfg.device_id, &h->comm_stream));
    if (!h->comm_side) NCCLCHK(h, ncclCommSplit(h->comm, 0, h->rank, &h->comm_side, nullptr));
    return DEMCZ_OK;
}

// The same set-up with the HOST carrying the handles (any transport: torch.distributed, MPI, Distributed.jl) and doing the two
// things a communicator does in mode 2: the ranks must meet (a barrier of the host's) between demcz_set_state and the first
// demcz_run, and between the last synchronising call and demcz_destroy; and a hand-off that timed out is an error on the rank
// that saw it (DEMCZ_ERR_STATE), not an automatic redo -- the ranks have no way here to agree on one.
extern "C" int32_t demcz_peer_export(demcz_handle* h, int32_t nranks, int32_t rank, void* handle_64B)
{
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    if (!h || !handle_64B || nranks < 2 || nranks > DEMCZ_MAX_PEERS + 1 || rank < 0 || rank >= nranks) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (h->comm || h->peer_mode != 0) return fail(h, DEMCZ_ERR_STATE, "demcz_peer_export: handle is already sharded");
    if (!peer_capable(h)) return fail(h, DEMCZ_ERR_STATE, "demcz_peer_export: this layout has no in-launch hand-off");
    if (h->cfg.chain_id0 != (int64_t)rank * h->cfg.N) return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_peer_export: chain_id0 must be rank * N");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (!h->live_log.empty()) { int32_t rcv = live_verify(h); if (rcv) return rcv; }
    hipIpcMemHandle_t mh;
    bool ok = false;
    int32_t rc = archive_make_fine(h, &mh, &ok);
    if (rc) return rc;
    if (!ok) return fail(h, DEMCZ_ERR_HIP, "demcz_peer_export: fine-grained allocation or hipIpcGetMemHandle refused");
    std::memcpy(handle_64B, &mh, sizeof(mh));
    h->nranks = nranks;
    h->rank = rank;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_peer_attach(demcz_handle* h, const void* handles_64B_each)
{
    if (!h || !handles_64B_each) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->archive_fine || h->peer_mode != 0 || h->nranks < 2) return fail(h, DEMCZ_ERR_STATE, "demcz_peer_attach: call demcz_peer_export first");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    std::vector<hipIpcMemHandle_t> hs((size_t)h->nranks);
    std::memcpy(hs.data(), handles_64B_each, sizeof(hipIpcMemHandle_t) * (size_t)h->nranks);
    if (!peers_open(h, hs.data(), h->nranks, h->rank)) return fail(h, DEMCZ_ERR_HIP, "demcz_peer_attach: hipIpcOpenMemHandle refused (peer access between the devices?)");
    peer_no_dual(h);
    h->peer_mode = 3;
    rec_invalidate(h);
    return DEMCZ_OK;
}

// Mode 3, the orderly end: a rank's exported archive may only be freed once no other rank has it mapped.  The host makes the
// ranks meet after their last synchronising call, every rank calls demcz_peer_detach (its results stay readable, it runs no
// further generations), the host makes them meet once more, and only then does any rank call demcz_destroy.
extern "C" int32_t demcz_peer_detach(demcz_handle* h)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (h->peer_mode != 3) return fail(h, DEMCZ_ERR_STATE, "demcz_peer_detach: host-mediated IPC peers only (demcz_peer_export / demcz_peer_attach)");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    int32_t rc = live_verify(h);
    if (rc) return rc;
    for (hipStream_t st : {h->stream, h->prod_stream, h->diag_stream})
        if (st) HIPCHK(h, hipStreamSynchronize(st));
    for (int r = 0; r < DEMCZ_MAX_PEERS; ++r)
        if (h->ipc_mapped[r]) { (void)hipIpcCloseMemHandle(h->ipc_mapped[r]); h->ipc_mapped[r] = nullptr; }
    h->peers_closed = true;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_get_peer_ping(const demcz_handle* h, int32_t* ok, double* wait_us)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (ok) *ok = h->ping_ok;
    if (wait_us) *wait_us = h->ping_wait_us;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_set_comm_timeout(demcz_handle* h, int64_t milliseconds)
{
    if (!h || milliseconds < 0) return DEMCZ_ERR_INVALID_ARGUMENT;
    h->comm_timeout_ms = milliseconds;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_debug_stall_exchange(demcz_handle* h, int32_t milliseconds)
{
    if (!h || milliseconds < 0 || milliseconds > 10000) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->comm) return fail(h, DEMCZ_ERR_STATE, "demcz_debug_stall_exchange: not a sharded handle (demcz_comm_init)");
    h->stall_next_ms = milliseconds;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_export_current_device(demcz_handle* h, double* X_device)
{
    if (!h || !X_device) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->has_state) return fail(h, DEMCZ_ERR_STATE, "demcz_export_current_device: no state");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    HIPCHK(h, hipMemcpyAsync(X_device, h->dX, (size_t)h->cfg.N * h->cfg.d * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return DEMCZ_OK;
}

extern "C" int32_t demcz_append_rows_device(demcz_handle* h, const double* rows_device, int64_t nrows, int64_t ldrows)
{
    if (!h || !rows_device || nrows < 1 || ldrows < nrows) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->has_state) return fail(h, DEMCZ_ERR_STATE, "demcz_append_rows_device: no state");
    if (h->M_app + nrows > h->cfg.Mcap) return fail(h, DEMCZ_ERR_CAPACITY, "demcz_append_rows_device: Z capacity exceeded");
    if (h->lag != 0) return fail(h, DEMCZ_ERR_STATE, "demcz_append_rows_device: caller-driven appends need append lag 0 on the handle");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    const int64_t tot = nrows * h->cfg.d;
    hipLaunchKernelGGL(append_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->dZ, h->ZS, h->M_app,
                       rows_device, nrows, ldrows, h->cfg.d);
    HIPCHK(h, hipGetLastError());
    h->M_app += nrows;
    h->M = h->M_app;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_append_rows(demcz_handle* h, const double* rows, int64_t nrows, int64_t ldrows)
{
    if (!h || !rows || nrows < 1 || ldrows < nrows) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->has_state) return fail(h, DEMCZ_ERR_STATE, "demcz_append_rows: no state");
    if (h->M_app + nrows > h->cfg.Mcap) return fail(h, DEMCZ_ERR_CAPACITY, "demcz_append_rows: Z capacity exceeded");
    if (h->lag != 0) return fail(h, DEMCZ_ERR_STATE, "demcz_append_rows: caller-driven appends need append lag 0 on the handle");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    int32_t rcz = ensure_scratch(h, nrows * h->cfg.d);
    if (rcz) return rcz;
    HIPCHK(h, hipMemcpy2DAsync(h->d_scratch, (size_t)nrows * sizeof(double), rows, (size_t)ldrows * sizeof(double),
                               (size_t)nrows * sizeof(double), (size_t)h->cfg.d, hipMemcpyHostToDevice, h->stream));
    const int64_t tot = nrows * h->cfg.d;
    hipLaunchKernelGGL(append_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->dZ, h->ZS, h->M_app,
                       (const double*)h->d_scratch, nrows, nrows, h->cfg.d);
    HIPCHK(h, hipGetLastError());
    SYNCCHK(h, h->stream);   // the caller may reuse `rows` on return
    h->M_app += nrows;
    h->M = h->M_app;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_set_external_append(demcz_handle* h, int32_t enabled)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->live_log.empty()) { int32_t rcv = live_verify(h); if (rcv) return rcv; }
    h->external_append = enabled != 0;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_set_append_lag(demcz_handle* h, int32_t batches)
{
    if (!h || batches < 0 || batches > 64) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->live_log.empty()) { int32_t rcv = live_verify(h); if (rcv) return rcv; }
    if (h->M_app != h->M || !h->pending.empty()) return fail(h, DEMCZ_ERR_STATE, "demcz_set_append_lag: rows are still pending");
    if (h->external_append && batches) return fail(h, DEMCZ_ERR_STATE, "demcz_set_append_lag: caller-driven appends schedule their own visibility");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (h->comm && batches > 0) {
        // side stream + double-buffered batch slabs: [E][d][n_loc] out, [R][E][d][n_loc] in
        if (!h->comm_stream) HIPCHK(h, stream_acquire(h->cfg.device_id, &h->comm_stream));
        // a communicator of its own for the side stream (collective over the parent: every rank makes this call)
        // (nothing of the parent communicator may still be queued on the compute stream when it is split)
        SYNCCHK(h, h->stream);
        SYNCCHK(h, h->comm_stream);
        if (!h->comm_side) NCCLCHK(h, ncclCommSplit(h->comm, 0, h->rank, &h->comm_side, nullptr));
        const size_t one = (size_t)h->cfg.N * h->cfg.d * sizeof(double);
        for (int b = 0; b < 2; ++b) {
            if (h->d_send[b]) HIPCHK(h, dev_free(h->cfg.device_id, h->d_send[b]));
            if (h->d_recv[b]) HIPCHK(h, dev_free(h->cfg.device_id, h->d_recv[b]));
            h->d_send[b] = h->d_recv[b] = nullptr;
            HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->d_send[b], one * batches));
            HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->d_recv[b], one * batches * h->nranks));
            if (!h->buf_done[b]) {
                HIPCHK(h, hipEventCreateWithFlags(&h->buf_done[b], hipEventDisableTiming));
                HIPCHK(h, hipEventRecord(h->buf_done[b], h->comm_stream));
            }
        }
    }
    h->lag = batches;
    rec_invalidate(h);
    h->batch_cnt = 0; h->batch_buf = 0; h->batch_J = -1;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_set_rng_offset(demcz_handle* h, int64_t generations)
{
    if (!h || generations < 0) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->live_log.empty()) { int32_t rcv = live_verify(h); if (rcv) return rcv; }
    h->rng_offset = generations;
    rec_invalidate(h);
    return DEMCZ_OK;
}

// ---- stateless diagnostics on caller arrays (src/utils.jl) ------------------------------------------
namespace {
struct ScratchHandle {
    demcz_handle h;
    int32_t open(int32_t device_id, int64_t N, int d, int64_t G, const double* chain, const double* log_obj)
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, DEMCZ_ERR_NO_DEVICE, "no HIP device visible (there is no CPU fallback)");
        if (device_id < 0 || device_id >= ndev || N < 1 || d < 1 || G < 1) return fail(nullptr, DEMCZ_ERR_INVALID_ARGUMENT, "bad device or shape");
        h.cfg.N = N; h.cfg.d = d; h.cfg.Gcap = G; h.cfg.device_id = device_id; h.g0 = 0; h.nranks = 1;
        HIPCHK(&h, hipSetDevice(device_id));
        HIPCHK(&h, hipStreamCreateWithFlags(&h.stream, hipStreamNonBlocking));
        h.own_stream = true;
        if (chain) {
            HIPCHK(&h, hipMalloc((void**)&h.dchain, (size_t)N * d * G * sizeof(double)));
            HIPCHK(&h, hipMemcpyAsync(h.dchain, chain, (size_t)N * d * G * sizeof(double), hipMemcpyHostToDevice, h.stream));
        }
        if (log_obj) {
            HIPCHK(&h, hipMalloc((void**)&h.dlogobj, (size_t)N * G * sizeof(double)));
            HIPCHK(&h, hipMemcpyAsync(h.dlogobj, log_obj, (size_t)N * G * sizeof(double), hipMemcpyHostToDevice, h.stream));
        }
        h.stage_cap = 4096 + (int64_t)d * (d + 1);
        HIPCHK(&h, hipHostMalloc((void**)&h.d_stage, (size_t)h.stage_cap * sizeof(double), hipHostMallocDefault));
        return DEMCZ_OK;
    }
    ~ScratchHandle()
    {
        if (h.stream) (void)hipStreamSynchronize(h.stream);
        free_all(&h);
    }
};
int32_t diag_fail(const ScratchHandle& s, int32_t rc)
{
    if (rc) g_create_error = s.h.err;
    return rc;
}
}  // namespace

extern "C" int32_t demcz_rhat_array(int32_t device_id, const double* chain, int64_t N, int32_t d, int64_t G, double* rhat)
{
    if (!chain || !rhat) return DEMCZ_ERR_INVALID_ARGUMENT;
    ScratchHandle s;
    int32_t rc = s.open(device_id, N, d, G, chain, nullptr);
    if (rc) return diag_fail(s, rc);
    return diag_fail(s, demcz_rhat(&s.h, 1, G, rhat));
}

extern "C" int32_t demcz_accept_ratio_array(int32_t device_id, const double* log_obj, int64_t N, int64_t G, double* ratio)
{
    if (!log_obj || !ratio) return DEMCZ_ERR_INVALID_ARGUMENT;
    ScratchHandle s;
    int32_t rc = s.open(device_id, N, 1, G, nullptr, log_obj);
    if (rc) return diag_fail(s, rc);
    return diag_fail(s, demcz_accept_ratio(&s.h, 1, G, ratio));
}

extern "C" int32_t demcz_mean_cov_array(int32_t device_id, const double* chain, int64_t N, int32_t d, int64_t G, double* mean, double* cov)
{
    if (!chain || !mean || !cov) return DEMCZ_ERR_INVALID_ARGUMENT;
    ScratchHandle s;
    int32_t rc = s.open(device_id, N, d, G, chain, nullptr);
    if (rc) return diag_fail(s, rc);
    return diag_fail(s, demcz_mean_cov(&s.h, 1, G, mean, cov));
}

extern "C" int32_t demcz_get_info(const demcz_handle* h, int64_t* M, int64_t* launches_window, int32_t* lanes_per_chain)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (M) *M = h->M_app;
    if (launches_window) *launches_window = h->launches;
    if (lanes_per_chain) *lanes_per_chain = (h->split_kind == 4) ? DEMCZ_LAYOUT_SPLIT_WAVE : h->lanes;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_selftest_draws(int32_t device_id, uint64_t seed, uint64_t chain, uint64_t blk0, int32_t n,
                                        uint64_t* words, double* normals, double* logu)
{
    if (n < 1 || !words || !normals || !logu) return DEMCZ_ERR_INVALID_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, DEMCZ_ERR_NO_DEVICE, "demcz_selftest_draws: no HIP device visible");
    if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, DEMCZ_ERR_HIP, "hipSetDevice failed");
    uint64_t* dw = nullptr; double* dn = nullptr; double* dl = nullptr;
    int32_t rc = DEMCZ_OK;
    if (hipMalloc((void**)&dw, (size_t)n * 16) != hipSuccess || hipMalloc((void**)&dn, (size_t)n * 16) != hipSuccess ||
        hipMalloc((void**)&dl, (size_t)n * 8) != hipSuccess) {
        rc = fail(nullptr, DEMCZ_ERR_HIP, "demcz_selftest_draws: hipMalloc failed");
    } else {
        hipLaunchKernelGGL(selftest_draws_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, seed, chain, blk0, (int)n, dw, dn, dl);
        if (hipMemcpy(words, dw, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(normals, dn, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(logu, dl, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(nullptr, DEMCZ_ERR_HIP, "demcz_selftest_draws: kernel or copy failed");
    }
    if (dw) (void)hipFree(dw);
    if (dn) (void)hipFree(dn);
    if (dl) (void)hipFree(dl);
    return rc;
}

#ifdef DEMCZ_STAMPS
// Diagnostic build only: the stamps of the last split-layout launch (16 per workgroup, first `n_wg` workgroups).
extern "C" int32_t demcz_debug_read_stamps(demcz_handle* h, unsigned long long* out, int64_t n_wg)
{
    if (!h || !out || n_wg < 0 || n_wg > DEMCZ_STAMP_WGS || !h->d_stamps) return DEMCZ_ERR_INVALID_ARGUMENT;
    SYNCCHK(h, h->stream);
    HIPCHK(h, hipMemcpy(out, h->d_stamps, (size_t)n_wg * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return DEMCZ_OK;
}
#endif

static int32_t run_checked_body(demcz_handle* h, int64_t g_from, int64_t g_to, double gamma, const double* temperature,
                                int64_t every, double threshold, int64_t* g_stop, int32_t* n_checks,
                                double* rhat_max, int32_t n_max, double* rhat_last)
{
    const int d = h->cfg.d;
    int32_t checks = 0;
    int64_t g = g_from;
    if (g_stop) *g_stop = g_to;
    // threshold <= 0: nothing is decided on the statistic, so the checks are only enqueued between the slabs
    // (results land in pinned host memory) and read once at the end: the GPU never waits for the host.
    const bool monitor = !(threshold > 0.0);
    const int64_t max_checks = (g_to - g_from + 1) / every + 2;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    if (max_checks * d > h->pinned_cap) {
        { int32_t rcq = quiesce_all(h); if (rcq) return rcq; }
        SYNCCHK(h, h->stream);
        if (h->pinned_rhat) (void)host_free(h->pinned_rhat);
        h->pinned_rhat = nullptr; h->pinned_cap = 0;
        const int64_t cap = std::max<int64_t>(max_checks, 1024) * d;       // (no reallocation between calls of different length)
        HIPCHK(h, host_malloc((void**)&h->pinned_rhat, (size_t)cap * sizeof(double)));
        h->pinned_cap = cap;
    }
    double* pinned = h->pinned_rhat;
    auto max_of = [d](const double* r) {
        double mx = r[0];
        for (int p = 1; p < d; ++p) mx = (r[p] > mx || r[p] != r[p]) ? r[p] : mx;      // a NaN stays
        return mx;
    };
    int32_t rc = DEMCZ_OK;
    // With a threshold the decision needs the statistic on the host.  Where the run can be rolled back cheaply (one GPU,
    // immediate visibility, appends owned by the library, no temperature upload per slab) the next slab is enqueued
    // BEFORE waiting for it, so the GPU does not idle over the host's round trip; a stop then discards that slab:
    // X, log_obj and the archive size go back to what they were, the rows it appended become unwritten again.
    const bool speculate = !monitor && !h->comm && h->lag == 0 && !h->external_append && !temperature;
    const int64_t N = h->cfg.N;
    if (speculate) {
        if (!h->d_spec_X) HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->d_spec_X, (size_t)N * d * sizeof(double)));
        if (!h->d_spec_lp) HIPCHK(h, dev_malloc(h->cfg.device_id, (void**)&h->d_spec_lp, (size_t)N * sizeof(double)));
        if (!h->spec_ev) HIPCHK(h, hipEventCreateWithFlags(&h->spec_ev, hipEventDisableTiming));
    }
    bool ahead = false;                 // the slab starting at g has already been enqueued
    while (g <= g_to && rc == DEMCZ_OK) {
        const int64_t nxt = std::min(g_to, ((g - 1) / every + 1) * every);
        if (!ahead) rc = demcz_run(h, g, nxt, gamma, temperature ? temperature + (g - g_from) : nullptr);
        ahead = false;
        if (rc) break;
        if (nxt % every == 0 && nxt - every >= h->g0) {          // demcz.jl:39-41
            double* slot = pinned + (size_t)checks * d;
            // (the call's last slab: nothing follows it on the compute stream, so its check goes there -- no hop to the side
            //  stream, ~10 us of event latency, at the end of every call)
            rc = rhat_enqueue(h, nxt - every + 1, nxt, slot, /*side=*/monitor && nxt < g_to);
            if (rc) break;
            ++checks;
            if (!monitor) {
                int64_t M_before = 0;
                if (speculate && nxt < g_to) {
                    if (hipEventRecord(h->spec_ev, h->stream) != hipSuccess) { rc = fail(h, DEMCZ_ERR_HIP, "demcz_run_checked: event failed"); break; }
                    if (hipMemcpyAsync(h->d_spec_X, h->dX, (size_t)N * d * sizeof(double), hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                        hipMemcpyAsync(h->d_spec_lp, h->dlp, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, h->stream) != hipSuccess) {
                        rc = fail(h, DEMCZ_ERR_HIP, "demcz_run_checked: snapshot failed");
                        break;
                    }
                    M_before = h->M_app;
                    const int64_t nn = std::min(g_to, (nxt / every + 1) * every);
                    rc = demcz_run(h, nxt + 1, nn, gamma, nullptr);
                    if (rc) break;
                    ahead = true;
                    if ((rc = sync_event(h, h->spec_ev, "demcz_run_checked")) != DEMCZ_OK) break;
                } else if ((rc = sync_stream(h, h->stream, "demcz_run_checked")) != DEMCZ_OK) {
                    break;
                }
                if (max_of(slot) < threshold) {                  // demcz.jl:43
                    if (g_stop) *g_stop = nxt;
                    if (ahead) {        // discard the slab that ran ahead
                        if ((rc = sync_stream(h, h->stream, "demcz_run_checked")) != DEMCZ_OK) break;
                        rc = check_live_err(h);
                        if (rc) break;
                        if (hipMemcpyAsync(h->dX, h->d_spec_X, (size_t)N * d * sizeof(double), hipMemcpyDeviceToDevice, h->stream) != hipSuccess ||
                            hipMemcpyAsync(h->dlp, h->d_spec_lp, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, h->stream) != hipSuccess) {
                            rc = fail(h, DEMCZ_ERR_HIP, "demcz_run_checked: restore failed");
                            break;
                        }
                        const size_t rest = (size_t)(h->M_app - M_before) * (size_t)h->ZS;
                        if (rest > 0) {
                            hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, h->stream,
                                               reinterpret_cast<unsigned long long*>(h->dZ + (size_t)M_before * h->ZS), rest, LIVE_SENTINEL);
                            if (hipGetLastError() != hipSuccess) { rc = fail(h, DEMCZ_ERR_HIP, "demcz_run_checked: restore failed"); break; }
                        }
                        // its history slots read as never written again (zeros, demcz.jl:24)
                        const int64_t s_lo = nxt - h->g0, s_hi = std::min<int64_t>(h->g_done - h->g0, h->cfg.Gcap);
                        if (s_hi > s_lo &&
                            (hipMemsetAsync(h->dchain + (size_t)N * d * s_lo, 0, (size_t)N * d * (s_hi - s_lo) * sizeof(double), h->stream) != hipSuccess ||
                             hipMemsetAsync(h->dlogobj + (size_t)N * s_lo, 0, (size_t)N * (s_hi - s_lo) * sizeof(double), h->stream) != hipSuccess)) {
                            rc = fail(h, DEMCZ_ERR_HIP, "demcz_run_checked: restore failed");
                            break;
                        }
                        if (h->hs_on) { h->after_launch_ev = nullptr; rc = stream_history(h, nxt + 1, h->g_done); if (rc) break; }   // (the mirrors read zeros there too)
                        h->M_app = M_before;
                        h->M = M_before;
                        h->g_done = nxt;
                        if ((rc = rec_scrub(h)) != DEMCZ_OK) break;
                        if (!h->live_log.empty() && h->live_log.back().g_from == nxt + 1) h->live_log.pop_back();     // the discarded slab is not to be redone
                        while (!h->acc_log.empty() && h->acc_log.back().g_last > nxt) h->acc_log.pop_back();
                    }
                    break;
                }
            }
        }
        g = nxt + 1;
    }
    // the verdict on the call's row hand-offs (demcz_run_checked, below) needs the error word on the host: its copy goes behind
    // the last launch NOW, so that it travels while the last slab's statistic is still being made -- not as a blocking copy of
    // its own after everything else (one host round trip less at the end of every call: ~30 us)
    h->pinned_err_launches = -1;
    if (rc == DEMCZ_OK && !h->live_log.empty() && h->peer_mode != 2) {
        if (!h->pinned_err && host_malloc((void**)&h->pinned_err, 4 * sizeof(unsigned int)) != hipSuccess) { h->pinned_err = nullptr; (void)hipGetLastError(); }
        if (h->pinned_err && hipMemcpyAsync(h->pinned_err, h->d_live_err, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream) == hipSuccess)
            h->pinned_err_launches = h->launches;
    }
    if (rc == DEMCZ_OK) rc = sync_stream(h, h->stream, "demcz_run_checked");
    if (rc == DEMCZ_OK && h->diag_stream) rc = sync_stream(h, h->diag_stream, "demcz_run_checked");
    if (rc == DEMCZ_OK && h->prod_stream) rc = sync_stream(h, h->prod_stream, "demcz_run_checked");
    if (rc == DEMCZ_OK && h->comm_stream) rc = sync_stream(h, h->comm_stream, "demcz_run_checked");
    if (rc == DEMCZ_OK) {
        for (int32_t i = 0; i < checks; ++i)
            if (rhat_max && i < n_max) rhat_max[i] = max_of(pinned + (size_t)i * d);
        if (rhat_last && checks > 0) std::copy(pinned + (size_t)(checks - 1) * d, pinned + (size_t)checks * d, rhat_last);
        if (n_checks) *n_checks = checks;
    }
    return rc;
}

extern "C" int32_t demcz_run_checked(demcz_handle* h, int64_t g_from, int64_t g_to, double gamma, const double* temperature,
                                     int64_t every, double threshold, int64_t* g_stop, int32_t* n_checks,
                                     double* rhat_max, int32_t n_max, double* rhat_last)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    DEADCHK(h);
    if (every < 4 || g_from < 1 || g_to < g_from) return fail(h, DEMCZ_ERR_INVALID_ARGUMENT, "demcz_run_checked: need every >= 4 and 1 <= g_from <= g_to");
    if (!h->has_state) return fail(h, DEMCZ_ERR_STATE, "demcz_run_checked: call demcz_set_state first");
    if (h->peer_mode == 1 || h->peer_mode == 3) return fail(h, DEMCZ_ERR_STATE, "demcz_run_checked: replica groups and host-mediated peers are driven call by call (demcz_run, demcz_rhat_partial)");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    // everything before this call is verified first, so that a failed LIVE hand-off inside it rolls back to HERE
    int32_t rc = live_verify(h);
    if (rc) return rc;
    // (a handle waiting to go LIVE again does so at this call's first slab or -- in a redo -- at a later one, never at a point of
    //  a fresh call that is not its entry: the snapshot a failure goes back to must be the entry's)
    if (h->no_live && h->rearm_from >= g_from) h->rearm_from = g_from - 1;
    const bool pacing = (getenv("DEMCZ_NO_HOST_PACING") == nullptr);
    for (;;) {
        h->host_paced = pacing;
        h->in_checked = true;
        rc = run_checked_body(h, g_from, g_to, gamma, temperature, every, threshold, g_stop, n_checks, rhat_max, n_max, rhat_last);
        h->host_paced = false;
        h->in_checked = false;
        if (h->live_log.empty()) break;
        // the statistics and the stop decision above may rest on a slab whose row hand-off failed: look, and if so undo the whole
        // call and make it again -- one launch per K-window up to the slab that holds the generation whose row never came, LIVE
        // again behind it while the handle has re-arms left (live_rollback), every slab of the redo logged against THIS call's
        // entry (`replaying`).  A redo that fails too is redone the same way; without a re-arm left it cannot fail.
        // (a run that failed for another reason on THIS rank only must not leave the others waiting in the reduction below: the
        //  communicator's deadline covers it)
        if (rc != DEMCZ_OK && h->peer_mode == 2) break;
        bool failed = false;
        { int32_t rcf = live_failed(h, failed); if (rcf) { rc = rcf; break; } }
        if (!failed) { h->live_log.clear(); break; }
        std::vector<demcz_handle::RunCall> dropped;
        rc = live_rollback(h, dropped);
        if (rc) break;
        h->replaying = true;
    }
    h->replaying = false;
    return rc;
}

// ---- diagnostic entry points (not part of the contract of SURVEY.md 8(b); benchmarks and tests only) ----------
extern "C" int32_t demcz_set_live_spin_limit(demcz_handle* h, int32_t polls)
{
    if (!h || polls < 0) return DEMCZ_ERR_INVALID_ARGUMENT;
    h->live_spin_limit = (unsigned int)polls;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_set_live_rearms(demcz_handle* h, int32_t n)
{
    if (!h || n < 0) return DEMCZ_ERR_INVALID_ARGUMENT;
    h->live_rearms_left = n;
    if (h->peer_mode == 1 && h->group) {
        h->group->rearms_left = n;
        for (demcz_handle* m : h->group->members) m->live_rearms_left = n;
    }
    if (n == 0) h->rearm_from = -1;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_get_live_rearms(const demcz_handle* h, int32_t* rearms, int32_t* left)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (rearms) *rearms = h->live_rearms;
    if (left) *left = h->live_rearms_left;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_debug_set_live_fault(demcz_handle* h, int32_t polls, int64_t g_from)
{
    if (!h || polls < -1 || g_from < 0) return DEMCZ_ERR_INVALID_ARGUMENT;
    h->live_fault_polls = polls;
    h->live_fault_g = g_from;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_debug_kernel_counts(const demcz_handle* h, int64_t* counts)
{
    if (!h || !counts) return DEMCZ_ERR_INVALID_ARGUMENT;
    const bool wave = h->lanes == DEMCZ_LAYOUT_SPLIT && h->split_kind == 4;
    counts[0] = h->kernel_counts[0];
    counts[1] = wave ? h->launches - h->kernel_counts[0] : 0;
    counts[2] = wave ? 0 : h->launches;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_debug_kernel_name(const demcz_handle* h, char* buf, int32_t cap)
{
    if (!h || !buf || cap < 1) return DEMCZ_ERR_INVALID_ARGUMENT;
    const int d = h->cfg.d;
    const char* tg = h->cfg.target_kind == DEMCZ_TARGET_MVNORMAL ? "MVNORMAL" : h->cfg.target_kind == DEMCZ_TARGET_ISO_QUAD ? "ISO_QUAD"
                     : h->cfg.target_kind == DEMCZ_TARGET_LINREG_SSE ? "LINREG_SSE" : "HOST";
    const char* lv = h->last_live ? "true" : "false";
    const char* tm = h->last_temper ? "true" : "false";
    char tmp[160];
    const bool lr = h->cfg.target_kind == DEMCZ_TARGET_LINREG_SSE;
    if (h->lanes == DEMCZ_LAYOUT_SPLIT) {
        if (h->lr_spec) snprintf(tmp, sizeof tmp, "window_kernel_lr8s<%d, %s>", d, lv);
        else if (h->split_kind == 4 && d <= 5) snprintf(tmp, sizeof tmp, "%s<%s, %d, %s, %s>", (h->last_ps2 && h->last_dual) ? "window_kernel_ps2d" : h->last_ps2 ? "window_kernel_ps2" : "window_kernel_ps", tg, d, lv, tm);
        else if (h->split_kind == 4 && h->last_live && pw_matrix_form(h)) snprintf(tmp, sizeof tmp, "window_kernel_pw<%s, %d, %s, %s, true>", tg, d, lv, tm);
        else if (h->split_kind == 4 && h->last_pw_reg) snprintf(tmp, sizeof tmp, "window_kernel_pw<%s, %d, %s, %s, false, true>", tg, d, lv, tm);
        else if (h->split_kind == 4) snprintf(tmp, sizeof tmp, "window_kernel_pw<%s, %d, %s, %s>", tg, d, lv, tm);
        else if (h->split_kind == 3 && h->mlb_qb > 0 && h->split_lanes == 16) snprintf(tmp, sizeof tmp, "window_kernel_mlb<%s, %d, %d, true, %s, %d>", tg, d, h->split_lanes, lv, h->mlb_qb);
        else if (h->split_kind == 3) snprintf(tmp, sizeof tmp, "window_kernel_mlb<%s, %d, %d, true, %s>", tg, d, h->split_lanes, lv);
        else if (h->split_kind == 2 && lr) snprintf(tmp, sizeof tmp, "window_kernel_lr16<%d, true, %s>", d, lv);
        else if (h->split_kind == 2) snprintf(tmp, sizeof tmp, "window_kernel_ml<%s, %d, 16, true, %s>", tg, d, lv);
        else snprintf(tmp, sizeof tmp, "window_kernel_pc8<%s, %d, %s, %s>", tg, d, lv, tm);
    } else if (h->lanes > 1) {
        if (!h->full_block) snprintf(tmp, sizeof tmp, "window_kernel_mlb<%s, %d, %d>", tg, d, h->lanes);
        else if (lr && uses_lr16(h)) snprintf(tmp, sizeof tmp, "window_kernel_lr16<%d, false, false>", d);
        else if (lr && ml_coop(h)) snprintf(tmp, sizeof tmp, "window_kernel_ml<%s, %d, %d, false, false, true>", tg, d, h->lanes);
        else snprintf(tmp, sizeof tmp, "window_kernel_ml<%s, %d, %d>", tg, d, h->lanes);
    } else {
        snprintf(tmp, sizeof tmp, "window_kernel<%s, %d, %s>", tg, d, h->full_block ? "true" : "false");
    }
    snprintf(buf, (size_t)cap, "demcz::%s", tmp);
    return DEMCZ_OK;
}

extern "C" int32_t demcz_get_live_status(const demcz_handle* h, int32_t* live_enabled, int32_t* redos)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (live_enabled) *live_enabled = (h->lanes == DEMCZ_LAYOUT_SPLIT && !h->no_live && h->live_claimed) ? 1 : 0;
    if (redos) *redos = h->live_redos;
    return DEMCZ_OK;
}

// The scatter kernels of a sharded run -- append_gathered_kernel (synchronous exchange) and append_batch_kernel
// (batched exchange) -- applied to a slab the CALLER built in the layout an all-gather over R ranks delivers,
// [R][cnt][d][n_loc] with n_loc = this handle's N: lets a one-GPU box check the index arithmetic for R > 1.
extern "C" int32_t demcz_debug_append_slab(demcz_handle* h, const double* slab, int32_t R, int32_t cnt, int32_t batched)
{
    if (!h || !slab || R < 1 || cnt < 1 || (!batched && cnt != 1)) return DEMCZ_ERR_INVALID_ARGUMENT;
    if (!h->has_state) return fail(h, DEMCZ_ERR_STATE, "demcz_debug_append_slab: no state");
    if (h->lag != 0 || !h->pending.empty()) return fail(h, DEMCZ_ERR_STATE, "demcz_debug_append_slab: rows are pending");
    const int64_t N = h->cfg.N, rows = N * R * cnt;
    const int d = h->cfg.d;
    if (h->M_app + rows > h->cfg.Mcap) return fail(h, DEMCZ_ERR_CAPACITY, "demcz_debug_append_slab: Z capacity exceeded");
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    int32_t rc = ensure_scratch(h, rows * d);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_scratch, slab, (size_t)rows * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
    const int64_t tot = rows * d;
    if (batched)
        hipLaunchKernelGGL(append_batch_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->dZ, h->ZS,
                           h->M_app, (const double*)h->d_scratch, N, (int)R, (int)cnt, d);
    else
        hipLaunchKernelGGL(append_gathered_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, h->dZ, h->ZS,
                           h->M_app, (const double*)h->d_scratch, N, (int)R, d);
    HIPCHK(h, hipGetLastError());
    SYNCCHK(h, h->stream);
    h->M_app += rows;
    h->M = h->M_app;
    rec_invalidate(h);
    return DEMCZ_OK;
}

extern "C" int32_t demcz_set_kernel_timing(demcz_handle* h, int32_t enabled)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    h->timing = enabled != 0;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_get_kernel_time(demcz_handle* h, int64_t* launches, double* milliseconds)
{
    if (!h) return DEMCZ_ERR_INVALID_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    SYNCCHK(h, h->stream);
    double total = 0.0;
    h->after_launch_ev = nullptr;         // (may be one of the events destroyed below)
    h->series_start_ms.clear(); h->series_dur_ms.clear();
    for (auto& pr : h->timed) {
        float ms = 0.0f, st = 0.0f;
        HIPCHK(h, hipEventElapsedTime(&ms, pr.first, pr.second));
        if (&pr != &h->timed.front()) HIPCHK(h, hipEventElapsedTime(&st, h->timed.front().first, pr.first));   // (an event against itself faults in the runtime)
        h->series_start_ms.push_back(st); h->series_dur_ms.push_back(ms);
        total += ms;
    }
    for (auto& pr : h->timed) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (launches) *launches = h->timed_launches;
    if (milliseconds) *milliseconds = total;
    h->timed.clear();
    h->timed_launches = 0;
    return DEMCZ_OK;
}

extern "C" int32_t demcz_get_kernel_time_series(demcz_handle* h, int32_t cap, double* start_ms, double* duration_ms, int32_t* n)
{
    if (!h || cap < 0 || !n) return DEMCZ_ERR_INVALID_ARGUMENT;
    const int32_t m = (int32_t)std::min<size_t>(h->series_start_ms.size(), (size_t)cap);
    for (int32_t i = 0; i < m; ++i) {
        if (start_ms) start_ms[i] = h->series_start_ms[i];
        if (duration_ms) duration_ms[i] = h->series_dur_ms[i];
    }
    *n = (int32_t)h->series_start_ms.size();
    return DEMCZ_OK;
}
