# DEMCHip.jl -- `ccall` shim over libdemcz_hip.so (C ABI: include/demcz.h) with the call surface of
# chrished/DEMC.jl: demcopt / demcz_sample / demcz_anneal returning (mc::MC, Z::Matrix{Float64}).
#
# STATUS: written against include/demcz.h but NOT executed -- there is no Julia in the build image or
# on the GPU box (every ccall is checked against the header by tests/test_julia_shim.py).  The identical call sequence is what the Python host (demc.jl_amd/sampler.py) runs
# and what tests/ verify; keep the two in step.  Field order of DemczConfig must match demcz_config.
module DEMCHip

using LinearAlgebra

const libdemcz = get(ENV, "DEMCZ_LIB", "libdemcz_hip")

# ---- types of src/DEMC.jl:10-43 ---------------------------------------------------------------------
struct MC
    chain::Array{Float64,3}
    log_obj::Array{Float64,2}
    Xcurrent::Array{Float64,2}
    log_objcurrent::Array{Float64,1}
end

mutable struct DEMCopt
    N::Int; K::Int; Ngeneration::Int; Nblocks::Int; blockindex::Array; eps_scale::Array{Float64,1}; γ::Float64
    verbose::Bool; print_step::Int; T0::Float64; TN::Float64; autostop::Symbol; autostop_every::Int; autostop_Rhat
end

demcopt(Npar; N=4, K=10, Ngeneration=5000, Nblocks=1, blockindex=[1:Npar], eps_scale=1e-4 * ones(Npar), γ=2.38,
        verbose=true, print_step=100, T0=3, TN=1e-3, autostop=:Rhat, autostop_every=1000, autostop_Rhat=1.05) =
    DEMCopt(N, K, Ngeneration, Nblocks, blockindex, eps_scale, γ, verbose, print_step, T0, TN, autostop, autostop_every, autostop_Rhat)

# ---- device targets (the closures of the reference's tests) -------------------------------------------
abstract type DeviceTarget end
struct MvNormalTarget <: DeviceTarget       # logpdf(MvNormal(μ, Σ), x), test/example_normpdf.jl:13-16
    μ::Vector{Float64}; W::Matrix{Float64}; c0::Float64
end
function MvNormalTarget(μ, Σ)
    L = cholesky(Symmetric(Matrix{Float64}(Σ))).L
    MvNormalTarget(Vector{Float64}(μ), Matrix(inv(L)), -0.5 * (length(μ) * log(2π) + 2 * sum(log, diag(L))))
end
struct IsoQuadTarget <: DeviceTarget; μ::Vector{Float64}; end                       # -sum((x .- μ).^2)
struct LinRegSSETarget <: DeviceTarget; X::Matrix{Float64}; y::Vector{Float64}; end  # -0.5*sum((y .- X*b).^2)

# ---- C ABI ----------------------------------------------------------------------------------------------
struct DemczConfig
    N::Int64; chain_id0::Int64; d::Int32; K::Int32; Mcap::Int64; Gcap::Int64; Nblocks::Int32
    block_offsets::Ptr{Int32}; block_indices::Ptr{Int32}; eps_scale::Ptr{Float64}
    seed::UInt64; device_id::Int32; target_kind::Int32
    mu::Ptr{Float64}; W::Ptr{Float64}; c0::Float64; design::Ptr{Float64}; yobs::Ptr{Float64}; nobs::Int64
    stream::Ptr{Cvoid}; lanes_per_chain::Int32; reserved0::Int32
end

# lanes_per_chain beyond 0 (the library chooses) / 1 / 8 / 16: the producer-consumer split layouts (include/demcz.h)
const LAYOUT_SPLIT = Int32(100)
const LAYOUT_SPLIT_WAVE = Int32(164)

struct DemczError <: Exception; code::Int32; msg::String; end
lasterr(h) = unsafe_string(ccall((:demcz_last_error, libdemcz), Cstring, (Ptr{Cvoid},), h))
chk(rc, h=C_NULL) = rc == 0 ? nothing : throw(DemczError(rc, lasterr(h)))

# `logobj` of the reference's surface: one of the device targets above, or ANY Julia function x::Vector{Float64} -> Float64
# (demcz.jl:189 calls it once per block-step per chain; such a closure runs on the host, see run_closure! below)
const LogObj = Union{DeviceTarget,Function}

function create(t::LogObj, N, d, K, Mcap, Gcap, blockindex, eps_scale, seed; device_id=0, chain_id0=0)
    offs = Int32[0; cumsum(length.(blockindex))]
    idx = Int32[i - 1 for b in blockindex for i in b]                       # 1-based -> 0-based
    eps = Vector{Float64}(eps_scale)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve offs idx eps t begin
        kind, mu, W, c0, design, y, nobs =
            t isa MvNormalTarget ? (Int32(0), pointer(t.μ), pointer(t.W), t.c0, Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), 0) :
            t isa IsoQuadTarget ? (Int32(1), pointer(t.μ), Ptr{Float64}(C_NULL), 0.0, Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), 0) :
            t isa LinRegSSETarget ? (Int32(2), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), 0.0, pointer(t.X), pointer(t.y), size(t.X, 1)) :
            (Int32(3), Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), 0.0, Ptr{Float64}(C_NULL), Ptr{Float64}(C_NULL), 0)   # DEMCZ_TARGET_HOST_CALLBACK
        cfg = DemczConfig(N, chain_id0, d, K, Mcap, Gcap, length(blockindex), pointer(offs), pointer(idx), pointer(eps),
                          UInt64(seed), device_id, kind, mu, W, c0, design, y, nobs, C_NULL, 0, 0)
        chk(ccall((:demcz_create, libdemcz), Int32, (Ref{Ptr{Cvoid}}, Ref{DemczConfig}), h, cfg))   # config is copied
    end
    h[]
end
destroy(h) = ccall((:demcz_destroy, libdemcz), Int32, (Ptr{Cvoid},), h)

set_state(h, X::Matrix{Float64}, logp, Z::Matrix{Float64}) =
    chk(ccall((:demcz_set_state, libdemcz), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Int64),
              h, X, logp === nothing ? C_NULL : logp, Z, size(Z, 1), size(Z, 1)), h)
run!(h, g_from, g_to, γ, temperature=nothing) =
    chk(ccall((:demcz_run, libdemcz), Int32, (Ptr{Cvoid}, Int64, Int64, Float64, Ptr{Float64}),
              h, g_from, g_to, γ, temperature === nothing ? C_NULL : temperature), h)
set_rng_offset(h, g) = chk(ccall((:demcz_set_rng_offset, libdemcz), Int32, (Ptr{Cvoid}, Int64), h, g), h)
# the loop demcz.jl:30-55 (generations + the R-hat test every `every`) as ONE call; returns the generation it stopped at
function run_checked!(h, g_from, g_to, γ, every, threshold)
    g_stop = Ref{Int64}(0); n = Ref{Int32}(0)
    chk(ccall((:demcz_run_checked, libdemcz), Int32,
              (Ptr{Cvoid}, Int64, Int64, Float64, Ptr{Float64}, Int64, Float64, Ref{Int64}, Ref{Int32}, Ptr{Float64}, Int32, Ptr{Float64}),
              h, g_from, g_to, γ, C_NULL, every, threshold, g_stop, n, C_NULL, 0, C_NULL), h)
    g_stop[]
end
function rhat(h, g_from, g_to, d)
    r = zeros(d)
    chk(ccall((:demcz_rhat, libdemcz), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}), h, g_from, g_to, r), h); r
end
function changed(h, g_from, g_to)
    c = zeros(Int64, g_to - g_from + 1)
    chk(ccall((:demcz_get_changed, libdemcz), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}), h, g_from, g_to, c), h); c
end
# sum(changed(h, g_from, g_to)) from the window kernels' ballot counters (no pass over the history)
function changed_total(h, g_from, g_to)
    t = Ref{Int64}(0)
    chk(ccall((:demcz_get_changed_total, libdemcz), Int32, (Ptr{Cvoid}, Int64, Int64, Ref{Int64}, Ptr{Int32}), h, g_from, g_to, t, C_NULL), h); t[]
end
function accept_ratio(h, g_from, g_to, N)
    a = zeros(N)
    chk(ccall((:demcz_accept_ratio, libdemcz), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}), h, g_from, g_to, a), h); a
end
function history(h, N, d, g_from, g_to)
    G = g_to - g_from + 1
    chain = Array{Float64,3}(undef, N, d, G); log_obj = Matrix{Float64}(undef, N, G)
    chk(ccall((:demcz_get_history, libdemcz), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}), h, g_from, g_to, chain, log_obj), h)
    chain, log_obj
end
# Streamed history (demcz_history_stream): the library copies every slab to pinned host mirrors while the next slab computes;
# take_history wraps the mirrors as Julia arrays WITHOUT copying (unsafe_wrap), detaches them from the handle and hands them back
# to the library's pool when the arrays are garbage-collected: each array has a finalizer of its own that releases ITS mirror only,
# so a caller may keep `mc.log_obj` and drop `mc.chain` (or the other way round) without either reading freed memory.
history_stream(h, on=true) = chk(ccall((:demcz_history_stream, libdemcz), Int32, (Ptr{Cvoid}, Int32), h, on ? 1 : 0), h)
function take_history(h, N, d, g_from, g_to)
    G = g_to - g_from + 1
    pc = Ref{Ptr{Float64}}(C_NULL); pl = Ref{Ptr{Float64}}(C_NULL)
    chk(ccall((:demcz_get_history_view, libdemcz), Int32, (Ptr{Cvoid}, Int64, Int64, Ref{Ptr{Float64}}, Ref{Ptr{Float64}}), h, g_from, g_to, pc, pl), h)
    bc = Ref{Ptr{Cvoid}}(C_NULL); bl = Ref{Ptr{Cvoid}}(C_NULL)
    chk(ccall((:demcz_detach_history, libdemcz), Int32, (Ptr{Cvoid}, Ref{Ptr{Cvoid}}, Ref{Ptr{Cvoid}}), h, bc, bl), h)
    chain = unsafe_wrap(Array, pc[], (N, d, G); own=false)
    log_obj = unsafe_wrap(Array, pl[], (N, G); own=false)
    let b = bc[]
        finalizer(_ -> ccall((:demcz_release_host_buffer, libdemcz), Int32, (Ptr{Cvoid},), b), chain)
    end
    let b = bl[]
        finalizer(_ -> ccall((:demcz_release_host_buffer, libdemcz), Int32, (Ptr{Cvoid},), b), log_obj)
    end
    chain, log_obj
end
# gives the buffers the library keeps for the next handle back to the runtime (device bytes, pinned bytes)
function pool_trim()
    a = Ref{Int64}(0); b = Ref{Int64}(0)
    ccall((:demcz_pool_trim, libdemcz), Int32, (Ref{Int64}, Ref{Int64}), a, b)
    a[], b[]
end
function state(h, N, d)
    M = Ref{Int64}(0); X = Matrix{Float64}(undef, N, d); lp = Vector{Float64}(undef, N)
    chk(ccall((:demcz_get_state, libdemcz), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{Int64}), h, X, lp, C_NULL, 0, M), h)
    Z = Matrix{Float64}(undef, M[], d)
    chk(ccall((:demcz_get_state, libdemcz), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int64}), h, C_NULL, C_NULL, Z, M[], C_NULL), h)
    X, lp, Z
end

# ---- host-closure mode: update_demcz_chain_block cut at the closure call demcz.jl:189 ----------------------
# demcz_propose draws (i1, i2, normals, log u) for block `ib` of generation `g` and returns the N proposals; the
# closure is evaluated here, on the host; demcz_accept_commit applies demcz.jl:190-194 (tempered:
# demcz_anneal.jl:165,172-178); demcz_end_generation does runchain!'s bookkeeping demcz.jl:84-91.
# `ib` is 1-based like the reference's loop variable (demcz.jl:168); the C ABI is 0-based.
propose!(h, Xprop::Matrix{Float64}, g, ib, γ) =
    chk(ccall((:demcz_propose, libdemcz), Int32, (Ptr{Cvoid}, Int64, Int32, Float64, Ptr{Float64}), h, g, ib - 1, γ, Xprop), h)
accept_commit!(h, lp::Vector{Float64}) =
    chk(ccall((:demcz_accept_commit, libdemcz), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h, lp, C_NULL), h)
accept_commit!(h, lp::Vector{Float64}, temperature::Real) =
    chk(ccall((:demcz_accept_commit, libdemcz), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}), h, lp, Float64(temperature)), h)
end_generation!(h, g) = chk(ccall((:demcz_end_generation, libdemcz), Int32, (Ptr{Cvoid}, Int64), h, g), h)
# Round 5: the same round trip without its copies.  The library owns pinned, device-mapped host buffers (Xprop: N x d, logp: N);
# the propose kernel writes the proposals INTO Xprop and raises a flag the call spins on, accept_commit only enqueues its kernel,
# which reads logp through the mapped pointer (include/demcz.h, demcz_closure_buffers).  The arrays are views of library memory:
# valid until destroy(h); do not write logp between accept_commit!(h) and the return of the next propose!(h, g, ib, γ).
function closure_buffers(h, N, d)
    px = Ref{Ptr{Float64}}(C_NULL); pl = Ref{Ptr{Float64}}(C_NULL)
    chk(ccall((:demcz_closure_buffers, libdemcz), Int32, (Ptr{Cvoid}, Ref{Ptr{Float64}}, Ref{Ptr{Float64}}), h, px, pl), h)
    unsafe_wrap(Array, px[], (Int(N), Int(d))), unsafe_wrap(Array, pl[], Int(N))
end
propose!(h, g, ib, γ) =                                                                  # proposals land in closure_buffers' Xprop
    chk(ccall((:demcz_propose, libdemcz), Int32, (Ptr{Cvoid}, Int64, Int32, Float64, Ptr{Float64}), h, g, ib - 1, γ, C_NULL), h)
accept_commit!(h) =                                                                      # log-densities read from closure_buffers' logp
    chk(ccall((:demcz_accept_commit, libdemcz), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h, C_NULL, C_NULL), h)
accept_commit!(h, temperature::Real) =
    chk(ccall((:demcz_accept_commit, libdemcz), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}), h, C_NULL, Float64(temperature)), h)

function run_closure!(h, logobj::Function, N, d, Nblocks, g_from, g_to, γ, temperature=nothing)
    Xp, lp = closure_buffers(h, N, d)                                                  # pinned, device-mapped (no copies, no syncs)
    for g in g_from:g_to                                                               # demcz.jl:30
        for ib in 1:Nblocks                                                            # update_blocks, demcz.jl:168
            propose!(h, g, ib, γ)                                                      # demcz.jl:176-188 for all N chains -> Xp
            for ic in 1:N
                lp[ic] = logobj(Xp[ic, :])                                             # demcz.jl:189
            end
            temperature === nothing ? accept_commit!(h) : accept_commit!(h, temperature[g-g_from+1])
        end
        end_generation!(h, g)                                                          # demcz.jl:84-91
    end
end

# generations g_from..g_to: one library call for a device target, the propose / evaluate / commit loop for a closure
advance!(h, t::DeviceTarget, N, d, Nblocks, g_from, g_to, γ, temperature=nothing) = run!(h, g_from, g_to, γ, temperature)
advance!(h, f::Function, N, d, Nblocks, g_from, g_to, γ, temperature=nothing) = run_closure!(h, f, N, d, Nblocks, g_from, g_to, γ, temperature)

# start state of a run, demcz.jl:13-22 (chains start at the last N rows of Zmat -- the documented intent of :15)
function start_state(t::LogObj, Zmat, N, prevrun)
    if prevrun === nothing
        X = Matrix{Float64}(Zmat[end-N+1:end, :])
        lp = t isa Function ? Float64[t(X[i, :]) for i in 1:N] : nothing               # demcz.jl:17 (device targets: on the device)
        return X, lp, 0
    end
    Matrix{Float64}(prevrun.chain[:, :, end]), Vector{Float64}(prevrun.log_objcurrent[:, end]), size(prevrun.chain, 3)   # :20-21
end

# ---- multi-GPU: one Julia process per GPU (Distributed.jl workers, MPI.jl ranks ...), chains split in rank order, Z replicated.
# The role of src/demcz.jl:101-165 (demcz_sample_par: one chain per worker process around a SharedArray Z): the library
# all-gathers the K-boundary rows over RCCL and all-reduces the R-hat moments itself; the host only carries the 128-byte id.
function comm_unique_id()
    id = zeros(UInt8, 128)
    chk(ccall((:demcz_comm_unique_id, libdemcz), Int32, (Ptr{Cvoid},), id)); id
end
comm_init(h, id::Vector{UInt8}, nranks, rank) =
    chk(ccall((:demcz_comm_init, libdemcz), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32), h, id, nranks, rank), h)
# E = 0: rows appended after generation jK are drawn from generation jK+1 on (a synchronous all-gather every K generations);
# E >= 1: batches of E boundaries travel in one all-gather on a side stream and become visible E windows after the batch closes
set_append_lag(h, E) = chk(ccall((:demcz_set_append_lag, libdemcz), Int32, (Ptr{Cvoid}, Int32), h, E), h)
# deadline of every wait behind a collective; past it the communicators are aborted and calls throw DemczError(6 = DEMCZ_ERR_COMM)
set_comm_timeout(h, ms) = chk(ccall((:demcz_set_comm_timeout, libdemcz), Int32, (Ptr{Cvoid}, Int64), h, ms), h)
synchronize(h) = chk(ccall((:demcz_synchronize, libdemcz), Int32, (Ptr{Cvoid},), h), h)
# (LIVE launches in use, times an in-launch row hand-off timed out and was redone with one launch per K-window)
# Rows handed over inside the launches (include/demcz.h): mode 0 = exchange through RCCL (or unsharded), 1 = replica group of this
# process, 2 = IPC peers set up by comm_init, 3 = IPC peers set up by the host (peer_export / peer_attach)
function peer_status(h)
    m = Ref{Int32}(0); n = Ref{Int32}(0)
    chk(ccall((:demcz_get_peer_status, libdemcz), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), h, m, n), h)
    Int(m[]), Int(n[])
end
# hosts that shard without the library's communicator carry the 64-byte IPC handles themselves (Distributed.jl, MPI ...)
function peer_export(h, nranks, rank)
    buf = zeros(UInt8, 64)
    chk(ccall((:demcz_peer_export, libdemcz), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}), h, nranks, rank, buf), h)
    buf
end
peer_attach(h, handles::Vector{Vector{UInt8}}) = (blob = reduce(vcat, handles);
    chk(ccall((:demcz_peer_attach, libdemcz), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), h, blob), h))
function live_status(h)
    on = Ref{Int32}(0); redos = Ref{Int32}(0)
    chk(ccall((:demcz_get_live_status, libdemcz), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), h, on, redos), h)
    on[] != 0, Int(redos[])
end
# a timed-out hand-off is redone and the handle goes LIVE again behind the failed boundary, at most `n` times (default 3)
set_live_rearms(h, n) = chk(ccall((:demcz_set_live_rearms, libdemcz), Int32, (Ptr{Cvoid}, Int32), h, n), h)
function live_rearms(h)
    done = Ref{Int32}(0); left = Ref{Int32}(0)
    chk(ccall((:demcz_get_live_rearms, libdemcz), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Int32}), h, done, left), h)
    Int(done[]), Int(left[])
end
# comm_init's first-contact check: (ok, wait_us) -- ok = 1 every rank saw every peer's token, 0 failed somewhere, -1 not made
function peer_ping(h)
    ok = Ref{Int32}(0); us = Ref{Float64}(0.0)
    chk(ccall((:demcz_get_peer_ping, libdemcz), Int32, (Ptr{Cvoid}, Ref{Int32}, Ref{Float64}), h, ok, us), h)
    Int(ok[]), us[]
end
# hosts that attached the peers themselves (peer_attach): barrier, peer_detach on every rank, barrier, then destroy
peer_detach(h) = chk(ccall((:demcz_peer_detach, libdemcz), Int32, (Ptr{Cvoid},), h), h)
warn_live_redos(h) = (r = live_status(h)[2]; r > 0 && @warn "DEMCz: $r in-launch row hand-off(s) timed out and were redone with one launch per K-window (results unchanged; is another process using this GPU?)"; nothing)

"""
    demcz_sample_par(t, Zmat, opts; sync_every=1000, prevrun=nothing, rank, nranks, unique_id, device_id=rank, append_lag=0, seed=0)

The sharded counterpart of `demcz_sample_par(logobj, Zmat, opts; sync_every, prevrun)` (src/demcz.jl:101-103): every rank
(one process per GPU) makes this call with the same arguments; rank `r` runs chains `r*N/nranks+1 : (r+1)*N/nranks` of the
`opts.N` chains against its own replica of Z.  `unique_id` = `comm_unique_id()` of rank 0, broadcast by the caller's transport
(`Distributed.remotecall_fetch`, `MPI.Bcast!` ...).  The R-hat test runs every `sync_every` generations over ALL chains, as the
reference's does per slab (demcz.jl:141-156); every rank takes the same stop decision.  Returns `(mc, Z)` with `mc` holding
THIS rank's chains and `Z` the full archive (identical on all ranks).  `prevrun` = this rank's shard of an earlier result.
"""
function demcz_sample_par(t::DeviceTarget, Zmat, opts::DEMCopt; sync_every=1000, prevrun=nothing, rank::Integer, nranks::Integer,
                          unique_id::Vector{UInt8}, device_id=rank, append_lag=0, seed=0, comm_timeout_ms=60000)
    N, K, G = opts.N, opts.K, opts.Ngeneration
    N % nranks == 0 || error("opts.N must be divisible by the number of ranks")
    nloc = N ÷ nranks
    nrowZ, d = size(Zmat)
    Mcap = nrowZ + Int(ceil(N * G / K))                                                # demcz.jl:109
    c0 = rank * nloc
    if prevrun === nothing
        X = Matrix{Float64}(Zmat[end-N+1+c0:end-N+c0+nloc, :]); lp = nothing; drawn = 0   # demcz.jl:113: the last N rows, this rank's part
    else
        X = Matrix{Float64}(prevrun.chain[:, :, end]); lp = Vector{Float64}(prevrun.log_objcurrent[:, end]); drawn = size(prevrun.chain, 3)
    end
    h = create(t, nloc, d, K, Mcap, G, opts.blockindex, opts.eps_scale, seed; device_id=device_id, chain_id0=c0)
    try
        comm_init(h, unique_id, nranks, rank)
        set_comm_timeout(h, comm_timeout_ms)
        append_lag == 0 || set_append_lag(h, append_lag)
        set_state(h, X, lp, Matrix{Float64}(Zmat))
        drawn == 0 || set_rng_offset(h, drawn)
        ig = opts.autostop == :Rhat ? run_checked!(h, 1, G, opts.γ, sync_every, opts.autostop_Rhat) :      # demcz.jl:129-156
                                      (run!(h, 1, G, opts.γ); synchronize(h); G)
        chain, log_obj = history(h, nloc, d, 1, ig)
        Xc, lpc, Z = state(h, nloc, d)
        mc = prevrun === nothing ? MC(chain, log_obj, Xc, lpc) :
             MC(cat(prevrun.chain, chain, dims=3), cat(prevrun.log_obj, log_obj, dims=2), Xc, lpc)
        return mc, Z
    finally
        destroy(h)
    end
end

# ---- diagnostics of src/utils.jl with the reference's names, reduced on the device ----------------------------------
# (host arrays in, as the reference's examples pass them: test/example_normpdf.jl:35-47; the arrays are uploaded once and
#  reduced by the kernels the autostop uses)
function Rhat_gelman(chain::Array{Float64,3}, Npop=size(chain, 1), Ngeneration=size(chain, 3), Npar=size(chain, 2); device_id=0)   # utils.jl:2-20
    r = zeros(Npar)
    c = Ngeneration == size(chain, 3) ? chain : chain[:, :, 1:Ngeneration]
    chk(ccall((:demcz_rhat_array, libdemcz), Int32, (Int32, Ptr{Float64}, Int64, Int32, Int64, Ptr{Float64}), device_id, c, Npop, Npar, Ngeneration, r)); r
end
function flatten_chain(chain, Npop=size(chain, 1), Ngeneration=size(chain, 3), Npar=size(chain, 2))          # utils.jl:22-32
    reshape(permutedims(chain[:, :, 1:Ngeneration], (2, 1, 3)), Npar, Npop * Ngeneration)                    # column ig, ic -> (ig-1)*Npop + ic
end
function accept_ratio(log_obj::Matrix{Float64}; device_id=0)                                                 # utils.jl:61
    a = zeros(size(log_obj, 1))
    chk(ccall((:demcz_accept_ratio_array, libdemcz), Int32, (Int32, Ptr{Float64}, Int64, Int64, Ptr{Float64}), device_id, log_obj, size(log_obj, 1), size(log_obj, 2), a)); a
end
function mean_cov_chain(chain::Array{Float64,3}, Npop=size(chain, 1), Ngeneration=size(chain, 3), Npar=size(chain, 2); device_id=0)   # utils.jl:96-111
    @assert size(chain) == (Npop, Npar, Ngeneration)
    b = zeros(Npar); cov = zeros(Npar, Npar)
    chk(ccall((:demcz_mean_cov_array, libdemcz), Int32, (Int32, Ptr{Float64}, Int64, Int32, Int64, Ptr{Float64}, Ptr{Float64}), device_id, chain, Npop, Npar, Ngeneration, b, cov))
    b, cov
end
function convergence_check(chain::Array{Float64,3}, log_obj::Matrix{Float64}, figure_path=nothing; verbose=true, parnames=[], device_id=0)   # utils.jl:34-94
    Npop, Npar, Ngeneration = size(chain)
    @assert size(log_obj) == (Npop, Ngeneration) "log_obj must be Npop x Ngeneration"                        # utils.jl:40-46
    ar = accept_ratio(log_obj; device_id=device_id)
    Rhat = Rhat_gelman(chain; device_id=device_id)
    if verbose
        println("Summary Checks\n\nAcceptance Ratio of each chain:"); println(ar); println("\nRhat Gelman: $Rhat\n")
    end
    ar, Rhat                                                                                                 # (the plotting part is commented out in the reference)
end
# on-device forms over a live handle's history (no N x d x G download): mean / covariance of generations g_from..g_to
function mean_cov(h, g_from, g_to, d)
    b = zeros(d); cov = zeros(d, d)
    chk(ccall((:demcz_mean_cov, libdemcz), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}), h, g_from, g_to, b, cov), h); b, cov
end

# ---- checkpoint: what a resumed run needs (the reference resumes in memory only, demcz.jl:18-22) ----------------------------
struct Checkpoint
    prevrun::MC                 # last generation only: chain[:, :, end:end], log_obj[:, end:end], Xcurrent, log_objcurrent
    Z::Matrix{Float64}
    generations_done::Int       # how far the chains' random streams have advanced (pass to set_rng_offset)
    seed::UInt64
end
checkpoint(mc::MC, Z, generations_done, seed) =
    Checkpoint(MC(mc.chain[:, :, end:end], mc.log_obj[:, end:end], copy(mc.Xcurrent), copy(mc.log_objcurrent)), Matrix{Float64}(Z), generations_done, UInt64(seed))
function save_checkpoint(path, ck::Checkpoint)                       # a flat little-endian file: 5 Int64 header words, then the arrays
    N, d = size(ck.prevrun.Xcurrent); M = size(ck.Z, 1)
    open(path, "w") do io
        write(io, Int64[N, d, M, ck.generations_done, reinterpret(Int64, ck.seed)])
        write(io, ck.prevrun.chain); write(io, ck.prevrun.log_obj); write(io, ck.prevrun.Xcurrent); write(io, ck.prevrun.log_objcurrent); write(io, ck.Z)
    end
end
function load_checkpoint(path)
    open(path, "r") do io
        hdr = Vector{Int64}(undef, 5); read!(io, hdr); N, d, M = hdr[1], hdr[2], hdr[3]
        chain = Array{Float64,3}(undef, N, d, 1); lo = Matrix{Float64}(undef, N, 1); X = Matrix{Float64}(undef, N, d)
        lp = Vector{Float64}(undef, N); Z = Matrix{Float64}(undef, M, d)
        read!(io, chain); read!(io, lo); read!(io, X); read!(io, lp); read!(io, Z)
        Checkpoint(MC(chain, lo, X, lp), Z, hdr[4], reinterpret(UInt64, hdr[5]))
    end
end
# resume: demcz_sample(t, ck.Z, opts; prevrun=ck.prevrun, seed=ck.seed, drawn=ck.generations_done)

# ---- drivers: src/demcz.jl:1-63 and src/demcz_anneal.jl:14-65 with the device below runchain! -------------
demcz_sample(t::LogObj, Zmat, opts::DEMCopt; prevrun=nothing, seed=0, drawn=nothing) =
    demcz_sample(t, Zmat, opts.N, opts.K, opts.Ngeneration, opts.Nblocks, opts.blockindex, opts.eps_scale, opts.γ;
                 prevrun=prevrun, verbose=opts.verbose, print_step=opts.print_step, autostop=opts.autostop,
                 autostop_Rhat=opts.autostop_Rhat, autostop_every=opts.autostop_every, seed=seed, drawn=drawn)

function demcz_sample(t::LogObj, Zmat, N=4, K=10, Ngeneration=5000, Nblocks=1, blockindex=[1:size(Zmat, 2)],
                      eps_scale=1e-4 * ones(size(Zmat, 2)), γ=2.38; prevrun=nothing, verbose=true, print_step=100,
                      autostop=:no, autostop_Rhat=1.01, autostop_every=1000, seed=0, drawn=nothing)
    nrowZ, d = size(Zmat)
    Mcap = nrowZ + Int(ceil(N * Ngeneration / K))                                     # demcz.jl:11
    X, lp, drawn0 = start_state(t, Zmat, N, prevrun)
    drawn = drawn === nothing ? drawn0 : drawn                                         # (a checkpoint's prevrun keeps one generation: its stream position is given)
    h = create(t, N, d, K, Mcap, Ngeneration, blockindex, eps_scale, seed)
    try
        set_state(h, X, lp, Matrix{Float64}(Zmat))
        drawn == 0 || set_rng_offset(h, drawn)                                         # the chains' streams continue
        streamed = t isa DeviceTarget
        streamed && history_stream(h)                                                  # slabs leave for the host while the next ones run
        ig = 0
        if autostop == :Rhat && t isa DeviceTarget                                     # demcz.jl:30-53 in one call
            ig = run_checked!(h, 1, Ngeneration, γ, autostop_every, autostop_Rhat)
        elseif autostop == :Rhat                                                       # closure: the same loop on the host
            while ig < Ngeneration
                nxt = min(Ngeneration, (ig ÷ autostop_every + 1) * autostop_every)
                advance!(h, t, N, d, Nblocks, ig + 1, nxt, γ); ig = nxt
                ig % autostop_every == 0 && maximum(rhat(h, ig - autostop_every + 1, ig, d)) < autostop_Rhat && break   # demcz.jl:39-43
            end
        else
            advance!(h, t, N, d, Nblocks, 1, Ngeneration, γ); ig = Ngeneration
        end
        if autostop == :Rhat && ig % autostop_every == 0 && maximum(rhat(h, ig - autostop_every + 1, ig, d)) < autostop_Rhat
            sum(accept_ratio(h, ig - autostop_every + 1, ig, N)) / N < 0.1 && println("Warning: accept ratio below 10% on average")   # demcz.jl:42-46
        end
        warn_live_redos(h)
        Xc, lpc, Z = state(h, N, d)                                                    # Z[1:M,:], demcz.jl:51
        chain, log_obj = streamed ? take_history(h, N, d, 1, ig) : history(h, N, d, 1, ig)   # demcz.jl:47 (no copy when streamed)
        mc = prevrun === nothing ? MC(chain, log_obj, Xc, lpc) :
             MC(cat(prevrun.chain, chain, dims=3), cat(prevrun.log_obj, log_obj, dims=2), Xc, lpc)   # demcz.jl:58-59
        return mc, Z
    finally
        destroy(h)
    end
end

tempbaseline(ig, Ng, T0, TN) = T0 * (TN / T0)^(ig / Ng)                                 # demcz_anneal.jl:1-3

demcz_anneal(t::LogObj, Zmat, opts::DEMCopt; prevrun=nothing, temperaturefun::Function=tempbaseline,
             adaptγ=Dict("adapt" => true, "minγ" => 0.1, "maxγ" => 4.0, "adapt_every" => 500), seed=0) =   # demcz_anneal.jl:14-16
    demcz_anneal(t, Zmat, opts.N, opts.K, opts.Ngeneration, opts.Nblocks, opts.blockindex, opts.eps_scale, opts.γ;
                 prevrun=prevrun, verbose=opts.verbose, print_step=opts.print_step, temperaturefun=temperaturefun,
                 T0=opts.T0, TN=opts.TN, adaptγ=adaptγ, seed=seed)

function demcz_anneal(t::LogObj, Zmat, N=4, K=10, Ngeneration=5000, Nblocks=1, blockindex=[1:size(Zmat, 2)],
                      eps_scale=1e-4 * ones(size(Zmat, 2)), γ=2.38; prevrun=nothing, verbose=true, print_step=100,
                      temperaturefun::Function=tempbaseline, T0=3, TN=0.0,
                      adaptγ=Dict("adapt" => true, "minγ" => 0.1, "maxγ" => 4.0, "adapt_every" => 500), seed=0)
    nrowZ, d = size(Zmat)
    Mcap = nrowZ + Int(ceil(N * Ngeneration / K))
    X, lp, drawn = start_state(t, Zmat, N, prevrun)
    h = create(t, N, d, K, Mcap, Ngeneration, blockindex, eps_scale, seed)
    try
        set_state(h, X, lp, Matrix{Float64}(Zmat))
        drawn == 0 || set_rng_offset(h, drawn)
        ae = adaptγ["adapt_every"]; ig = 0
        while ig < Ngeneration                                                         # demcz_anneal.jl:39
            nxt = adaptγ["adapt"] ? min(Ngeneration, (ig ÷ ae + 1) * ae) : Ngeneration
            temps = Float64[temperaturefun(g, Ngeneration, T0, TN) for g in ig+1:nxt]   # demcz_anneal.jl:69
            advance!(h, t, N, d, Nblocks, ig + 1, nxt, γ, temps); ig = nxt
            if adaptγ["adapt"] && ig % ae == 0                                         # demcz_anneal.jl:48-57
                accept = (ae > 1 ? changed_total(h, ig - ae + 2, ig) : 0) / (N * ae)
                if accept < 0.1; γ = max(adaptγ["minγ"], γ * 0.5) elseif accept > 0.5; γ = min(adaptγ["maxγ"], γ * 1.5) end
            end
        end
        chain, log_obj = history(h, N, d, 1, Ngeneration)
        Xc, lpc, Z = state(h, N, d)
        mc = prevrun === nothing ? MC(chain, log_obj, Xc, lpc) :
             MC(cat(prevrun.chain, chain, dims=3), cat(prevrun.log_obj, log_obj, dims=2), Xc, lpc)
        return mc, Z
    finally
        destroy(h)
    end
end

end # module
