#!/usr/bin/env python3
"""bench.py -- chain-updates/s of the DEMCz hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is what the boundary is crossed per: ONE AUTOSTOP SLAB of the reference's driver loop
(src/demcz.jl:30-55 with the `demcopt` defaults, DEMC.jl:41) = `autostop_every` = 1000 generations of
all chains -- every chain proposes, evaluates its log-density and takes its Metropolis decision
1000 times; the full history is written, Z is appended to every K = 10 generations -- followed by the
split-R-hat check of that slab (Rhat_gelman, utils.jl:2-20, demcz.jl:41).  `--steps 20` therefore
times 20 000 generations and 20 R-hat checks; `value` stays N x generations / second.  At least one
untimed slab of exactly the timed shape always runs first.  Workload at N=1: BASELINE config C2
(MvNormal d=5, correlated Sigma, N=1024 chains, K=10).  At N>1 each GPU holds 1024 more chains
(weak scaling), Z is replicated and the K-boundary rows are all-gathered over RCCL.

Prints ONE JSON line (rank 0).  `value` = N_total * generations / seconds with inputs resident in HBM.
`roofline` prices the window kernel (the dominant kernel) against HBM from HIP events recorded by the
library on the kernel's own stream; `cpu_baseline` is the CPU oracle ("port": this repo's C
restatement of src/demcz.jl, NOT the Julia package) timed on one host core on the same workload.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np


def algorithmic_bytes_per_update(d, K):
    """two Z rows gathered (16 d) + history row (8 d) + log_obj (8) + the
    amortised Z append (8 d / K).  State and RNG stay in registers across a window."""
    return 8.0 * (3 * d + 1 + d / K)


def measured_traffic(n_loc, d, K, lanes, gens_per_launch):
    """HBM-side bytes per window-kernel launch from the committed rocprofv3 PMC passes of THIS command
    (scripts/collect_profiles.sh + collect_calibration.sh + summarize_profiles.py: FETCH_SIZE and WRITE_SIZE in separate
    passes, KiB units, and FETCH_SIZE calibrated on the kernel's own two read patterns -- 64-byte row gathers are counted in
    full, the draw-record pieces at half their bytes: profiles/fetch_calibration.json).  A counter cannot be read from inside
    the run it counts, so the figure comes from the profile file -- and is only reported when the profiled launches had the
    shape of the launches just timed (same chains, d, K, layout, generations per launch); otherwise null.
    Returns (calibrated bytes per launch, source) or None."""
    f = ROOT / "profiles" / "latest_traffic.json"
    if not f.exists():
        return None
    prof = json.loads(f.read_text())
    shape = prof.get("launch_shape") or {}
    if (shape.get("chains"), shape.get("dim"), shape.get("K"), shape.get("lanes_per_chain")) != (n_loc, d, K, lanes):
        return None
    if abs(float(shape.get("generations_per_launch", -1)) - gens_per_launch) > 1e-9:
        return None
    if shape.get("bytes_per_launch_calibrated") is None:
        return None
    return float(shape["bytes_per_launch_calibrated"]), f"profiles/{prof.get('tag')}_traffic.json"


def throughput_point(demc, N, d, K, seed, gens, device_id, reps=1):
    """One point of the chain-count sweep (not the headline): the same generation loop at N chains,
    history and appends included, to show where the path leaves the latency regime.  `reps` calls of `gens` generations are
    timed back to back (small populations: one call of 1000 generations is a quarter host latency)."""
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    G = (1 + reps) * gens
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (G // K + 1), Gcap=G, blockindex=[range(d)],
                       eps_scale=w["eps_scale"], seed=seed, target=w["target"], device_id=device_id)
    e.set_state(w["Zinit"][-N:], None, w["Zinit"])
    e.run(1, gens, w["gamma"])
    e.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        e.run((1 + i) * gens + 1, (2 + i) * gens, w["gamma"])
    e.synchronize()
    dt = time.perf_counter() - t0
    lanes = e.info()["lanes_per_chain"]
    e.close()
    gbs = N * gens * reps / dt * algorithmic_bytes_per_update(d, K) / 1e9
    return {"chains": N, "lanes_per_chain": lanes, "generations_timed": gens * reps, "value": N * gens * reps / dt, "achieved_GBps": gbs,
            "frac_of_8TBps": gbs / 8000.0}


FP64_MFMA_PEAK_TFLOPS = 78.6    # v_mfma_f64_16x16x4_f64: 2048 flop / 65 clocks x 1024 SIMDs x 2.4 GHz (measured instruction rate,
                                # scripts/probes/mfma_f64_rate.hip, profiles/r03c_mfma_f64_rate.txt)


def config_row(demc, name, w, N, d, K, blocks, seed, device_id, gens=2000, anneal=False, flops_per_update=None):
    """One of the other BASELINE configs that fit one GPU (C3, C4's per-GPU shard, C5): `gens` generations after a warm slab of
    the same length, history and appends included, window kernels timed by HIP events on their own stream inside the library
    (demcz_set_kernel_timing).  Priced like the headline: algorithmic bytes per launch / average launch
    duration against HBM -- or, for the regression target, algorithmic flops against the FP64 matrix rate."""
    M0 = w["Zinit"].shape[0]
    e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (2 * gens // K + 1), Gcap=2 * gens, blockindex=blocks, eps_scale=w["eps_scale"],
                       seed=seed, target=w["target"], device_id=device_id)
    try:
        e.set_state(w["Zinit"][-N:], None, w["Zinit"])
        temps = np.array([demc.tempbaseline(g, 2 * gens, 3.0, 1e-3) for g in range(1, 2 * gens + 1)]) if anneal else None
        e.run(1, gens, w["gamma"], None if temps is None else temps[:gens])
        e.synchronize()
        e.set_kernel_timing(True)
        t0 = time.perf_counter()
        e.run(gens + 1, 2 * gens, w["gamma"], None if temps is None else temps[gens:])
        e.synchronize()
        dt = time.perf_counter() - t0
        launches, ev_ms = e.get_kernel_time()
        e.set_kernel_timing(False)
        rh = e.rhat(gens + 1, 2 * gens)
        live_on, redos = e.live_status()
        B = algorithmic_bytes_per_update(d, K)
        gpl = gens / max(launches, 1)
        avg_s = ev_ms / 1e3 / max(launches, 1)
        row = {"workload": name, "chains": N, "dim": d, "K": K, "generations_timed": gens, "value": N * gens / dt,
               "unit": "chain-updates/s", "us_per_K_window": dt / (gens / K) * 1e6,
               "us_per_K_window_kernels": (ev_ms / 1e3) / (gens / K) * 1e6,
               "kernel": e.kernel_name(), "lanes_per_chain": e.info()["lanes_per_chain"], "live_launches": live_on, "live_redos": redos,
               "launches": launches, "generations_per_launch": gpl, "avg_launch_us": avg_s * 1e6, "max_rhat": float(np.max(rh))}
        if flops_per_update is None:
            ach = B * N * gpl / avg_s / 1e9
            row["roofline"] = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                               "bytes_per_chain_update": B, "algorithmic_bytes_per_launch": B * N * gpl}
        else:
            ach = flops_per_update * N * gpl / avg_s / 1e12
            row["roofline"] = {"bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / FP64_MFMA_PEAK_TFLOPS, "flops_per_chain_update": flops_per_update,
                               "peak_source": "v_mfma_f64_16x16x4_f64 measured: 2048 flop / 65 clocks x 1024 SIMDs x 2.4 GHz "
                                              "(profiles/r03c_mfma_f64_rate.txt); dtype f64"}
        return row
    finally:
        e.close()


def config_rows(demc, seed, device_id):
    """BASELINE configs C3, C4 (one GPU's shard of it) and C5 -- and C1 on the HIP path (its CPU row is in cpu_baseline)."""
    out = {}
    K = 10
    try:
        w = demc.workloads.mvnormal_problem(20, 4096)
        out["C3"] = config_row(demc, "C3: MvNormal d=20, Nblocks=4 (4 x 5), N=4096", w, 4096, 20, K,
                               [range(0, 5), range(5, 10), range(10, 15), range(15, 20)], seed, device_id)
        w = demc.workloads.mvnormal_problem(20, 1024)
        out["C4_shard"] = config_row(demc, "C4: one GPU's shard of MvNormal d=20 N=8192 over 8 GPUs: N=1024, Z all local", w, 1024, 20, K,
                                     [range(20)], seed, device_id)
        w = demc.workloads.linreg_problem(10, 2048)
        nobs = w["design"].shape[0]
        out["C5"] = config_row(demc, "C5: demcz_anneal on linear-regression SSE, d=10, nobs=1000, N=2048, T0=3 -> TN=1e-3", w, 2048, 10, K,
                               [range(10)], seed, device_id, anneal=True, flops_per_update=2.0 * nobs * 10 + 3.0 * nobs)
        w = demc.workloads.mvnormal_problem(5, 4)
        out["C1_hip"] = config_row(demc, "C1 on the HIP path: MvNormal d=5, N=4, 10 000 generations (BASELINE runs C1 on the CPU: cpu_baseline.c1)",
                                   w, 4, 5, K, [range(5)], seed, device_id, gens=10000)
    except Exception as e:          # reporting only
        out["error"] = f"{type(e).__name__}: {e}"[:300]
    return out


def d_sweep_rows(demc, seed, device_id, gens=1000):
    """`configs.d_sweep` (round 5): C2's population (N = 1024, K = 10, one full block) across dimensions, so that the line shows
    the fast path is not a whitelist of shapes -- the dimensions rounds 2-4 built (5, 8, 10, 20) next to the ones in between and
    beyond, which the reference's own scripts use (README.md:16 "10-20 dimensions"; test/example_linreg.jl:9: 26;
    test/test_anneal_parallel.jl:16: 30) -- and the isotropic quadratic of test/test_anneal.jl:7-10 at d = 10 / 30, tempered.
    Every row: kernel name, events-based launch time, HBM roofline fraction, like the BASELINE configs above."""
    rows = []
    try:
        for d in (5, 6, 7, 8, 10, 12, 16, 20, 26, 30):
            w = demc.workloads.mvnormal_problem(d, 1024)
            rows.append(config_row(demc, f"MvNormal d={d}, N=1024", w, 1024, d, 10, [range(d)], seed, device_id, gens=gens))
        for d in (10, 30):
            w = demc.workloads.iso_quad_problem(d, 1024)
            rows.append(config_row(demc, f"isotropic quadratic d={d}, N=1024, tempered T0=3 -> TN=1e-3", w, 1024, d, 10, [range(d)], seed,
                                   device_id, gens=gens, anneal=True))
        # the reference's regression example at ITS dimension (test/example_linreg.jl:9: 25 regressors + intercept, nobs = 1000), annealed:
        # window_kernel_ml<LINREG_SSE, 26, 16> -- FP64 vector fmas, priced against the FP64 rate like C5's matrix kernel
        w = demc.workloads.linreg_problem(26, 1024)
        nobs = w["design"].shape[0]
        rows.append(config_row(demc, "linear-regression SSE d=26, nobs=1000, N=1024, tempered T0=3 -> TN=1e-3", w, 1024, 26, 10, [range(26)], seed,
                               device_id, gens=min(gens, 400), anneal=True, flops_per_update=2.0 * nobs * 26 + 3.0 * nobs))
    except Exception as e:          # reporting only
        rows.append({"error": f"{type(e).__name__}: {e}"[:300]})
    return rows


def closure_row(demc, seed, device_id, gens=300):
    """`configs.closure` (round 5): the reference's defining feature is an ARBITRARY log-density (`logobj`, src/demcz.jl:189,
    README.md:14).  Behind this ABI that is the host-closure mode: per block-step the device draws and forms the N proposals
    (demcz_propose), the host evaluates its closure on them, the device takes the N Metropolis decisions (demcz_accept_commit).
    C2's shape (MvNormal d = 5, N = 1024, K = 10) with a vectorised NumPy closure, two ways:
      copies     rounds 1-4: propose kernel -> D2H copy -> stream sync | closure | H2D copy -> accept kernel -> stream sync
      pipelined  round 5 (demcz_closure_buffers): the kernels write / read pinned host memory directly, the host spins on a flag
                 the propose kernel's last workgroup raises, the commit is only enqueued
    Per block-step: microseconds inside each of the three calls and inside the closure (host clock), and updates/s."""
    N, d, K = 1024, 5, 10
    w = demc.workloads.mvnormal_problem(d, N)
    M0 = w["Zinit"].shape[0]
    spec = w["target"].spec()
    mu, c0 = np.asarray(w["mu"], dtype=np.float64), float(spec["c0"])
    Wl = np.asarray(spec["W"], dtype=np.float64).reshape(d, d, order="F")

    def closure(X):                       # log N(x; mu, Sigma) for the rows of X, W = chol(Sigma)^-1 (lower)
        Y = (X - mu) @ Wl.T
        return c0 - 0.5 * np.einsum("ij,ij->i", Y, Y)

    out = {"workload": f"host-closure mode, C2's shape: MvNormal d={d}, N={N}, K={K}, vectorised NumPy closure", "generations_timed": gens}
    try:
        for mode in ("copies", "pipelined"):
            e = demc.HipEngine(N=N, d=d, K=K, Mcap=M0 + N * (2 * gens // K + 2), Gcap=2 * gens, blockindex=[range(d)],
                               eps_scale=w["eps_scale"], seed=seed, target=closure, device_id=device_id)
            try:
                X0 = np.asfortranarray(w["Zinit"][-N:])
                e.set_state(X0, closure(X0), w["Zinit"])
                bufs = e.closure_buffers() if mode == "pipelined" else None
                t_prop = t_clo = t_acc = t_end = 0.0
                t_all = 0.0
                for g in range(1, 2 * gens + 1):
                    timed = g > gens
                    a = time.perf_counter()
                    Xp = e.propose(g, 0, w["gamma"])
                    b = time.perf_counter()
                    if bufs is not None:
                        bufs[1][:] = closure(Xp)
                        c = time.perf_counter()
                        e.accept_commit(None)
                    else:
                        lp = closure(Xp)
                        c = time.perf_counter()
                        e.accept_commit(lp)
                    dd = time.perf_counter()
                    e.end_generation(g)
                    f = time.perf_counter()
                    if timed:
                        t_prop += b - a; t_clo += c - b; t_acc += dd - c; t_end += f - dd; t_all += f - a
                e.synchronize()
                out[mode] = {"value": N * gens / t_all, "unit": "chain-updates/s", "us_per_block_step": t_all / gens * 1e6,
                             "us_propose_call": t_prop / gens * 1e6, "us_closure": t_clo / gens * 1e6,
                             "us_accept_commit_call": t_acc / gens * 1e6, "us_end_generation_call": t_end / gens * 1e6}
            finally:
                e.close()
    except Exception as ex:          # reporting only
        out["error"] = f"{type(ex).__name__}: {ex}"[:300]
    return out


def cpu_c1(K=10, seed=31953150):
    """BASELINE config C1 as the reference runs it (test/example_normpdf.jl:20-30 plumbing): MvNormal d=5, N=4 chains, 10 000
    generations, ONE core, chains updated in the reference's order (chain ic+1 sees chain ic's fresh archive row inside a
    generation divisible by K, src/demcz.jl:30-33, 88-91), chains started at the zero vector (src/demcz.jl:11, 15: the slice of
    the zero-padded matrix).  The oracle -- this repo's C restatement -- not the Julia package."""
    import demc_jl_amd as demc
    import oracle_py as O
    try:
        O.build(native=True)
        native = True
    except Exception:
        native = False
    N, d, G = 4, 5, 10000
    w = demc.workloads.mvnormal_problem(d, N)
    Z0 = w["Zinit"]
    M0 = Z0.shape[0]
    Mcap = M0 + -(-N * G // K)
    prob = O.Problem(N, d, K, Mcap, w["eps_scale"], seed, target=w["target"].spec())
    X = np.zeros((N, d), order="F")                       # init="reference_zeros"
    lp = O.logp(prob, X)
    Z = np.zeros((Mcap, d), order="F")
    Z[:M0] = Z0
    t0 = time.perf_counter()
    O.run(prob, X, lp, Z, M0, 1, G, w["gamma"], history=True, native=native, schedule=O.SCHED_SEQUENTIAL)
    dt = time.perf_counter() - t0
    return {"value": N * G / dt, "unit": "chain-updates/s", "cores": 1, "seconds": dt,
            "sample": "C1 whole: N=4, d=5, K=10, 10 000 generations, reference order, chains start at zero (demcz.jl:11,15), O(1) index draw"}


def cpu_baseline(w, N, d, K, seed, budget_updates, M_final):
    """Time the oracle (test infrastructure, used here only as the reported CPU baseline)."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle_py as O
    try:
        O.build(native=True)
        native = True
    except Exception:
        native = False
    Z0 = w["Zinit"]
    M0 = Z0.shape[0]
    # the sample is sized by TIME: a 1000-generation calibration run, then as many generations as ~3 s of this host's
    # single core make (at most budget_updates); the sequential row runs half as many, the OpenMP row twice as many
    # -- about 10 s of wall time in all, whatever the host
    Gc = 1000 // K * K
    probc = O.Problem(N, d, K, M0 + -(-N * Gc // K), w["eps_scale"], seed, target=w["target"].spec())
    Xc = np.array(Z0[M0 - N:], order="F")
    lpc = O.logp(probc, Xc)
    Zc = np.zeros((M0 + -(-N * Gc // K), d), order="F")
    Zc[:M0] = Z0
    t0 = time.perf_counter()
    O.run(probc, Xc, lpc, Zc, M0, 1, Gc, w["gamma"], history=True, native=native)
    rate_c = N * Gc / (time.perf_counter() - t0)
    G = max(K, int(min(budget_updates, 3.0 * rate_c) // N) // K * K)
    Mcap = M0 + -(-N * G // K)
    prob = O.Problem(N, d, K, Mcap, w["eps_scale"], seed, target=w["target"].spec())
    X = np.array(Z0[M0 - N:], order="F")
    lp = O.logp(prob, X)
    Z = np.zeros((Mcap, d), order="F")
    Z[:M0] = Z0
    t0 = time.perf_counter()
    O.run(prob, X, lp, Z, M0, 1, G, w["gamma"], history=True, native=native)
    dt = time.perf_counter() - t0
    out = {"value": N * G / dt, "unit": "chain-updates/s", "cores": 1, "kind": "port",
           "sample": f"oracle/demcz_oracle.c ({'-O3 -march=native' if native else '-O2'}), synchronous schedule, "
                     f"O(1) index draw, N={N} d={d} K={K}, {G} generations incl. history writes, {dt:.1f} s"}
    # the reference's own order (chain ic+1 sees chain ic's fresh row inside a generation
    # divisible by K, src/demcz.jl:30-33, 88-91), one thread; and the same with the reference's O(M) index draw
    # (collect(1:M) + deleteat!, src/demcz.jl:176-178) costed at the archive size this bench run ends with
    try:
        Gs = max(K, G // 2 // K * K)
        Mcap_s = M0 + -(-N * Gs // K)
        prob_s = O.Problem(N, d, K, Mcap_s, w["eps_scale"], seed, target=w["target"].spec())
        Xs = np.array(Z0[M0 - N:], order="F")
        lps = O.logp(prob_s, Xs)
        Zs = np.zeros((Mcap_s, d), order="F")
        Zs[:M0] = Z0
        t0 = time.perf_counter()
        O.run(prob_s, Xs, lps, Zs, M0, 1, Gs, w["gamma"], history=True, native=native, schedule=O.SCHED_SEQUENTIAL)
        dts = time.perf_counter() - t0
        seq = N * Gs / dts
        out["sequential"] = {"value": seq, "cores": 1, "sample": f"reference order (Gauss-Seidel within a generation), O(1) index draw, {Gs} generations, {dts:.1f} s"}
        L = O.lib(native)
        n_rep = max(3, int(1.5e8 // max(M_final, 1)))
        t0 = time.perf_counter()
        for i in range(n_rep):
            L.oracle_faithful_index_cost(int(M_final), (i * 7919) % int(M_final))
        per = (time.perf_counter() - t0) / n_rep
        out["faithful"] = {"value": 1.0 / (1.0 / seq + per), "cores": 1, "archive_rows": int(M_final), "index_draw_us": per * 1e6,
                           "sample": f"sequential row + the reference's O(M) index draw (collect(1:M); deleteat!) emulated at M = {int(M_final)} "
                                     f"rows (the archive at the end of this bench run): {per * 1e6:.0f} us per proposal, {n_rep} repetitions"}
    except Exception as e:
        out["sequential"] = {"value": None, "error": str(e)[:200]}
    # the same loop spread over the host's cores (OpenMP over chains): the "CPU-omp" row, reported beside the
    # single-core figure (`value`/`cores` stay the scalar port)
    try:
        # at most 8 threads: the box's CPU share is smaller than its affinity mask suggests, and spinning OpenMP
        # barriers on oversubscribed cores take minutes instead of seconds
        thr = max(1, min(len(os.sched_getaffinity(0)), 8))
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        X = np.array(Z0[M0 - N:], order="F")
        lp = O.logp(prob, X)
        Gm = 2 * G
        Mcap2 = M0 + -(-N * Gm // K)
        prob2 = O.Problem(N, d, K, Mcap2, w["eps_scale"], seed, target=w["target"].spec())
        Z2 = np.zeros((Mcap2, d), order="F")
        Z2[:M0] = Z0
        t0 = time.perf_counter()
        O.run(prob2, X, lp, Z2, M0, 1, Gm, w["gamma"], history=True, native=native, threads=thr)
        dt2 = time.perf_counter() - t0
        out["omp"] = {"value": N * Gm / dt2, "cores": thr, "sample": f"{Gm} generations, {dt2:.1f} s"}
    except Exception as e:  # the OpenMP row is optional; the scalar port above is the contract
        out["omp"] = {"value": None, "error": str(e)[:200]}
    try:
        out["c1"] = cpu_c1(K)
    except Exception as e:
        out["c1"] = {"value": None, "error": str(e)[:200]}
    return out


# Window-kernel time of one batch of the deferred schedule at 1024 chains, d = 5, K = 10 (one launch per batch of E
# boundaries), measured on one MI355X with `bench.py --append-lag E`: 23.4 us at E = 10, 47.1 at 25, 87.5 at 50
#  -- what a batch's all-gather has to hide behind.
def deferred_batch_us(E, n_loc):
    return (7.5 + 1.6 * E) * (n_loc / 1024.0)


def sharded_selfcheck(demc, sharding, rank, local_rank, stream_ptr, all_min, sharded_engine_factory=None):
    """Before any multi-GPU number is quoted: a small run over the communicator (this rank's shard of 128 chains per rank, 120
    generations, the R-hat of the whole population) against the SAME run made by this GPU alone over all the chains -- results
    do not depend on the sharding (a chain's random stream is its global id, every rank holds the whole archive), so the shard
    must equal its rows of the unsharded history bit for bit, the archives must be equal, and so must the statistic.  Both
    schedules: rows visible from the next generation on, and batches of two boundaries.  `all_min`: reduces an int over the
    ranks with MIN.  `sharded_engine_factory`: tests on a one-GPU box build the sharded leg's engine with a one-rank communicator.
    Returns {schedule: bool}."""
    n_loc, d, K, G, seed = 128, 5, 10, 120, 4242
    world = sharding.world_size if sharding else 1
    N = n_loc * world
    w = demc.workloads.mvnormal_problem(d, N)
    X, logp = demc.initial_state(w["target"], w["Zinit"], N, G, K, None, "last_rows")
    out = {}
    for name, lag in (("every_K", 0), ("batches_of_2", 2)):
        res = []
        for sh in (sharding, None):
            r = demc.make_runner(w["target"], w["Zinit"], N, K, G, [range(d)], w["eps_scale"], X, logp, seed=seed, sharding=sh,
                                 device_id=local_rank, engine_factory=sharded_engine_factory if sh is not None else None,
                                 lanes_per_chain=0, stream=stream_ptr, append_lag=lag)
            r.run(1, G, w["gamma"])
            rh = r.rhat(1, G)
            ch, lo = r.history(1, G)
            _, _, Z, M = r.state()
            r.close()
            res.append((np.array(ch), np.array(lo), np.array(Z), int(M), np.array(rh)))
        (cs, ls, Zs, Ms, rs), (c1, l1, Z1, M1, r1) = res
        lo_, hi_ = rank * n_loc, (rank + 1) * n_loc
        ok = (np.array_equal(cs, c1[lo_:hi_]) and np.array_equal(ls, l1[lo_:hi_]) and Ms == M1 and np.array_equal(Zs, Z1)
              and bool(np.all(np.abs(rs - r1) <= 1e-9 * np.maximum(1.0, np.abs(r1)))))
        out[name] = bool(all_min(1 if ok else 0) == 1)
    return out


def tune_append_lag(dist, torch, n_loc, d, K, every, device):
    """Sharded runs: boundaries per all-gather (demcz_set_append_lag).  A batch's rows travel while the next batch computes;
    if the all-gather takes longer than that the run is bound by the links' latency, not by the kernels.  The latency of an
    8-rank all-gather of a few hundred KB is the node's, not ours to assume: time it here, once, for the batch sizes offered
    (they divide a slab), and take the smallest one whose all-gather (with a quarter to spare) fits behind its batch's compute.
    Every rank sees the same (max-reduced) timings, so every rank takes the same schedule.  Returns (E, [(E, us), ...])."""
    world = dist.get_world_size()
    # (E = 10 is not offered: the exchange's fixed cost per batch on the compute stream -- two cross-stream waits, the snapshot --
    #  is ~14 us even at nranks = 1, scripts/rccl_single_rank_lag.py: 60 % of a 10-boundary batch, 30 % of a 25-boundary one)
    cands = [c for c in (25, 50) if (every // K) % c == 0] or [10]
    rows = []
    for E in cands:
        n = E * n_loc * d
        src = torch.zeros(n, dtype=torch.float64, device=device)
        dst = torch.empty(n * world, dtype=torch.float64, device=device)
        for _ in range(3):
            dist.all_gather_into_tensor(dst, src)
        if device != "cpu":
            torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            dist.all_gather_into_tensor(dst, src)
        if device != "cpu":
            torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rows.append((E, float(t.item()) * 1e6))
    chosen = cands[-1]
    for E, us in rows:
        # (the batch-time model was measured at d = 5; other dimensions keep the largest batch)
        if d == 5 and us * 1.25 < deferred_batch_us(E, n_loc):
            chosen = E
            break
    return chosen, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed autostop slabs (1000 generations + R-hat check each)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed slabs of the same shape before them (at least 1 is run)")
    ap.add_argument("--slab-generations", type=int, default=1000, help="generations per step: autostop_every (DEMC.jl:41)")
    ap.add_argument("--chains-per-gpu", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the chain-count sweep object")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (C3, C4's shard, C5, C1 on the HIP path)")
    ap.add_argument("--lanes-per-chain", type=int, default=0)
    ap.add_argument("--append-lag", type=int, default=-1,
                    help="demcz_set_append_lag batches for `value`.  Default: 0 (rows visible from the next generation on -- the "
                         "reference's schedule with a synchronous exchange); when sharded the tuned deferred schedule is measured "
                         "as well and reported as `value_deferred`")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the rendezvous (nccl = RCCL; gloo with --dry-run)")
    ap.add_argument("--dry-run", action="store_true",
                    help="everything up to the first GPU call: arguments, rendezvous, sharding plan, the exchange of the 128-byte "
                         "communicator id -- printed as JSON by rank 0 (lets a host without GPUs check the N > 1 launch path)")
    args = ap.parse_args()

    import torch
    import demc_jl_amd as demc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if args.steps < 1 or args.warmup < 0 or args.slab_generations < 4:
        raise SystemExit("need --steps >= 1, --warmup >= 0, --slab-generations >= 4")
    if not args.dry_run:
        torch.cuda.set_device(local_rank)
    sharding = None
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group(args.backend)
        else:
            dist.init_process_group(args.backend, device_id=torch.device("cuda", local_rank))
        from demc_jl_amd.dist import torch_sharding
        sharding = torch_sharding(mode="rccl")

    # sharded default: boundaries per all-gather chosen by tune_append_lag (25 if the probe fails): a batch's compute
    # (23.5 us per 100 generations) has to outlast a latency-bound all-gather over the node's links
    d, K, n_loc = args.dim, 10, args.chains_per_gpu
    lag_probe = None
    if args.append_lag >= 0:
        lag = args.append_lag
    elif world == 1:
        lag = 0
    else:
        lag = 25 if (args.slab_generations // K) % 25 == 0 else 10
        ok = 1
        try:      # (a failed probe leaves the documented default; a probe cannot change results, only the schedule that is reported)
            cand, lag_probe = tune_append_lag(dist, torch, n_loc, d, K, args.slab_generations, "cpu" if args.dry_run else "cuda")
        except Exception as e:
            ok, cand, lag_probe = 0, lag, f"failed: {e}"
        # Every rank must run the same batch size (mismatched all-gather counts hang): the probe counts only if it worked on ALL
        # ranks, and then rank 0's choice is the one taken.
        dev = "cpu" if args.dry_run else "cuda"
        flag = torch.tensor([ok], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        pick = torch.tensor([cand if int(flag.item()) == 1 else lag], dtype=torch.int64, device=dev)
        dist.broadcast(pick, src=0)
        lag = int(pick.item())
        if int(flag.item()) != 1 and not isinstance(lag_probe, str):
            lag_probe = "failed on another rank"
    N = n_loc * world
    every = args.slab_generations
    S = args.steps
    W = max(args.warmup, 1)          # one slab of the timed shape always runs first (code objects, record buffers, draws)
    G = (W + S) * every
    thr, seed = 1.05, 31953150
    w = demc.workloads.mvnormal_problem(d, N)
    if args.dry_run:
        # what make_runner / demcz_comm_init would be given, without touching a GPU
        X, logp = demc.initial_state(w["target"], w["Zinit"], N, G, K, None, "last_rows")
        plan = {"world": world, "rank": rank, "local_rank": local_rank, "chains_total": N, "chains_per_gpu": n_loc,
                "chain_id0": rank * n_loc, "append_lag": lag, "generations": G, "warmup_slabs": W, "timed_slabs": S,
                "Mcap": int(w["Zinit"].shape[0] + -(-N * G // K)), "X_shard_shape": list(X[rank * n_loc:(rank + 1) * n_loc].shape),
                "mode": sharding.mode if sharding else "single GPU", "append_lag_probe_us": lag_probe}
        if sharding is not None:
            uid = bytes(range(128)) if rank == 0 else None         # stands in for demcz_comm_unique_id's ncclUniqueId
            got = sharding.broadcast_bytes(uid)
            plan["unique_id_ok"] = (got == bytes(range(128)))
            parts = sharding.all_gather(np.full(3, float(rank)))
            plan["all_gather_ranks"] = [float(p[0]) for p in parts]
            plans = [None] * world
            dist.all_gather_object(plans, plan)
            if rank == 0:
                print(json.dumps({"dry_run": True, "ranks": plans}))
            dist.destroy_process_group()
        else:
            print(json.dumps({"dry_run": True, "ranks": [plan]}))
        return
    stream = torch.cuda.Stream()
    selfcheck = None
    exchange_note = None
    if world > 1:
        def all_min(v):
            t = torch.tensor([v], dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item())
        try:
            selfcheck = sharded_selfcheck(demc, sharding, rank, local_rank, stream.cuda_stream, all_min)
        except demc.DemczError as e:
            if e.code == 6:            # DEMCZ_ERR_COMM: same exit as below
                sys.stderr.write(json.dumps({"error": "DEMCZ_ERR_COMM in the sharded self-check", "message": str(e), "rank": rank}) + "\n")
                sys.stderr.flush()
                os._exit(3)
            raise
        exchange_note = None
        if not all(selfcheck.values()) and not os.environ.get("DEMCZ_NO_PEER"):
            # The in-launch hand-off between the GPUs (every rank's publisher waves store boundary rows into every replica over
            # IPC-opened pointers, demcz_comm_init) has never met real peers before this line runs: if the sharded run does not
            # reproduce the unsharded one WITH it, it is switched off -- the rows then travel through ncclAllGather per K-window,
            # the schedule of rounds 1-3 -- and the check is made again.  (The verdict is the MIN over ranks: every rank is here.)
            exchange_note = f"peer hand-off failed the sharded self-check ({selfcheck}); fell back to the ncclAllGather exchange"
            os.environ["DEMCZ_NO_PEER"] = "1"
            selfcheck = sharded_selfcheck(demc, sharding, rank, local_rank, stream.cuda_stream, all_min)
        if not all(selfcheck.values()):
            # a sharded run that does not reproduce the unsharded one is not a measurement of anything: say so and stop
            if rank == 0:
                print(json.dumps({"metric": "chain-updates/sec (N x gens/s) + gens-to-Rhat<1.05, MvNormal d=5 N=1024", "value": None,
                                  "unit": "chain-updates/s", "n_gpus": world, "error": "sharded self-check failed: the sharded run "
                                  "differs from the same run on one GPU", "sharded_selfcheck": selfcheck}))
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(4)
    X, logp = demc.initial_state(w["target"], w["Zinit"], N, G, K, None, "last_rows")

    def measure(lag_m):
        """One complete measurement with `lag_m` boundaries per exchange (0: the north star's schedule -- rows visible from the
        next generation on): runner, W untimed slabs, the fenced timed region of S slabs.  Returns a dict."""
        runner = demc.make_runner(w["target"], w["Zinit"], N, K, G, [range(d)], w["eps_scale"], X, logp, seed=seed,
                                  sharding=sharding, device_id=local_rank, engine_factory=None,
                                  lanes_per_chain=args.lanes_per_chain, stream=stream.cuda_stream, append_lag=lag_m)
        eng = runner.engines[0]
        res = {"gens_to_rhat": None, "rhat_trace": []}

        def advance(g_from, g_to, timed):
            """Generations g_from..g_to with the R-hat check every `every` generations: one library call
            (demcz_run_checked = the driver loop demcz.jl:30-55 without its stop), the window kernels timed by
            HIP events on their own stream inside the library.  Returns (ms spent in window kernels, launches)."""
            _, trace, _ = runner.run_checked(g_from, g_to, w["gamma"], every, 0.0)      # (returns with the streams drained)
            return trace

        def note(trace, g_from):
            first = ((g_from - 1) // every + 1) * every
            for i, mx in enumerate(trace):
                gchk = first + i * every
                res["rhat_trace"].append((gchk, float(mx)))
                if res["gens_to_rhat"] is None and mx < thr:
                    res["gens_to_rhat"] = gchk

        stamps = {}

        def fence(drain_library=True, name=None):
            if drain_library:
                runner.synchronize()
            torch.cuda.synchronize()          # (the whole device: the library's streams included)
            if name:
                stamps[name + "_arrive"] = time.perf_counter()      # (CLOCK_MONOTONIC: one clock for all processes of the node)
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize()
            if name:
                stamps[name + "_leave"] = time.perf_counter()

        note(advance(1, W * every, False), 1)
        eng.set_kernel_timing(True)
        fence(name="pre")
        # the timed region holds the S steps and nothing else: one library call that returns when its last slab's statistic is on
        # the host, and the fence.  Reading the events out, bookkeeping in Python: after it.
        t0 = time.perf_counter()
        trace = advance(W * every + 1, G, True)
        stamps["timed_done"] = time.perf_counter()
        fence(drain_library=False)            # (demcz_run_checked returned with the last slab's statistic on the host)
        dt = time.perf_counter() - t0
        note(trace, W * every + 1)
        launches, ev_ms = eng.get_kernel_time()
        eng.set_kernel_timing(False)
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        # per-step times as the GPU saw them: start-to-start of consecutive slabs' brackets (device clock; gaps between
        # launches and the R-hat kernels beside them included), the last step by its own bracket -- SURVEY 8(d): median of repeats
        try:
            st, du = eng.get_kernel_time_series()
            steps_ms = [float(x) for x in np.diff(st)] + ([float(du[-1])] if len(du) else [])
        except Exception:
            steps_ms = []
        res.update(dt=dt, ev_ms=ev_ms, launches=launches, steps_ms=steps_ms,
                   acc=runner.accept_ratio_mean(G - every + 1, G))          # (a cheap end-of-run sanity check, not timed)
        res["live_on"], res["live_redos"] = eng.live_status()
        res["lanes"] = eng.info()["lanes_per_chain"]
        res["peer"] = eng.peer_status()
        res["kernel"] = eng.kernel_name()
        res["kernel_counts"] = eng.kernel_counts()
        # what a SCALE line needs to explain itself (round 5): per rank, did the in-launch hand-off hold (redos), did the handle go
        # LIVE again (re-arms), what demcz_comm_init's first-contact check saw, when the rank reached the rendezvous in front of
        # the timed region and how long its own timed call took
        res["rank_diag"] = {"rank": rank, "live_launches": bool(res["live_on"]), "live_redos": int(res["live_redos"]),
                            "live_rearms": list(eng.live_rearms()) if hasattr(eng, "live_rearms") else None,
                            "peer_status": list(res["peer"]), "peer_ping": list(eng.peer_ping()) if hasattr(eng, "peer_ping") else None,
                            "window_launches_timed": int(launches), "rendezvous_arrive_s": stamps.get("pre_arrive"),
                            "rendezvous_leave_s": stamps.get("pre_leave"), "own_timed_call_ms": (stamps["timed_done"] - t0) * 1e3}
        runner.close()        # (frees the device's LIVE slot for the next measurement / the sweep's handles)
        return res

    from demc_jl_amd._lib import DemczError, ERR_COMM
    try:
        # N > 1: `value` is the north star's schedule ("all-gather ... every K steps": synchronous exchange, rows visible in the
        # next generation, the sampler the N = 1 line runs); the tuned deferred schedule is measured right after it and reported
        # beside it as `value_deferred` -- a different sampler (rows visible lag..2 lag windows late), so it never is `value`.
        both = world > 1 and args.append_lag < 0
        m = measure(0 if both else lag)
        md = measure(lag) if both else None
    except DemczError as e:
        if e.code == ERR_COMM:
            # a peer died or stalled: the handle's communicators are aborted.  A fresh process is the only retry -- and no
            # torch.distributed call here: the other ranks may be gone
            print(json.dumps({"error": "DEMCZ_ERR_COMM", "rank": rank, "detail": str(e)}), file=sys.stderr, flush=True)
            os._exit(3)
        raise
    rank_rows, skew = [m["rank_diag"]], None
    if dist is not None:
        rank_rows = [None] * world
        dist.all_gather_object(rank_rows, m["rank_diag"])
        arr = [r["rendezvous_arrive_s"] for r in rank_rows if r and r.get("rendezvous_arrive_s") is not None]
        lea = [r["rendezvous_leave_s"] for r in rank_rows if r and r.get("rendezvous_leave_s") is not None]
        if arr and lea:
            skew = {"arrival_spread_us": (max(arr) - min(arr)) * 1e6, "release_spread_us": (max(lea) - min(lea)) * 1e6,
                    "what": "spread over the ranks of when each reached / left the barrier in front of the timed region (host clock, one node): "
                            "what the ranks' first launches start apart by -- the in-launch hand-off tolerates ~0.1-0.3 s (its poll limit)"}
        for r in rank_rows:
            if r:
                r.pop("rendezvous_arrive_s", None); r.pop("rendezvous_leave_s", None)
    dt, ev_ms, launches, lanes = m["dt"], m["ev_ms"], m["launches"], m["lanes"]
    gens_to_rhat, rhat_trace, acc, live_on, live_redos = m["gens_to_rhat"], m["rhat_trace"], m["acc"], m["live_on"], m["live_redos"]
    lag_value = 0 if both else lag

    if rank == 0:
        B = algorithmic_bytes_per_update(d, K)
        gens = S * every
        gens_per_launch = gens / max(launches, 1)
        bytes_per_launch = B * n_loc * gens_per_launch
        avg_launch_s = (ev_ms / 1e3) / max(launches, 1)
        achieved = bytes_per_launch / avg_launch_s / 1e9
        # PMC bytes per launch from the committed rocprofv3 passes of this same command, FETCH_SIZE calibrated on the kernel's own
        # read patterns (one number); null unless the profiled launches had this shape
        traffic = measured_traffic(n_loc, d, K, lanes, gens_per_launch) or (None, None)
        out = {
            "metric": "chain-updates/sec (N x gens/s) + gens-to-Rhat<1.05, MvNormal d=5 N=1024",
            "value": N * gens / dt, "unit": "chain-updates/s", "n_gpus": world, "steps": S, "warmup": W,
            "ms_per_step": dt / S * 1e3,
            "ms_per_step_median": float(np.median(m["steps_ms"])) if m["steps_ms"] else None,
            "value_median": (N * every / (float(np.median(m["steps_ms"])) / 1e3)) if m["steps_ms"] else None,
            "ms_per_step_device": {"n": len(m["steps_ms"]), "min": float(np.min(m["steps_ms"])) if m["steps_ms"] else None,
                                   "median": float(np.median(m["steps_ms"])) if m["steps_ms"] else None,
                                   "max": float(np.max(m["steps_ms"])) if m["steps_ms"] else None,
                                   "what": "start-to-start of consecutive slabs on the kernel's stream (HIP events), last slab by its own bracket"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C2: MvNormal d={d} correlated Sigma, N={n_loc} chains/GPU x {world} GPU, K={K}, "
                                   f"gamma=2.38, eps=1e-5; step = one autostop slab = {every} generations with full "
                                   f"history + Z append every K + the slab's split-Rhat check (demcz.jl:30-55); "
                                   + ("`value`: rows visible from the next generation on (append_lag 0"
                                      + (", synchronous RCCL all-gather every K generations)" if world > 1 else ")")
                                      if lag_value == 0 else f"`value`: deferred visibility, append_lag {lag_value}"),
                       "chains_total": N, "dim": d, "K": K, "generations_per_step": every, "generations_timed": gens,
                       "lanes_per_chain": lanes, "append_lag": lag_value, "append_lag_probe_us": lag_probe,
                       "live_launches": live_on, "live_redos": live_redos, "sharded_selfcheck": selfcheck,
                       "exchange": (None if world == 1 else
                                    "in-launch hand-off: every rank's publisher waves store a boundary's rows into every replica over IPC-opened "
                                    f"pointers ({m['peer'][1]} peers); no collective per K-window" if m["peer"][0] == 2 and live_on else
                                    "ncclAllGather + scatter per K-window" + (" (the in-launch hand-off timed out and the run was redone)" if m["peer"][0] == 2 else "")),
                       "exchange_note": exchange_note,
                       "ranks": rank_rows if world > 1 else None, "rendezvous_skew": skew,
                       "live_rearms": m["rank_diag"]["live_rearms"],
                       "parallelism": f"chains sharded x{world}, Z replicated" if world > 1 else "single GPU"},
            "gens_to_rhat_1p05": gens_to_rhat, "rhat_trace": rhat_trace[-12:], "accept_ratio_mean": acc,
            "value_window_kernels_only": N * gens / (ev_ms / 1e3) if ev_ms > 0 else None,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "frac_of_measured_copy": achieved / 6290.0,        # MI355X_MICROARCH.md: 6.29 TB/s measured copy rate
                         "traffic": traffic[0], "traffic_over_algorithmic": (traffic[0] / bytes_per_launch) if traffic[0] else None,
                         "traffic_source": traffic[1],
                         "kernel": m["kernel"] + (f" (+ demcz::produce_kernel<{d}> beside it)" if lanes == 164 else ""),
                         "kernel_counts": m["kernel_counts"],
                         "launches": launches, "generations_per_launch": gens_per_launch,
                         "avg_launch_us": avg_launch_s * 1e6, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "bytes_per_chain_update": B},
        }
        if md is not None:
            out["value_deferred"] = N * gens / md["dt"]
            out["deferred"] = {"append_lag": lag, "ms_per_step": md["dt"] / S * 1e3, "gens_to_rhat_1p05": md["gens_to_rhat"],
                               "rhat_trace": md["rhat_trace"][-6:], "accept_ratio_mean": md["acc"],
                               "ms_per_step_median": float(np.median(md["steps_ms"])) if md["steps_ms"] else None,
                               "what": f"the same run with the K-boundary rows of {lag} boundaries travelling in one all-gather on a side "
                                       f"stream and becoming visible {lag}..{2 * lag} windows later (demcz_set_append_lag): a different "
                                       "sampler from `value`'s -- compare its gens_to_rhat_1p05; `--gpus 1 --append-lag E` is its N = 1 point"}
        if world == 1 and not args.no_sweep and (n_loc, d) == (1024, 5):
            try:     # reporting only: throughput regime of the same path
                out["chain_count_sweep"] = [throughput_point(demc, n, d, K, seed, g, local_rank, reps)
                                            for n, g, reps in ((2048, 1000, 5), (4096, 1000, 5), (16384, 400, 5), (131072, 200, 2), (1048576, 100, 1))]
            except Exception as e:
                out["chain_count_sweep"] = f"failed: {e}"
        if world == 1 and not args.no_configs:
            out["configs"] = config_rows(demc, seed, local_rank)
            out["configs"]["d_sweep"] = d_sweep_rows(demc, seed, local_rank)
            out["configs"]["closure"] = closure_row(demc, seed, local_rank)
        if not args.no_cpu_baseline and world == 1:      # (the contract: rank 0 at N = 1 only)
            try:
                out["cpu_baseline"] = cpu_baseline(demc.workloads.mvnormal_problem(d, n_loc), n_loc, d, K, seed, 8.0e7, w["Zinit"].shape[0] + N * (G // K))
            except Exception as e:   # the baseline is reporting only; never fail the bench on it
                out["cpu_baseline"] = {"value": None, "unit": "chain-updates/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e}"}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
