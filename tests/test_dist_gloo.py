"""world_size-2 (and 2x2 shards) run of the sharded host path over torch.distributed/gloo on CPU.

Each rank holds half the chains (oracle-backed test engine), all-gathers its rows at every K
boundary, all-reduces the R-hat moments and the accept counts; the result must equal the
single-process run bit for bit, and every rank must take the same autostop decision."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, local_shards, outdir, lag=0):
    for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
        sys.path.insert(0, str(p))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import demc_jl_amd as demc
    from demc_jl_amd.dist import torch_sharding
    from oracle_engine import OracleEngine
    d, N = 5, 16
    w = demc.workloads.mvnormal_problem(d, N)
    opts = demc.demcopt(d, N=N, K=10, Ngeneration=2000, eps_scale=w["eps_scale"], verbose=False, autostop="Rhat",
                        autostop_every=250, autostop_Rhat=1.15)
    sh = torch_sharding(mode="host", local_shards=local_shards)
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=21, engine_factory=OracleEngine, sharding=sh, append_lag=lag)
    np.savez(Path(outdir) / f"rank{rank}.npz", chain=mc.chain, log_obj=mc.log_obj, Z=Z)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("local_shards", [1, 2])
def test_two_ranks_equal_single_process(tmp_path, local_shards):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), local_shards, str(tmp_path)), nprocs=world, join=True)
    import demc_jl_amd as demc
    from oracle_engine import OracleEngine
    d, N = 5, 16
    w = demc.workloads.mvnormal_problem(d, N)
    opts = demc.demcopt(d, N=N, K=10, Ngeneration=2000, eps_scale=w["eps_scale"], verbose=False, autostop="Rhat",
                        autostop_every=250, autostop_Rhat=1.15)
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=21, engine_factory=OracleEngine)
    G = mc.chain.shape[2]
    assert G < 2000, "autostop should trigger so that the agreement of the decision is tested"
    parts = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for r, p in enumerate(parts):
        assert p["chain"].shape[2] == G                              # same stop generation on every rank
        assert np.array_equal(p["Z"], Z)                             # replicated archive identical
        assert np.array_equal(p["chain"], mc.chain[r * N // 2:(r + 1) * N // 2])
        assert np.array_equal(p["log_obj"], mc.log_obj[r * N // 2:(r + 1) * N // 2])


def test_two_ranks_deferred_exchange_equal_single_process(tmp_path):
    """The deferred-visibility schedule (append_lag=2) across two gloo ranks == one process applying the
    same rule: what a rank may draw from is a function of the generation only."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), 1, str(tmp_path), 2), nprocs=world, join=True)
    import demc_jl_amd as demc
    from oracle_engine import OracleEngine
    d, N = 5, 16
    w = demc.workloads.mvnormal_problem(d, N)
    opts = demc.demcopt(d, N=N, K=10, Ngeneration=2000, eps_scale=w["eps_scale"], verbose=False, autostop="Rhat",
                        autostop_every=250, autostop_Rhat=1.15)
    one = demc.Sharding(mode="host", local_shards=1, host_exchange_always=True)
    mc, Z = demc.demcz_sample(w["target"], w["Zinit"], opts, seed=21, engine_factory=OracleEngine, sharding=one, append_lag=2)
    G = mc.chain.shape[2]
    for r in range(world):
        p = np.load(tmp_path / f"rank{r}.npz")
        assert p["chain"].shape[2] == G and np.array_equal(p["Z"], Z)
        assert np.array_equal(p["chain"], mc.chain[r * N // 2:(r + 1) * N // 2])


def test_bench_multi_gpu_launch_path_dry_run():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per GPU) up to the first GPU
    call: argument handling, rendezvous, the sharding plan (equal shards in rank order, global chain ids), the
    broadcast of the 128-byte communicator id and an all-gather -- over gloo, on a host without GPUs."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0",
           "--dry-run", "--backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["dry_run"] and len(out["ranks"]) == 2
    for rk, p in enumerate(out["ranks"]):
        assert (p["world"], p["rank"], p["chains_total"], p["chains_per_gpu"], p["chain_id0"]) == (2, rk, 2048, 1024, rk * 1024)
        # boundaries per all-gather: probed (every rank takes the same one), from the batch sizes that divide a slab
        assert p["append_lag"] in (25, 50) and p["append_lag"] == out["ranks"][0]["append_lag"]
        assert [e for e, _ in p["append_lag_probe_us"]] == [25, 50] and all(us > 0 for _, us in p["append_lag_probe_us"])
        assert p["mode"] == "rccl" and p["unique_id_ok"] and p["all_gather_ranks"] == [0.0, 1.0]
        assert p["warmup_slabs"] == 1 and p["timed_slabs"] == 3 and p["generations"] == 4000      # one untimed slab always runs
        assert p["X_shard_shape"] == [1024, 5] and p["Mcap"] == 2048 + 2048 * 400
    # a world size that does not match --gpus is refused before anything else happens
    bad = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                         timeout=120, env={k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}, cwd=str(ROOT))
    assert bad.returncode != 0 and "torch.distributed.run" in (bad.stderr + bad.stdout)
