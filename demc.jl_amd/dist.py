"""torch.distributed plumbing for sharded runs (one process per GPU).

PyTorch is used for what it is good at here -- rendezvous, process groups, a broadcast of the
128-byte RCCL id -- not for compute.  ``backend="nccl"`` IS RCCL on ROCm; ``"gloo"`` works on
hosts without a GPU (used by the world_size-2 CPU tests of the host-driven exchange).
"""
from __future__ import annotations

import numpy as np

from .sampler import Sharding


def torch_sharding(mode: str = "rccl", local_shards: int = 1, group=None) -> Sharding:
    """Build a ``Sharding`` from the initialised default process group.

    ``mode="rccl"``: the library's own communicator does the data-path collectives; torch only
    broadcasts the unique id.  ``mode="host"``: collectives go through torch.distributed on host
    tensors (any backend)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    cuda_backend = dist.get_backend(group) == "nccl"

    def _dev():
        return torch.device("cuda", torch.cuda.current_device()) if cuda_backend else torch.device("cpu")

    def all_gather(a: np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(_dev())
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        return [o.cpu().numpy() for o in outs]

    def all_reduce_sum(a: np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(a).copy()).to(_dev())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()

    def broadcast_bytes(b):
        obj = [b]
        dist.broadcast_object_list(obj, src=0, group=group)
        return obj[0]

    return Sharding(rank=rank, world_size=world, mode=mode, local_shards=local_shards, all_gather=all_gather,
                    all_reduce_sum=all_reduce_sum, broadcast_bytes=broadcast_bytes)
