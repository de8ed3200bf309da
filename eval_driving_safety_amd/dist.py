"""One process per GPU.  PGD shards by image with no communication; the universal-patch attack has
one exchange per inner iteration: an all-reduce(SUM) of the [3,D,D] float32 patch delta (71-122 KB,
latency-bound - RCCL picks its low-latency protocol at this size).

``torch.distributed`` backend "nccl" is RCCL over xGMI on ROCm; the same code runs on "gloo" for the
CPU tests (tests/test_dist_gloo.py)."""
import os

import torch
import torch.distributed as dist


class Comm:
    """Thin view of the default process group; world 1 needs no initialisation at all."""

    def __init__(self, rank=0, world=1, initialised=False):
        self.rank, self.world, self.initialised = rank, world, initialised

    @staticmethod
    def from_env(backend=None, device=None):
        """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as torch.distributed.run exports them."""
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        if world == 1:
            return Comm()
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
            if backend is None:   # ADV_COMM_BACKEND=gloo: tests that put two ranks on ONE GPU (RCCL refuses that)
                backend = os.environ.get("ADV_COMM_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
        return Comm(rank, world, True)

    def shard(self, n):
        """indices of the units (stereo pairs) this rank owns: i % world == rank"""
        return range(self.rank, n, self.world)

    def rounds(self, n):
        """number of lock-step rounds needed to cover n units, identical on every rank"""
        return (n + self.world - 1) // self.world

    def all_reduce_sum_(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def all_reduce_max_(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def close(self):
        if self.initialised and dist.is_initialized():
            dist.destroy_process_group()
