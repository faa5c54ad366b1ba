"""Multi-GPU layer: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The reference restores every image independently (own B, beta, gamma, J: sucre.py:204,243), so images of a
scene shard across ranks with NO data-path collective (``shard_images``; results are bit-identical to a 1-GPU
run).  ``fit_shared_water`` is the north-star extension: all ranks step in lock-step and share the nine water
parameters; the only exchange is one all-reduce of 10 float64 sums per iteration (80 bytes -> latency-bound on
xGMI, so it is issued once per iteration on the compute stream, never bucketed).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world() -> tuple[int, int, int]:
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init_process_group(backend: str | None = None) -> tuple[int, int, int]:
    """Reads RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment (torchrun contract)."""
    rank, local_rank, world = env_rank_world()
    if (world > 1 or (backend is not None and COLLECTIVE_AT_WORLD_1)) and not dist.is_initialized():
        if backend is None:
            # RCCL needs one GPU per rank; several ranks sharing a GPU (a 1-GPU test box) talk over gloo instead.
            # device_count() does not initialise the GPU.
            enough = torch.cuda.device_count() >= int(os.environ.get('LOCAL_WORLD_SIZE', world))
            backend = os.environ.get('SUCRE_DIST_BACKEND', 'nccl' if (torch.cuda.is_available() and enough) else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


# A process group of ONE rank has nothing to sum, so the collective is skipped -- unless this is set: then the call goes
# through the backend anyway (RCCL reduces the tensor in place on its own stream and hands it back to the launch stream).
# tests/test_gpu_rccl.py uses it to put the real RCCL path (the float64 view into the engine's buffer, the stream
# hand-off around every iteration) under test on a one-GPU box.
COLLECTIVE_AT_WORLD_1 = False


def all_reduce_sum(t: torch.Tensor, group=None) -> None:
    """In-place sum over ranks.  RCCL reduces device tensors directly; under gloo (CPU tests, or several ranks
    sharing one GPU) a device tensor is staged through the host."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not COLLECTIVE_AT_WORLD_1):
        return
    if t.is_cuda and dist.get_backend(group) != 'nccl':
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def shard_images(image_ids: list, rank: int, world: int) -> list:
    """Contiguous, balanced partition of the target images over ranks (first ``len % world`` ranks get one
    extra).  Every image belongs to exactly one rank; order inside a rank is preserved."""
    n = len(image_ids)
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return list(image_ids[start:start + base + (1 if rank < extra else 0)])


class WaterBackend:
    """What ``fit_shared_water`` drives: one in-flight restoration on this rank.  ``sucre_amd.engine`` provides
    the HIP implementation (``HipWaterBackend``); CPU tests plug in an oracle-backed fake."""

    def n_obs(self) -> int: raise NotImplementedError
    def set_n_obs_total(self, n: int) -> None: raise NotImplementedError
    def grad(self, step: int) -> torch.Tensor: raise NotImplementedError      # float64[>=10] sums, on the comm device
    def step(self, step: int) -> None: raise NotImplementedError              # consumes the (all-reduced) sums


def fit_shared_water(backend: WaterBackend, num_iter: int, group=None) -> None:
    """Lock-step fit of one image per rank with shared B, beta, gamma: objective
    sum_ranks sum_obs r^2 / (3 sum_ranks n_obs).  Per iteration: local gradient pass -> all-reduce(sum) of the
    ten float64 sums -> identical Adam step on every rank (J updates stay local)."""
    n = torch.tensor([backend.n_obs()], dtype=torch.int64)
    if hasattr(backend, 'grad_device') and dist.is_initialized() and dist.get_backend(group) == 'nccl':
        n = n.to(backend.grad_device())
    all_reduce_sum(n, group)
    backend.set_n_obs_total(int(n.item()))
    for it in range(1, num_iter + 1):
        sums = backend.grad(it)
        all_reduce_sum(sums, group)
        backend.step(it)
    if hasattr(backend, 'finish'):
        backend.finish()     # a backend that defers its parameter step to the next launch takes the last one here
