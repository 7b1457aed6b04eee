"""Multi-GPU plumbing of the path: rays shard, the scene is broadcast once.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in CPU tests).  The
path has exactly one exchange step: rank `src` built and flattened the tree; its scene blob
(bvhgpu_scene_export: traversal array + shape AABBs) goes to every peer in ONE broadcast.  Hit lists
stay on the GPU that produced them — there is no gather/all-reduce on the data path.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(rank: int, world: int, rays_per_gpu: int) -> Tuple[int, int]:
    """Weak-scaling shard of the seed-0 create_ray stream (testbase.rs:687-691): rank r owns rays
    [r*R, (r+1)*R).  splitmix64's state after j draws is j*GAMMA, so a shard starts in O(1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * rays_per_gpu, rays_per_gpu


def broadcast_scene(blob, src: int = 0):
    """Broadcast the scene blob tensor in place (uint8 tensor on the GPU for nccl/RCCL, CPU for gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src)
    return blob


def broadcast_nbytes(nbytes: int, device, src: int = 0) -> int:
    """Peers learn the blob size before allocating the receive buffer."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(nbytes)], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src)
    return int(t.item())
