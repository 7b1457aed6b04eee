"""Multi-GPU plumbing of the path: rays shard, the scene is broadcast once.

One process per GPU.  The path has exactly one exchange step: rank `root` built and flattened the tree; its traversal
array, shape AABBs and LDS slot table go to every peer in RCCL broadcasts issued by the C ABI itself (bvhgpu_bcast /
bvhgpu_bcast_known, csrc/comm.hip: ncclBroadcast straight from / into the trees' HBM buffers over xGMI).  Hit lists stay
on the GPU that produced them — there is no gather / all-reduce on the data path.  torch.distributed (or anything else
that can move 128 bytes) is only the launcher's rendezvous for the RCCL unique id, and the bench's barrier.

The older transport — export a scene blob, broadcast it with torch.distributed, import it on the peers — is kept
(broadcast_scene) for backends without RCCL (the gloo CPU tests) and as the fallback of bench.py.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

from . import _lib
from ._lib import check


def shard_range(rank: int, world: int, rays_per_gpu: int) -> Tuple[int, int]:
    """Weak-scaling shard of the seed-0 create_ray stream (testbase.rs:687-691): rank r owns rays
    [r*R, (r+1)*R).  splitmix64's state after j draws is j*GAMMA, so a shard starts in O(1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank * rays_per_gpu, rays_per_gpu


def strong_shard(rank: int, world: int, total_rays: int) -> Tuple[int, int]:
    """Strong-scaling shard (BASELINE.json configs[3]: 100 M rays sharded over the GPUs): rank r owns the contiguous slice
    [r*T/W, (r+1)*T/W) of the stream — the slices tile [0, T) exactly for any T, W."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    lo = rank * total_rays // world
    hi = (rank + 1) * total_rays // world
    return lo, hi - lo


class Communicator:
    """bvhgpu_comm: the RCCL communicator of the C ABI, one rank per process (bvhgpu_comm_init_rank)."""

    def __init__(self, ctx, nranks: int, rank: int, unique_id: bytes):
        lib = _lib.load()
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError("RCCL unique id must be 128 bytes")
        h = C.c_void_p()
        buf = (C.c_char * _lib.COMM_ID_BYTES).from_buffer_copy(unique_id)
        check(lib.bvhgpu_comm_init_rank(ctx._h, int(nranks), int(rank), C.cast(buf, C.c_void_p), C.byref(h)), ctx._h)
        self._h = h
        self.ctx = ctx
        self.nranks = int(nranks)
        self.rank = int(rank)
        ctx._child_add()      # (bvhgpu_comm_destroy synchronises the ctx's stream: the ctx must outlive the communicator, api.Context)
        self._counted = True

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * _lib.COMM_ID_BYTES)()
        check(_lib.load().bvhgpu_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    @staticmethod
    def from_torch_distributed(ctx, device=None) -> "Communicator":
        """Rendezvous through an initialised torch.distributed process group of any backend: rank 0 draws the id and the
        128 bytes are broadcast as a tensor (on `device` for the nccl backend, on the CPU for gloo)."""
        import torch
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        on = device if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
        t = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8, device=on)
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(Communicator.unique_id()), dtype=torch.uint8))
        dist.broadcast(t, 0)
        return Communicator(ctx, world, rank, bytes(t.cpu().numpy().tobytes()))

    def bcast(self, tree, root: int = 0, dtype: Optional[str] = None, n_shapes: Optional[int] = None, triangles: bool = False):
        """Collective.  `tree`: on the root the Bvh / FlatBvh to send; on a peer None (a FlatBvh is created) or the FlatBvh an
        earlier bcast returned (its HBM is reused).  With dtype ("f32"/"f64") and n_shapes known to every rank the broadcast
        is enqueued without any host round trip (bvhgpu_bcast_known); otherwise a header travels first.  Returns the tree."""
        from .api import FlatBvh
        lib = _lib.load()
        arr = (C.c_void_p * 1)(tree._t if tree is not None else None)
        if dtype is not None and n_shapes is not None:
            check(lib.bvhgpu_bcast_known(self._h, arr, int(root), _lib.F32 if dtype == "f32" else _lib.F64, int(n_shapes),
                                         _lib.BCAST_TRIANGLES if triangles else 0), self.ctx._h)
        else:
            check(lib.bvhgpu_bcast(self._h, arr, int(root)), self.ctx._h)
        if tree is not None:
            return tree
        dt = C.c_int()
        h = C.c_void_p(arr[0])
        check(lib.bvhgpu_tree_info(h, C.byref(dt), None, None, None), self.ctx._h)
        return FlatBvh(self.ctx, h, "f32" if dt.value == _lib.F32 else "f64")

    def info(self) -> dict:
        """what the communicator itself says (bvhgpu_comm_info) + which RCCL it runs over (bvhgpu_rccl_info)"""
        return comm_info(self._h)

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().bvhgpu_comm_destroy(self._h)
        self._h = None
        if getattr(self, "_counted", False):
            self._counted = False
            self.ctx._child_drop()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_info() -> dict:
    """bvhgpu_rccl_info: version code and file of the RCCL the C ABI's broadcasts go through (loads it on first use)"""
    lib = _lib.load()
    ver, shared = C.c_int(0), C.c_int(0)
    path = C.create_string_buffer(512)
    check(lib.bvhgpu_rccl_info(C.byref(ver), C.byref(shared), path, 512))
    v = int(ver.value)
    return {"version_code": v, "version": f"{v // 10000}.{v // 100 % 100}.{v % 100}" if v else None,
            "library": path.value.decode(errors="replace"), "shared_with_process": bool(shared.value)}


def comm_info(handle) -> dict:
    lib = _lib.load()
    nr, fr, nl = C.c_int(0), C.c_int(0), C.c_int(0)
    check(lib.bvhgpu_comm_info(handle, C.byref(nr), C.byref(fr), C.byref(nl)))
    out = {"nranks": int(nr.value), "first_rank": int(fr.value), "n_local": int(nl.value)}
    try:
        out.update(rccl_info())
    except _lib.BvhGpuError as e:   # (cannot happen on a live communicator; never let a report take a run down)
        out["rccl_info_error"] = str(e)
    return out


def broadcast_step(comm: "Communicator", rank: int, tree, aabbs, rays, dtype: str, n_shapes: int, hits=None, flags: int = 0, root: int = 0):
    """One step of the sharded path with the tree rebuilt on `root` every step (bench.py's N > 1 step; one process per GPU):

        root:  Bvh::build_par + flatten enqueued (rebuild_flat_async) → bvhgpu_bcast_known straight out of the tree's buffers (the
               status header is composed on the device from the build's own outcome) → its own shard's walk enqueued
        peer:  the receive enqueued → its shard's walk enqueued
        all :  ONE host wait at the end — no host synchronisation before it on any rank.

    `tree`: on the root the Bvh (rebuilt from `aabbs`, a device tensor or None to send the tree as it is); on a peer None or the FlatBvh
    the previous step returned (its HBM is reused).  BVHGPU_REBROADCAST — the root's optimistic build needed the slow path, which
    every rank sees at its wait — repeats the exchange with the finished tree (the root's own batch is complete by then).
    Returns (tree, stats of this rank's batch, number of rebroadcasts)."""
    from ._lib import REBROADCAST, BvhGpuError
    rebroadcasts = 0
    stats = None
    for attempt in range(3):
        if rank == root:
            if attempt == 0 and aabbs is not None:
                # the arrays that travel (folded binary array + its slot table) are wanted at once: one complete flatten pass instead of
                # the lazy pair (wide walk's arrays now, the rest on first use — which would be the very next call)
                lazy = tree.ctx.get_tuning(_lib.TUNE_FLATTEN_LAZY)
                tree.ctx.set_tuning(_lib.TUNE_FLATTEN_LAZY, 0)
                try:
                    tree.rebuild_async(aabbs)
                finally:
                    tree.ctx.set_tuning(_lib.TUNE_FLATTEN_LAZY, lazy)
            comm.bcast(tree, root, dtype, n_shapes)
            if attempt > 0:                      # the root's batch was completed by the wait that raised
                return tree, stats, rebroadcasts
            h = tree.traverse_async(rays, hits, flags=flags)
        else:
            tree = comm.bcast(tree, root, dtype, n_shapes)
            h = tree.traverse_async(rays, hits, flags=flags)
        try:
            return tree, h.wait(), rebroadcasts
        except BvhGpuError as e:
            if e.status != REBROADCAST:
                raise
            rebroadcasts += 1
            if rank == root:
                stats = h.wait()                 # (complete already: the wait that raised replayed it on the finished tree)
    raise RuntimeError("the broadcast did not settle after two rebroadcasts")


class LocalCommunicator:
    """bvhgpu_comm over several ctxs of ONE process (bvhgpu_comm_init_all = ncclCommInitAll): ctx i is rank i.  `bcast`
    takes one tree per ctx — the root's is the source, a peer's entry is None or a FlatBvh an earlier bcast returned — and
    returns the list of trees (peers: FlatBvh).  What a C++ / Rust host that drives all GPUs of a node from one thread uses."""

    def __init__(self, ctxs: Sequence):
        lib = _lib.load()
        self.ctxs = list(ctxs)
        arr = (C.c_void_p * len(self.ctxs))(*[c._h for c in self.ctxs])
        h = C.c_void_p()
        check(lib.bvhgpu_comm_init_all(arr, len(self.ctxs), C.byref(h)), self.ctxs[0]._h)
        self._h = h
        self.nranks = len(self.ctxs)
        for c in self.ctxs:
            c._child_add()
        self._counted = True

    def bcast(self, trees: Sequence, root: int = 0, dtype: Optional[str] = None, n_shapes: Optional[int] = None, triangles: bool = False,
              raise_on_error: bool = True):
        from .api import FlatBvh
        lib = _lib.load()
        arr = (C.c_void_p * self.nranks)(*[(t._t if t is not None else None) for t in trees])
        if dtype is not None and n_shapes is not None:
            rc = lib.bvhgpu_bcast_known(self._h, arr, int(root), _lib.F32 if dtype == "f32" else _lib.F64, int(n_shapes),
                                        _lib.BCAST_TRIANGLES if triangles else 0)
        else:
            rc = lib.bvhgpu_bcast(self._h, arr, int(root))
        out = []
        for i, t in enumerate(trees):
            if t is not None or not arr[i]:
                out.append(t)
                continue
            dt = C.c_int()
            h = C.c_void_p(arr[i])
            lib.bvhgpu_tree_info(h, C.byref(dt), None, None, None)
            out.append(FlatBvh(self.ctxs[i], h, "f32" if dt.value == _lib.F32 else "f64"))
        self.last_status = rc
        if rc != _lib.OK and raise_on_error:
            check(rc, self.ctxs[min(max(root, 0), self.nranks - 1)]._h)
        return out

    def info(self) -> dict:
        return comm_info(self._h)

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().bvhgpu_comm_destroy(self._h)
        self._h = None
        if getattr(self, "_counted", False):
            self._counted = False
            for c in self.ctxs:
                c._child_drop()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def broadcast_scene(blob, src: int = 0):
    """Fallback transport: broadcast the scene blob tensor in place with torch.distributed (uint8 tensor on the GPU for
    nccl/RCCL, CPU for gloo)."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src)
        # The receive lands on torch's stream; bvhgpu_scene_import reads the blob on the ctx's stream, which is another one unless the ctx
        # was created on torch's (a ctx made on the DEFAULT stream — handle 0 — gets a stream of its own).  This transport is the
        # host-synchronised fallback anyway: wait for the blob.  (Found by an 8-rank rehearsal on one GPU: a peer imported a blob that
        # had not arrived yet — "not a bvhgpu scene blob".)
        if getattr(blob, "is_cuda", False):
            torch.cuda.current_stream(blob.device).synchronize()
    return blob


def broadcast_nbytes(nbytes: int, device, src: int = 0) -> int:
    """Peers learn the blob size before allocating the receive buffer."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(nbytes)], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src)
    return int(t.item())
