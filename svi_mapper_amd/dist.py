"""Multi-GPU plumbing for the landmark-sharded BA: one process per GPU,
torch.distributed over RCCL (backend "nccl" on ROCm) carrying the one exchange step of the path —
the sum of the reduced camera normal equations — through the C ABI's all-reduce hook.

The library calls the hook with a device pointer, a count of doubles and the HIP stream it is
working on; the hook wraps the pointer as a torch tensor (no copy) and issues dist.all_reduce with
that stream current, so the collective is stream-ordered between the Schur kernels before it and the
Cholesky after it without any host synchronisation.
"""
import numpy as np


def landmark_shards(edge_lm_slot, n_landmarks, n_ranks):
    """Contiguous landmark-slot ranges balanced by projection-edge count: the partition the library
    applies internally (ba_host.cpp build_structure). Returns bounds[n_ranks + 1]."""
    deg = np.zeros(n_landmarks + 1, np.int64)
    np.add.at(deg, np.asarray(edge_lm_slot, np.int64) + 1, 1)
    cum = np.cumsum(deg)
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, n_ranks):
        bounds.append(int(min(max(np.searchsorted(cum, total * r // n_ranks, side="left"), 0), n_landmarks)))
    bounds.append(n_landmarks)
    return np.asarray(bounds, np.int64)


class _DevPtr:
    """Minimal __cuda_array_interface__ carrier so torch can alias library-owned HBM."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def make_allreduce_hook(group=None):
    """Returns fn(ptr, count, stream) for BundleAdjuster.set_allreduce, summing over `group`."""
    import torch
    import torch.distributed as dist

    cache = {}
    streams = {}
    gloo = dist.get_backend(group) == "gloo"

    def hook(ptr, count, stream):
        key = (ptr, count)
        t = cache.get(key)
        if t is None:
            from . import _dlpack
            t = _dlpack.alias(ptr, (count,), torch.float64, torch.device("cuda", torch.cuda.current_device()))
            cache[key] = t
        ext = streams.get(stream)
        if ext is None:
            ext = streams[stream] = torch.cuda.ExternalStream(stream)
        if gloo:
            # rehearsal path (several ranks sharing one GPU, or no RCCL): through the host, synchronously
            ext.synchronize()
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            with torch.cuda.stream(ext):
                t.copy_(host)
            ext.synchronize()
            return 0
        with torch.cuda.stream(ext):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return 0

    return hook


class NativeRccl:
    """An RCCL communicator owned by libsvi_hot.so (svi_rccl_*): the C++ hook a torchless host would use.  The 128-byte unique
    id of rank 0 reaches the other ranks through `exchange(id_bytes_or_None) -> id_bytes` (here: torch.distributed's object
    broadcast, any side channel will do); after that no Python sits between the library and RCCL."""

    def __init__(self, rank, n_ranks, device, exchange=None, agree=None):
        """agree(ok) -> bool (optional with a custom `exchange`): True iff EVERY rank passed ok=True.  ncclCommInitRank blocks
        until all ranks have entered it, so the ranks first agree that librccl resolved on every one of them (and that rank 0
        got its unique id); if not, all of them raise here - none is left waiting inside the collective initialisation - and
        the caller falls back (bench.py: the torch.distributed hook) on every rank alike."""
        import ctypes as C

        from . import _capi
        lib = _capi.load_library()
        buf = (C.c_char * 128)()
        ok = lib.svi_rccl_available() == 1
        failed = None if ok else "librccl.so could not be loaded on rank %d" % rank
        if ok and rank == 0 and lib.svi_rccl_unique_id(buf) != 0:
            failed = lib.svi_last_error().decode()   # (reported AFTER the exchange: the other ranks are waiting in it)
        raw = bytes(buf)
        if n_ranks > 1:
            if exchange is None:
                import torch.distributed as dist
                votes = [None] * n_ranks
                dist.all_gather_object(votes, failed)
                bad = [v for v in votes if v]
                if bad:
                    raise RuntimeError("native RCCL hook unavailable: %s" % "; ".join(bad))
                box = [raw if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                raw = box[0]
            else:
                if agree is not None and not agree(failed is None):
                    raise RuntimeError("native RCCL hook unavailable: %s" % (failed or "another rank could not load librccl"))
                raw = exchange((None if failed else raw) if rank == 0 else None)
        if failed or raw is None:
            raise RuntimeError("svi_rccl setup failed: %s" % (failed or "see rank 0"))
        idb = (C.c_char * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        _capi.check(lib.svi_rccl_create(idb, int(rank), int(n_ranks), int(device), C.byref(h)), "svi_rccl_create")
        self._lib, self.handle = lib, h

    def close(self):
        if getattr(self, "handle", None):
            self._lib.svi_rccl_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def init_from_env(backend="nccl"):
    """torchrun-style init: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment."""
    import os

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if backend == "nccl":
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local
