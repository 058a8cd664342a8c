"""detectron2.utils.comm members AMPIS uses — synchronize() = barrier (ampis/data_utils.py:27,107) — and the data-parallel
gradient exchange of the training path (what DefaultTrainer gets from DDP + NCCL under
ampis/data_utils.py:135).

One process per GPU.  The device collectives are RCCL calls made by libampis_hip.so itself (include/ampis_hip.h "Multi-GPU
exchange": amp_comm_init / amp_barrier / amp_allreduce, the bucketed gradient all-reduce issued from inside
amp_model_forward_backward); torch.distributed is only the side channel that carries the 128-byte RCCL id to the ranks (any
backend, normally gloo) and the CPU-side world / rank bookkeeping.  The gradient arena goes out in AMP_GRAD_BUCKETS buckets in
the order the backward pass completes them (mask head, box head, RPN, FPN, res5, res4, res3) so that the exchange overlaps
the remaining weight-gradient kernels.

`AMP_COMM_BACKEND=staged` (rehearsal on a one-GPU box, where RCCL refuses two ranks on one device, and the CPU tests) drives
the SAME bucket plan through torch.distributed on host staging buffers instead; it is never selected implicitly.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from .. import _lib

_rccl_ctx = None        # the context that carries this process's RCCL communicator (attach_rccl)


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main_process():
    return get_rank() == 0


def backend():
    """'rccl' (the C-ABI communicator) or 'staged' (torch.distributed on host buffers: rehearsal / CPU tests only)."""
    return "staged" if os.environ.get("AMP_COMM_BACKEND", "rccl") == "staged" else "rccl"


def attach_rccl(ctx):
    """Create this rank's RCCL communicator on `ctx` (world / rank from torch.distributed, which also carries the id)."""
    global _rccl_ctx
    world, rank = get_world_size(), get_rank()
    ids = [_lib.Context.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(ids, src=0)
    ctx.comm_init(rank, world, ids[0])
    _rccl_ctx = ctx
    return ctx.comm_info()


def detach_rccl():
    global _rccl_ctx
    if _rccl_ctx is not None:
        _rccl_ctx.comm_destroy()
        _rccl_ctx = None


def synchronize():
    """Barrier over all ranks (detectron2.utils.comm.synchronize).  On the device communicator when there is one."""
    if get_world_size() == 1:
        return
    if _rccl_ctx is not None:
        _rccl_ctx.barrier()
    else:
        dist.barrier()


class _DeviceArray:
    """Zero-copy view of raw device memory for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def arena_as_tensor(ptr, n, device):
    """torch view of `n` floats of device memory at `ptr` (tests and tools; the exchange itself never goes through torch)."""
    return torch.as_tensor(_DeviceArray(ptr, n), device=device)


# ---- the bucket plan (host only; shared by the library and by the staged backend) ----

def plan_buckets(names, offsets, sizes, max_gap=64):
    """[(bucket, offset, n)] merged float ranges in exchange order for tensors `names` at `offsets` (floats) of `sizes`."""
    L = _lib.lib()
    nt = len(names)
    b = (C.c_int * nt)(*[L.amp_grad_bucket_of(str(n).encode()) for n in names])
    o = (C.c_size_t * nt)(*[int(x) for x in offsets])
    n = (C.c_size_t * nt)(*[int(x) for x in sizes])
    cap = 64 * 7
    ob, oo, on, cnt = (C.c_int * cap)(), (C.c_size_t * cap)(), (C.c_size_t * cap)(), C.c_int()
    _lib.check(L.amp_plan_grad_buckets(nt, b, o, n, int(max_gap), cap, ob, oo, on, C.byref(cnt)), "amp_plan_grad_buckets")
    return [(ob[i], oo[i], on[i]) for i in range(cnt.value)]


def arena_layout(shapes, align=64):
    """Offsets (floats) of tensors laid out back to back in dict order, each aligned to `align` floats like the library's
    parameter / gradient arena.  Returns (names, offsets, sizes, total)."""
    names, offs, sizes, off = [], [], [], 0
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        off = (off + align - 1) // align * align
        names.append(k); offs.append(off); sizes.append(n)
        off += n
    return names, offs, sizes, off


def all_reduce_buckets_(flat, plan):
    """In-place SUM all-reduce of a flat CPU tensor, bucket by bucket in plan order (async, joined at the end): the staged
    twin of what RCCL does on the device arena.  Returns the 1/world factor that turns the sum into DDP's mean."""
    ws = get_world_size()
    if ws > 1:
        work = [dist.all_reduce(flat[o:o + n], op=dist.ReduceOp.SUM, async_op=True) for _, o, n in plan]
        for w in work:
            w.wait()
    return 1.0 / ws


def all_reduce_sum_(t):
    """In-place SUM all-reduce of a CPU tensor (no-op for a single process); returns 1/world."""
    ws = get_world_size()
    if ws > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return 1.0 / ws


def all_reduce_gradients(model, ctx):
    """Make the gradient arena of `model` the SUM over all ranks; returns the 1/world factor the SGD step applies (DDP
    averages).  With the RCCL communicator the buckets were already handed to RCCL from inside amp_model_forward_backward
    (they overlap the backward pass; amp_model_sgd_step waits for them on the device); with the overlap switched off
    (`set_grad_overlap(False)`) they are handed over here, all at once."""
    ws = get_world_size()
    if ws == 1 and ctx.comm_info()[1] == 0:
        return 1.0
    if ctx.comm_info()[1] > 0:
        if not model.grads_exchanged():          # set_grad_overlap(False): nothing was exchanged inside forward_backward
            model.allreduce_grads()
        return 1.0 / ctx.comm_info()[1]
    if backend() != "staged":
        raise _lib.AmpError("world size > 1 but the context has no RCCL communicator: call comm.attach_rccl(ctx) "
                            "(or set AMP_COMM_BACKEND=staged for a rehearsal without RCCL)")
    ptr, _ = model.grad_arena()
    ctx.sync()
    for _, o, n in model.grad_buckets():
        host = np.empty(n, dtype=np.float32)
        ctx.d2h(host, ptr + 4 * o)
        dist.all_reduce(torch.from_numpy(host), op=dist.ReduceOp.SUM)
        ctx.h2d(ptr + 4 * o, host)
    return 1.0 / ws
