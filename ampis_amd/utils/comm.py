"""detectron2.utils.comm members AMPIS uses: synchronize() = barrier (ampis/data_utils.py:27,107); one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU) when initialised, no-op otherwise.  Also the
data-parallel gradient exchange of the training path: one SUM all-reduce over the flat fp32 gradient arena
(replaces DDP's bucketed NCCL all-reduce; 43.7 M floats = 175 MB per step)."""
import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main_process():
    return get_rank() == 0


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


class _DeviceArray:
    """Zero-copy view of raw device memory for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def arena_as_tensor(ptr, n, device):
    return torch.as_tensor(_DeviceArray(ptr, n), device=device)


def all_reduce_sum_(t):
    """In-place SUM all-reduce (no-op for a single process). Returns the factor that turns the sum into the mean."""
    ws = get_world_size()
    if ws > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return 1.0 / ws


def all_reduce_gradients(model, ctx):
    """Sum the gradient arena of `model` over all ranks; returns the 1/world factor the SGD step applies (DDP averages)."""
    ws = get_world_size()
    if ws == 1:
        return 1.0
    ptr, n = model.grad_arena()
    ctx.sync()                                   # our stream produced the gradients
    t = arena_as_tensor(ptr, n, torch.device("cuda", ctx.device))
    scale = all_reduce_sum_(t)
    torch.cuda.synchronize(ctx.device)           # RCCL ran on torch's stream
    return scale
