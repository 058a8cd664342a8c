"""detectron2.utils.comm members AMPIS uses: synchronize() = barrier (ampis/data_utils.py:27,107); one process per GPU,
torch.distributed over RCCL when initialised, no-op otherwise."""
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
