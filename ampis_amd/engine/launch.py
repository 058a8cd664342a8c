"""detectron2.engine.launch: one process per GPU on this machine.  AMPIS itself never calls it (its notebooks are single-GPU);
it is what a user of DefaultTrainer / AmpisTrainer (ampis/data_utils.py:135) reaches for to train on several GPUs.

The workers are spawned (never forked) before the parent has touched the GPU; each gets RANK / LOCAL_RANK / WORLD_SIZE and a
torch.distributed process group on gloo -- the side channel that carries the RCCL id (utils/comm.attach_rccl); the device
collectives themselves are RCCL calls made by libampis_hip.so."""
import os
import socket


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(local_rank, main_func, world, dist_url, args):
    import torch.distributed as dist
    os.environ["RANK"] = os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ["WORLD_SIZE"] = str(world)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", init_method=dist_url, rank=local_rank, world_size=world)
    try:
        main_func(*args)
    finally:
        from ..utils import comm
        comm.detach_rccl()
        dist.destroy_process_group()


def launch(main_func, num_gpus_per_machine, num_machines=1, machine_rank=0, dist_url=None, args=()):
    """launch(main_func, num_gpus_per_machine, args=(...)): run main_func(*args) in num_gpus_per_machine processes (one node)."""
    if num_machines != 1 or machine_rank != 0:
        raise NotImplementedError("ampis_amd.engine.launch: one node (8 GPUs over xGMI) is the supported topology")
    world = int(num_gpus_per_machine)
    if world <= 1:
        return main_func(*args)
    if dist_url in (None, "auto"):
        dist_url = f"tcp://127.0.0.1:{_free_port()}"
    import torch.multiprocessing as mp
    mp.start_processes(_worker, nprocs=world, args=(main_func, world, dist_url, tuple(args)), start_method="spawn")
