"""One process per GPU under `python -m torch.distributed.run` (BASELINE configs 4 and 5): process-group and device set-up
for the command-line entry points.  Everything here runs BEFORE the first GPU call of the process - the device is chosen from
LOCAL_RANK first and the RCCL communicator is bound to it (`device_id=`), never the other way round, and nothing re-execs.

The reference has no multi-GPU code (its sweeps are one job per GPU, slurm_bc.py:123); these entry points keep its
single-process behaviour when RANK / WORLD_SIZE are absent."""
import os

import torch


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them).  Returns
    (rank, local_rank, world).  world == 1: nothing is initialised.  backend: 'nccl' (= RCCL over xGMI on ROCm; default when a GPU
    is visible), 'gloo' (CPU tests; also $PVR_DIST_BACKEND)."""
    rank, local_rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    n_dev = torch.cuda.device_count()                       # counting devices does not initialise the GPU runtime
    if n_dev > 0:
        # PVR_ONE_GPU=1: every rank on cuda:0 (a test aid for 1-GPU boxes; RCCL refuses two ranks on one device, so gloo)
        one_gpu = os.environ.get('PVR_ONE_GPU', '0') == '1'
        torch.cuda.set_device(0 if one_gpu else local_rank % n_dev)
    if world <= 1:
        return rank, local_rank, 1
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), local_rank, dist.get_world_size()
    backend = backend or os.environ.get('PVR_DIST_BACKEND') or ('nccl' if n_dev > 0 and os.environ.get('PVR_ONE_GPU', '0') != '1' else 'gloo')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', torch.cuda.current_device()))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def finalize_distributed():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1
