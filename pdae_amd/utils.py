"""Host-side helpers with the reference's semantics (utils/utils.py, trainer/base_trainer.py)."""
import os
import random

import numpy as np
import torch
import yaml


def set_seed(inc, base_seed=666666666):
    """utils/utils.py:30-35: python / numpy / torch / torch.cuda seeded with base+inc (+0,+1,+2,+3)."""
    seed = base_seed + inc
    random.seed(seed)
    np.random.seed(seed + 1)
    torch.manual_seed(seed + 2)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed + 3)


def dispatch_num_samples_for_process(num_samples, num_process, rank):
    """Share of `num_samples` that process `rank` of `num_process` generates: floor(n / w) each, the last rank also takes the n mod w left
    over (the contract of trainer/base_trainer.py:143-153; table-tested in tests/test_ddp_cpu.py)."""
    if not (1 <= num_process <= num_samples and 0 <= rank < num_process):
        raise AssertionError(f"cannot split {num_samples} samples over {num_process} processes for rank {rank}")
    share, left = divmod(num_samples, num_process)
    return share + (left if rank == num_process - 1 else 0)


def load_yaml(filename):
    with open(filename, "r") as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def save_yaml(filename, data):
    with open(filename, "w") as f:
        yaml.dump(data, f, Dumper=yaml.Dumper)


def init_distributed_mode(backend=None):
    """utils/utils.py:18-27, with the backend parametrised (gloo for CPU plumbing tests, nccl == RCCL on ROCm)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or "MASTER_ADDR" in os.environ:
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if not torch.distributed.is_initialized():
            torch.distributed.init_process_group(backend=backend)
        torch.distributed.barrier()
    return rank, world, local_rank
