"""Data parallelism: one process per GPU, gradient averaging by ONE RCCL all-reduce of the flat gradient
arena per step (issued from the library's side stream), parameters + optimiser slots broadcast from rank 0
once -- the Horovod behaviour of the reference (training/base.py:97-107; supervised.py:338-369;
cgan.py:608-637) re-expressed for RCCL over xGMI.

The control plane (rank discovery, barrier, exchanging the 128-byte RCCL unique id) rides on whatever
launcher started the processes; with ``torch.distributed.run`` that is a gloo (CPU) process group.
"""
import ctypes
import os

from . import _lib


def rank_world_from_env():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def init_with_id(rank, world, id_bytes):
    assert len(id_bytes) == 128
    buf = ctypes.create_string_buffer(bytes(id_bytes), 128)
    lib = _lib.lib()
    # RCCL prints a banner ("Hostname : ...", "Librccl path : ...") on stdout at communicator creation; keep stdout
    # clean for callers that print machine-readable results (bench.py's single JSON line): send it to stderr.
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        st = lib.dl4ds_dist_init(int(rank), int(world), buf)
        ctypes.CDLL(None).fflush(None)        # the banner sits in C stdio's buffer: push it out while fd 1 is stderr
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    _lib.check(st)


def unique_id():
    buf = ctypes.create_string_buffer(128)
    _lib.check(_lib.lib().dl4ds_dist_unique_id(buf))
    return buf.raw


def init_from_torch_distributed(dist, rank, world):
    """Exchange the RCCL unique id over an existing (gloo) torch.distributed group, then ncclCommInitRank."""
    import torch
    t = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        t = torch.frombuffer(bytearray(unique_id()), dtype=torch.uint8).clone()
    dist.broadcast(t, src=0)
    init_with_id(rank, world, bytes(t.numpy().tobytes()))


def world():
    r, w = ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib().dl4ds_dist_world(ctypes.byref(r), ctypes.byref(w)))
    return r.value, w.value


def broadcast_trainer(engine, root=0):
    """hvd.callbacks.BroadcastGlobalVariablesCallback(0) / hvd.broadcast_variables: params + Adam m, v."""
    _lib.check(_lib.lib().dl4ds_dist_broadcast_trainer(engine.h, int(root)))


def finalize():
    _lib.check(_lib.lib().dl4ds_dist_finalize())


def shard_indices(n_samples, rank, world, seed, epoch=0):
    """Rank-strided slice of one seeded permutation (the reference lets every rank shuffle independently,
    dataloader.py:463; a shared seeded permutation is the reproducible equivalent)."""
    import numpy as np
    perm = np.random.default_rng(seed + epoch).permutation(n_samples)
    return perm[rank::world]
