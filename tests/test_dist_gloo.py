"""world_size=2 gloo (CPU) test of the multi-process control plane used by bench.py / the trainers:
RCCL unique-id exchange, barrier + max-over-ranks timing, identical sharding decisions.  The data plane
(ncclAllReduce on the gradient arena) needs GPUs and is exercised by the driver's multi-GPU bench."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dl4ds_amd import parallel
    got = {}
    # stand-ins for the two RCCL calls (no GPU here); everything else is the production code path
    parallel.unique_id = lambda: bytes(range(128))
    parallel.init_with_id = lambda r, w, idb: got.update(rank=r, world=w, id=bytes(idb))
    parallel.init_from_torch_distributed(dist, rank, world)
    # max-over-ranks timing as in bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    # emulate the gradient average that the RCCL all-reduce + Adam(1/world) performs
    g = torch.full((8,), float(rank + 1))
    dist.all_reduce(g)
    idx = parallel.shard_indices(10, rank, world, seed=3)
    q.put((rank, got, float(t[0]), (g / world).tolist(), idx.tolist(), parallel.rank_world_from_env()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_control_plane():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ids = [r[1]['id'] for r in res]
    assert ids[0] == ids[1] == bytes(range(128))
    assert [r[1]['rank'] for r in res] == [0, 1] and all(r[1]['world'] == 2 for r in res)
    assert all(r[2] == 2.0 for r in res)                       # MAX over ranks
    assert all(r[3] == [1.5] * 8 for r in res)                 # average of per-rank gradients
    assert sorted(res[0][4] + res[1][4]) == list(range(10))    # disjoint shards covering the data
    assert res[1][5] == (1, 2, 1)
