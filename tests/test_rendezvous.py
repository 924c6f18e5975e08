"""CPU tests of the multi-process control plane that needs no GPU: the TCP hand-over of the 128-byte RCCL id
(dl4ds_amd.parallel.exchange_bytes -- the id exchange of init_from_env, this package's hvd.init()), equal-length
rank shards, and bench.py's own rank launcher.  Two real processes, world_size 2 (and 3)."""
import multiprocessing as mp
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rdzv_worker(rank, world, port, q, delay):
    import time
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    os.environ.pop('DL4DS_RDZV_PORT', None)
    from dl4ds_amd import parallel
    time.sleep(delay)                                   # rank 0 late / early: both orders must work
    assert parallel.rendezvous_endpoint() == ('127.0.0.1', port + 1)
    payload = bytes(range(128)) if rank == 0 else b''
    got = parallel.exchange_bytes(payload, rank, world, timeout=60.0)
    q.put((rank, got, parallel.rank_world_from_env()))


@pytest.mark.parametrize('world,late_root', [(2, False), (2, True), (3, False)])
def test_tcp_id_exchange(world, late_root):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rdzv_worker, args=(r, world, port, q, (0.5 if (r == 0) == late_root else 0.0)))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    assert all(r[1] == bytes(range(128)) for r in res)
    assert [r[2] for r in res] == [(i, world, i) for i in range(world)]


def test_rendezvous_times_out_when_a_rank_is_missing():
    from dl4ds_amd import parallel
    port = _free_port()
    with pytest.raises(TimeoutError, match='only 1 of 2 ranks'):
        parallel.exchange_bytes(b'x' * 128, 0, 2, timeout=0.5, endpoint=('127.0.0.1', port))
    with pytest.raises(TimeoutError, match='could not reach rank 0'):
        parallel.exchange_bytes(b'', 1, 2, timeout=0.5, endpoint=('127.0.0.1', port))


def test_rendezvous_port_in_use_is_reported():
    from dl4ds_amd import parallel
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    s.listen(1)
    try:
        with pytest.raises(RuntimeError, match='DL4DS_RDZV_PORT'):
            parallel.exchange_bytes(b'x' * 128, 0, 2, timeout=1.0, endpoint=('127.0.0.1', s.getsockname()[1]))
    finally:
        s.close()


def test_equal_shards():
    from dl4ds_amd import parallel
    from dl4ds_amd.dataloader import equal_shard
    for n in (10, 11, 17, 64):
        for world in (1, 2, 3, 8):
            perm = np.random.default_rng(n).permutation(n)
            shards = [equal_shard(perm, r, world) for r in range(world)]
            assert len({len(s) for s in shards}) == 1                     # same step count on every rank
            allv = np.concatenate(shards)
            assert len(set(allv.tolist())) == len(allv) == (n // world) * world
            for r in range(world):
                np.testing.assert_array_equal(parallel.equal_shard(perm, r, world), shards[r])
            a = parallel.shard_indices(n, 0, world, seed=3, epoch=1)
            assert len(a) == n // world


def test_data_generators_have_equal_length_on_every_rank():
    from dl4ds_amd.dataloader import DataGenerator
    data = np.random.default_rng(0).random((23, 8, 8, 1)).astype(np.float32)
    lens, seen = [], []
    for r in range(3):
        g = DataGenerator(data, None, 'resnet', 'spc', 2, batch_size=2, seed=1, rank=r, world=3)
        lens.append(len(g))
        seen += g.indices.tolist()
    assert lens == [3, 3, 3] and len(set(seen)) == 21


def test_bench_launcher_refuses_more_ranks_than_devices():
    """`python bench.py --gpus 2` invoked directly (no launcher): it must start the ranks itself; in this container
    there is no HIP device, so it has to stop with a clear message instead of an argparse / launcher error."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip('two GPUs visible: covered by the gpu tests')
    assert r.returncode != 0
    assert 'HIP device(s) visible' in (r.stderr + r.stdout) or 'no HIP device' in (r.stderr + r.stdout)


def test_bench_port_pair_is_free():
    sys.path.insert(0, ROOT)
    import bench
    p = bench._free_port_pair()
    for port in (p, p + 1):
        s = socket.socket()
        s.bind(('127.0.0.1', port))
        s.close()


def test_bad_peers_do_not_take_rank_zero_down(monkeypatch):
    """A stale client of another job (wrong world size), a peer that hangs up in the middle of the hello and a port scanner
    reach rank 0 before the real rank 1 does: rank 0 rejects them, keeps serving and completes the exchange."""
    import struct
    import threading
    import time
    from dl4ds_amd import parallel
    port = _free_port()
    ep = ('127.0.0.1', port)
    out = {}
    t = threading.Thread(target=lambda: out.setdefault('root', parallel.exchange_bytes(b'I' * 128, 0, 2, timeout=30.0, endpoint=ep)))
    t.start()

    def connect():
        for _ in range(200):
            try:
                return socket.create_connection(ep, timeout=2.0)
            except OSError:
                time.sleep(0.02)
        raise AssertionError('rank 0 never listened')
    with connect() as c:                                     # another job's rank: "rank 1 of 4"
        c.sendall(parallel._MAGIC + struct.pack('<ii', 1, 4) + parallel._token_digest())
        assert struct.unpack('<i', parallel._recv_exact(c, 4))[0] == -1
    with connect() as c:                                     # dies half-way through the hello
        c.sendall(parallel._MAGIC[:4])
    with connect() as c:                                     # not this protocol at all
        c.sendall(b'GET / HTTP/1.0\r\n\r\n' + b' ' * 48)
    got = parallel.exchange_bytes(b'', 1, 2, timeout=30.0, endpoint=ep)
    t.join(timeout=30)
    assert got == b'I' * 128 and out['root'] == b'I' * 128


def test_rendezvous_token_must_match(monkeypatch):
    import threading
    from dl4ds_amd import parallel
    port = _free_port()
    ep = ('127.0.0.1', port)
    monkeypatch.setenv('DL4DS_RDZV_TOKEN', 'job-1234')
    err = {}

    def root():
        try:
            parallel.exchange_bytes(b'I' * 128, 0, 2, timeout=3.0, endpoint=ep)
        except TimeoutError as e:
            err['root'] = str(e)
    t = threading.Thread(target=root)
    t.start()
    import struct, time
    for _ in range(200):
        try:
            c = socket.create_connection(ep, timeout=2.0)
            break
        except OSError:
            time.sleep(0.02)
    with c:                                                  # right rank and world, wrong secret: no id for this peer
        import hashlib
        c.sendall(parallel._MAGIC + struct.pack('<ii', 1, 2) + hashlib.sha256(b'job-9999').digest())
        assert struct.unpack('<i', parallel._recv_exact(c, 4))[0] == -1
    t.join(timeout=30)
    assert 'wrong token' in err['root']


@pytest.mark.parametrize('root_token,peer_token', [(None, 'job-1'), ('job-1', None), ('job-1', 'job-12'), ('job-12', 'job-1')])
def test_rendezvous_token_is_symmetric_and_not_a_prefix_match(monkeypatch, root_token, peer_token):
    """A rank 0 without a token must not accept a peer that sends one (and vice versa), and a token that merely starts
    with rank 0's does not pass: what is compared is a fixed-size digest, present in every hello."""
    import hashlib
    import struct
    import threading
    import time
    from dl4ds_amd import parallel
    ep = ('127.0.0.1', _free_port())
    if root_token is None:
        monkeypatch.delenv('DL4DS_RDZV_TOKEN', raising=False)
    else:
        monkeypatch.setenv('DL4DS_RDZV_TOKEN', root_token)
    err = {}

    def root():
        try:
            parallel.exchange_bytes(b'I' * 128, 0, 2, timeout=3.0, endpoint=ep)
        except TimeoutError as e:
            err['root'] = str(e)
    t = threading.Thread(target=root)
    t.start()
    for _ in range(200):
        try:
            c = socket.create_connection(ep, timeout=2.0)
            break
        except OSError:
            time.sleep(0.02)
    with c:
        c.sendall(parallel._MAGIC + struct.pack('<ii', 1, 2) + hashlib.sha256((peer_token or '').encode()).digest())
        assert struct.unpack('<i', parallel._recv_exact(c, 4))[0] == -1
    t.join(timeout=30)
    assert 'wrong token' in err['root']


def test_bench_spawns_eight_ranks_with_the_launcher_environment_and_propagates_failure(tmp_path):
    """bench.spawn_ranks for N = 8 with a stand-in child (no GPU here): every child sees RANK = LOCAL_RANK = i, WORLD_SIZE = 8,
    one MASTER_ADDR / MASTER_PORT for all, HSA_ENABLE_IPC_MODE_LEGACY=0; a failing rank's code is what the launcher returns
    and the surviving ranks are stopped instead of waiting in a collective for ever."""
    import json
    import time
    sys.path.insert(0, ROOT)
    import bench
    child = tmp_path / 'child.py'
    child.write_text(
        'import json, os, sys, time\n'
        'r = int(os.environ["RANK"])\n'
        'keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")\n'
        'json.dump({k: os.environ.get(k) for k in keys} | {"argv": sys.argv[1:]}, open(os.path.join(sys.argv[1], f"env{r}.json"), "w"))\n'
        'if len(sys.argv) > 2 and sys.argv[2] == "fail" and r == 5:\n'
        '    time.sleep(0.3); sys.exit(7)\n'
        'if len(sys.argv) > 2 and sys.argv[2] == "fail":\n'
        '    time.sleep(60)\n')
    assert bench.spawn_ranks(8, [str(tmp_path)], script=str(child), device_count=8) == 0
    envs = [json.load(open(tmp_path / f'env{r}.json')) for r in range(8)]
    for r, e in enumerate(envs):
        assert e['RANK'] == e['LOCAL_RANK'] == str(r) and e['WORLD_SIZE'] == e['LOCAL_WORLD_SIZE'] == '8'
        assert e['MASTER_ADDR'] == '127.0.0.1' and e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert len({e['MASTER_PORT'] for e in envs}) == 1
    t0 = time.time()
    assert bench.spawn_ranks(8, [str(tmp_path), 'fail'], script=str(child), device_count=8) == 7
    assert time.time() - t0 < 30                              # the other seven were terminated, not waited for
    with pytest.raises(SystemExit, match='only 4 HIP device'):
        bench.spawn_ranks(8, [], script=str(child), device_count=4)


@pytest.mark.parametrize('value,expect', [(None, False), ('', False), ('0', False), ('false', False), ('No', False), ('off', False),
                                          ('1', True), ('yes', True), ('true', True)])
def test_allow_unsynced_reads_the_variable_like_the_library(monkeypatch, value, expect):
    from dl4ds_amd import parallel
    if value is None:
        monkeypatch.delenv('DL4DS_ALLOW_UNSYNCED', raising=False)
    else:
        monkeypatch.setenv('DL4DS_ALLOW_UNSYNCED', value)
    assert parallel.allow_unsynced() is expect
