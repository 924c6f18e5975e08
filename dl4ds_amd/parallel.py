"""Data parallelism: one process per GPU, gradient averaging by RCCL all-reduces of the flat gradient
arena (issued from the library's side stream, overlapped with the backward pass), parameters + optimiser
slots broadcast from rank 0 once -- the Horovod behaviour of the reference (training/base.py:97-107;
supervised.py:338-369; cgan.py:608-637) re-expressed for RCCL over xGMI.

Bring-up is self-contained (``init_from_env`` is this package's ``hvd.init()``): the 128-byte RCCL unique id
travels from rank 0 to the other ranks over a plain TCP socket on ``MASTER_ADDR``; every later exchange
(barriers, validation-loss averages, the early-stopping decision, max-over-ranks timing) is an RCCL collective
on a few floats.  Any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT works
(``torch.distributed.run``, ``bench.py --gpus N``'s own spawner, srun/mpirun wrappers); torch is not involved.
"""
import ctypes
import hashlib
import hmac
import os
import socket
import struct
import time

from . import _lib

_MAGIC = b'DL4DSRCCL1'


def rank_world_from_env():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))


def allow_unsynced():
    """DL4DS_ALLOW_UNSYNCED as the library reads it (csrc/dist.cpp::dist_expected_world): unset, empty, '0', 'false',
    'no' and 'off' mean NO -- one parser on both sides, so '0' cannot opt out in Python and be refused in C++."""
    v = os.environ.get('DL4DS_ALLOW_UNSYNCED', '')
    return v.strip().lower() not in ('', '0', 'false', 'no', 'off')


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


# ------------------------------------------------------------------------------------------------ rendezvous
def rendezvous_endpoint():
    """(host, port) rank 0 listens on for the id exchange: MASTER_ADDR and DL4DS_RDZV_PORT, by default MASTER_PORT + 1
    (MASTER_PORT itself belongs to the launcher's own store under torch.distributed.run)."""
    host = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = os.environ.get('DL4DS_RDZV_PORT')
    if port is None:
        port = int(os.environ.get('MASTER_PORT', '29500')) + 1
    return host, int(port)


def _token_digest():
    return hashlib.sha256(os.environ.get('DL4DS_RDZV_TOKEN', '').encode()).digest()


def _recv_exact(conn, n):
    buf = b''
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    return buf


def exchange_bytes(payload, rank, world, timeout=300.0, endpoint=None):
    """Rank 0 hands `payload` (bytes) to every other rank; returns the payload on all ranks.  Rank 0 returns once all
    world-1 peers have fetched it, so the call also orders the ranks (nobody proceeds before everyone arrived).
    DL4DS_RDZV_TOKEN (optional, the same on all ranks of a job) must match: a guard against two jobs sharing a port by
    accident.  What travels is a fixed 32-byte SHA-256 of it (of the empty string when unset), so a rank WITH a token never
    passes a rank 0 without one or vice versa, prefixes do not match, the hello has one length, and the comparison is
    constant-time."""
    if world <= 1:
        return payload
    host, port = endpoint or rendezvous_endpoint()
    deadline = time.time() + timeout
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            srv.bind((host, port))
        except OSError as e:
            srv.close()
            raise RuntimeError(f'dl4ds_amd.parallel: rank 0 cannot listen on {host}:{port} ({e}); set DL4DS_RDZV_PORT '
                               'to a free port (the same on all ranks)') from e
        srv.listen(world)
        seen = set()
        rejected = []                                 # (peer address, reason): reported if the deadline passes
        token = _token_digest()
        try:
            while len(seen) < world - 1:
                srv.settimeout(max(deadline - time.time(), 0.01))
                try:
                    conn, addr = srv.accept()
                except socket.timeout:
                    raise TimeoutError(f'dl4ds_amd.parallel: only {len(seen) + 1} of {world} ranks reached the rendezvous '
                                       f'on {host}:{port} within {timeout:.0f} s'
                                       + (f'; rejected peers: {rejected[-4:]}' if rejected else '')) from None
                # one bad peer (a stale client of an earlier job on this port, a port scanner, a rank of another job, a
                # connection that dies half-way) must not take rank 0 down: reject it, keep serving, fail at the deadline
                try:
                    with conn:
                        conn.settimeout(5.0)             # (the accept loop is serial: a silent client may hold it this long)
                        hello = _recv_exact(conn, len(_MAGIC) + 8 + 32)
                        if hello[:len(_MAGIC)] != _MAGIC:
                            continue
                        r, w = struct.unpack('<ii', hello[len(_MAGIC):len(_MAGIC) + 8])
                        token_ok = hmac.compare_digest(hello[len(_MAGIC) + 8:], token)
                        if w != world or not (0 < r < world) or r in seen or not token_ok:
                            rejected.append((addr[0], f'says rank {r} of {w}' + ('' if token_ok else ', wrong token')))
                            conn.sendall(struct.pack('<i', -1))
                            continue
                        conn.sendall(struct.pack('<i', len(payload)) + payload)
                        _recv_exact(conn, 1)              # ack: the peer holds the payload
                        seen.add(r)
                except (ConnectionError, socket.timeout, OSError) as e:
                    rejected.append((addr[0], repr(e)))
                    continue
        finally:
            srv.close()
        return payload
    last = None
    while time.time() < deadline:
        try:
            with socket.create_connection((host, port), timeout=5.0) as conn:
                conn.settimeout(max(deadline - time.time(), 1.0))
                conn.sendall(_MAGIC + struct.pack('<ii', rank, world) + _token_digest())
                n = struct.unpack('<i', _recv_exact(conn, 4))[0]
                if n < 0:
                    raise RuntimeError('dl4ds_amd.parallel: rank 0 rejected this rank (rank / world / DL4DS_RDZV_TOKEN '
                                       'mismatch: is another job using this MASTER_ADDR / port?)')
                data = _recv_exact(conn, n)
                conn.sendall(b'\x01')
                return data
        except (ConnectionRefusedError, ConnectionResetError, socket.timeout, ConnectionError, OSError) as e:
            last = e                                   # rank 0 is not listening yet
            time.sleep(0.05)
    raise TimeoutError(f'dl4ds_amd.parallel: rank {rank} could not reach rank 0 on {host}:{port} within {timeout:.0f} s '
                       f'({last!r})')


def is_initialized():
    n = ctypes.c_int()
    r = ctypes.c_int()
    d = ctypes.c_int()
    _lib.check(_lib.lib().dl4ds_dist_comm_info(ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)))
    return n.value > 0


def comm_info():
    """What RCCL reports: {'nranks', 'rank', 'device'} (nranks 0 without a communicator)."""
    n, r, d = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib().dl4ds_dist_comm_info(ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)))
    return {'nranks': n.value, 'rank': r.value, 'device': d.value}


def init_from_env(timeout=300.0):
    """hvd.init() + set_visible_gpus(hvd.local_rank()) (training/base.py:97-107): bind LOCAL_RANK's GPU, create the RCCL
    communicator over all WORLD_SIZE ranks.  Idempotent; a no-op for a single process.  Returns (rank, world)."""
    rank, world_size, _ = rank_world_from_env()
    _lib.lib()                                         # LOCAL_RANK -> device (raises without a GPU)
    if world_size <= 1:
        return rank, 1
    if is_initialized():
        r, w = world()
        if (r, w) != (rank, world_size):
            raise RuntimeError(f'dl4ds_amd.parallel: communicator is rank {r}/{w} but the environment says '
                               f'{rank}/{world_size}')
        return rank, world_size
    payload = unique_id() if rank == 0 else b''
    init_with_id(rank, world_size, exchange_bytes(payload, rank, world_size, timeout))
    info = comm_info()
    if info['nranks'] != world_size or info['rank'] != rank:
        raise RuntimeError(f'dl4ds_amd.parallel: RCCL reports {info}, expected rank {rank} of {world_size}')
    return rank, world_size


def init_from_torch_distributed(dist, rank, world):
    """Exchange the RCCL unique id over an existing torch.distributed group (for callers that already have one)."""
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


def allreduce_host(values, op='sum'):
    """Reduce a short list of Python floats across the ranks (RCCL, fp32): 'sum' | 'max' | 'min' | 'mean'."""
    vals = [float(v) for v in values]
    arr = (ctypes.c_float * len(vals))(*vals)
    code = {'sum': 0, 'mean': 0, 'max': 1, 'min': 2}[op]
    _lib.check(_lib.lib().dl4ds_dist_allreduce_host(arr, len(vals), code))
    out = [float(v) for v in arr]
    if op == 'mean':
        w = world()[1]
        out = [v / w for v in out]
    return out


def barrier():
    _lib.check(_lib.lib().dl4ds_dist_barrier())


def broadcast_trainer(engine, root=0):
    """hvd.callbacks.BroadcastGlobalVariablesCallback(0) / hvd.broadcast_variables: params + Adam m, v, iterations."""
    _lib.check(_lib.lib().dl4ds_dist_broadcast_trainer(engine.h, int(root)))


def finalize():
    _lib.check(_lib.lib().dl4ds_dist_finalize())


def shard_indices(n_samples, rank, world, seed, epoch=0):
    """Rank-strided slice of one seeded permutation (the reference lets every rank shuffle independently,
    dataloader.py:463; a shared seeded permutation is the reproducible equivalent).  Every rank gets the SAME number
    of samples (the remainder n % world is dropped for this epoch): ranks with unequal step counts would leave the
    others waiting in the gradient all-reduce."""
    import numpy as np
    perm = np.random.default_rng(seed + epoch).permutation(n_samples)
    return equal_shard(perm, rank, world)


def equal_shard(perm, rank, world):
    if world <= 1:
        return perm
    n = (len(perm) // world) * world
    return perm[:n][rank::world]
