"""Two ranks through SupervisedTrainer.run() on CPU: the HOST-side collective sequence of the data-parallel trainers
(parameter broadcast, the rank-averaged train / validation / test losses, the early-stopping decision they feed) must be
the same on every rank -- a rank that enters a different collective, or leaves the epoch loop at another epoch, strands
the others inside RCCL, which shows up as a hang, not as an error.  The GPU pieces (model builders, the engine, RCCL) are
replaced by stand-ins; the trainer code, the host data generator and the sharding are the production code.  The stand-in
bus checks that both ranks enter the SAME collective at every step and fails (instead of hanging) when one does not
arrive."""
import threading

import numpy as np
import pytest


class Bus:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=30)
        self.slots = [None] * world
        self.log = [[] for _ in range(world)]

    def collective(self, rank, tag, payload):
        self.slots[rank] = (tag, payload)
        self.log[rank].append(tag)
        self.barrier.wait()
        tags = {s[0] for s in self.slots}
        vals = [s[1] for s in self.slots]
        self.barrier.wait()
        assert len(tags) == 1, f'ranks entered different collectives: {[s[0] for s in self.slots]}'
        return vals


def _run_two_ranks(monkeypatch, tmp_path, **trainer_kw):
    import dl4ds_amd.parallel as parallel
    import dl4ds_amd.training.supervised as sup
    import dl4ds_amd.training.base as base
    bus = Bus(2)
    tl = threading.local()
    monkeypatch.setattr(parallel, 'rank_world_from_env', lambda: (tl.rank, 2, tl.rank))
    monkeypatch.setattr(parallel, 'allow_unsynced', lambda: False)
    monkeypatch.setattr(parallel, 'init_from_env', lambda timeout=300.0: (tl.rank, 2))

    def allreduce_host(values, op='sum'):
        vals = bus.collective(tl.rank, f'allreduce_host[{len(values)}]:{op}', [float(v) for v in values])
        red = {'sum': np.sum, 'mean': np.mean, 'max': np.max, 'min': np.min}[op]
        return [float(red([v[i] for v in vals])) for i in range(len(values))]
    monkeypatch.setattr(parallel, 'allreduce_host', allreduce_host)
    monkeypatch.setattr(parallel, 'broadcast_trainer', lambda eng, root=0: bus.collective(tl.rank, 'broadcast_trainer', None))
    monkeypatch.setattr(parallel, 'barrier', lambda: bus.collective(tl.rank, 'barrier', None))

    class StubModel:
        name = 'stub'

        def get_weights(self):
            return {'w': np.zeros(1, np.float32)}

        def summary(self, **kw):
            pass

    class StubEngine:
        """Losses that differ between the ranks: on its OWN validation loss rank 1 would stop two epochs before rank 0."""
        def __init__(self, model, **kw):
            self.n = 0

        def step(self, x, y):
            self.n += 1
            return 1.0 / self.n + 0.1 * tl.rank

        def evaluate(self, x, y):
            return (1.0 / self.n if tl.rank == 0 else 0.5) + 0.01 * tl.rank

        def load_checkpoint(self, p):
            pass

        def save_checkpoint(self, p):
            pass
    monkeypatch.setattr(sup, 'SupervisedEngine', StubEngine)
    for b in ('net_postupsampling', 'net_pin', 'unet_pin', 'recnet_postupsampling', 'recnet_pin'):
        monkeypatch.setattr(sup.M, b, lambda **kw: StubModel())
    rng = np.random.default_rng(0)
    fields = lambda n: rng.random((n, 16, 16, 1)).astype(np.float32)
    data = fields(16), fields(8), fields(8)
    out, errs = {}, []

    def worker(rank):
        tl.rank = rank
        try:
            t = sup.SupervisedTrainer('resnet', 'spc', *data, scale=2, batch_size=2, device_data=False, verbose=False,
                                      save=rank == 0 and trainer_kw.pop('save', False), save_path=str(tmp_path), **trainer_kw)
            t.run()
            out[rank] = t
        except BaseException as e:          # noqa: BLE001 -- reported below, and the other rank's barrier breaks
            errs.append((rank, repr(e)))
            bus.barrier.abort()
    th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errs, errs
    assert not any(t.is_alive() for t in th), 'a rank is stranded'
    return out, bus


def test_both_ranks_walk_the_same_collective_sequence_and_stop_at_the_same_epoch(monkeypatch, tmp_path):
    out, bus = _run_two_ranks(monkeypatch, tmp_path, epochs=8, early_stopping=True, patience=2, min_delta=0.05)
    assert bus.log[0] == bus.log[1] and bus.log[0][0] == 'broadcast_trainer'
    h0, h1 = out[0].fithist, out[1].fithist
    assert h0['val_loss'] == h1['val_loss'] and h0['loss'] == h1['loss']          # rank-averaged, identical everywhere
    assert len(h0['val_loss']) < 8                                                # early stopping did fire ...
    assert out[0].test_loss == out[1].test_loss                                  # ... and the test score is shared
    # every epoch: one train-loss and one validation-loss reduction; then the test-loss reduction
    assert bus.log[0].count('allreduce_host[2]:sum') == 2 * len(h0['val_loss']) + 1
    assert out[0].running_on_first_worker and not out[1].running_on_first_worker
    # LR x world (supervised.py:338-352) and equal shards: both ranks took the same number of steps
    assert out[0].engine.n == out[1].engine.n > 0


def test_steps_per_epoch_is_divided_by_the_world_size(monkeypatch, tmp_path):
    out, bus = _run_two_ranks(monkeypatch, tmp_path, epochs=2, steps_per_epoch=4)
    assert out[0].engine.n == out[1].engine.n == 2 * (4 // 2)                     # supervised.py:393-394
    assert bus.log[0] == bus.log[1]
