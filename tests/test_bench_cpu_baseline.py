"""bench.py's CPU baseline, second phase (round 6): concurrent replicas of the oracle step on disjoint core sets.  Host logic only --
tiny replicas (B = 1, four cores each) so that it runs in the CPU suite; the GPU box runs it at B = 16 on 16-core sets."""
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_replicas_aggregate_over_the_common_window():
    import bench
    from oracle import models as M
    by = bench.physical_cores_by_socket()
    if not by or sum(len(v) for v in by.values()) < 8:
        pytest.skip('needs eight physical cores visible through sysfs')
    P = M.init_params('net_postupsampling', (1, 128, 128, 1), None, seed=7, backbone_block='resnet', upsampling='spc', scale=4)
    w = {k: np.asarray(v) for k, v in P.items()}
    with tempfile.TemporaryDirectory() as tmp:
        r = bench.cpu_replicas(w, tmp, batch=1, per=4, nsteps=1)
    assert r['replicas'] >= 2 and r['cores'] == 4 * r['replicas'] and len(r['per_replica_samples_per_s']) == r['replicas']
    # all samples over the common window: never more than the sum of the replicas' own rates, and more than the slowest one's
    assert min(r['per_replica_samples_per_s']) < r['value'] <= sum(r['per_replica_samples_per_s']) * 1.001
