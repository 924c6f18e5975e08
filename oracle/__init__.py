"""CPU oracle for the dl4ds conv-SR train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dl4ds_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker / the timed CPU baseline.

PARITY UNPINNED: the reference (carlos-gg/dl4ds @ 1.8.0) ships no tests, no
fixtures and no golden vectors, and its arithmetic lives in TensorFlow/Keras
(un-vendored, un-pinned; ``setup.py:46-58``), which cannot be installed in the
build container (no wheel, no network).  This oracle is therefore a
*restatement* of

  * the graph topology / hyper-parameters in ``dl4ds/models/*.py``,
    ``dl4ds/losses.py`` and ``dl4ds/training/{supervised,cgan}.py`` (cited
    file:line at each function), and
  * the published TF/Keras 2.6-2.15 semantics of each primitive
    (SURVEY.md appendix A), stated as assumptions in each docstring.

It is pinned only by (1) analytic known-answer tests, (2) fp64 finite
differences, (3) agreement between two independent backends written here
(``np_ops`` -- plain numpy loops/tensordot; ``torch_ops`` -- torch CPU
functional ops with autograd), see ``tests/test_oracle_*.py``.
"""
