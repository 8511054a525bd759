"""-m gpu: the persistent queue kernel of the NDT batch (csrc/ndt_align.hip, ndt_queue_kernel) -- ONE launch per align in which the
pairs advance independently -- against the launch-per-evaluation path.  The slice count of a round is a fixed function of the batch
shape and the round number, so the launch-per-evaluation path can be made to cut every round the same way (DGS_NDT_SCHEDULE=1) and
the two must then agree BIT FOR BIT: same transforms, scores, iteration / evaluation counts, trajectories and fitness scores.  That
pins the queue kernel's hand-offs (queue words, per-round record slots, rows, tickets) on every search method and batch shape.
The queue kernel is an EXPERIMENT (measured slower): it is compiled into libdgs_reg_exp.so only and off by default."""
import os

import numpy as np
import pytest

from delta_graph_slam_amd import _lib as L
from delta_graph_slam_amd import synth

pytestmark = pytest.mark.gpu


def _reg(env, **kw):
    from delta_graph_slam_amd.registration import Registration
    kw.setdefault("ndt_strict_order", 0)      # the queue / schedule kernels serve the fast order only
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return Registration("NDT_OMP", lib_path=L.EXPERIMENTS_LIB_PATH, **kw)   # the queue kernel lives in the experiments build
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _queue(**kw):
    return _reg({"DGS_NDT_QUEUE": "1"}, **kw)


def _lockstep_same_schedule(**kw):
    return _reg({"DGS_NDT_QUEUE": "0", "DGS_NDT_SCHEDULE": "1"}, **kw)


def _same(a, b):
    assert len(a) == len(b)
    for c, (x, y) in enumerate(zip(a, b)):
        assert x["status"] == y["status"] == 0
        assert np.array_equal(x["T"], y["T"]), c
        assert (x["converged"], x["iterations"], x["evaluations"], x["score"]) == (y["converged"], y["iterations"], y["evaluations"], y["score"]), c
        assert x["fitness"] == y["fitness"], c


@pytest.fixture(scope="module")
def shard():
    return synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=8)


def test_queue_kernel_equals_launch_per_evaluation_bit_for_bit_on_the_bench_shard(shard):
    tgt, sources, guesses, _ = shard
    q, l = _queue(ndt_resolution=1.0), _lockstep_same_schedule(ndt_resolution=1.0)
    for r in (q, l):
        r.setInputTarget(tgt)
    a, b = q.align_batch(sources, guesses), l.align_batch(sources, guesses)
    _same(a, b)
    for c in (0, 7, 31):
        assert np.array_equal(q.ndt_trajectory(c), l.ndt_trajectory(c))
    _same(q.align_batch(sources, guesses), a)                      # and reproducible run to run (no timing in the partition of the sums)
    assert sum(x["evaluations"] for x in a) > 32 * 10


@pytest.mark.parametrize("search", ["DIRECT1", "DIRECT26", "KDTREE"])
def test_queue_kernel_other_search_methods(shard, search):
    tgt, sources, guesses, _ = shard
    kw = dict(ndt_resolution=1.0, ndt_search_method=L.NDT_SEARCH[search])
    q, l = _queue(**kw), _lockstep_same_schedule(**kw)
    for r in (q, l):
        r.setInputTarget(tgt)
    _same(q.align_batch(sources[:6], guesses[:6]), l.align_batch(sources[:6], guesses[:6]))


def test_queue_kernel_on_ragged_tiny_and_wide_batches():
    """2 pairs, 3 ragged pairs with an empty and a 5-point source, and 200 small pairs (more pairs than a wave has lanes: the
    claim scans several words per lane)."""
    rng = np.random.default_rng(3)
    tgt, sources, guesses, _ = synth.loop_batch(n_candidates=4, n_points=16384, seed=9, distinct_scans=4)
    q, l = _queue(ndt_resolution=1.0), _lockstep_same_schedule(ndt_resolution=1.0)
    for r in (q, l):
        r.setInputTarget(tgt)
    _same(q.align_batch(sources[:2], guesses[:2]), l.align_batch(sources[:2], guesses[:2]))
    ragged = [sources[0][:9000], np.zeros((0, 4), np.float32), sources[1][:5], sources[2]]
    a, b = q.align_batch(ragged, guesses), l.align_batch(ragged, guesses)
    assert a[1]["status"] == b[1]["status"] == 4 and not a[1]["converged"]
    for k in (0, 2, 3):
        assert np.array_equal(a[k]["T"], b[k]["T"]) and a[k]["evaluations"] == b[k]["evaluations"] and a[k]["fitness"] == b[k]["fitness"]
    wide = [sources[k % 4][: 1500 + 37 * k] for k in range(200)]
    gw = np.stack([guesses[k % 4] for k in range(200)])
    _same(q.align_batch(wide, gw), l.align_batch(wide, gw))


def test_single_align_keeps_the_launch_per_evaluation_path_and_a_batch_of_one_can_use_the_queue(shard):
    tgt, sources, guesses, _ = shard
    d = _reg({}, ndt_resolution=1.0)                               # default: one launch per evaluation
    q1 = _reg({"DGS_NDT_QUEUE": "1", "DGS_NDT_QUEUE_MIN_PAIRS": "1"}, ndt_resolution=1.0)
    l1 = _reg({"DGS_NDT_QUEUE": "0", "DGS_NDT_SCHEDULE": "1"}, ndt_resolution=1.0)
    for r in (d, q1, l1):
        r.setInputTarget(tgt)
        r.setInputSource(sources[3])
        r.align(guesses[3])
    assert np.array_equal(q1.getFinalTransformation(), l1.getFinalTransformation()) and q1.last_result.evaluations == l1.last_result.evaluations
    assert d.hasConverged() and q1.hasConverged()
