"""-m gpu: `DGS_EARLY_FITNESS=1` -- dgs_align_batch walks the fitness (loop_detector.hpp:148) of the candidates that have finished on
the side stream while the others still iterate.  Off by default (it measured slower, csrc/handle.h); while it is in the library every
record of a batch must EQUAL the plain order's: same transforms, and the same fitness sums bit for bit (a pair's partial rows and their
order do not depend on which launch walked it)."""
import os

import numpy as np
import pytest

from delta_graph_slam_amd import synth

pytestmark = pytest.mark.gpu


def _records(env, tgt, sources, guesses, order, rounds=3):
    from delta_graph_slam_amd.registration import Registration
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        reg = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=order)   # the switches are read when the handle is made
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    out = []
    for _ in range(rounds):       # the second and third batch find the target's index already built: no side-stream build ahead of the walks
        reg.setInputTarget(tgt)
        res = reg.align_batch(sources, guesses)
        out.append(np.concatenate([np.concatenate([x["T"].ravel(), [x["fitness"], x["iterations"], x["evaluations"], float(x["converged"])]]) for x in res]))
    return out


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("min_pairs,lds_kb", [(1, 0), (8, 0), (3, 54)])
def test_early_fitness_walks_give_the_records_of_the_plain_order(min_pairs, lds_kb, order):
    tgt, sources, guesses, _ = synth.loop_batch(n_candidates=24, n_points=32768, seed=77, distinct_scans=24)
    sources = list(sources)
    sources[5] = sources[5][:20001]       # ragged sizes
    sources[11] = sources[11][:777]
    plain = _records({"DGS_EARLY_FITNESS": "0"}, tgt, sources, guesses, order)
    early = _records({"DGS_EARLY_FITNESS": "1", "DGS_EARLY_FITNESS_MIN_PAIRS": str(min_pairs), "DGS_EARLY_FITNESS_LDS_KB": str(lds_kb)}, tgt, sources, guesses, order)
    evals = plain[0].reshape(len(sources), -1)[:, 18]
    assert evals.max() - evals.min() >= 8, "the candidates must finish at different launches for the early walks to have anything to do"
    for a, b in zip(plain, early):
        assert np.array_equal(a, b)
