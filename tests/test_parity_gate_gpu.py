"""-m gpu: the parity contract on the shards bench.py times (seeds 40 / 1040 / 2040: ranks 0-2 of `bench.py --gpus N`, 32 distinct
65,536-point scans each).

north_star's gate is "final pose within 1e-4 m / 1e-5 rad of the reference CPU path".

  * The UPSTREAM order (dgs_params.ndt_strict_order = 1: dgs_params_init's default and the mode bench.py times since round 4) meets it on every pair of every shard,
    with float32-bit-equal transforms and the oracle's iteration counts -- asserted here unconditionally.
  * The FAST order (ndt_strict_order = 0, opt-in) re-associates the per-point float math; NDT's damped Newton iteration with its loose stop (|step| < 0.01)
    amplifies that on a few ill-conditioned pairs per shard -- exactly where the oracle's own answer moves by more under perturbations
    that carry no information.  Round 3 asserted that with bounds fitted to the measurement ("measured - 1" pairs, twice the band): a
    regression of the same size would have stayed green.  Now the measurement itself is the assertion: the SET of pairs outside the gate
    must equal the committed list (tests/golden/fast_order_gate.json, made by scripts/r4_fast_variants.py), every one of them must sit
    inside ONE times the oracle's own 34-twin band recorded there, the band of the first of them is re-derived here (6 twins, a lower
    bound of the 34), over all other pairs the gate holds pair by pair, and what the caller consumes -- converged flags, the chosen loop
    candidate and its score (loop_detector.hpp:149-155) -- is the reference's.  Any change to the kernel that moves a pair across the
    gate turns this red and has to be re-measured."""
import json
import os

import numpy as np
import pytest

from tests.helpers import TOL_ROT, TOL_TRANS, oracle_shard, pose_error, sequential_best

pytestmark = pytest.mark.gpu

GATE = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fast_order_gate.json")))["shards"]


@pytest.mark.parametrize("seed", [40, 1040, 2040])
def test_upstream_order_meets_the_gate_on_every_pair_of_a_bench_shard(oracle_lib, seed):
    from delta_graph_slam_amd.registration import Registration
    tgt, sources, guesses, ref, fit_ref = oracle_shard(oracle_lib, seed)
    r = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=1)
    r.setInputTarget(tgt)
    res = r.align_batch(sources, guesses)
    for c in range(len(sources)):
        assert res[c]["converged"] == ref[c]["converged"] and res[c]["iterations"] == ref[c]["iterations"], (seed, c)
        assert np.array_equal(res[c]["T"], ref[c]["T"]), (seed, c, pose_error(res[c]["T"], ref[c]["T"]))
        assert abs(res[c]["fitness"] - fit_ref[c]) <= 1e-9 * fit_ref[c], (seed, c)
    # evaluation counts: equal on every pair but for line searches that sit on their clamped minimum step, where the sufficient-decrease
    # test is decided at the 1e-14 level of the sums' association: measured 0 of 96 with the round-4 kernels
    assert sum(res[c]["evaluations"] != ref[c]["evaluations"] for c in range(len(sources))) == 0
    b_ref, s_ref = sequential_best([x["converged"] for x in ref], fit_ref)
    b_gpu, s_gpu = sequential_best([x["converged"] for x in res], [x["fitness"] for x in res])
    assert b_ref >= 0 and b_gpu == b_ref and abs(s_gpu - s_ref) <= 1e-9 * s_ref


@pytest.mark.parametrize("seed", [40, 1040, 2040])
def test_fast_order_on_a_bench_shard(oracle_lib, seed):
    from delta_graph_slam_amd.registration import Registration
    tgt, sources, guesses, ref, fit_ref = oracle_shard(oracle_lib, seed)
    f = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=0)
    f.setInputTarget(tgt)
    fast = f.align_batch(sources, guesses)
    n = len(sources)
    err = np.array([pose_error(fast[c]["T"], ref[c]["T"]) for c in range(n)])
    ok = (err[:, 0] <= TOL_TRANS) & (err[:, 1] <= TOL_ROT)
    want = GATE[str(seed)]
    outside = sorted(int(c) for c in np.nonzero(~ok)[0])
    assert outside == sorted(int(k) for k in want["outside"]), (seed, outside, err[~ok])      # the exact set, not a count
    assert int(ok.sum()) == want["pairs_inside"]
    for c in outside:
        bt, br = want["outside"][str(c)]["oracle_band_m_rad"]
        assert bt > TOL_TRANS or br > TOL_ROT
        assert err[c, 0] <= bt + TOL_TRANS and err[c, 1] <= br + TOL_ROT, ("outside ONE times the oracle's own band", seed, c, err[c], bt, br)
    if outside:   # the committed band of the first outside pair is at least what six of its 34 twins give today (the oracle is deterministic)
        c = outside[0]
        _, bt6, br6 = oracle_lib.ndt_band(tgt, sources[c], guesses[c], resolution=1.0)
        bt, br = want["outside"][str(c)]["oracle_band_m_rad"]
        assert bt6 <= bt * (1 + 1e-9) + 1e-12 and br6 <= br * (1 + 1e-9) + 1e-12, (seed, c, bt6, bt, br6, br)
    for c in range(n):
        assert fast[c]["converged"] == ref[c]["converged"], c
    assert np.sqrt(np.mean(err[ok, 0] ** 2)) <= TOL_TRANS and np.sqrt(np.mean(err[ok, 1] ** 2)) <= TOL_ROT
    # ---- what the caller sees: the chosen loop candidate and its score
    b_ref, s_ref = sequential_best([x["converged"] for x in ref], fit_ref)
    b_gpu, s_gpu = sequential_best([x["converged"] for x in fast], [x["fitness"] for x in fast])
    assert b_ref >= 0 and b_gpu == b_ref, (seed, b_gpu, b_ref, s_gpu, s_ref)
    assert abs(s_gpu - s_ref) <= 1e-3 * s_ref, (seed, s_gpu, s_ref)
    for c in range(n):   # every candidate's score: 1e-3 inside the gate (measured <= 1e-4), an optimum of the same quality (1 %) on the listed pairs
        assert abs(fast[c]["fitness"] - fit_ref[c]) <= (1e-3 if ok[c] else 1e-2) * fit_ref[c], (seed, c, fast[c]["fitness"], fit_ref[c])
