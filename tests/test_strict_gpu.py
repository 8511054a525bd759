"""-m gpu parity of the NDT validation modes (dgs_params.ndt_strict_order, include/dgs_reg.h) against the CPU oracle.

ndt_strict_order = 1 (UPSTREAM): every float operation of upstream's per-voxel update in upstream's order; only the order in
which the per-point double totals are summed differs from the oracle -> evaluations agree to ~1e-14, final transforms are
asserted EQUAL (float32 bit patterns), unconditionally, on every pair.
ndt_strict_order = 2 (UPSTREAM_SEQUENTIAL): the sums are formed in point-index order too -> score, gradient and Hessian of
every evaluation are asserted bit-identical doubles.
The voxel table (means, inverse covariances) is bit-identical in every mode."""
import numpy as np
import pytest

from delta_graph_slam_amd import _lib as L
from delta_graph_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shard40(oracle_lib):
    from tests.helpers import oracle_shard
    return oracle_shard(oracle_lib, 40)


def _reg(mode, **kw):
    from delta_graph_slam_amd.registration import Registration
    kw.setdefault("ndt_resolution", 1.0)
    return Registration("NDT_OMP", ndt_strict_order=mode, **kw)


@pytest.mark.parametrize("res", [0.5, 1.0, 2.0])
def test_voxel_table_is_bit_identical(oracle_lib, res):
    tgt, _, _ = synth.kitti_pair(n_points=32768)
    o = oracle_lib.NdtOracle(resolution=res)
    o.set_target(tgt)
    r = _reg(0, ndt_resolution=res)
    r.setInputTarget(tgt)
    vo, vg = o.voxels(), r.ndt_voxels()
    for k in ("keys", "counts", "valid"):
        assert np.array_equal(vo[k], vg[k]), k
    assert np.array_equal(vo["mean"], vg["mean"])
    v = vo["valid"]
    assert v.sum() > 100
    assert np.array_equal(vo["icov"][v], vg["icov"][v])   # same eigen solver, clamp and inverse, operation for operation


@pytest.mark.parametrize("search", ["DIRECT7", "DIRECT1", "DIRECT26", "KDTREE"])
def test_evaluations_are_bit_identical(oracle_lib, search):
    tgt, src, _ = synth.planar_pair(n=16384)
    o = oracle_lib.NdtOracle(resolution=1.0, search_method=search)
    o.set_target(tgt)
    o.set_source(src)
    regs = {m: _reg(m, ndt_search_method=L.NDT_SEARCH[search]) for m in (1, 2)}
    for r in regs.values():
        r.setInputTarget(tgt)
        r.setInputSource(src)
    for p in ([0, 0, 0, 0, 0, 0], [0.2, -0.05, 0.03, 0.02, -0.03, 0.04], [0.3, -0.1, 0.05, 0.01, -0.02, 0.05], [5.0, 3.0, 0.5, 0.3, -0.2, 1.0]):
        p = np.array(p, float)
        so, go, Ho = o.derivatives(p)
        sg, gg, Hg = regs[2].ndt_derivatives(p)
        assert so == sg and np.array_equal(go, gg) and np.array_equal(Ho, Hg), (search, p)
        assert not np.array_equal(Ho, Ho.T) or np.abs(Ho).max() == 0     # upstream's float Hessian is not exactly symmetric: all 36 entries travel
        s1, g1, H1 = regs[1].ndt_derivatives(p)
        assert abs(so - s1) <= 1e-12 * abs(so) + 1e-300
        assert np.abs(go - g1).max() <= 1e-11 * (np.abs(go).max() + 1e-300) and np.abs(Ho - H1).max() <= 1e-11 * (np.abs(Ho).max() + 1e-300)


def _same_run(r, ro):
    assert r.hasConverged() == ro["converged"]
    assert (r.last_result.iterations, r.last_result.evaluations) == (ro["iterations"], ro["evaluations"])
    assert np.array_equal(r.getFinalTransformation(), ro["T"]), np.abs(r.getFinalTransformation() - ro["T"]).max()
    tg = r.ndt_trajectory()
    assert tg.shape == ro["trajectory"].shape and np.abs(tg - ro["trajectory"]).max() <= 1e-11


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("line_search", [0, 1])
def test_cfg1_align_is_bit_identical(oracle_lib, mode, line_search):
    tgt, src, _ = synth.planar_pair()
    o = oracle_lib.NdtOracle(resolution=1.0, line_search=line_search)
    o.set_target(tgt)
    o.set_source(src)
    r = _reg(mode, ndt_line_search=line_search)
    r.setInputTarget(tgt)
    r.setInputSource(src)
    r.align()
    _same_run(r, o.align())


@pytest.mark.parametrize("mode", [1, 2])
def test_cfg2_align_is_bit_identical_from_identity_and_from_a_prediction(oracle_lib, mode):
    tgt, src, Tgt = synth.kitti_pair()
    o = oracle_lib.NdtOracle(resolution=1.0)
    o.set_target(tgt)
    o.set_source(src)
    r = _reg(mode)
    r.setInputTarget(tgt)
    r.setInputSource(src)
    r.align()                                         # the identity guess: an ill-conditioned start (DESIGN.md "Parity")
    _same_run(r, o.align())
    guess = Tgt.copy()
    guess[0, 3] -= 0.25
    guess[1, 3] += 0.10
    r.align(guess.astype(np.float32))
    _same_run(r, o.align(guess.astype(np.float32)))
    for eps in (1e-6,):                               # also where both sit on the same fixed point
        o2 = oracle_lib.NdtOracle(resolution=1.0, transformation_epsilon=eps)
        o2.set_target(tgt)
        o2.set_source(src)
        r2 = _reg(mode, transformation_epsilon=eps)
        r2.setInputTarget(tgt)
        r2.setInputSource(src)
        r2.align(guess.astype(np.float32))
        _same_run(r2, o2.align(guess.astype(np.float32)))


def test_cfg5_dense_indoor_is_bit_identical(oracle_lib):
    tgt, src, _ = synth.indoor_pair()
    o = oracle_lib.NdtOracle(resolution=0.5)
    o.set_target(tgt)
    o.set_source(src)
    r = _reg(1, ndt_resolution=0.5)
    r.setInputTarget(tgt)
    r.setInputSource(src)
    r.align()
    _same_run(r, o.align())


def test_cfg4_shard_shape_32_candidates_of_65536_points(oracle_lib, shard40):
    """configs[3]'s per-GPU shard (= bench.py's step, the same 32 distinct scans and guesses): 32 candidates x 65,536 points against
    one target, yaw / xy guesses perturbed by up to 1 m / 5 deg.  Upstream-order mode: every final transform equals the oracle's,
    unconditionally, and so does the caller's arg-min.  (The default order on the same shard: tests/test_parity_gate_gpu.py.)"""
    tgt, sources, guesses, ref, fit_ref = shard40
    r = _reg(1)
    r.setInputTarget(tgt)
    res = r.align_batch(sources, guesses)
    for c in range(32):
        assert res[c]["converged"] == ref[c]["converged"] and res[c]["iterations"] == ref[c]["iterations"]
        assert np.array_equal(res[c]["T"], ref[c]["T"]), c
        assert abs(res[c]["fitness"] - fit_ref[c]) <= 1e-11 * fit_ref[c]
    from tests.helpers import sequential_best
    assert sequential_best([x["converged"] for x in res], [x["fitness"] for x in res])[0] == sequential_best([x["converged"] for x in ref], fit_ref)[0]


def test_structural_zero_shortcuts_are_bit_identical():
    """The upstream-order kernel skips upstream's multiplications by the structural zeros / ones of the point gradient and point
    Hessian (exact for finite operands); `DGS_NDT_STRICT_LITERAL=1` multiplies them out.  Same bits: score, gradient, the full
    non-symmetric 6x6 Hessian at several poses, and the final transforms of a batch."""
    import os
    import subprocess
    import sys
    code = r'''
import json, sys, numpy as np
sys.path.insert(0, %r)
from delta_graph_slam_amd import synth, _lib as L
from delta_graph_slam_amd.registration import Registration
tgt, sources, guesses, _ = synth.loop_batch(n_candidates=4, n_points=30000, seed=3, distinct_scans=4)
out = {}
for method in ("DIRECT7", "DIRECT1", "KDTREE"):
    r = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=1, ndt_search_method=L.NDT_SEARCH[method])
    r.setInputTarget(tgt)
    r.setInputSource(sources[0])
    for k, p in enumerate(([0, 0, 0, 0, 0, 0], [0.3, -0.2, 0.05, 0.01, -0.02, 0.4])):
        s, g, H = r.ndt_derivatives(np.array(p, np.float64))
        out["%%s_%%d" %% (method, k)] = [float(s).hex()] + [float(v).hex() for v in g] + [float(v).hex() for v in H.ravel()]
r = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=1)
r.setInputTarget(tgt)
res = r.align_batch(sources, guesses)
out["batch"] = [[float(v).hex() for v in np.asarray(x["T"], np.float64).ravel()] + [x["iterations"], x["evaluations"]] for x in res]
print(json.dumps(out))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for lit in ("0", "1"):
        env = dict(os.environ, DGS_NDT_STRICT_LITERAL=lit)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        res[lit] = p.stdout.strip().splitlines()[-1]
    assert res["0"] == res["1"]
