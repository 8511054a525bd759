"""-m gpu parity at BASELINE.json's full sizes, plus size-independent properties and the committed goldens.

Tolerance (north_star): final pose within 1e-4 m / 1e-5 rad of the reference CPU path.
* In upstream operation order (dgs_params.ndt_strict_order >= 1) the final transform EQUALS the oracle's on every pair,
  unconditionally: tests/test_strict_gpu.py, and the loop-closure test below.
* The default (fast) order re-associates the float per-point math (1/3 of the flops).  NDT's damped Newton iteration with its
  loose stop amplifies float rounding on ill-conditioned pairs -- the oracle itself moves by more than the tolerance there when
  it is compiled with FMA contraction, uses the host libm's expf, or its float32 guess moves by an ulp (tests/helpers.py).  The
  fast order is therefore asserted per evaluation everywhere (score / gradient / Hessian at the oracle's own iterates) and on
  the final pose wherever the oracle's own band is inside the tolerance."""
import os

import numpy as np
import pytest

from delta_graph_slam_amd import _lib as L
from delta_graph_slam_amd import synth
from tests.helpers import TOL_ROT, TOL_TRANS, ndt_oracle_band, pose_error

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "registration_small.npz"))


def _reg(method="NDT_OMP", **kw):
    from delta_graph_slam_amd.registration import Registration
    return Registration(method, **kw)


CFG4_FAST_ORDER_OUTSIDE_THE_ORACLE_BAND = [2]   # test_cfg4_loop_batch_sharded_api_matches_sequential: pairs of the 6 where the fast order (ndt_strict_order = 0) leaves the 34-twin band


def _check_against_band(T_gpu, T_oracle, bt, br):
    et, er = pose_error(T_gpu, T_oracle)
    if bt <= TOL_TRANS and br <= TOL_ROT:
        assert et <= TOL_TRANS and er <= TOL_ROT, ("well-conditioned pair outside tolerance", et, er, bt, br)
        return True
    # The reference algorithm itself is not reproducible to the tolerance on this pair (its answer moves by more than the
    # tolerance when its float32 guess moves by an ulp): no final-pose statement is defined; callers fall back on
    # evaluation-level parity along the oracle's trajectory.
    return False


def test_cfg2_kitti_pair_ndt(oracle_lib):
    tgt, src, Tgt = synth.kitti_pair()
    assert tgt.shape == (65536, 4)
    r = _reg(ndt_resolution=1.0)
    r.setInputTarget(tgt)
    r.setInputSource(src)
    r.align()
    ro, bt, br = ndt_oracle_band(oracle_lib, tgt, src, resolution=1.0)
    assert r.hasConverged() == ro["converged"]
    assert r.last_result.iterations == ro["iterations"] and r.last_result.evaluations == ro["evaluations"]
    et, er = pose_error(r.getFinalTransformation(), ro["T"])
    assert et <= TOL_TRANS and er <= TOL_ROT, (et, er, bt, br)
    # from the identity guess NDT at 1 m resolution locks y / yaw but not the 1 m along-street offset (the ground rings
    # carry most points and do not constrain x); with an odometry-like prediction it reaches the ground truth
    guess = Tgt.copy()
    guess[0, 3] -= 0.25
    guess[1, 3] += 0.10
    r.align(guess.astype(np.float32))
    o = oracle_lib.NdtOracle(resolution=1.0)
    o.set_target(tgt)
    o.set_source(src)
    ro = o.align(guess.astype(np.float32))
    et, er = pose_error(r.getFinalTransformation(), ro["T"])
    assert et <= TOL_TRANS and er <= TOL_ROT and r.last_result.iterations == ro["iterations"]
    assert pose_error(r.getFinalTransformation(), Tgt)[0] < 0.05
    fo, n, inl = oracle_lib.fitness_score(tgt, src, r.getFinalTransformation())
    assert abs(r.getFitnessScore() - fo) <= 1e-11 * fo
    assert abs(r.getInlierFraction(0.25) - inl / src.shape[0]) < 1e-12
    c = r.counts()
    assert c["target_points"] == 65536 and c["valid_voxels"] > 500


def test_cfg5_dense_indoor_ndt_half_metre(oracle_lib):
    tgt, src, Tgt = synth.indoor_pair()
    assert tgt.shape == (200000, 4)
    r = _reg(ndt_resolution=0.5)
    r.setInputTarget(tgt)
    r.setInputSource(src)
    r.align()
    ro, bt, br = ndt_oracle_band(oracle_lib, tgt, src, resolution=0.5)
    assert r.hasConverged() == ro["converged"] and r.last_result.iterations == ro["iterations"]
    assert _check_against_band(r.getFinalTransformation(), ro["T"], bt, br)
    assert pose_error(r.getFinalTransformation(), Tgt)[0] < 0.02
    # derivative evaluation at full size: linear in the cloud (two halves add up to the whole)
    p = np.array([0.08, 0.04, 0.0, 0.0, 0.0, 0.025])
    s_all, g_all, H_all = r.ndt_derivatives(p)
    parts = []
    for half in (src[:100000], src[100000:]):
        r.setInputSource(half)
        parts.append(r.ndt_derivatives(p))
    assert abs(parts[0][0] + parts[1][0] - s_all) <= 1e-12 * abs(s_all)
    assert np.abs(parts[0][1] + parts[1][1] - g_all).max() <= 1e-10 * np.abs(g_all).max() + 1e-9
    assert np.abs(parts[0][2] + parts[1][2] - H_all).max() <= 1e-10 * np.abs(H_all).max()


def test_cfg3_vlp16_stream_gicp_odometry(oracle_lib):
    """FAST_GICP with the launch-file values (delta_graph_slam.launch:60-69), driven by the odometry mirror."""
    from delta_graph_slam_amd.odometry import ScanMatchingOdometry
    from tests.oracle_engine import OracleRegistration
    clouds, poses = synth.vlp16_stream(n_frames=32)
    kw = dict(keyframe_delta_trans=1.0, keyframe_delta_angle=1.0, keyframe_delta_time=1e9)
    gpu = ScanMatchingOdometry(_reg("FAST_GICP", gicp_max_correspondence_distance=2.0, transformation_epsilon=0.1), kw)
    cpu = ScanMatchingOdometry(OracleRegistration("FAST_GICP", max_correspondence_distance=2.0, transformation_epsilon=0.1), kw)
    for k, c in enumerate(clouds):
        og = gpu.matching(0.1 * k, c, want_status=True)
        oc = cpu.matching(0.1 * k, c, want_status=True)
        dt, dr = pose_error(og, oc)
        assert dt <= TOL_TRANS and dr <= TOL_ROT, (k, dt, dr)
        if k:
            assert gpu.last_status.has_converged == cpu.last_status.has_converged
            assert abs(gpu.last_status.matching_error - cpu.last_status.matching_error) <= 1e-9 * cpu.last_status.matching_error
            assert abs(gpu.last_status.inlier_fraction - cpu.last_status.inlier_fraction) < 1e-9
    assert gpu.n_keyframes == cpu.n_keyframes and gpu.n_keyframes >= 4      # >= 3 keyframe switches inside the 32 frames


def test_cfg4_loop_batch_sharded_api_matches_sequential(oracle_lib):
    """LoopDetector.matching over the HIP engine == the reference's sequential candidate loop on the oracle."""
    from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
    from tests.oracle_engine import OracleEngine
    tgt, sources, guesses, gts = synth.loop_batch(n_candidates=6, n_points=16384, seed=77, distinct_scans=3)
    new = KeyFrame(tgt, np.eye(3), 100.0, 0)
    cands = []
    for c, G in enumerate(guesses):
        est = np.eye(3)
        est[:2, :2] = G[:2, :2]
        est[:2, 2] = G[:2, 3]
        cands.append(KeyFrame(sources[c], est, 0.0, c + 1))
    dc = LoopDetector({"fitness_score_thresh": 1e9}, registration=OracleEngine("NDT_OMP", resolution=1.0))
    lc = dc.matching(cands, new)
    # ---- upstream operation order: every record equals the sequential reference loop's, and so does the chosen loop
    ds = LoopDetector({"fitness_score_thresh": 1e9}, registration=_reg(ndt_resolution=1.0, ndt_strict_order=1))
    ls = ds.matching(cands, new)
    for c in range(6):
        assert (ds.last_records[c, 1] > 0.5) == (dc.last_records[c, 1] > 0.5)
        assert np.array_equal(ds.last_records[c, 4:20].astype(np.float32), dc.last_records[c, 4:20].astype(np.float32)), c
        assert abs(ds.last_records[c, 2] - dc.last_records[c, 2]) <= 1e-11 * dc.last_records[c, 2]
    assert (ls is None) == (lc is None) and (ls is None or (ls.key2.id == lc.key2.id and np.array_equal(ls.relative_pose, lc.relative_pose)))
    # ---- the opt-in fast order (dgs_params.ndt_strict_order = 0)
    dg = LoopDetector({"fitness_score_thresh": 1e9}, registration=_reg(ndt_resolution=1.0, ndt_strict_order=0))
    lg = dg.matching(cands, new)
    r = _reg(ndt_resolution=1.0, ndt_strict_order=0)
    r.setInputTarget(tgt)
    o = oracle_lib.NdtOracle(resolution=1.0)
    o.set_target(tgt)
    n_loose = 0
    outside_band = []
    for c in range(6):
        assert (dg.last_records[c, 1] > 0.5) == (dc.last_records[c, 1] > 0.5)
        ro, bt, br = ndt_oracle_band(oracle_lib, tgt, sources[c], guesses[c], resolution=1.0)
        if not _check_against_band(dg.last_records[c, 4:20].reshape(4, 4), dc.last_records[c, 4:20].reshape(4, 4), bt, br):
            # The oracle is not reproducible to the gate on this pair (a street scene on a 1 m grid leaves the along-street offset weakly
            # constrained: moving the oracle's float32 guess by a few ulps moves its answer by up to a metre on such pairs; 4-5 of these 6
            # pairs are of that kind, against 1-4 of 32 on the bench shards, which tests/test_parity_gate_gpu.py pins pair by pair).  The
            # statement there is the band statement: the device's answer lies within ONE times the oracle's own 34-twin band.
            twins = ((True, 0, 0), (False, 1, 0)) + tuple((False, 0, k) for k in range(-16, 17) if k)
            _, bt34, br34 = ndt_oracle_band(oracle_lib, tgt, sources[c], guesses[c], twins=twins, resolution=1.0)
            et, er = pose_error(dg.last_records[c, 4:20].reshape(4, 4), dc.last_records[c, 4:20].reshape(4, 4))
            if not (et <= bt34 + TOL_TRANS and er <= br34 + TOL_ROT):
                outside_band.append(c)
            n_loose += 1
        # whatever the conditioning, every evaluation along the oracle's own trajectory agrees tightly
        r.setInputSource(sources[c])
        o.set_source(sources[c])
        for p in ro["trajectory"][1:6]:
            so, go, Ho = o.derivatives(p)
            sg, gg, Hg = r.ndt_derivatives(p)
            assert abs(so - sg) <= 1e-6 * abs(so) and np.abs(go - gg).max() <= 5e-6 * np.abs(go).max() and np.abs(Ho - Hg).max() <= 5e-6 * np.abs(Ho).max()
    # The band is 34 samples of a chaotic map, not a bound: the re-associated order can end in an optimum none of the twins visits.  Which pairs
    # do is pinned (measured on an MI355X, round 4: pair 2, 0.25 m / 0.017 rad against a band of 0.66 m / 0.0077 rad) -- the FAST order
    # does not meet north_star's gate, which is why the upstream order is the default and the one bench.py times; a change that moves another pair out shows here.
    assert outside_band == CFG4_FAST_ORDER_OUTSIDE_THE_ORACLE_BAND, outside_band
    # the caller-level result: the same loop candidate as the reference's sequential loop, its score to 1e-3 relative
    assert lg is not None and lc is not None and lg.key2.id == lc.key2.id
    assert abs(lg.score - lc.score) <= 1e-3 * lc.score


def test_committed_goldens_on_the_device():
    tgt, src = GOLD["tgt"], GOLD["src"]
    poses = np.array([[0, 0, 0, 0, 0, 0], [0.2, -0.05, 0.03, 0.02, -0.03, 0.04]], float)
    for search in ("DIRECT7", "DIRECT1", "KDTREE"):
        r = _reg(ndt_resolution=2.0, ndt_search_method=L.NDT_SEARCH[search])
        r.setInputTarget(tgt)
        r.setInputSource(src)
        for k, p in enumerate(poses):
            s, g, H = r.ndt_derivatives(p)
            assert abs(s - GOLD[f"ndt_{search}_p{k}_score"]) <= 2e-6 * abs(s)
            assert np.abs(g - GOLD[f"ndt_{search}_p{k}_grad"]).max() <= 1e-5 * np.abs(g).max() + 1e-9
            assert np.abs(H - GOLD[f"ndt_{search}_p{k}_hess"]).max() <= 1e-5 * np.abs(H).max()
        r.align()
        it = GOLD[f"ndt_{search}_iters"]
        assert [r.last_result.iterations, r.last_result.evaluations, int(r.hasConverged())] == list(it)
        dt, dr = pose_error(r.getFinalTransformation(), GOLD[f"ndt_{search}_T"])
        assert dt <= TOL_TRANS and dr <= TOL_ROT
        traj = r.ndt_trajectory()
        assert np.abs(traj - GOLD[f"ndt_{search}_traj"]).max() < 1e-4
    r = _reg(ndt_resolution=2.0, ndt_line_search=0)
    r.setInputTarget(tgt)
    r.setInputSource(src)
    r.align()
    assert [r.last_result.iterations, r.last_result.evaluations, int(r.hasConverged())] == list(GOLD["ndt_fixedstep_iters"])
    for reg in ("PLANE", "FROBENIUS"):
        g = _reg("FAST_GICP", gicp_max_correspondence_distance=2.0, gicp_regularization=L.GICP_REG[reg])
        g.setInputTarget(tgt)
        g.setInputSource(src)
        e, H, b = g.gicp_linearize(np.eye(4))
        assert abs(e - GOLD[f"gicp_{reg}_lin_err"]) <= 1e-9 * e and np.abs(H - GOLD[f"gicp_{reg}_lin_H"]).max() <= 1e-9 * np.abs(H).max()
        g.align()
        assert [g.last_result.iterations, g.last_result.evaluations, int(g.hasConverged())] == list(GOLD[f"gicp_{reg}_iters"])
        dt, dr = pose_error(g.getFinalTransformation(), GOLD[f"gicp_{reg}_T"])
        assert dt <= TOL_TRANS and dr <= TOL_ROT


def test_results_are_bit_reproducible_run_to_run():
    tgt, src, _ = synth.kitti_pair(n_points=32768)
    outs = []
    for _ in range(2):
        r = _reg(ndt_resolution=1.0)
        r.setInputTarget(tgt)
        r.setInputSource(src)
        r.align()
        outs.append((r.getFinalTransformation().tobytes(), r.getFitnessScore(), r.last_result.score))
    assert outs[0] == outs[1]          # fixed-order reductions, no atomics in the sums


def test_edge_cases_on_the_device():
    r = _reg(ndt_resolution=1.0)
    rng = np.random.default_rng(0)
    tiny = np.ones((5, 4), np.float32)
    tiny[:, :3] = rng.normal(size=(5, 3))
    r.setInputTarget(tiny)                      # fewer than 6 points per voxel: no valid voxel at all
    r.setInputSource(tiny)
    r.align()
    assert r.hasConverged() and np.array_equal(r.getFinalTransformation(), np.eye(4, dtype=np.float32))   # zero gradient: converged at the guess
    assert r.getFitnessScore() == 0.0
    bad = synth.planar_pair(n=4096)[0]
    bad[7, 0] = np.nan
    bad[9, 2] = np.inf
    r.setInputTarget(bad)                       # non-finite points are skipped by the voxel filter and the NN index
    assert r.counts()["occupied_voxels"] > 0
    far = np.ones((3, 4), np.float32)
    far[:, :3] = [[1e5, 0, 0], [0, 1e5, 0], [0, 0, 0]]
    idx, sq = r.nearestKSearch(far)
    assert np.all(np.isfinite(sq)) and sq[0] > 1e9      # unbounded exact search (fitness_score_max_range = DBL_MAX)
    huge = np.ones((64, 4), np.float32)
    huge[:, :3] = rng.uniform(-3e4, 3e4, size=(64, 3))
    from delta_graph_slam_amd.registration import DgsError
    with pytest.raises(DgsError) as e:
        r2 = _reg(ndt_resolution=0.01)
        r2.setInputTarget(huge)                 # "Leaf size is too small for the input dataset"
    assert e.value.status == 5


@pytest.mark.parametrize("method", ["FAST_GICP", "FAST_VGICP"])
def test_cfg2_kitti_pair_gicp_family(oracle_lib, method):
    """configs[1]'s 65,536-point pair through the two GICP back-ends: same iterations and the same pose as the restatement."""
    tgt, src, Tgt = synth.kitti_pair()
    guess = Tgt.copy()
    guess[0, 3] -= 0.25
    guess[1, 3] += 0.10
    guess = guess.astype(np.float32)
    if method == "FAST_GICP":
        o = oracle_lib.GicpOracle(max_correspondence_distance=2.5)
        r = _reg(method, gicp_max_correspondence_distance=2.5)
    else:
        o = oracle_lib.VgicpOracle(resolution=1.0)
        r = _reg(method, vgicp_resolution=1.0)
    o.set_target(tgt)
    o.set_source(src)
    ro = o.align(guess)
    r.setInputTarget(tgt)
    r.setInputSource(src)
    r.align(guess)
    assert r.hasConverged() == ro["converged"]
    assert r.last_result.iterations == ro["iterations"] and r.last_result.evaluations == ro["evaluations"]
    et, er = pose_error(r.getFinalTransformation(), ro["T"])
    assert et <= 1e-6 and er <= 1e-7, (et, er)          # double-precision optimiser: far inside the 1e-4 m / 1e-5 rad gate
    assert pose_error(r.getFinalTransformation(), Tgt)[0] < 0.05
    fo, _, _ = oracle_lib.fitness_score(tgt, src, r.getFinalTransformation())
    assert abs(r.getFitnessScore() - fo) <= 1e-11 * fo
