"""-m gpu parity tests for the NDT hot path: HIP (through the C ABI) vs the CPU oracle on identical inputs."""
import numpy as np
import pytest

from delta_graph_slam_amd import synth
from tests.helpers import TOL_ROT, TOL_TRANS, f32_sqdist, f32_transform, pose_error

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[1, 0], ids=["upstream_order", "fast_order"])
def reg_cls(request):
    """The registration factory with the NDT evaluation order pinned: every test of this file runs in the default (upstream) order and
    in the opt-in fast order, against the same oracle numbers and the same tolerances."""
    from delta_graph_slam_amd.registration import Registration

    def make(method, **kw):
        if method == "NDT_OMP":
            kw.setdefault("ndt_strict_order", request.param)
        return Registration(method, **kw)
    return make


def _pair(reg_cls, oracle_lib, tgt, src, **kw):
    okw = dict(resolution=kw.get("ndt_resolution", 1.0), transformation_epsilon=kw.get("transformation_epsilon", 0.01),
               search_method=kw.get("search", "DIRECT7"), line_search=kw.get("ndt_line_search", 1),
               fix_hessian_d1=kw.get("ndt_fix_hessian_d1", 0), max_iterations=kw.get("maximum_iterations", 64))
    o = oracle_lib.NdtOracle(**okw)
    o.set_target(tgt)
    o.set_source(src)
    from delta_graph_slam_amd import _lib as L
    gkw = {k: v for k, v in kw.items() if k != "search"}
    gkw.setdefault("ndt_resolution", 1.0)
    gkw["ndt_search_method"] = L.NDT_SEARCH[kw.get("search", "DIRECT7")]
    r = reg_cls("NDT_OMP", **gkw)
    r.setInputTarget(tgt)
    r.setInputSource(src)
    return o, r


def test_voxel_table_matches_oracle(reg_cls, oracle_lib):
    tgt, src, _ = synth.planar_pair(n=16384)
    o, r = _pair(reg_cls, oracle_lib, tgt, src)
    vo, vg = o.voxels(), r.ndt_voxels()
    assert np.array_equal(vo["keys"], vg["keys"])
    assert np.array_equal(vo["counts"], vg["counts"])
    assert np.array_equal(vo["valid"], vg["valid"])
    # moments are accumulated in the same (point-index) order in double -> means are bit-identical
    assert np.array_equal(vo["mean"], vg["mean"])
    v = vo["valid"]
    rel = np.abs(vo["icov"][v] - vg["icov"][v]).max() / np.abs(vo["icov"][v]).max()
    assert rel < 1e-9
    c = r.counts()
    assert c["valid_voxels"] == v.sum() and c["occupied_voxels"] == len(vo["keys"])


@pytest.mark.parametrize("search", ["DIRECT7", "DIRECT1", "DIRECT26", "KDTREE"])
def test_derivatives_match_oracle(reg_cls, oracle_lib, search):
    tgt, src, _ = synth.planar_pair(n=16384)
    o, r = _pair(reg_cls, oracle_lib, tgt, src, search=search)
    for p in ([0, 0, 0, 0, 0, 0], [0.2, -0.05, 0.03, 0.02, -0.03, 0.04], [0.3, -0.1, 0.05, 0.01, -0.02, 0.05]):
        so, go, Ho = o.derivatives(np.array(p, float))
        sg, gg, Hg = r.ndt_derivatives(np.array(p, float))
        assert abs(so - sg) <= 2e-6 * abs(so)
        # gradient / Hessian entries are sums with cancellation: compare against the largest entry
        assert np.abs(go - gg).max() <= 1e-5 * np.abs(go).max() + 1e-9, (p, go, gg)
        assert np.abs(Ho - Hg).max() <= 1e-5 * np.abs(Ho).max() + 1e-9, (p, np.abs(Ho - Hg).max() / np.abs(Ho).max())


@pytest.mark.parametrize("line_search", [0, 1])
def test_align_cfg1_matches_oracle(reg_cls, oracle_lib, line_search):
    tgt, src, Tgt = synth.planar_pair()
    o, r = _pair(reg_cls, oracle_lib, tgt, src, ndt_line_search=line_search)
    ro = o.align()
    r.align()
    assert r.hasConverged() == ro["converged"]
    assert r.last_result.iterations == ro["iterations"]
    assert r.last_result.evaluations == ro["evaluations"]
    dt, dr = pose_error(r.getFinalTransformation(), ro["T"])
    assert dt <= TOL_TRANS and dr <= TOL_ROT, (dt, dr)
    # and both are near the ground truth (sanity, not parity)
    dt, dr = pose_error(r.getFinalTransformation(), Tgt)
    assert dt < 0.02 and dr < 2e-3


def test_align_tight_epsilon_same_fixed_point(reg_cls, oracle_lib):
    tgt, src, _ = synth.planar_pair()
    o, r = _pair(reg_cls, oracle_lib, tgt, src, transformation_epsilon=1e-6)
    ro = o.align()
    r.align()
    dt, dr = pose_error(r.getFinalTransformation(), ro["T"])
    assert dt <= TOL_TRANS and dr <= TOL_ROT, (dt, dr)


def test_align_with_guess_and_aligned_cloud(reg_cls, oracle_lib):
    tgt, src, Tgt = synth.planar_pair()
    guess = synth.make_transform((0.25, -0.05, 0.0), (0.0, 0.0, 0.04)).astype(np.float32)
    o, r = _pair(reg_cls, oracle_lib, tgt, src)
    ro = o.align(guess)
    aligned = r.align(guess, want_aligned=True)
    dt, dr = pose_error(r.getFinalTransformation(), ro["T"])
    assert dt <= TOL_TRANS and dr <= TOL_ROT, (dt, dr)
    ref = f32_transform(r.getFinalTransformation(), src)
    assert np.abs(aligned[:, :3] - ref).max() < 1e-4
    assert np.all(aligned[:, 3] == 1.0)


def test_fitness_and_nn_exact(reg_cls, oracle_lib):
    from scipy.spatial import cKDTree
    tgt, src, _ = synth.planar_pair(n=8192)
    o, r = _pair(reg_cls, oracle_lib, tgt, src)
    r.align()
    T = r.getFinalTransformation()
    xt = f32_transform(T, src)
    tree = cKDTree(tgt[:, :3].astype(np.float64))
    _, nn = tree.query(xt.astype(np.float64), k=1)
    d2 = f32_sqdist(xt, tgt[nn, :3])
    # exact search: float distances to the GPU's answer can only be <= the kd-tree's (ties/rounding), never larger
    q = np.ones((xt.shape[0], 4), np.float32)
    q[:, :3] = xt
    idx, sq = r.nearestKSearch(q)
    d2_gpu_idx = f32_sqdist(xt, tgt[idx, :3])
    assert np.array_equal(sq, d2_gpu_idx)          # reported distance is the float distance to the reported index
    assert np.all(sq <= d2)                         # exactness
    assert (idx != nn).mean() < 1e-3                # differences only on float ties
    fit = r.getFitnessScore()
    assert abs(fit - float(np.mean(sq.astype(np.float64)))) <= 1e-12 * fit
    mr = 0.05
    fit_r = r.getFitnessScore(mr)
    sel = sq <= np.float32(mr)
    assert abs(fit_r - float(np.mean(sq[sel].astype(np.float64)))) <= 1e-12 * fit_r
    assert r.getFitnessScore(-1.0) == 1.7976931348623157e308   # DBL_MAX when nothing qualifies
    inl = r.getInlierFraction(0.25)
    assert abs(inl - float((sq < np.float32(0.25)).mean())) < 1e-12


def test_batch_matches_single(reg_cls, oracle_lib):
    tgt, src, _ = synth.planar_pair(n=8192)
    rng = np.random.default_rng(3)
    sources, guesses = [], []
    for k in range(5):
        n = int(rng.integers(3000, 8192))
        sources.append(src[:n].copy())
        guesses.append(synth.make_transform((0.3 + rng.uniform(-0.1, 0.1), -0.1 + rng.uniform(-0.1, 0.1), 0.0), (0, 0, 0.05 + rng.uniform(-0.02, 0.02))).astype(np.float32))
    sources.append(np.zeros((0, 4), np.float32))   # ragged edge: empty candidate
    guesses.append(np.eye(4, dtype=np.float32))
    o, r = _pair(reg_cls, oracle_lib, tgt, src)
    res = r.align_batch(sources, guesses, compute_fitness=True)
    assert res[-1]["status"] == 4 and not res[-1]["converged"]
    for k in range(5):
        r.setInputSource(sources[k])
        r.align(guesses[k])
        # same kernels; only the number of partial rows per pair (fixed-order double sums) differs between batch and single
        dt, dr = pose_error(res[k]["T"], r.getFinalTransformation())
        assert dt <= 1e-6 and dr <= 1e-7
        assert res[k]["converged"] == r.hasConverged()
        assert abs(res[k]["fitness"] - r.getFitnessScore()) <= 1e-6 * res[k]["fitness"]
        o.set_source(sources[k])
        ro = o.align(guesses[k])
        dt, dr = pose_error(res[k]["T"], ro["T"])
        assert dt <= TOL_TRANS and dr <= TOL_ROT, (k, dt, dr)
        assert res[k]["iterations"] == ro["iterations"]


def test_errors_surface_as_not_converged(reg_cls):
    r = reg_cls("NDT_OMP", ndt_resolution=1.0)
    assert r.align() is None and not r.hasConverged()       # no target: PCL prints an error and returns
    tgt, src, _ = synth.planar_pair(n=2048)
    r.setInputTarget(tgt)
    assert r.align() is None and not r.hasConverged()       # no source
    assert np.array_equal(r.getFinalTransformation(), np.eye(4, dtype=np.float32))


def test_calc_fitness_score_between_two_clouds(reg_cls, oracle_lib):
    """InformationMatrixCalculator::calc_fitness_score on the device == the oracle's PCL loop."""
    from delta_graph_slam_amd.information_matrix import InformationMatrixCalculator
    tgt, src, Tgt = synth.planar_pair(n=8192)
    r = reg_cls("NDT_OMP", ndt_resolution=1.0)
    r.setInputTarget(src)                                    # the registration's own state must survive the auxiliary call
    r.setInputSource(tgt)
    r.align()
    T_before, fit_before = r.getFinalTransformation(), r.getFitnessScore()
    for mr in (1.7976931348623157e308, 0.05):
        fo, n, _ = oracle_lib.fitness_score(tgt, src, Tgt.astype(np.float32), mr)
        fg = r.calc_fitness_score(tgt, src, Tgt, mr)
        assert abs(fg - fo) <= 1e-12 * fo
    assert r.calc_fitness_score(tgt, src, Tgt, -1.0) == 1.7976931348623157e308
    assert r.calc_fitness_score(tgt, np.zeros((0, 4), np.float32), Tgt) == 1.7976931348623157e308
    assert np.array_equal(r.getFinalTransformation(), T_before) and r.getFitnessScore() == fit_before
    calc = InformationMatrixCalculator({}, registration=r)
    inf = calc.calc_information_matrix(tgt, src, Tgt)
    fo, _, _ = oracle_lib.fitness_score(tgt, src, Tgt.astype(np.float32))
    wx = calc.weight(20.0, 0.5, 0.1 ** 2, 5.0 ** 2, fo)
    assert inf.shape == (3, 3) and abs(inf[0, 0] - 1.0 / np.float32(wx)) < 1e-9 and inf[0, 1] == 0
    const = InformationMatrixCalculator({"use_const_inf_matrix": True}).calc_information_matrix(None, None, None)
    assert np.allclose(np.diag(const), [2.0, 2.0, 10.0])


@pytest.mark.parametrize("leaf", [0.1, 0.25, 1.0])
def test_voxel_grid_filter_matches_oracle(reg_cls, oracle_lib, leaf):
    """pcl::VoxelGrid centroid down-sampling on the device, bit-exact against the restatement."""
    import torch
    xyz, _ = synth.street_scan((0.0, 0.0, 0.0), 64, (2.0, -24.8), 1024, 5)
    cloud = synth._xyz1(xyz)
    cloud[11, 1] = np.nan                                   # non-finite points are dropped, as with is_dense = false
    r = reg_cls("NDT_OMP")
    ref = oracle_lib.voxel_grid(cloud, leaf)
    out = r.voxel_grid_filter(cloud, leaf)
    assert out.shape == ref.shape and np.array_equal(out, ref)           # same cells, same order, same float sums
    dout = r.voxel_grid_filter(torch.from_numpy(cloud).cuda(), leaf)     # device in -> device out
    assert dout.is_cuda and np.array_equal(dout.cpu().numpy(), ref)
    assert r.voxel_grid_filter(np.zeros((0, 4), np.float32), leaf).shape == (0, 4)


def test_voxel_grid_filter_leaves_the_ndt_model_bookkeeping_alone(reg_cls):
    """The odometry flow down-samples on the same registration object between setInputTarget and align
    (scan_matching_odometry_nodelet.cpp:184): counts / voxel dump must still describe the NDT target afterwards."""
    tgt, src, _ = synth.planar_pair(n=16384)
    r = reg_cls("NDT_OMP", ndt_resolution=1.0)
    r.setInputTarget(tgt)
    before, vox_before = r.counts(), r.ndt_voxels()
    small = r.voxel_grid_filter(src, 0.5)
    assert 0 < small.shape[0] < src.shape[0]
    after, vox_after = r.counts(), r.ndt_voxels()
    assert before == after
    assert np.array_equal(vox_before["keys"], vox_after["keys"]) and np.array_equal(vox_before["mean"], vox_after["mean"])
    r2 = reg_cls("NDT_OMP", ndt_resolution=1.0)      # also when the counts were never read before the filter ran
    r2.setInputTarget(tgt)
    r2.voxel_grid_filter(src, 0.5)
    assert r2.counts() == before


def test_device_tensors_may_die_right_after_the_call(reg_cls, oracle_lib):
    """include/dgs_reg.h ordering contract: the library has finished reading a device buffer when set_input_* returns, so a
    temporary tensor may be recycled by the caching allocator immediately (the pattern of __graft_entry__.smoke())."""
    import torch
    tgt, src, _ = synth.planar_pair(n=16384)
    r = reg_cls("NDT_OMP", ndt_resolution=1.0)
    r.setInputTarget(torch.from_numpy(tgt).cuda())
    junk = [torch.full((16384, 4), float(k), device="cuda") for k in range(8)]      # re-uses the freed blocks at once
    r.setInputSource(torch.from_numpy(src).cuda())
    junk += [torch.full((16384, 4), -1.0, device="cuda") for _ in range(8)]
    torch.cuda.synchronize()
    r.align()
    h = reg_cls("NDT_OMP", ndt_resolution=1.0)
    h.setInputTarget(tgt)
    h.setInputSource(src)
    h.align()
    assert np.array_equal(r.getFinalTransformation(), h.getFinalTransformation()) and len(junk) == 16


@pytest.mark.parametrize("leaf", [0.1, 0.5, 2.0])
def test_approx_voxel_grid_filter_matches_oracle(reg_cls, oracle_lib, leaf):
    """pcl::ApproximateVoxelGrid (scan_matching_odometry_nodelet.cpp:90-96) on the device: the sequential history-table pass is
    reproduced exactly -- same centroids, same float bits, same output ORDER -- by sorts over slots and flush triggers."""
    import torch
    xyz, _ = synth.street_scan((0.0, 0.0, 0.0), 64, (2.0, -24.8), 1024, 5)
    scan = synth._xyz1(xyz)
    rng = np.random.default_rng(1)
    rnd = np.ones((30000, 4), np.float32)
    rnd[:, :3] = (rng.normal(0, 8, (30000, 3)) * [1, 1, 0.1]).astype(np.float32)
    one_cell = np.ones((700, 4), np.float32)
    one_cell[:, :3] = rng.uniform(0.01, 0.09, (700, 3))
    r = reg_cls("NDT_OMP")
    for cloud in (scan, rnd, rnd[rng.permutation(30000)], one_cell, rnd[:1]):
        ref = oracle_lib.approx_voxel_grid(cloud, leaf)
        out = r.voxel_grid_filter(cloud, leaf, approximate=True)
        assert out.shape == ref.shape and np.array_equal(out, ref)
        dout = r.voxel_grid_filter(torch.from_numpy(cloud).cuda(), leaf, approximate=True)
        assert dout.is_cuda and np.array_equal(dout.cpu().numpy(), ref)
    assert r.voxel_grid_filter(np.zeros((0, 4), np.float32), leaf, approximate=True).shape == (0, 4)


def test_odometry_runs_with_the_approximate_voxel_grid(reg_cls):
    """downsample_method = APPROX_VOXELGRID (scan_matching_odometry_nodelet.cpp:90-96) through the odometry mirror."""
    from delta_graph_slam_amd.odometry import ScanMatchingOdometry
    clouds, poses = synth.vlp16_stream(n_frames=5)
    odo = ScanMatchingOdometry(reg_cls("FAST_GICP", gicp_max_correspondence_distance=2.0),
                               dict(downsample_method="APPROX_VOXELGRID", downsample_resolution=0.2, keyframe_delta_trans=1.0, keyframe_delta_angle=1.0,
                                    keyframe_delta_time=1e9))
    ref = ScanMatchingOdometry(reg_cls("FAST_GICP", gicp_max_correspondence_distance=2.0),
                               dict(downsample_method="VOXELGRID", downsample_resolution=0.2, keyframe_delta_trans=1.0, keyframe_delta_angle=1.0,
                                    keyframe_delta_time=1e9))
    a = [odo.matching(0.1 * k, c) for k, c in enumerate(clouds)]       # matching() filters the scan itself (:184)
    b = [ref.matching(0.1 * k, c) for k, c in enumerate(clouds)]
    moved = np.linalg.norm(b[-1][:3, 3])
    assert moved > 0.05
    assert np.linalg.norm(a[-1][:3, 3] - b[-1][:3, 3]) < 0.05 + 0.1 * moved    # two down-samplings of the same scans: the same motion
