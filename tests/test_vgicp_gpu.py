"""-m gpu parity tests for FAST_VGICP: HIP (through the C ABI) vs the CPU restatement on identical inputs."""
import numpy as np
import pytest

from delta_graph_slam_amd import synth
from tests.helpers import pose_error

pytestmark = pytest.mark.gpu


def _pair(oracle_lib, tgt, src, search="DIRECT1", res=1.0, **kw):
    from delta_graph_slam_amd import _lib as L
    from delta_graph_slam_amd.registration import Registration
    o = oracle_lib.VgicpOracle(resolution=res, search_method=search, transformation_epsilon=kw.get("eps", 0.01), optimizer=kw.get("optimizer", 1),
                               k_correspondences=kw.get("k", 20))
    o.set_target(tgt)
    o.set_source(src)
    r = Registration("FAST_VGICP", vgicp_resolution=res, vgicp_search_method=L.VGICP_SEARCH[search], transformation_epsilon=kw.get("eps", 0.01),
                     gicp_optimizer=kw.get("optimizer", 1), gicp_correspondence_randomness=kw.get("k", 20))
    r.setInputTarget(tgt)
    r.setInputSource(src)
    return o, r


@pytest.mark.parametrize("res", [1.0, 0.35])
def test_voxelmap_matches_oracle(oracle_lib, res):
    tgt, src, _ = synth.planar_pair(n=8192)
    o, r = _pair(oracle_lib, tgt, src, res=res)
    co, no, mo, vo = o.voxels()
    cg, ng, mg, vg = r.vgicp_voxels()
    assert np.array_equal(co, cg) and np.array_equal(no, ng)
    assert np.array_equal(mo, mg)                    # same sums in the same (point-index) order
    err = np.abs(vo - vg).max(axis=(1, 2)) / np.abs(vo).max(axis=(1, 2))
    assert np.quantile(err, 0.99) < 1e-9             # covariances inherit the k-NN tie caveat of test_gicp_gpu
    assert err.max() < 1e-3


@pytest.mark.parametrize("search", ["DIRECT1", "DIRECT7", "DIRECT27"])
def test_linearize_and_error_match_oracle(oracle_lib, search):
    tgt, src, _ = synth.planar_pair(n=8192)
    o, r = _pair(oracle_lib, tgt, src, search=search)
    for t, rot in (((0, 0, 0), (0, 0, 0)), ((0.25, -0.08, 0.04), (0.01, -0.015, 0.04))):
        T = synth.make_transform(t, rot)
        eo, Ho, bo = o.linearize(T)
        eg, Hg, bg = r.gicp_linearize(T)
        assert abs(eo - eg) <= 1e-8 * abs(eo)
        assert np.abs(Ho - Hg).max() <= 1e-8 * np.abs(Ho).max()
        assert np.abs(bo - bg).max() <= 1e-8 * np.abs(bo).max()
        T2 = synth.make_transform((t[0] + 0.01, t[1], t[2] - 0.005), (rot[0], rot[1] + 0.002, rot[2]))
        assert abs(o.compute_error(T2) - r.gicp_linearize(T2, error_only=True)[0]) <= 1e-8 * abs(eo)


@pytest.mark.parametrize("search,optimizer", [("DIRECT1", 1), ("DIRECT7", 1), ("DIRECT1", 0)])
def test_align_cfg1_matches_oracle(oracle_lib, search, optimizer):
    tgt, src, Tgt = synth.planar_pair()
    o, r = _pair(oracle_lib, tgt, src, search=search, optimizer=optimizer)
    ro = o.align()
    r.align()
    assert r.hasConverged() == ro["converged"]
    assert r.last_result.iterations == ro["iterations"] and r.last_result.evaluations == ro["evaluations"]
    dt, dr = pose_error(r.getFinalTransformation(), ro["T"])
    assert dt <= 1e-6 and dr <= 1e-7, (dt, dr)
    dt, dr = pose_error(r.getFinalTransformation(), Tgt)
    assert dt < 0.01 and dr < 1e-3
    fit = r.getFitnessScore()
    fo, _, _ = oracle_lib.fitness_score(tgt, src, r.getFinalTransformation())
    assert abs(fit - fo) <= 1e-12 * fo


def test_align_street_scan_with_guess(oracle_lib):
    tgt, src, Tgt = synth.kitti_pair(n_points=16384)
    guess = Tgt.astype(np.float32).copy()
    guess[0, 3] -= 0.3
    guess[1, 3] += 0.1
    o, r = _pair(oracle_lib, tgt, src, search="DIRECT7")
    ro = o.align(guess)
    r.align(guess)
    assert r.hasConverged() == ro["converged"] and r.last_result.iterations == ro["iterations"]
    dt, dr = pose_error(r.getFinalTransformation(), ro["T"])
    assert dt <= 1e-6 and dr <= 1e-7, (dt, dr)
    dt, dr = pose_error(r.getFinalTransformation(), Tgt)
    assert dt < 0.05 and dr < 5e-3


def test_batch_and_resident_clouds(oracle_lib):
    from delta_graph_slam_amd.registration import Registration
    tgt, src, _ = synth.planar_pair(n=8192)
    rng = np.random.default_rng(2)
    sources = [src[rng.permutation(8192)[:m]].copy() for m in (8192, 5000, 700)] + [np.zeros((0, 4), np.float32)]
    guesses = [synth.make_transform(rng.uniform(-0.1, 0.1, 3), rng.uniform(-0.02, 0.02, 3)).astype(np.float32) for _ in sources]
    r = Registration("FAST_VGICP")
    r.setInputTarget(tgt)
    res = r.align_batch(sources, guesses, compute_fitness=True)
    assert res[3]["status"] == 4 and not res[3]["converged"]
    o = oracle_lib.VgicpOracle(resolution=1.0)
    o.set_target(tgt)
    for k in range(3):
        o.set_source(sources[k])
        ro = o.align(guesses[k])
        assert res[k]["converged"] == ro["converged"] and res[k]["iterations"] == ro["iterations"]
        dt, dr = pose_error(res[k]["T"], ro["T"])
        assert dt <= 1e-6 and dr <= 1e-7, (k, dt, dr)
    clouds = [r.make_cloud(s) for s in sources]
    res2 = r.align_batch(clouds, guesses, compute_fitness=True)
    for a, b in zip(res, res2):
        assert np.array_equal(a["T"], b["T"]) and a["converged"] == b["converged"]
        assert (a["fitness"] == b["fitness"]) or (np.isnan(a["fitness"]) and np.isnan(b["fitness"]))
    # a new target invalidates the voxel map
    r.setInputTarget(tgt[:4000])
    assert r.vgicp_voxels()[1].sum() == 4000


def test_factory_selects_vgicp():
    from delta_graph_slam_amd.registration import select_registration_method
    tgt, src, Tgt = synth.planar_pair(n=4096)
    reg = select_registration_method({"registration_method": "FAST_VGICP", "reg_resolution": 1.0})
    assert reg.params.method == 2 and reg.params.vgicp_resolution == 1.0
    reg.setInputTarget(tgt)
    reg.setInputSource(src)
    reg.align()
    assert reg.hasConverged()
    dt, dr = pose_error(reg.getFinalTransformation(), Tgt)
    assert dt < 0.01 and dr < 1e-3


def test_committed_goldens_on_the_device():
    import os
    from delta_graph_slam_amd import _lib as L
    from delta_graph_slam_amd.registration import Registration
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vgicp_small.npz"))
    for search in ("DIRECT1", "DIRECT7", "DIRECT27"):
        r = Registration("FAST_VGICP", vgicp_resolution=1.0, vgicp_search_method=L.VGICP_SEARCH[search])
        r.setInputTarget(G["tgt"])
        r.setInputSource(G["src"])
        if search == "DIRECT1":
            coords, counts, means, covs = r.vgicp_voxels()
            assert np.array_equal(coords, G["vox_coords"]) and np.array_equal(counts, G["vox_counts"])
            assert np.array_equal(means, G["vox_means"])
        e, H, b = r.gicp_linearize(G["T1"])
        assert abs(e - float(G[f"{search}_lin_err"])) <= 1e-8 * abs(e)
        assert np.abs(H - G[f"{search}_lin_H"]).max() <= 1e-8 * np.abs(H).max()
        r.align()
        assert [r.last_result.iterations, r.last_result.evaluations, int(r.hasConverged())] == list(G[f"{search}_iters"])
        dt, dr = pose_error(r.getFinalTransformation(), G[f"{search}_T"])
        assert dt <= 1e-6 and dr <= 1e-7
