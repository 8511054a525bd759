"""CPU tests pinning the NDT oracle (C++ restatement of pclomp::NormalDistributionsTransform).

The reference holds no fixtures for this path ("parity unpinned"), so the oracle is pinned by
(i) an independent numpy float64 statement of the score function and its finite differences, (ii) known-answer cases,
(iii) numpy.linalg for its small linear algebra, (iv) committed self-goldens (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from delta_graph_slam_amd import synth
from oracle import ndt_ref
from oracle import oracle as orc
from tests.helpers import pose_error

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "registration_small.npz"))


@pytest.fixture(scope="module")
def small():
    tgt, src, Tgt = synth.planar_pair(n=3000)
    return tgt, src, Tgt


def test_gauss_constants_and_voxel_model_vs_numpy(small):
    tgt, src, _ = small
    o = orc.NdtOracle(resolution=2.0)
    o.set_target(tgt)
    o.set_source(src)
    model = ndt_ref.VoxelModel(tgt, 2.0)
    v = o.voxels()
    assert v["valid"].sum() == len(model.cells)
    assert sorted(model.all_counts) == list(v["keys"])
    assert [model.all_counts[k] for k in v["keys"]] == list(v["counts"])
    for key, mean, icov, ok in zip(v["keys"], v["mean"], v["icov"], v["valid"]):
        if ok:
            m, _cov, ic, _n = model.cells[int(key)]
            assert np.allclose(mean, m, rtol=0, atol=1e-12)
            assert np.allclose(icov, ic, rtol=1e-8, atol=1e-10)
    mn, mx, dv = o.grid()
    assert np.array_equal(mn, model.min_b) and np.array_equal(dv, model.div_b)


def test_score_gradient_hessian_vs_finite_differences(small):
    tgt, src, _ = small
    p0 = np.array([0.2, -0.05, 0.03, 0.02, -0.03, 0.04])
    model = ndt_ref.VoxelModel(tgt, 2.0)
    sets = ndt_ref.neighbour_sets(model, src, p0)
    f = lambda p: ndt_ref.score(model, src, p, sets)
    for fix in (1, 0):
        o = orc.NdtOracle(resolution=2.0, fix_hessian_d1=fix)
        o.set_target(tgt)
        o.set_source(src)
        s, g, H = o.derivatives(p0)
        assert abs(s - f(p0)) < 5e-7 * abs(s)                                  # float per-point math vs float64
        g_fd = np.array([(f(p0 + d) - f(p0 - d)) / 2e-5 for d in np.eye(6) * 1e-5])
        assert np.abs(g - g_fd).max() < 2e-6 * np.abs(g_fd).max()
        # Hessian by Richardson-extrapolated central differences of the float64 score
        def fdH(h):
            Hf = np.zeros((6, 6))
            for i in range(6):
                for j in range(i, 6):
                    di, dj = np.eye(6)[i] * h, np.eye(6)[j] * h
                    Hf[i, j] = Hf[j, i] = (f(p0 + di + dj) - f(p0 + di - dj) - f(p0 - di + dj) + f(p0 - di - dj)) / (4 * h * h)
            return Hf
        H_fd = (4 * fdH(2e-4) - fdH(4e-4)) / 3
        rel = np.abs(H - H_fd) / np.abs(H_fd).max()
        if fix:
            assert rel.max() < 5e-6
        else:
            # upstream's h_ang_d1 z-term (+sy, Magnusson eq. 6.21 as printed) perturbs H(ry, ry) only
            assert rel[4, 4] > 1e-5
            rel[4, 4] = 0
            assert rel.max() < 5e-6
        assert np.abs(H - H.T).max() < 1e-6 * np.abs(H).max()


def test_identical_clouds_stay_at_identity(small):
    tgt, _, _ = small
    o = orc.NdtOracle(resolution=2.0)
    o.set_target(tgt)
    o.set_source(tgt)
    r = o.align()
    assert r["converged"]
    dt, dr = pose_error(r["T"], np.eye(4))
    assert dt < 2e-2 and dr < 2e-3     # the NDT optimum sits within a small fraction of the 2 m voxel of the true pose


def test_known_rigid_motion_is_recovered_without_noise():
    rng = np.random.default_rng(5)
    tgt = synth._xyz1(synth._planar_surfaces(6000, rng, 0.02))
    T = synth.make_transform((0.25, -0.15, 0.05), (0.01, -0.015, 0.04))
    src = synth.apply_transform(np.linalg.inv(T), tgt)                          # exactly the target, moved
    o = orc.NdtOracle(resolution=2.0, transformation_epsilon=1e-5)
    o.set_target(tgt)
    o.set_source(src)
    r = o.align()
    dt, dr = pose_error(r["T"], T)
    assert r["converged"] and dt < 2e-2 and dr < 2e-3


def test_euler_and_pose_matrix_helpers():
    for r in [(0.01, -0.02, 0.05), (-0.3, 0.2, -0.1), (-2.0, 0.4, 1.0)]:
        T = synth.make_transform((1, 2, 3), r).astype(np.float32)
        e = orc.euler_angles_012(T)
        assert np.allclose(synth.euler_to_matrix(*e.astype(np.float64)), T[:3, :3], atol=3e-6)
        P = orc.pose_to_matrix_f32(np.array([1, 2, 3, *r], float))
        assert np.allclose(P, T, atol=2e-7) and P.dtype == np.float32


def test_small_linear_algebra_vs_numpy():
    rng = np.random.default_rng(0)
    for _ in range(20):
        A = rng.normal(size=(6, 6))
        A = A + A.T
        b = rng.normal(size=6)
        assert np.allclose(orc.svd_solve6(A, b), np.linalg.solve(A, b), rtol=1e-9, atol=1e-11)
        Apd = A @ A.T + np.eye(6)
        assert np.allclose(orc.ldlt_solve6(Apd, b), np.linalg.solve(Apd, b), rtol=1e-10, atol=1e-12)
        S = rng.normal(size=(3, 3))
        S = S @ S.T
        ev, V = orc.sym_eig3(S)
        assert np.allclose(ev, np.linalg.eigvalsh(S), rtol=1e-12, atol=1e-14)
        assert np.allclose(V @ np.diag(ev) @ V.T, S, atol=1e-12)
    # rank-deficient: JacobiSVD::solve gives the minimum-norm least-squares solution
    A = np.diag([3.0, 2.0, 1.0, 0.0, 0.0, 0.0])
    b = np.arange(1.0, 7.0)
    assert np.allclose(orc.svd_solve6(A, b), np.linalg.pinv(A) @ b, atol=1e-12)
    assert np.allclose(orc.svd_solve6(np.zeros((6, 6)), b), 0)


def test_line_search_modes_and_iteration_cap(small):
    tgt, src, Tgt = small
    res = {}
    for ls in (0, 1):
        o = orc.NdtOracle(resolution=2.0, line_search=ls)
        o.set_target(tgt)
        o.set_source(src)
        res[ls] = o.align()
    assert res[0]["evaluations"] == res[0]["iterations"] + 1          # fixed step: one evaluation per iteration
    assert res[1]["evaluations"] >= res[1]["iterations"] + 1
    for r in res.values():
        assert pose_error(r["T"], Tgt)[0] < 0.05
    o = orc.NdtOracle(resolution=2.0, max_iterations=2, transformation_epsilon=1e-9)
    o.set_target(tgt)
    o.set_source(src)
    r = o.align()
    assert r["iterations"] == 4 and r["converged"]          # nr_iterations_ > max_iterations_ is tested before the increment


def test_matches_committed_goldens():
    tgt, src = GOLD["tgt"], GOLD["src"]
    poses = np.array([[0, 0, 0, 0, 0, 0], [0.2, -0.05, 0.03, 0.02, -0.03, 0.04]], float)
    for search in ("DIRECT7", "DIRECT1", "KDTREE"):
        o = orc.NdtOracle(resolution=2.0, search_method=search)
        o.set_target(tgt)
        o.set_source(src)
        for k, p in enumerate(poses):
            s, g, H = o.derivatives(p)
            assert np.isclose(s, GOLD[f"ndt_{search}_p{k}_score"], rtol=1e-12)
            assert np.allclose(g, GOLD[f"ndt_{search}_p{k}_grad"], rtol=1e-10, atol=1e-9)
            assert np.allclose(H, GOLD[f"ndt_{search}_p{k}_hess"], rtol=1e-10, atol=1e-8)
        r = o.align()
        assert np.array_equal(np.array([r["iterations"], r["evaluations"], int(r["converged"])]), GOLD[f"ndt_{search}_iters"])
        assert np.allclose(r["T"], GOLD[f"ndt_{search}_T"], atol=1e-6)
        assert np.allclose(r["trajectory"], GOLD[f"ndt_{search}_traj"], atol=1e-6)
    v = orc.NdtOracle(resolution=2.0)
    v.set_target(tgt)
    vox = v.voxels()
    assert np.array_equal(vox["keys"], GOLD["ndt_vox_keys"]) and np.array_equal(vox["counts"], GOLD["ndt_vox_counts"])
    assert np.allclose(vox["mean"], GOLD["ndt_vox_mean"], atol=1e-12)


def test_approx_voxel_grid_restatement_vs_a_plain_python_pass():
    """pcl::ApproximateVoxelGrid (oracle/cpu/voxelgrid_cpu.cpp) against an independent statement of the same pass written with a
    Python dict of table slots: same points, same order, same float bits."""
    from oracle import oracle as orc
    rng = np.random.default_rng(3)
    cloud = np.ones((4000, 4), np.float32)
    cloud[:, :3] = (rng.normal(0, 6, (4000, 3)) * [1, 1, 0.2]).astype(np.float32)
    cloud[1000:2000, :3] = cloud[:1000, :3] + np.float32(0.01)          # revisits of the same cells much later
    for leaf in (0.25, 1.0, 3.0):
        inv = np.float32(1.0) / np.float32(leaf)
        table = {}
        out = []

        def flush(e):
            out.append([e[4] / np.float32(e[3]), e[5] / np.float32(e[3]), e[6] / np.float32(e[3]), np.float32(1.0)])

        for p in cloud:
            ix, iy, iz = (int(np.floor(p[a] * inv)) for a in range(3))
            slot = (ix * 7171 + iy * 3079 + iz * 4231) & 511
            e = table.get(slot)
            if e is not None and e[3] and (e[0], e[1], e[2]) != (ix, iy, iz):
                flush(e)
                e = None
            if e is None:
                e = [ix, iy, iz, 0, np.float32(0), np.float32(0), np.float32(0)]
                table[slot] = e
            e[0], e[1], e[2] = ix, iy, iz
            e[3] += 1
            e[4] = np.float32(e[4] + p[0]); e[5] = np.float32(e[5] + p[1]); e[6] = np.float32(e[6] + p[2])
        for slot in sorted(table):
            if table[slot][3]:
                flush(table[slot])
        want = np.array(out, np.float32)
        got = orc.approx_voxel_grid(cloud, leaf)
        assert got.shape == want.shape and np.array_equal(got, want)
        assert got.shape[0] < cloud.shape[0]
