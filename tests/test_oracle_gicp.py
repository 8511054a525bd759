"""CPU tests pinning the FAST_GICP / fitness-score oracle (parity unpinned upstream, see test_oracle_ndt.py)."""
import os

import numpy as np
import pytest
from scipy.linalg import expm
from scipy.spatial import cKDTree

from delta_graph_slam_amd import synth
from oracle import oracle as orc
from tests.helpers import f32_sqdist, f32_transform, pose_error

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "registration_small.npz"))


@pytest.fixture(scope="module")
def small():
    return synth.planar_pair(n=3000)


def test_knn_is_exact(small):
    tgt, src, _ = small
    idx, d2 = orc.knn(tgt, src[:1500], 20)
    dd, ii = cKDTree(tgt[:, :3].astype(np.float64)).query(src[:1500, :3].astype(np.float64), k=20)
    assert np.mean([set(a) == set(b) for a, b in zip(idx, ii)]) > 0.999
    assert np.all(np.diff(d2, axis=1) >= 0)                                   # ascending
    assert np.array_equal(d2, f32_sqdist(src[:1500, None, :3], tgt[idx, :3]))  # FLANN's float accumulation order


def test_covariance_regularisations(small):
    tgt, src, _ = small
    idx, _ = orc.knn(src, src, 20)
    nb = src[idx, :3].astype(np.float64)
    raw = np.einsum("nki,nkj->nij", nb - nb.mean(1, keepdims=True), nb - nb.mean(1, keepdims=True)) / 20
    g = orc.GicpOracle(regularization="NONE")
    g.set_target(tgt)
    g.set_source(src)
    assert np.allclose(g.covariances("source"), raw, rtol=1e-10, atol=1e-14)
    g = orc.GicpOracle(regularization="PLANE")
    g.set_target(tgt)
    g.set_source(src)
    cov = g.covariances("source")
    ev = np.linalg.eigvalsh(cov)
    assert np.allclose(ev, [1e-3, 1, 1], atol=1e-9)
    # the flattened direction is the raw covariance's smallest eigenvector
    n_raw = np.linalg.eigh(raw)[1][:, :, 0]
    n_reg = np.linalg.eigh(cov)[1][:, :, 0]
    assert np.quantile(np.abs(np.abs(np.einsum("ni,ni->n", n_raw, n_reg)) - 1), 0.99) < 1e-6
    g = orc.GicpOracle(regularization="FROBENIUS")
    g.set_target(tgt)
    g.set_source(src)
    Ci = np.linalg.inv(raw + 1e-3 * np.eye(3))
    ref = np.linalg.inv(Ci / np.linalg.norm(Ci, axis=(1, 2), keepdims=True))
    assert np.allclose(g.covariances("source"), ref, rtol=1e-8, atol=1e-12)


def test_se3_exp_vs_expm():
    rng = np.random.default_rng(2)
    for a in list(rng.normal(size=(10, 6))) + [np.zeros(6), np.array([1e-7, 0, 0, 1, 2, 3.0])]:
        W = np.zeros((4, 4))
        W[:3, :3] = [[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]]
        W[:3, 3] = a[3:]
        assert np.allclose(orc.se3_exp(a), expm(W), atol=1e-12)


def test_linearize_gradient_is_consistent_with_the_error(small):
    """E(exp(xi) T) = sum e^T M e with fixed correspondences: dE/dxi at 0 is 2 b, and H is the Gauss-Newton matrix."""
    tgt, src, _ = small
    g = orc.GicpOracle(max_correspondence_distance=2.0)
    g.set_target(tgt)
    g.set_source(src)
    T = synth.make_transform((0.25, -0.08, 0.04), (0.01, -0.015, 0.04))
    E0, H, b = g.linearize(T)
    assert abs(g.compute_error(T) - E0) < 1e-9 * E0
    grad = np.array([(g.compute_error(orc.se3_exp(d) @ T) - g.compute_error(orc.se3_exp(-d) @ T)) / 2e-6 for d in np.eye(6) * 1e-6])
    assert np.abs(grad - 2 * b).max() < 1e-5 * np.abs(b).max()
    assert np.allclose(H, H.T, rtol=1e-12) and np.all(np.linalg.eigvalsh(H) > 0)
    corr, sq = g.correspondences()
    xt = f32_transform(T.astype(np.float32), src)
    _, nn = cKDTree(tgt[:, :3].astype(np.float64)).query(xt.astype(np.float64))
    ok = corr >= 0
    assert np.mean(corr[ok] == nn[ok]) > 0.999 and np.all(sq[ok] < 4.0) and np.all(sq[~ok] >= 4.0)


def test_align_recovers_known_motion_and_matches_goldens():
    tgt, src, Tgt = GOLD["tgt"], GOLD["src"], GOLD["T_gt"]
    for reg in ("PLANE", "FROBENIUS"):
        g = orc.GicpOracle(regularization=reg, max_correspondence_distance=2.0)
        g.set_target(tgt)
        g.set_source(src)
        assert np.allclose(g.covariances("source"), GOLD[f"gicp_{reg}_cov_source"], rtol=1e-9, atol=1e-12)
        e, H, b = g.linearize(np.eye(4))
        assert np.isclose(e, GOLD[f"gicp_{reg}_lin_err"], rtol=1e-10)
        assert np.allclose(H, GOLD[f"gicp_{reg}_lin_H"], rtol=1e-9) and np.allclose(b, GOLD[f"gicp_{reg}_lin_b"], rtol=1e-9, atol=1e-9)
        r = g.align()
        assert np.array_equal(np.array([r["iterations"], r["evaluations"], int(r["converged"])]), GOLD[f"gicp_{reg}_iters"])
        assert np.allclose(r["T"], GOLD[f"gicp_{reg}_T"], atol=1e-6)
        dt, dr = pose_error(r["T"], Tgt)
        assert r["converged"] and dt < 0.02 and dr < 3e-3


def test_gauss_newton_and_lm_agree_on_an_easy_pair(small):
    tgt, src, Tgt = small
    out = []
    for opt in (0, 1):
        g = orc.GicpOracle(optimizer=opt, max_correspondence_distance=2.0, transformation_epsilon=1e-4, rotation_epsilon=1e-5)
        g.set_target(tgt)
        g.set_source(src)
        out.append(g.align())
    dt, dr = pose_error(out[0]["T"], out[1]["T"])
    assert dt < 1e-3 and dr < 1e-4


def test_fitness_score_semantics(small):
    """pcl::Registration::getFitnessScore: squared distances compared with max_range (information_matrix_calculator.cpp:97)."""
    tgt, src, Tgt = small
    T = Tgt.astype(np.float32)
    xt = f32_transform(T, src)
    _, nn = cKDTree(tgt[:, :3].astype(np.float64)).query(xt.astype(np.float64))
    d2 = f32_sqdist(xt, tgt[nn, :3]).astype(np.float64)
    s, n, inl = orc.fitness_score(tgt, src, T)
    assert n == src.shape[0] and abs(s - d2.mean()) < 1e-9 * s and inl == int((d2 < 0.25).sum())
    s, n, _ = orc.fitness_score(tgt, src, T, max_range=0.01)
    assert n == int((d2 <= 0.01).sum()) and abs(s - d2[d2 <= 0.01].mean()) < 1e-9 * s
    s, n, _ = orc.fitness_score(tgt, src, T, max_range=-1.0)
    assert n == 0 and s == np.finfo(np.float64).max
    assert np.allclose(GOLD["fitness"], orc.fitness_score(GOLD["tgt"], GOLD["src"], GOLD["gicp_PLANE_T"]), rtol=1e-9)
