"""CPU tests pinning the FAST_VGICP restatement (oracle/cpu/vgicp_cpu.cpp) with independent numpy statements.
The reference ships no fixtures for this path: parity unpinned."""
import numpy as np
import pytest

from delta_graph_slam_amd import synth


@pytest.fixture(scope="module")
def pair():
    return synth.planar_pair(n=4096)


def _numpy_voxelmap(tgt, covs, res):
    coord = np.floor(tgt[:, :3].astype(np.float64) / res - 0.5).astype(np.int64)
    keys, inv = np.unique(coord, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    order = np.lexsort((keys[:, 0], keys[:, 1], keys[:, 2]))
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    V = keys.shape[0]
    counts = np.bincount(inv, minlength=V)
    means = np.zeros((V, 3))
    cov = np.zeros((V, 3, 3))
    np.add.at(means, inv, tgt[:, :3].astype(np.float64))
    np.add.at(cov, inv, covs)
    means /= counts[:, None]
    cov /= counts[:, None, None]
    return keys[order], counts[order], means[order], cov[order]


@pytest.mark.parametrize("res", [1.0, 0.7])
def test_voxelmap_is_the_mean_of_points_and_covariances(oracle_lib, pair, res):
    tgt, src, _ = pair
    o = oracle_lib.VgicpOracle(resolution=res)
    o.set_target(tgt)
    o.set_source(src)
    coords, counts, means, covs = o.voxels()
    k, c, m, cv = _numpy_voxelmap(tgt, o.covariances("target"), res)
    assert np.array_equal(coords, k) and np.array_equal(counts, c)
    assert np.allclose(means, m, rtol=1e-13, atol=1e-13)
    assert np.allclose(covs, cv, rtol=1e-12, atol=1e-15)
    assert counts.sum() == tgt.shape[0]


@pytest.mark.parametrize("search,noff", [("DIRECT1", 1), ("DIRECT7", 7), ("DIRECT27", 27)])
def test_cost_matches_a_numpy_statement(oracle_lib, pair, search, noff):
    """E = sum over (point, offset voxel) of sqrt(n_voxel) e^T (C_voxel + R C_p R^T)^-1 e, written independently."""
    tgt, src, _ = pair
    o = oracle_lib.VgicpOracle(resolution=1.0, search_method=search)
    o.set_target(tgt)
    o.set_source(src)
    T = synth.make_transform((0.12, -0.07, 0.03), (0.01, -0.02, 0.03))
    e, H, b = o.linearize(T)
    coords, counts, means, covs = o.voxels()
    lut = {tuple(c): i for i, c in enumerate(coords)}
    cs = o.covariances("source")
    p = src[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    base = np.floor(p / 1.0 - 0.5).astype(np.int64)
    if noff == 1:
        offs = [(0, 0, 0)]
    elif noff == 7:
        offs = [(0, 0, 0), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    else:
        offs = [(i - 1, j - 1, k - 1) for i in range(3) for j in range(3) for k in range(3)]
    R = T[:3, :3]
    total = 0.0
    n_corr = 0
    for i in range(src.shape[0]):
        RCR = R @ cs[i] @ R.T
        for d in offs:
            v = lut.get((base[i, 0] + d[0], base[i, 1] + d[1], base[i, 2] + d[2]))
            if v is None:
                continue
            err = means[v] - p[i]
            total += np.sqrt(counts[v]) * err @ np.linalg.solve(covs[v] + RCR, err)
            n_corr += 1
    assert n_corr > 0.5 * src.shape[0]
    assert abs(total - e) <= 1e-9 * abs(total)
    # with the correspondences frozen, dE/dxi = 2 b (left perturbation exp(xi) T) and H is the Gauss-Newton matrix
    g = np.zeros(6)
    for k in range(6):
        d = np.zeros(6)
        d[k] = 1e-6
        ep = o.compute_error(oracle_lib.se3_exp(d) @ T)
        em = o.compute_error(oracle_lib.se3_exp(-d) @ T)
        g[k] = (ep - em) / 2e-6
    assert np.abs(g - 2 * b).max() <= 1e-6 * np.abs(b).max()
    assert np.allclose(H, H.T, rtol=1e-12) and np.all(np.linalg.eigvalsh(H) > 0)


def test_more_offsets_give_more_correspondences_and_the_same_optimum(oracle_lib, pair):
    tgt, src, Tgt = pair
    res = {}
    for search in ("DIRECT1", "DIRECT7", "DIRECT27"):
        o = oracle_lib.VgicpOracle(resolution=1.0, search_method=search)
        o.set_target(tgt)
        o.set_source(src)
        r = o.align()
        assert r["converged"]
        dt = np.linalg.norm(r["T"][:3, 3] - Tgt[:3, 3])
        assert dt < 5e-3, (search, dt)
        res[search] = o.linearize(r["T"].astype(np.float64))[0]
    assert res["DIRECT1"] < res["DIRECT7"] < res["DIRECT27"]


def test_identical_clouds_stay_at_identity(oracle_lib, pair):
    tgt, _, _ = pair
    o = oracle_lib.VgicpOracle(resolution=1.0)
    o.set_target(tgt)
    o.set_source(tgt)
    r = o.align()
    assert r["converged"] and np.abs(r["T"] - np.eye(4)).max() < 1e-3      # point-to-voxel-mean: near, not at, identity


def test_set_target_rebuilds_the_voxelmap(oracle_lib, pair):
    tgt, src, _ = pair
    o = oracle_lib.VgicpOracle(resolution=1.0)
    o.set_target(tgt)
    o.set_source(src)
    n1 = o.voxels()[1].sum()
    o.set_target(tgt[:1000])
    assert o.voxels()[1].sum() == 1000 and n1 == tgt.shape[0]


def test_matches_committed_goldens(oracle_lib):
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "vgicp_small.npz"))
    for search in ("DIRECT1", "DIRECT7", "DIRECT27"):
        o = oracle_lib.VgicpOracle(resolution=1.0, search_method=search)
        o.set_target(G["tgt"])
        o.set_source(G["src"])
        if search == "DIRECT1":
            coords, counts, means, covs = o.voxels()
            assert np.array_equal(coords, G["vox_coords"]) and np.array_equal(counts, G["vox_counts"])
            assert np.allclose(means, G["vox_means"], rtol=0, atol=1e-12) and np.allclose(covs, G["vox_covs"], rtol=1e-10, atol=1e-14)
        e, H, b = o.linearize(G["T1"])
        assert abs(e - float(G[f"{search}_lin_err"])) <= 1e-10 * abs(e)
        assert np.allclose(H, G[f"{search}_lin_H"], rtol=1e-10, atol=1e-9) and np.allclose(b, G[f"{search}_lin_b"], rtol=1e-10, atol=1e-9)
        r = o.align()
        assert [r["iterations"], r["evaluations"], int(r["converged"])] == list(G[f"{search}_iters"])
        assert np.allclose(r["T"], G[f"{search}_T"], rtol=0, atol=2e-6)
