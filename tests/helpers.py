"""Shared test helpers: pose error metrics with the north-star tolerances."""
import numpy as np

TOL_TRANS = 1e-4  # metres  (BASELINE.json north_star: "within 1e-4 m / 1e-5 rad of reference")
TOL_ROT = 1e-5    # radians


def pose_error(Ta, Tb):
    """(translation error [m], rotation angle error [rad]) between two 4x4 transforms."""
    Ta = np.asarray(Ta, np.float64)
    Tb = np.asarray(Tb, np.float64)
    dt = np.linalg.norm(Ta[:3, 3] - Tb[:3, 3])
    R = Ta[:3, :3].T @ Tb[:3, :3]
    # angle from the skew part (accurate for tiny angles, unlike acos of the trace)
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(w)
    c = 0.5 * (np.trace(R) - 1.0)
    return dt, float(np.arctan2(s, c))


def f32_sqdist(a, b):
    """FLANN L2_Simple order in float32: (dx*dx + dy*dy) + dz*dz."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    d = a - b
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def f32_transform(T, xyz):
    """pcl::transformPointCloud in float32: ((m0 x + m1 y) + m2 z) + m3."""
    T = np.asarray(T, np.float32)
    x, y, z = (np.asarray(xyz[:, k], np.float32) for k in range(3))
    out = np.empty((xyz.shape[0], 3), np.float32)
    for r in range(3):
        out[:, r] = ((T[r, 0] * x + T[r, 1] * y) + T[r, 2] * z) + T[r, 3]
    return out


def ndt_oracle_band(orc, tgt, src, guess=None, twins=None, **kw):
    """The reference algorithm's own reproducibility on one pair (oracle.ndt_band): the largest deviation of the oracle's answer
    under perturbations that carry no information.  Returns (result of the unperturbed oracle, band_translation, band_rotation)."""
    return orc.ndt_band(tgt, src, guess, twins, **kw)


def sequential_best(converged, fitness):
    """The arg-min of LoopDetector::matching (loop_detector.hpp:126-156) as the reference runs it, one candidate after the other:
    a candidate is skipped iff it did not converge or its score is greater than the best so far (on a tie the LATER one wins)."""
    best, best_score = -1, 1.7976931348623157e308
    for c, (ok, s) in enumerate(zip(converged, fitness)):
        if (not ok) or s > best_score:
            continue
        best, best_score = c, s
    return best, best_score


_SHARDS = {}


def oracle_shard(orc, seed, n_candidates=32, n_points=65536, resolution=1.0):
    """One rank's shard of bench.py's workload (synth.loop_batch(seed = 40 + 1000 * rank), 32 distinct scans) with the oracle's
    sequential candidate loop over it: (target, sources, guesses, [oracle align results], [oracle fitness scores]).  Cached per
    process (three -m gpu modules share the shards)."""
    key = (seed, n_candidates, n_points, resolution)
    if key not in _SHARDS:
        from concurrent.futures import ThreadPoolExecutor
        from delta_graph_slam_amd import synth
        tgt, sources, guesses, _ = synth.loop_batch(n_candidates=n_candidates, n_points=n_points, seed=seed, distinct_scans=n_candidates)
        o = orc.NdtOracle(resolution=resolution)
        o.set_target(tgt)
        ref = []
        for c in range(n_candidates):
            o.set_source(sources[c])
            ref.append(o.align(guesses[c]))
        # pcl::Registration::getFitnessScore per candidate (single-threaded upstream; here a few at a time, each on one thread)
        with ThreadPoolExecutor(max_workers=8) as ex:
            fit = list(ex.map(lambda c: orc.fitness_score(tgt, sources[c], ref[c]["T"])[0], range(n_candidates)))
        _SHARDS[key] = (tgt, sources, guesses, ref, fit)
    return _SHARDS[key]
