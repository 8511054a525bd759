"""Randomised cross-check of the two round-2 index changes, beyond the fixed cases of the test suite:
  * k-NN covariances through the wave-per-leaf search (default) against the per-query walk (DGS_KNN_LEAF=0);
  * nearest-neighbour answers over the k-d ordered target index (DGS_NN_KD_ALL=1) against the Hilbert ordered one.
Clouds: random sizes 1 .. 150,000 and shapes (gaussian blobs, planes, lines, lattices with duplicates, clusters far apart, huge
offsets).  usage: python scripts/fuzz_index.py [cases]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from delta_graph_slam_amd import _lib as L
from delta_graph_slam_amd.registration import Registration


def make(env, method="NDT_OMP", **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return Registration(method, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def cloud(rng):
    n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 5000), rng.integers(5000, 150000)]))
    kind = rng.integers(0, 7)
    p = rng.normal(0, rng.uniform(0.1, 30), (n, 3))
    if kind == 1:
        p[:, 2] *= 1e-3                                              # plane
    elif kind == 2:
        p[:, 1:] = 0                                                 # line
    elif kind == 3:
        p = np.round(p / 0.5) * 0.5                                  # lattice: duplicates, ties
    elif kind == 4:
        p[: n // 10] += rng.uniform(200, 2000, 3)                    # a far cluster
    elif kind == 5:
        p += rng.uniform(-1e4, 1e4, 3)                               # far from the origin
    elif kind == 6:
        p = p[rng.integers(0, max(n // 20, 1), n)]                   # a few distinct points many times
    c = np.ones((n, 4), np.float32)
    c[:, :3] = p
    return c, kind


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(2026)
    h = Registration("FAST_GICP")                      # a GICP handle builds no voxel grid: any extent is fine
    kd = make({"DGS_NN_KD_ALL": "1"}, "FAST_GICP")
    bad = 0
    for it in range(cases):
        c, kind = cloud(rng)
        n = c.shape[0]
        q = np.ones((4000, 4), np.float32)
        q[:, :3] = c[rng.integers(0, n, 4000), :3] + rng.normal(0, rng.uniform(0.01, 5), (4000, 3)).astype(np.float32)
        h.setInputTarget(c)
        kd.setInputTarget(c)
        ih, dh = h.nearestKSearch(q)
        ik, dk = kd.nearestKSearch(q)
        ok_nn = np.array_equal(ih, ik) and np.array_equal(dh, dk)
        k = int(rng.choice([5, 10, 20, 32]))
        a = make({"DGS_KNN_LEAF": "1"}, "FAST_GICP", gicp_regularization=L.GICP_REG["NONE"], gicp_correspondence_randomness=k)
        b = make({"DGS_KNN_LEAF": "0"}, "FAST_GICP", gicp_regularization=L.GICP_REG["NONE"], gicp_correspondence_randomness=k)
        for r in (a, b):
            r.setInputTarget(c)
            r.setInputSource(c)
        ca, cb = a.gicp_covariances("source", n), b.gicp_covariances("source", n)
        scale = np.maximum(np.abs(cb).max(axis=(1, 2)), 1e-300)
        err = np.abs(ca - cb).max(axis=(1, 2)) / scale
        ok_knn = bool(err.max() < 1e-10)
        a.close(); b.close()
        print(it, 'n', n, 'kind', int(kind), 'k', k, 'nn', ok_nn, 'knn', ok_knn, 'max err %.2e' % err.max(), flush=True)
        bad += (not ok_nn) + (not ok_knn)
    print('failures', bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
