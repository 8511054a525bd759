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


def _ulp_shift(G, k):
    """Move the guess translation by k float32 ulps (x up, y down): a perturbation below the input's own resolution."""
    Gp = np.asarray(G, np.float32).copy()
    for _ in range(abs(k)):
        Gp[0, 3] = np.nextafter(Gp[0, 3], np.float32(np.inf if k > 0 else -np.inf))
        Gp[1, 3] = np.nextafter(Gp[1, 3], np.float32(-np.inf if k > 0 else np.inf))
    return Gp


def ndt_oracle_band(orc, tgt, src, guess=None, twins=None, **kw):
    """The reference algorithm's own reproducibility on one pair: the largest deviation of the oracle's answer under
    perturbations that carry no information -- the same source compiled with FMA contraction, the host libm's expf instead
    of the platform-independent one, and the float32 initial guess moved by +-1 and +-2 ulps.  `twins` selects a subset
    (tuples (perturbed build, exp_libm, ulps)).  Returns (result of the unperturbed oracle, band_translation, band_rotation)."""
    G = np.eye(4, dtype=np.float32) if guess is None else np.asarray(guess, np.float32)
    if twins is None:
        twins = ((True, 0, 0), (False, 1, 0), (False, 0, 1), (False, 0, -1), (False, 0, 2), (False, 0, -2))
    runs = []
    for perturbed, libm, k in ((False, 0, 0),) + tuple(twins):
        o = orc.NdtOracle(perturbed=perturbed, exp_libm=libm, **kw)
        o.set_target(tgt)
        o.set_source(src)
        runs.append(o.align(_ulp_shift(G, k)))
    bt = max(pose_error(r["T"], runs[0]["T"])[0] for r in runs[1:])
    br = max(pose_error(r["T"], runs[0]["T"])[1] for r in runs[1:])
    return runs[0], bt, br
