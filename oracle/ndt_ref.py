"""ORACLE -- TEST INFRASTRUCTURE ONLY.  **Parity unpinned** (see oracle/oracle.py).

Independent numpy float64 statement of the NDT score function (Magnusson 2009 eq. 6.9-6.10; SURVEY.md
App. A), written directly from the formulas rather than from the C++ restatement, so that finite
differences of it can check the restatement's analytic gradient / Hessian tables and its voxel model.
Pure numpy; sized for clouds of a few thousand points.
"""
from __future__ import annotations

import numpy as np


def gauss_constants(resolution: float, outlier_ratio: float = 0.55):
    c1 = 10.0 * (1.0 - outlier_ratio)
    c2 = outlier_ratio / resolution ** 3
    d3 = -np.log(c2)
    d1 = -np.log(c1 + c2) - d3
    d2 = -2.0 * np.log((-np.log(c1 * np.exp(-0.5) + c2) - d3) / d1)
    return d1, d2


def rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


class VoxelModel:
    """Voxel-Gaussian target model: mean / covariance per >=6-point voxel, eigenvalue clamp 0.01*lmax."""

    def __init__(self, target_xyz: np.ndarray, resolution: float, min_points: int = 6, eig_mult: float = 0.01):
        xyz = np.asarray(target_xyz, np.float32)[:, :3]
        self.res = np.float32(resolution)
        inv = np.float32(1.0) / self.res
        self.min_b = np.floor(xyz.min(0) * inv).astype(np.int64)
        self.max_b = np.floor(xyz.max(0) * inv).astype(np.int64)
        self.div_b = self.max_b - self.min_b + 1
        ijk = (np.floor(xyz * inv) - self.min_b.astype(np.float32)).astype(np.int64)
        key = ijk[:, 0] + self.div_b[0] * (ijk[:, 1] + self.div_b[1] * ijk[:, 2])
        self.cells = {}
        self.all_counts = {}
        x64 = xyz.astype(np.float64)
        for k in np.unique(key):
            pts = x64[key == k]
            n = pts.shape[0]
            self.all_counts[int(k)] = n
            if n < min_points:
                continue
            mean = pts.mean(0)
            # PCL's single-pass form == biased covariance * (n-1)/n
            cov = (pts.T @ pts / n - np.outer(mean, mean)) * ((n - 1.0) / n)
            ev, V = np.linalg.eigh(cov)
            if ev[0] < 0 or ev[1] < 0 or ev[2] <= 0:
                continue
            lo = eig_mult * ev[2]
            if ev[0] < lo:
                ev = ev.copy()
                ev[0] = lo
                if ev[1] < lo:
                    ev[1] = lo
                cov = V @ np.diag(ev) @ np.linalg.inv(V)
            self.cells[int(k)] = (mean, cov, np.linalg.inv(cov), n)

    def lookup(self, ijk):
        if np.any(ijk < self.min_b) or np.any(ijk > self.max_b):
            return None
        r = ijk - self.min_b
        return self.cells.get(int(r[0] + self.div_b[0] * (r[1] + self.div_b[1] * r[2])))


_OFF7 = np.array([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])


def neighbour_sets(model: VoxelModel, source_xyz, p0, offsets=_OFF7):
    """Fix each point's neighbour-voxel set at pose p0 (so finite differences see a smooth function)."""
    R = rot_xyz(*p0[3:])
    xt = (np.asarray(source_xyz, np.float64)[:, :3] @ R.T + p0[:3]).astype(np.float32)
    ijk = np.floor(xt / model.res).astype(np.int64)
    sets = []
    for i in range(xt.shape[0]):
        cells = []
        for o in offsets:
            c = model.lookup(ijk[i] + o)
            if c is not None:
                cells.append(c)
        sets.append(cells)
    return sets


def score(model: VoxelModel, source_xyz, p, sets, outlier_ratio: float = 0.55) -> float:
    """score(p) = sum_i sum_{v in N(i)} -d1 exp(-d2/2 q^T Sigma^-1 q), q = T(p) x_i - mu_v   (float64)."""
    d1, d2 = gauss_constants(float(model.res), outlier_ratio)
    R = rot_xyz(*p[3:])
    xt = np.asarray(source_xyz, np.float64)[:, :3] @ R.T + np.asarray(p[:3], np.float64)
    s = 0.0
    for i, cells in enumerate(sets):
        for (mean, _cov, icov, _n) in cells:
            q = xt[i] - mean
            s += -d1 * np.exp(-0.5 * d2 * (q @ icov @ q))
    return s
