"""2-D <-> 3-D transform flattening used by the loop detector's initial guess.

Mirrors /root/reference/src/hdl_graph_slam/ros_utils.cpp:94-144 (normalize_euler_angs, transform2Dto3D,
transform3Dto2D); the Euler extraction is Eigen 3.3's eulerAngles(0,1,2) in float.
"""
from __future__ import annotations

import numpy as np

__all__ = ["transform2Dto3D", "transform2Dto3D_batch", "transform3Dto2D", "euler_angles_012", "normalize_euler_angs"]


def euler_angles_012(R) -> np.ndarray:
    """Eigen::Matrix3f::eulerAngles(0, 1, 2): R = Rx(a0) Ry(a1) Rz(a2), first angle in [0, pi]."""
    m = np.asarray(R, dtype=np.float32)
    f = np.float32
    r0 = np.arctan2(m[1, 2], m[2, 2]).astype(f)
    c2 = np.sqrt(m[0, 0] * m[0, 0] + m[0, 1] * m[0, 1]).astype(f)
    if r0 > 0:
        r0 = f(r0 - f(np.pi))
        r1 = np.arctan2(-m[0, 2], -c2).astype(f)
    else:
        r1 = np.arctan2(-m[0, 2], c2).astype(f)
    s1, c1 = np.sin(r0).astype(f), np.cos(r0).astype(f)
    r2 = np.arctan2(s1 * m[2, 0] - c1 * m[1, 0], c1 * m[1, 1] - s1 * m[2, 1]).astype(f)
    return np.array([-r0, -r1, -r2], dtype=np.float32)


def normalize_euler_angs(e) -> np.ndarray:
    """ros_utils.cpp:94-103: pick the smaller-norm of e and e -/+ pi per component."""
    e = np.asarray(e, dtype=np.float32)
    en = (e - np.float32(np.pi) * np.where(e >= 0, 1, -1)).astype(np.float32)
    return en if np.linalg.norm(en) < np.linalg.norm(e) else e


def transform2Dto3D(trans2D) -> np.ndarray:
    """ros_utils.cpp:105-126: yaw + xy only, z = roll = pitch = 0."""
    t = np.asarray(trans2D, dtype=np.float32)
    ang = np.arctan2(t[1, 0], t[0, 0]).astype(np.float32)   # Eigen::Rotation2Df(mat).angle()
    c, s = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    T = np.eye(4, dtype=np.float32)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1] = c, -s, s, c
    T[0, 3], T[1, 3] = t[0, 2], t[1, 2]
    return T


def transform2Dto3D_batch(trans2D) -> np.ndarray:
    """transform2Dto3D over a stack [n,3,3] -> [n,4,4] float32 (same arithmetic, one numpy pass)."""
    t = np.asarray(trans2D, dtype=np.float32)
    ang = np.arctan2(t[:, 1, 0], t[:, 0, 0]).astype(np.float32)
    c, s = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    T = np.zeros((t.shape[0], 4, 4), dtype=np.float32)
    T[:, 0, 0], T[:, 0, 1], T[:, 1, 0], T[:, 1, 1] = c, -s, s, c
    T[:, 2, 2] = 1.0
    T[:, 3, 3] = 1.0
    T[:, 0, 3], T[:, 1, 3] = t[:, 0, 2], t[:, 1, 2]
    return T


def transform3Dto2D(trans3D) -> np.ndarray:
    """ros_utils.cpp:128-144."""
    T = np.asarray(trans3D, dtype=np.float32)
    e = normalize_euler_angs(euler_angles_012(T[:3, :3]))
    c, s = np.cos(e[2]).astype(np.float32), np.sin(e[2]).astype(np.float32)
    out = np.eye(3, dtype=np.float32)
    out[0, 0], out[0, 1], out[1, 0], out[1, 1] = c, -s, s, c
    out[0, 2], out[1, 2] = T[0, 3], T[1, 3]
    return out
