"""Synthetic Velodyne-shaped clouds for the registration hot path.

Every cloud is float32 [N,4] (x, y, z, 1.0) -- the pcl::PointXYZ 16-byte layout the
reference hands to setInputTarget/setInputSource
(/root/reference/apps/scan_matching_odometry_nodelet.cpp:117-118,180,185).
All generators are seeded and deterministic; there is no dataset download.

Configs (BASELINE.json "configs"):
  cfg1  planar+noise 16k pair           -> planar_pair()
  cfg2  HDL-64E KITTI-shaped 65,536 pair -> kitti_pair()
  cfg3  VLP-16 stream (ragged ~30k)      -> vlp16_stream()
  cfg4  loop batch: 1 target + K sources -> loop_batch()
  cfg5  dense indoor 200k pair           -> indoor_pair()
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "euler_to_matrix", "make_transform", "apply_transform", "planar_pair", "street_scan",
    "kitti_pair", "vlp16_stream", "loop_batch", "indoor_pair", "voxel_centroid_downsample",
]


# ----------------------------------------------------------------------------- transforms
def euler_to_matrix(rx: float, ry: float, rz: float) -> np.ndarray:
    """R = Rx(rx) @ Ry(ry) @ Rz(rz) (the NDT pose parameterisation)."""
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def make_transform(t, r) -> np.ndarray:
    """4x4 float64 from translation (3) and euler angles (rx, ry, rz)."""
    T = np.eye(4)
    T[:3, :3] = euler_to_matrix(*r)
    T[:3, 3] = t
    return T


def apply_transform(T: np.ndarray, cloud: np.ndarray) -> np.ndarray:
    """T (4x4) applied to an [N,4] xyz1 cloud, result float32 xyz1."""
    xyz = cloud[:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    out = np.ones((cloud.shape[0], 4), dtype=np.float32)
    out[:, :3] = xyz.astype(np.float32)
    return out


def _xyz1(xyz: np.ndarray) -> np.ndarray:
    out = np.ones((xyz.shape[0], 4), dtype=np.float32)
    out[:, :3] = xyz.astype(np.float32)
    return out


# ----------------------------------------------------------------------------- cfg1
def _planar_surfaces(n: int, rng: np.random.Generator, sigma: float) -> np.ndarray:
    """20x20 m ground + two perpendicular 20x4 m walls (~50/25/25 %), N(0, sigma) noise."""
    n_g = n // 2
    n_w1 = n // 4
    n_w2 = n - n_g - n_w1
    g = np.stack([rng.uniform(-10, 10, n_g), rng.uniform(-10, 10, n_g), np.zeros(n_g)], 1)
    w1 = np.stack([rng.uniform(-10, 10, n_w1), np.full(n_w1, 10.0), rng.uniform(0, 4, n_w1)], 1)
    w2 = np.stack([np.full(n_w2, 10.0), rng.uniform(-10, 10, n_w2), rng.uniform(0, 4, n_w2)], 1)
    pts = np.concatenate([g, w1, w2], 0)
    pts += rng.normal(0.0, sigma, pts.shape)
    return pts


def planar_pair(n: int = 16384, seed_target: int = 1, seed_source: int = 2, sigma: float = 0.02,
                t_gt=(0.30, -0.10, 0.05), r_gt=(0.01, -0.02, 0.05)):
    """cfg1: returns (target, source, T_gt) with target ~= T_gt @ source."""
    tgt = _planar_surfaces(n, np.random.default_rng(seed_target), sigma)
    src_world = _planar_surfaces(n, np.random.default_rng(seed_source), sigma)
    T = make_transform(t_gt, r_gt)
    Tinv = np.linalg.inv(T)
    src = src_world @ Tinv[:3, :3].T + Tinv[:3, 3]
    return _xyz1(tgt), _xyz1(src), T


# ----------------------------------------------------------------------------- street scene
_FACADE_Y = 9.0
_FACADE_H = 12.0
_GAP_PERIOD = 15.0
_GAP_WIDTH = 3.0


def _street_scene(seed: int = 7):
    """12 car-sized boxes, 20 pole cylinders and 48 facade buttresses (bay fronts of random width / depth that break
    the along-street symmetry of the two planar facades), fixed by seed: the scene is shared by all scans."""
    rng = np.random.default_rng(seed)
    boxes = []
    for k in range(12):
        cx = -66.0 + 12.0 * k + rng.uniform(-2, 2)
        side = 1.0 if k % 2 == 0 else -1.0
        cy = side * rng.uniform(3.5, 6.0)
        lx, ly, lz = rng.uniform(3.8, 4.8), rng.uniform(1.6, 1.9), rng.uniform(1.4, 1.8)
        boxes.append((cx - lx / 2, cy - ly / 2, 0.0, cx + lx / 2, cy + ly / 2, lz))
    for k in range(48):
        side = 1.0 if k % 2 == 0 else -1.0
        cx = -118.0 + 5.0 * k + rng.uniform(-2.0, 2.0)
        w, d, hgt = rng.uniform(0.8, 3.0), rng.uniform(0.3, 1.2), rng.uniform(3.0, _FACADE_H)
        y0, y1 = (_FACADE_Y - d, _FACADE_Y) if side > 0 else (-_FACADE_Y, -_FACADE_Y + d)
        boxes.append((cx - w / 2, y0, 0.0, cx + w / 2, y1, hgt))
    poles = []
    for k in range(20):
        px = -76.0 + 8.0 * k + rng.uniform(-1, 1)
        side = 1.0 if k % 2 == 0 else -1.0
        poles.append((px, side * 7.5, 0.15, 6.0))
    return np.array(boxes), np.array(poles)


def _az_span(origin, yaw, xs, ys, n_az):
    """Inclusive azimuth-index span [i0, i0 + cnt) (modulo n_az) of the rays that can see points (xs, ys)."""
    ang = np.arctan2(ys - origin[1], xs - origin[0]) - yaw
    ref = ang[0]
    rel = np.mod(ang - ref + np.pi, 2 * np.pi) - np.pi          # unwrap around the first corner
    lo, hi = ref + rel.min(), ref + rel.max()
    step = 2 * np.pi / n_az
    i0 = int(np.floor(lo / step)) - 1
    cnt = int(np.ceil((hi - lo) / step)) + 3
    return i0, min(cnt, n_az)


def _raycast(origin: np.ndarray, yaw: float, dirs: np.ndarray, boxes: np.ndarray, poles: np.ndarray) -> np.ndarray:
    """Range to the nearest surface along unit dirs [beams, n_az, 3] (world frame) from origin; inf on miss. float64.
    Boxes and poles are only tested against the azimuth sector they subtend."""
    nb, na, _ = dirs.shape
    t_best = np.full((nb, na), np.inf)
    ox, oy, oz = origin
    dx, dy, dz = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        # ground z = 0
        t = np.where(dz < 0, -oz / dz, np.inf)
        t_best = np.minimum(t_best, np.where(t > 0, t, np.inf))
        # facades y = +-9 with door gaps
        for ysign in (1.0, -1.0):
            t = (ysign * _FACADE_Y - oy) / dy
            hx = ox + t * dx
            hz = oz + t * dz
            ok = (t > 0) & (hz >= 0) & (hz <= _FACADE_H) & (np.mod(hx, _GAP_PERIOD) >= _GAP_WIDTH)
            t_best = np.minimum(t_best, np.where(ok, t, np.inf))
        # boxes (slab test) on their azimuth sector
        for b in boxes:
            inside = b[0] <= ox <= b[3] and b[1] <= oy <= b[4]
            if inside:
                cols = np.arange(na)
            else:
                i0, cnt = _az_span(origin, yaw, np.array([b[0], b[0], b[3], b[3]]), np.array([b[1], b[4], b[1], b[4]]), na)
                cols = np.mod(np.arange(i0, i0 + cnt), na)
            d = dirs[:, cols, :]
            inv = 1.0 / d
            t0 = (b[:3] - origin) * inv
            t1 = (b[3:] - origin) * inv
            tmin = np.minimum(t0, t1).max(axis=2)
            tmax = np.maximum(t0, t1).min(axis=2)
            ok = (tmax >= tmin) & (tmin > 0)
            t_best[:, cols] = np.minimum(t_best[:, cols], np.where(ok, tmin, np.inf))
        # vertical cylinders on their azimuth sector
        for (px, py, r, h) in poles:
            i0, cnt = _az_span(origin, yaw, np.array([px - r, px + r, px - r, px + r]), np.array([py - r, py - r, py + r, py + r]), na)
            cols = np.mod(np.arange(i0, i0 + cnt), na)
            ddx, ddy, ddz = dx[:, cols], dy[:, cols], dz[:, cols]
            a = ddx * ddx + ddy * ddy
            fx, fy = ox - px, oy - py
            bq = 2 * (fx * ddx + fy * ddy)
            cq = fx * fx + fy * fy - r * r
            disc = bq * bq - 4 * a * cq
            t = (-bq - np.sqrt(np.maximum(disc, 0.0))) / (2 * a)
            hz = oz + t * ddz
            ok = (disc >= 0) & (t > 0) & (hz >= 0) & (hz <= h)
            t_best[:, cols] = np.minimum(t_best[:, cols], np.where(ok, t, np.inf))
    return t_best.reshape(-1)


def voxel_centroid_downsample(xyz: np.ndarray, leaf: float) -> np.ndarray:
    """pcl::VoxelGrid-style centroid filter (used upstream of the hot path,
    /root/reference/apps/scan_matching_odometry_nodelet.cpp:83-89). Output ordered by voxel key."""
    ijk = np.floor(xyz / leaf).astype(np.int64)
    ijk -= ijk.min(0)
    dims = ijk.max(0) + 1
    key = ijk[:, 0] + dims[0] * (ijk[:, 1] + dims[1] * ijk[:, 2])
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    xyz_s = xyz[order]
    heads = np.flatnonzero(np.r_[True, key_s[1:] != key_s[:-1]])
    counts = np.diff(np.r_[heads, key_s.size])
    sums = np.add.reduceat(xyz_s, heads, axis=0)
    return sums / counts[:, None]


def street_scan(pose_xy_yaw, beams: int, elev_deg, azimuths: int, seed: int, range_sigma: float = 0.02,
                rmin: float = 0.1, rmax: float = 100.0, sensor_z: float = 1.73, scene_seed: int = 7):
    """Ray-cast one spinning-LiDAR scan of the synthetic street; points in the SENSOR frame.

    Returns (xyz float64 [M,3], T_world_sensor 4x4).  Range clip matches
    /root/reference/launch/delta_graph_slam.launch:31-33 (distance filter 0.1-100 m).
    """
    boxes, poles = _street_scene(scene_seed)
    x, y, yaw = pose_xy_yaw
    elev = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], beams))
    az = np.linspace(0.0, 2 * np.pi, azimuths, endpoint=False)
    E, A = np.meshgrid(elev, az, indexing="ij")
    d_s = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
    T = make_transform((x, y, sensor_z), (0.0, 0.0, yaw))
    d_w = d_s @ T[:3, :3].T
    t = _raycast(T[:3, 3], yaw, d_w.reshape(beams, azimuths, 3), boxes, poles)
    rng = np.random.default_rng(seed)
    t = t + rng.normal(0.0, range_sigma, t.shape)
    ok = np.isfinite(t) & (t >= rmin) & (t <= rmax)
    return d_s[ok] * t[ok, None], T


def _fix_count(xyz: np.ndarray, n: int, seed: int) -> np.ndarray:
    """Deterministic thinning (seeded choice, order preserved) or padding (jittered repeats)."""
    m = xyz.shape[0]
    rng = np.random.default_rng(seed)
    if m >= n:
        keep = np.sort(rng.choice(m, n, replace=False))
        return xyz[keep]
    extra = rng.choice(m, n - m, replace=True)
    pad = xyz[extra] + rng.normal(0.0, 0.01, (n - m, 3))
    return np.concatenate([xyz, pad], 0)


def hdl64_scan(pose_xy_yaw, seed: int, n_points: int = 65536, leaf: float = 0.075, azimuths: int = 4096):
    """HDL-64E-shaped scan (64 beams, +2.0..-24.8 deg) -> voxel filter -> exactly n_points."""
    xyz, T = street_scan(pose_xy_yaw, 64, (2.0, -24.8), azimuths, seed)
    xyz = voxel_centroid_downsample(xyz, leaf)
    return _xyz1(_fix_count(xyz, n_points, seed + 1000)), T


def kitti_pair(n_points: int = 65536, seed_target: int = 10, seed_source: int = 11):
    """cfg2: target at the origin pose, source 1.0 m ahead + 2 deg yaw. T_gt maps source->target."""
    tgt, Ta = hdl64_scan((0.0, 0.0, 0.0), seed_target, n_points)
    src, Tb = hdl64_scan((1.0, 0.0, np.deg2rad(2.0)), seed_source, n_points)
    return tgt, src, np.linalg.inv(Ta) @ Tb


def vlp16_stream(n_frames: int = 100, seed: int = 20, speed_range=(1.0, 10.0), hz: float = 10.0):
    """cfg3: VLP-16-shaped frames (16 beams +-15 deg x 1875 azimuths = 30,000 rays; sky rays miss so
    clouds are ragged) along an S-curve. Returns (list of clouds, list of T_world_sensor)."""
    clouds, poses = [], []
    x = -40.0
    for k in range(n_frames):
        s = k / max(n_frames - 1, 1)
        v = speed_range[0] + (speed_range[1] - speed_range[0]) * 0.5 * (1 - np.cos(2 * np.pi * s))
        x += v / hz
        y = 2.0 * np.sin(2 * np.pi * s)
        yaw = np.arctan2(2.0 * 2 * np.pi * np.cos(2 * np.pi * s) / max(n_frames - 1, 1), v / hz)
        xyz, T = street_scan((x, y, yaw), 16, (15.0, -15.0), 1875, seed + k)
        clouds.append(_xyz1(xyz))
        poses.append(T)
    return clouds, poses


def loop_batch(n_candidates: int = 256, n_points: int = 65536, seed: int = 40, radius: float = 15.0,
               distinct_scans: int | None = None):
    """cfg4: one target keyframe + n_candidates source keyframes sampled within `radius` m.

    Guesses follow the reference loop detector (loop_detector.hpp:139-143): ground-truth relative pose
    perturbed by U(+-1 m, +-5 deg yaw), then flattened to yaw/xy.  When `distinct_scans` < n_candidates
    the scans are re-used round-robin (each re-use still gets its own guess perturbation).
    Returns (target, [sources], guesses [K,4,4] f32, T_gt [K,4,4] f64).
    """
    rng = np.random.default_rng(seed)
    tgt, Ta = hdl64_scan((0.0, 0.0, 0.0), seed, n_points)
    k_scans = n_candidates if distinct_scans is None else min(distinct_scans, n_candidates)
    scans = []
    for k in range(k_scans):
        r = radius * np.sqrt(rng.uniform())
        th = rng.uniform(0, 2 * np.pi)
        # keep the sensor on the carriageway (|y| < 3 m) so it is inside the street canyon
        pose = (r * np.cos(th), np.clip(r * np.sin(th), -3.0, 3.0), rng.uniform(-0.3, 0.3))
        s, Tb = hdl64_scan(pose, seed + 1 + k, n_points)
        scans.append((s, np.linalg.inv(Ta) @ Tb))
    sources, guesses, gts = [], [], []
    for c in range(n_candidates):
        s, Tgt = scans[c % k_scans]
        yaw_gt = np.arctan2(Tgt[1, 0], Tgt[0, 0])
        gx = Tgt[0, 3] + rng.uniform(-1.0, 1.0)
        gy = Tgt[1, 3] + rng.uniform(-1.0, 1.0)
        gyaw = yaw_gt + np.deg2rad(rng.uniform(-5.0, 5.0))
        G = make_transform((gx, gy, 0.0), (0.0, 0.0, gyaw)).astype(np.float32)
        sources.append(s)
        guesses.append(G)
        gts.append(Tgt)
    return tgt, sources, np.stack(guesses), np.stack(gts)


# ----------------------------------------------------------------------------- cfg5
def _indoor_surfaces(n: int, rng: np.random.Generator, sigma: float, scene_seed: int = 5) -> np.ndarray:
    """20x15x3 m room (floor, ceiling, 4 walls) + 10 furniture boxes, sampled uniformly by area."""
    srng = np.random.default_rng(scene_seed)
    rects = []  # (origin, u, v) parallelograms
    L, W, H = 20.0, 15.0, 3.0
    rects.append(((0, 0, 0), (L, 0, 0), (0, W, 0)))
    rects.append(((0, 0, H), (L, 0, 0), (0, W, 0)))
    rects.append(((0, 0, 0), (L, 0, 0), (0, 0, H)))
    rects.append(((0, W, 0), (L, 0, 0), (0, 0, H)))
    rects.append(((0, 0, 0), (0, W, 0), (0, 0, H)))
    rects.append(((L, 0, 0), (0, W, 0), (0, 0, H)))
    for _ in range(10):
        bx, by = srng.uniform(1, L - 3), srng.uniform(1, W - 3)
        lx, ly, lz = srng.uniform(0.5, 2.0), srng.uniform(0.5, 2.0), srng.uniform(0.4, 1.8)
        rects.append(((bx, by, lz), (lx, 0, 0), (0, ly, 0)))            # top
        rects.append(((bx, by, 0), (lx, 0, 0), (0, 0, lz)))
        rects.append(((bx, by + ly, 0), (lx, 0, 0), (0, 0, lz)))
        rects.append(((bx, by, 0), (0, ly, 0), (0, 0, lz)))
        rects.append(((bx + lx, by, 0), (0, ly, 0), (0, 0, lz)))
    o = np.array([r[0] for r in rects], float)
    u = np.array([r[1] for r in rects], float)
    v = np.array([r[2] for r in rects], float)
    area = np.linalg.norm(np.cross(u, v), axis=1)
    which = rng.choice(len(rects), n, p=area / area.sum())
    a, b = rng.uniform(size=n), rng.uniform(size=n)
    pts = o[which] + a[:, None] * u[which] + b[:, None] * v[which]
    pts -= np.array([L / 2, W / 2, 0.0])  # sensor-centred
    pts += rng.normal(0.0, sigma, pts.shape)
    return pts


def indoor_pair(n: int = 200000, seed_target: int = 50, seed_source: int = 51, sigma: float = 0.01,
                t_gt=(0.10, 0.05, 0.0), r_gt=(0.0, 0.0, 0.03)):
    """cfg5: dense indoor pair; returns (target, source, T_gt)."""
    tgt = _indoor_surfaces(n, np.random.default_rng(seed_target), sigma)
    src_world = _indoor_surfaces(n, np.random.default_rng(seed_source), sigma)
    T = make_transform(t_gt, r_gt)
    Tinv = np.linalg.inv(T)
    src = src_world @ Tinv[:3, :3].T + Tinv[:3, 3]
    return _xyz1(tgt), _xyz1(src), T
