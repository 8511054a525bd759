"""Host-side mirror of the reference's registration objects, over the C ABI (include/dgs_reg.h).

`Registration` keeps the pcl::Registration<PointXYZ,PointXYZ> method names the reference calls
(/root/reference/apps/scan_matching_odometry_nodelet.cpp:180,185,218,222,228,318,327 and
/root/reference/include/hdl_graph_slam/loop_detector.hpp:124,138,145,148,149,155), and
`select_registration_method` mirrors the factory /root/reference/src/hdl_graph_slam/registrations.cpp:22-124
(same `registration_method` strings, same `reg_*` parameter names and defaults).

Clouds are float32 [N,4] arrays (numpy on the host, or torch tensors already resident in HBM); transforms
are 4x4 float32 numpy arrays in ordinary row/column indexing (converted to Eigen's column-major at the ABI).
There is no CPU fallback: constructing a Registration without the HIP library / a gfx950 device raises.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib as L

try:  # torch is plumbing only: device memory + streams
    import torch
except Exception:  # pragma: no cover
    torch = None

__all__ = ["Registration", "RegistrationGroup", "DeviceCloud", "GroupCloud", "select_registration_method", "DgsError"]
DgsError = L.DgsError


def _is_tensor(x) -> bool:
    return torch is not None and isinstance(x, torch.Tensor)


def _cloud_ptr(cloud, sync: bool = True):
    """-> (pointer, n, on_device, keepalive).  sync=False: the caller orders torch's stream itself (once for a whole batch)."""
    if _is_tensor(cloud):
        if cloud.dtype != torch.float32 or cloud.dim() != 2 or cloud.shape[1] != 4:
            raise ValueError("cloud tensors must be float32 [N,4]")
        t = cloud.contiguous()
        if t.is_cuda:
            # the handle works on a stream of its own: whatever produced this tensor on torch's current stream must have finished
            # (include/dgs_reg.h, ordering contract for device pointers); the library is done with the buffer when its call returns
            if sync:
                torch.cuda.current_stream(t.device).synchronize()
            return C.c_void_p(t.data_ptr()), t.shape[0], 1, t
        a = t.numpy()
        return a.ctypes.data_as(C.c_void_p), a.shape[0], 0, a
    a = np.ascontiguousarray(cloud, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 4:
        raise ValueError("clouds must be float32 [N,4] (x, y, z, pad)")
    return a.ctypes.data_as(C.c_void_p), a.shape[0], 0, a


def _col16(T) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T.reshape(16))


def _from_col16(t16) -> np.ndarray:
    return np.array(t16, dtype=np.float32).reshape(4, 4).T.copy()


# numpy view of dgs_result (include/dgs_reg.h)
_RESULT_DTYPE = np.dtype([("T", np.float32, (16,)), ("converged", np.int32), ("iterations", np.int32), ("evaluations", np.int32),
                          ("status", np.int32), ("score", np.float64), ("fitness", np.float64)])
assert _RESULT_DTYPE.itemsize == C.sizeof(L.Result)


class DeviceCloud:
    """A cloud resident in HBM (dgs_cloud): KeyFrame::cloud of the loop detector kept on the device together with its
    NN index / GICP covariances, so a keyframe that is a loop candidate tick after tick is uploaded and indexed once."""

    def __init__(self, registration: "Registration", cloud):
        ptr, n, dev, keep = _cloud_ptr(cloud)
        self._lib = registration._lib
        self._c = C.c_void_p()
        registration._check(self._lib.dgs_cloud_create(registration._h, ptr, n, dev, C.byref(self._c)))
        self.n = n

    def __len__(self):
        return self.n

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self._lib.dgs_cloud_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Registration:
    """pcl::Registration-shaped object backed by one dgs_handle (HIP, gfx950)."""

    def __init__(self, method: str = "NDT_OMP", device: int | None = None, _borrow=None, lib_path: str | None = None, **params):
        lib = L.load(lib_path)   # lib_path: another build of the library (tests of the experiments build)
        self._lib = lib
        self._owns = _borrow is None
        if _borrow is not None:   # a dgs_group member's handle (RegistrationGroup.member): the group owns and destroys it
            self.method, self.params, self._h = method, None, C.c_void_p(_borrow)
            self._converged, self._final, self.last_result, self._keep = False, np.eye(4, dtype=np.float32), None, {}
            return
        exact = {"NDT_OMP": L.METHOD_NDT, "NDT_HIP": L.METHOD_NDT, "FAST_GICP": L.METHOD_GICP, "FAST_GICP_HIP": L.METHOD_GICP,
                 "FAST_VGICP": L.METHOD_VGICP, "FAST_VGICP_HIP": L.METHOD_VGICP}
        if method not in exact:   # pcl::ICP / GICP / NDT, pclomp::GICP, FAST_VGICP_CUDA are other algorithms: not served here
            raise NotImplementedError(f"registration_method {method!r} is not served by the HIP back-ends (served: {sorted(exact)})")
        m = exact[method]
        p = L.Params()
        rc = lib.dgs_params_init(C.byref(p), m)
        if rc:
            raise DgsError(rc, "dgs_params_init")
        if device is not None:
            p.device = int(device)
        for k, v in params.items():
            if not hasattr(p, k):
                raise TypeError(f"unknown registration parameter {k!r}")
            setattr(p, k, v)
        self.params = p
        self.method = method
        self._h = C.c_void_p()
        rc = lib.dgs_create(C.byref(p), C.byref(self._h))
        if rc:
            raise DgsError(rc, "dgs_create failed (this package has no CPU fallback): " + (lib.dgs_last_error(None) or b"").decode())
        self._converged = False
        self._final = np.eye(4, dtype=np.float32)
        self.last_result = None
        self._keep = {}

    # -- lifetime ---------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            if getattr(self, "_owns", True):
                self._lib.dgs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc:
            raise DgsError(rc, (self._lib.dgs_last_error(self._h) or b"").decode())

    def set_stream(self, stream=None):
        """Run on a torch.cuda.Stream (or the handle's own stream when None)."""
        ptr = None if stream is None else C.c_void_p(stream.cuda_stream)
        self._check(self._lib.dgs_set_stream(self._h, ptr))

    def synchronize(self):
        self._check(self._lib.dgs_synchronize(self._h))

    # -- pcl::Registration surface ------------------------------------------------------------------------
    def make_cloud(self, cloud) -> DeviceCloud:
        return DeviceCloud(self, cloud)

    def setInputTarget(self, cloud):
        if isinstance(cloud, DeviceCloud):
            self._keep["target"] = cloud
            self._check(self._lib.dgs_set_input_target_cloud(self._h, cloud._c))
            return
        ptr, n, dev, keep = _cloud_ptr(cloud)
        self._check(self._lib.dgs_set_input_target(self._h, ptr, n, dev))

    def setInputSource(self, cloud):
        if isinstance(cloud, DeviceCloud):
            self._keep["source"] = cloud
            self._n_source = cloud.n
            self._check(self._lib.dgs_set_input_source_cloud(self._h, cloud._c))
            return
        ptr, n, dev, keep = _cloud_ptr(cloud)
        self._n_source = n
        self._check(self._lib.dgs_set_input_source(self._h, ptr, n, dev))

    def align(self, guess=None, want_aligned: bool = False):
        """align(*aligned, guess).  Returns the aligned cloud ([N,4] float32) when want_aligned, else None.
        A failed registration never raises for numerical reasons: hasConverged() is False and the final
        transformation is the guess (the contract at scan_matching_odometry_nodelet.cpp:222-226)."""
        g = None if guess is None else _col16(guess)
        gp = None if g is None else g.ctypes.data_as(C.c_void_p)
        res = L.Result()
        out = None
        outp = None
        if want_aligned:
            out = np.empty((self._n_source, 4), dtype=np.float32)
            outp = out.ctypes.data_as(C.c_void_p)
        rc = self._lib.dgs_align(self._h, gp, C.byref(res), outp, 0)
        self.last_result = res
        self._converged = bool(res.converged)
        self._final = _from_col16(res.final_transformation)
        if rc in (L.DGS_OK,):
            return out
        if rc in (3, 4):  # PCL prints an error and returns; converged stays false
            return None
        self._check(rc)
        return out

    def hasConverged(self) -> bool:
        return self._converged

    def getFinalTransformation(self) -> np.ndarray:
        return self._final.copy()

    def getFitnessScore(self, max_range: float = 1.7976931348623157e308) -> float:
        s = C.c_double(0)
        self._check(self._lib.dgs_get_fitness_score(self._h, max_range, C.byref(s)))
        return s.value

    def getInlierFraction(self, max_sq_dist: float = 0.25) -> float:
        """The nearestKSearch loop of publish_scan_matching_status (scan_matching_odometry_nodelet.cpp:321-332)."""
        f = C.c_double(0)
        self._check(self._lib.dgs_get_inlier_fraction(self._h, max_sq_dist, C.byref(f)))
        return f.value

    def nearestKSearch(self, queries):
        """getSearchMethodTarget()->nearestKSearch(pt, 1, ...) for a batch of points -> (indices, sq_dists)."""
        ptr, m, dev, keep = _cloud_ptr(queries)
        if dev:
            raise ValueError("host queries only in the Python mirror")
        idx = np.empty(m, dtype=np.int32)
        sq = np.empty(m, dtype=np.float32)
        self._check(self._lib.dgs_nearest_search_target(self._h, ptr, m, 0, idx.ctypes.data_as(C.c_void_p), sq.ctypes.data_as(C.c_void_p)))
        return idx, sq

    def nn_fitness_distances(self, queries):
        """Test hook: squared 1-NN distances through the fitness pass's grid index (must equal nearestKSearch's)."""
        ptr, m, dev, keep = _cloud_ptr(queries)
        if dev:
            raise ValueError("host queries only in the Python mirror")
        sq = np.empty(m, dtype=np.float32)
        self._check(self._lib.dgs_nn_fitness_distances(self._h, ptr, m, 0, sq.ctypes.data_as(C.c_void_p)))
        return sq

    def find_loop_candidates(self, accum_distance, xy, new_accum_distance: float, new_xy, accum_distance_thresh: float, distance_thresh: float) -> np.ndarray:
        """LoopDetector::find_candidates (loop_detector.hpp:83-111) on the device: indices of the candidate keyframes, in keyframe order."""
        acc = np.ascontiguousarray(accum_distance, dtype=np.float64)
        p = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        q = np.ascontiguousarray(new_xy, dtype=np.float64)
        n = acc.shape[0]
        out = np.empty(max(n, 1), dtype=np.int32)
        m = C.c_int64(0)
        self._check(self._lib.dgs_find_loop_candidates(self._h, acc.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), n, 0, float(new_accum_distance),
                                                       q.ctypes.data_as(C.c_void_p), float(accum_distance_thresh), float(distance_thresh),
                                                       out.ctypes.data_as(C.c_void_p), n, C.byref(m)))
        return out[:m.value].copy()

    def calc_fitness_score(self, cloud1, cloud2, relpose=None, max_range: float = 1.7976931348623157e308) -> float:
        """InformationMatrixCalculator::calc_fitness_score (information_matrix_calculator.cpp:77-108) on the device."""
        p1, n1, d1, k1 = _cloud_ptr(cloud1)
        p2, n2, d2, k2 = _cloud_ptr(cloud2)
        if n1 and n2 and d1 != d2:
            raise ValueError("both clouds must live on the same side (host arrays or device tensors)")
        t = None if relpose is None else _col16(relpose)
        s = C.c_double(0)
        self._check(self._lib.dgs_calc_fitness_score(self._h, p1, n1, p2, n2, d1 if n1 else d2, None if t is None else t.ctypes.data_as(C.c_void_p),
                                                      max_range, C.byref(s)))
        return s.value

    def voxel_grid_filter(self, cloud, leaf_size: float, approximate: bool = False):
        """pcl::VoxelGrid centroid filter (scan_matching_odometry_nodelet.cpp:83-89,155-165) -- or, with approximate=True,
        pcl::ApproximateVoxelGrid (:90-96) -- on the device.
        numpy in -> numpy out; device tensor in -> device tensor out (no host round trip of the points)."""
        fn = self._lib.dgs_approx_voxel_grid_filter if approximate else self._lib.dgs_voxel_grid_filter
        ptr, n, dev, keep = _cloud_ptr(cloud)
        m = C.c_int64(0)
        if dev:
            out = torch.empty((max(n, 1), 4), dtype=torch.float32, device=cloud.device)
            self._check(fn(self._h, ptr, n, 1, leaf_size, C.c_void_p(out.data_ptr()), n, 1, C.byref(m)))
            return out[:m.value]
        out = np.empty((max(n, 1), 4), dtype=np.float32)
        self._check(fn(self._h, ptr, n, 0, leaf_size, out.ctypes.data_as(C.c_void_p), n, 0, C.byref(m)))
        return out[:m.value].copy()

    # -- batched candidates (loop_detector.hpp:137-156) ----------------------------------------------------
    def _align_batch_raw(self, sources, guesses, compute_fitness, fitness_max_range):
        """-> ctypes array of dgs_result, one per source"""
        n = len(sources)
        g = None
        gp = None
        if guesses is not None:
            ga = np.asarray(guesses, dtype=np.float32)
            if ga.ndim == 3:   # [n,4,4] row-major -> n x column-major float[16]
                g = np.ascontiguousarray(ga.transpose(0, 2, 1).reshape(n, 16))
            else:
                g = np.ascontiguousarray(np.stack([_col16(G) for G in guesses]))
            gp = g.ctypes.data_as(C.c_void_p)
        res = (L.Result * n)()
        if all(isinstance(s_, DeviceCloud) for s_ in sources):
            cl = (C.c_void_p * n)(*[s_._c.value for s_ in sources])
            self._check(self._lib.dgs_align_batch_clouds(self._h, n, cl, gp, 1 if compute_fitness else 0, fitness_max_range, res))
            return res
        ptrs = (C.c_void_p * n)()
        sizes = (C.c_int64 * n)()
        keep = []
        devs = set()
        cuda_devs = set()
        for i, s in enumerate(sources):
            ptr, m, dev, k = _cloud_ptr(s, sync=False)
            ptrs[i] = ptr.value if ptr.value else 0
            sizes[i] = m
            keep.append(k)
            if m:
                devs.add(dev)
            if dev:
                cuda_devs.add(k.device)
        if len(devs) > 1:
            raise ValueError("sources must be all host arrays or all device tensors")
        on_device = devs.pop() if devs else 0
        for d in cuda_devs:   # one ordering point for the whole batch (the device-pointer contract of include/dgs_reg.h)
            torch.cuda.current_stream(d).synchronize()
        self._check(self._lib.dgs_align_batch(self._h, n, ptrs, sizes, on_device, gp, 1 if compute_fitness else 0, fitness_max_range, res))
        return res

    def align_batch(self, sources, guesses=None, compute_fitness: bool = True, fitness_max_range: float = 1.7976931348623157e308):
        if len(sources) == 0:
            return []
        res = self._align_batch_raw(sources, guesses, compute_fitness, fitness_max_range)
        return [dict(T=_from_col16(r.final_transformation), converged=bool(r.converged), iterations=r.iterations,
                     evaluations=r.evaluations, status=r.status, score=r.score, fitness=r.fitness) for r in res]

    def align_batch_records(self, sources, guesses=None, compute_fitness: bool = True, fitness_max_range: float = 1.7976931348623157e308):
        """align_batch as the loop detector's exchange records: float64 [n, 20] = (-1, converged, fitness, status, T row-major),
        filled from the C ABI's result array in one numpy pass (column 0 is left for the caller's candidate index)."""
        n = len(sources)
        out = np.full((n, 20), -1.0, dtype=np.float64)
        if n == 0:
            return out
        res = self._align_batch_raw(sources, guesses, compute_fitness, fitness_max_range)
        a = np.frombuffer(res, dtype=_RESULT_DTYPE, count=n)
        out[:, 1] = a["converged"] != 0
        out[:, 2] = a["fitness"]
        out[:, 3] = a["status"]
        out[:, 4:20] = a["T"].reshape(n, 4, 4).transpose(0, 2, 1).reshape(n, 16)   # column-major -> row-major
        return out

    # -- measurement / test hooks ---------------------------------------------------------------------------
    def profile_enable(self, on: bool = True):
        self._check(self._lib.dgs_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._check(self._lib.dgs_profile_reset(self._h))

    def profile_get(self, kernel_id: int):
        ms = C.c_double(0)
        n = C.c_int64(0)
        self._check(self._lib.dgs_profile_get(self._h, kernel_id, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def counts(self):
        out = (C.c_int64 * 8)()
        self._check(self._lib.dgs_get_counts(self._h, out))
        return dict(target_points=out[0], source_points=out[1], valid_voxels=out[2], occupied_voxels=out[3],
                    grid_cells=out[4], evaluations=out[5])

    def vgicp_voxels(self):
        """FAST_VGICP target voxel map -> (coords [V,3], counts [V], means [V,3], covs [V,3,3]) in ascending (z, y, x) order"""
        n = C.c_int64(0)
        self._check(self._lib.dgs_vgicp_get_voxels(self._h, 0, None, None, None, None, C.byref(n)))
        coords = np.zeros((n.value, 3), np.int32)
        counts = np.zeros(n.value, np.int32)
        means = np.zeros((n.value, 3))
        covs = np.zeros((n.value, 3, 3))
        self._check(self._lib.dgs_vgicp_get_voxels(self._h, n.value, coords.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p),
                                                   means.ctypes.data_as(C.c_void_p), covs.ctypes.data_as(C.c_void_p), C.byref(n)))
        return coords, counts, means, covs

    def ndt_derivatives(self, p, T=None):
        p = np.ascontiguousarray(p, dtype=np.float64)
        t16 = None if T is None else _col16(T)
        s = C.c_double(0)
        g = np.zeros(6)
        H = np.zeros((6, 6))
        self._check(self._lib.dgs_ndt_derivatives(self._h, p.ctypes.data_as(C.c_void_p), None if t16 is None else t16.ctypes.data_as(C.c_void_p),
                                                   C.byref(s), g.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p)))
        return s.value, g, H

    def ndt_hessian_double(self, p):
        """computeHessian in PCL's double form at pose p (dgs_ndt_hessian_double; upstream evaluation orders only)."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        H = np.zeros((6, 6))
        self._check(self._lib.dgs_ndt_hessian_double(self._h, p.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p)))
        return H

    def gicp_covariances(self, which: str = "source", n: int | None = None):
        c = self.counts()   # the library writes one 3x3 per point of the cloud it holds: size the buffer from ITS count
        n = int(c["source_points"] if which == "source" else c["target_points"])
        out = np.zeros((n, 3, 3))
        self._check(self._lib.dgs_gicp_get_covariances(self._h, 0 if which == "source" else 1, out.ctypes.data_as(C.c_void_p)))
        return out

    def gicp_linearize(self, T, error_only: bool = False):
        T = np.ascontiguousarray(T, dtype=np.float64)
        e = C.c_double(0)
        H = np.zeros((6, 6))
        b = np.zeros(6)
        self._check(self._lib.dgs_gicp_linearize(self._h, T.ctypes.data_as(C.c_void_p), 1 if error_only else 0, C.byref(e),
                                                  H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))
        return (e.value,) if error_only else (e.value, H, b)

    def ndt_trajectory(self, pair: int = 0):
        buf = np.zeros((72, 6))
        n = C.c_int32(0)
        self._check(self._lib.dgs_ndt_get_trajectory(self._h, pair, buf.ctypes.data_as(C.c_void_p), C.byref(n)))
        return buf[:n.value].copy()

    def ndt_voxels(self):
        n = C.c_int64(0)
        self._check(self._lib.dgs_ndt_get_voxels(self._h, C.byref(n), None, None, None, None, None))
        nv = n.value
        keys = np.zeros(nv, np.int64)
        counts = np.zeros(nv, np.int32)
        valid = np.zeros(nv, np.int32)
        mean = np.zeros((nv, 3))
        icov = np.zeros((nv, 3, 3))
        if nv:
            self._check(self._lib.dgs_ndt_get_voxels(self._h, C.byref(n), keys.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p),
                                                      valid.ctypes.data_as(C.c_void_p), mean.ctypes.data_as(C.c_void_p),
                                                      icov.ctypes.data_as(C.c_void_p)))
        keep = keys >= 0
        o = np.argsort(keys[keep])
        return dict(keys=keys[keep][o], counts=counts[keep][o], valid=valid[keep][o].astype(bool), mean=mean[keep][o], icov=icov[keep][o])


class GroupCloud:
    """dgs_group_cloud: KeyFrame::cloud (keyframe.hpp:51) resident on a group's devices -- on one member (`owner` >= 0: member
    owner mod G) or on every member (`owner` None / -1: the new keyframe, every member's target)."""

    def __init__(self, group: "RegistrationGroup", cloud, owner=None):
        ptr, n, dev, keep = _cloud_ptr(cloud)
        if dev:
            raise ValueError("a group cloud is uploaded from the host (KeyFrame::cloud)")
        self._lib = group._lib
        self._group = group
        self._c = C.c_void_p()
        group._check(self._lib.dgs_group_cloud_create(group._g, ptr, n, -1 if owner is None or owner < 0 else int(owner), C.byref(self._c)))
        self.n = n

    def __len__(self):
        return self.n

    @property
    def copies(self) -> int:
        return int(self._lib.dgs_group_cloud_copies(self._c))

    def trim(self, owner: int = -1):
        """dgs_group_cloud_trim: keep the copy on member owner mod G (or the first holder's), drop the others -- a keyframe that was every
        member's target for a tick goes back to one copy on its owner."""
        self._group._check(self._lib.dgs_group_cloud_trim(self._group._g, self._c, int(owner)))

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self._lib.dgs_group_cloud_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RegistrationGroup:
    """dgs_group (include/dgs_reg.h): the candidate loop of LoopDetector::matching (loop_detector.hpp:137-156) sharded over
    several GPUs of ONE process -- candidate c goes to device c mod G, the result records come back through an RCCL all-gather
    in original candidate order.  Offers the batch surface LoopDetector drives (setInputTarget / align_batch[_records]), so a
    LoopDetector(registration=RegistrationGroup(...)) is the single-process counterpart of the one-process-per-GPU sharding."""

    def __init__(self, method: str = "NDT_OMP", devices=(0,), **params):
        lib = L.load()
        self._lib = lib
        exact = {"NDT_OMP": L.METHOD_NDT, "NDT_HIP": L.METHOD_NDT, "FAST_GICP": L.METHOD_GICP, "FAST_GICP_HIP": L.METHOD_GICP,
                 "FAST_VGICP": L.METHOD_VGICP, "FAST_VGICP_HIP": L.METHOD_VGICP}
        if method not in exact:
            raise NotImplementedError(f"registration_method {method!r} is not served by the HIP back-ends")
        p = L.Params()
        rc = lib.dgs_params_init(C.byref(p), exact[method])
        if rc:
            raise DgsError(rc, "dgs_params_init")
        for k, v in params.items():
            if not hasattr(p, k):
                raise TypeError(f"unknown registration parameter {k!r}")
            setattr(p, k, v)
        self.devices = [int(d) for d in devices]
        self.method = method
        dv = (C.c_int32 * len(self.devices))(*self.devices)
        self._g = C.c_void_p()
        rc = lib.dgs_group_create(C.byref(p), dv, len(self.devices), C.byref(self._g))
        if rc:
            raise DgsError(rc, "dgs_group_create failed (this package has no CPU fallback): " + (lib.dgs_last_error(None) or b"").decode())
        self.best_index = -1
        self.best_score = float("inf")

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            self._lib.dgs_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc:
            raise DgsError(rc, (self._lib.dgs_group_last_error(self._g) or b"").decode())

    @property
    def uses_rccl(self) -> bool:
        return bool(self._lib.dgs_group_uses_rccl(self._g))

    @property
    def rccl_ranks(self) -> int:
        """ncclCommCount of the group's communicator (0 without RCCL)."""
        return int(self._lib.dgs_group_rccl_ranks(self._g))

    @property
    def last_gather_used_rccl(self) -> bool:
        return bool(self._lib.dgs_group_last_gather_used_rccl(self._g))

    def member(self, k: int) -> "Registration":
        """Member k's handle as a (borrowed) Registration: measurement hooks (profile_*, counts) of one device of the group."""
        h = self._lib.dgs_group_member(self._g, int(k))
        if not h:
            raise IndexError(k)
        return Registration(self.method, _borrow=h)

    def make_cloud(self, cloud, owner=None) -> GroupCloud:
        return GroupCloud(self, cloud, owner)

    def setInputTarget(self, cloud):
        if isinstance(cloud, GroupCloud):
            self._keep_target = cloud
            self._check(self._lib.dgs_group_set_input_target_cloud(self._g, cloud._c))
            return
        ptr, n, dev, keep = _cloud_ptr(cloud)
        if dev:
            raise ValueError("a group takes host clouds (KeyFrame::cloud) or GroupCloud objects; each member holds its own copy")
        self._check(self._lib.dgs_group_set_input_target(self._g, ptr, n))

    def _raw(self, sources, guesses, compute_fitness, fitness_max_range):
        n = len(sources)
        g = gp = None
        if guesses is not None:
            ga = np.asarray(guesses, dtype=np.float32)
            g = np.ascontiguousarray(ga.transpose(0, 2, 1).reshape(n, 16)) if ga.ndim == 3 else np.ascontiguousarray(np.stack([_col16(G) for G in guesses]))
            gp = g.ctypes.data_as(C.c_void_p)
        temps = []
        if n and any(isinstance(s_, GroupCloud) for s_ in sources) and not all(isinstance(s_, GroupCloud) for s_ in sources):
            # a mixed list (a keyframe without an id among cached ones): the raw clouds are uploaded for this call, candidate c to member c mod G
            sources = list(sources)
            for i, s_ in enumerate(sources):
                if not isinstance(s_, GroupCloud):
                    sources[i] = GroupCloud(self, s_, owner=i)
                    temps.append(sources[i])
        if n and all(isinstance(s_, GroupCloud) for s_ in sources):   # resident keyframes: nothing is uploaded
            cl = (C.c_void_p * n)(*[s_._c.value for s_ in sources])
            res = (L.Result * n)()
            bi = C.c_int32(-1)
            bs = C.c_double(0)
            try:
                self._check(self._lib.dgs_group_align_batch_clouds(self._g, n, cl, gp, 1 if compute_fitness else 0, fitness_max_range, res, C.byref(bi), C.byref(bs)))
            finally:
                for t in temps:
                    t.close()
            self.best_index, self.best_score = bi.value, bs.value
            return res
        ptrs = (C.c_void_p * n)()
        sizes = (C.c_int64 * n)()
        keep = []
        for i, s_ in enumerate(sources):
            ptr, m, dev, k = _cloud_ptr(s_)
            if dev:
                raise ValueError("a group takes host clouds")
            ptrs[i] = ptr.value if ptr.value else 0
            sizes[i] = m
            keep.append(k)
        res = (L.Result * n)()
        bi = C.c_int32(-1)
        bs = C.c_double(0)
        self._check(self._lib.dgs_group_align_batch(self._g, n, ptrs, sizes, gp, 1 if compute_fitness else 0, fitness_max_range, res, C.byref(bi), C.byref(bs)))
        self.best_index, self.best_score = bi.value, bs.value
        return res

    def align_batch(self, sources, guesses=None, compute_fitness: bool = True, fitness_max_range: float = 1.7976931348623157e308):
        if len(sources) == 0:
            return []
        res = self._raw(sources, guesses, compute_fitness, fitness_max_range)
        return [dict(T=_from_col16(r.final_transformation), converged=bool(r.converged), iterations=r.iterations, evaluations=r.evaluations,
                     status=r.status, score=r.score, fitness=r.fitness) for r in res]

    def align_batch_records(self, sources, guesses=None, compute_fitness: bool = True, fitness_max_range: float = 1.7976931348623157e308):
        n = len(sources)
        out = np.full((n, 20), -1.0, dtype=np.float64)
        if n == 0:
            return out
        res = self._raw(sources, guesses, compute_fitness, fitness_max_range)
        a = np.frombuffer(res, dtype=_RESULT_DTYPE, count=n)
        out[:, 1] = a["converged"] != 0
        out[:, 2] = a["fitness"]
        out[:, 3] = a["status"]
        out[:, 4:20] = a["T"].reshape(n, 4, 4).transpose(0, 2, 1).reshape(n, 16)
        return out


def select_registration_method(params: dict | None = None, device: int | None = None) -> Registration:
    """Mirror of hdl_graph_slam::select_registration_method (registrations.cpp:22-124) for the HIP back-ends.

    `params` plays the role of the private NodeHandle: keys are the reference's rosparam names
    (registration_method, reg_num_threads, reg_transformation_epsilon, reg_maximum_iterations,
    reg_max_correspondence_distance, reg_correspondence_randomness, reg_resolution, reg_nn_search_method).
    "NDT_HIP" selects the HIP NDT; "FAST_GICP_HIP" / "FAST_GICP" select GICP (:27-36); "FAST_VGICP_HIP" / "FAST_VGICP" the voxelised
    GICP (:48-56).  Every other name goes through the reference's own chain of tests, in its order: "ICP" (:59-64) and any name
    containing "GICP" (:66-87: pcl::GICP, or pclomp::GICP when it also contains "OMP") are different algorithms -> NotImplementedError;
    what is left is the NDT branch (:88-123): a name without "NDT" warns "unknown registration type ... use NDT" (:89-92), and then
    a name without "OMP" -- plain "NDT" as well as an unknown name such as "FOO" -- is pcl::NormalDistributionsTransform (:94-100),
    another algorithm -> NotImplementedError, while a name containing "OMP" ("NDT_OMP", and e.g. "FOO_OMP" after the warning) is
    pclomp::NormalDistributionsTransform (:101-120), which this library serves.
    """
    pr = dict(params or {})
    method = pr.get("registration_method", "NDT_OMP")
    common = dict(num_threads=int(pr.get("reg_num_threads", 0)),
                  transformation_epsilon=float(pr.get("reg_transformation_epsilon", 0.01)),
                  maximum_iterations=int(pr.get("reg_maximum_iterations", 64)))
    if method in ("FAST_GICP", "FAST_GICP_HIP"):
        return Registration("FAST_GICP", device=device,
                            gicp_max_correspondence_distance=float(pr.get("reg_max_correspondence_distance", 2.5)),
                            gicp_correspondence_randomness=int(pr.get("reg_correspondence_randomness", 20)), **common)
    if method in ("FAST_VGICP", "FAST_VGICP_HIP"):      # registrations.cpp:48-56
        return Registration("FAST_VGICP", device=device, vgicp_resolution=float(pr.get("reg_resolution", 1.0)),
                            gicp_correspondence_randomness=int(pr.get("reg_correspondence_randomness", 20)), **common)
    if method == "ICP" or "GICP" in method:             # :59-64, :66-87 (FAST_VGICP_CUDA lands here too when the reference is built without CUDA)
        raise NotImplementedError(f"registration_method {method!r} is served by the reference's own factory branch, not by the HIP back-ends")
    if method != "NDT_HIP":
        if "NDT" not in method:                         # :89-92
            import sys
            print(f"warning: unknown registration type({method})\n       : use NDT", file=sys.stderr)
        if "OMP" not in method:                         # :94-100 pcl::NormalDistributionsTransform
            raise NotImplementedError(f"registration_method {method!r} selects pcl::NormalDistributionsTransform in the reference "
                                      "(registrations.cpp:94-100), a different algorithm from NDT_OMP: not served by the HIP back-ends")
    nn = pr.get("reg_nn_search_method", "DIRECT7")
    search = L.NDT_SEARCH["KDTREE"] if nn == "KDTREE" else L.NDT_SEARCH["DIRECT1"] if nn == "DIRECT1" else L.NDT_SEARCH["DIRECT7"]
    return Registration("NDT_OMP", device=device, ndt_resolution=float(pr.get("reg_resolution", 0.5)), ndt_search_method=search, **common)
