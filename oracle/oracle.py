"""ORACLE -- TEST INFRASTRUCTURE ONLY.  **Parity unpinned.**

ctypes front-end of the CPU restatement (oracle/cpu/*.cpp -> oracle/_build/liboracle.so).

Why "unpinned": the arithmetic of the hot path lives in koide3/ndt_omp, SMRT-AIST/fast_gicp and PCL, which
the reference clones at un-pinned HEAD (/root/reference/README.md:21-22, docker/noetic/Dockerfile:14-15) and
which are absent from this image; the reference ships no tests, fixtures or golden vectors (SURVEY.md §4,
§8c).  The restatement follows the published algorithms (Magnusson 2009 ch. 6; More & Thuente 1994;
Segal et al. 2009 "Generalized-ICP"; fast_gicp's LM driver as recalled in SURVEY.md App. A/B) and is
anchored on the reference's own call sites (registrations.cpp:22-124, scan_matching_odometry_nodelet.cpp:
173-270, loop_detector.hpp:119-173, information_matrix_calculator.cpp:77-108) plus analytic known-answer
tests in tests/ (finite differences, scipy cKDTree, numpy.linalg).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_LIB_FMA_PATH = os.path.join(_HERE, "_build", "liboracle_fma.so")  # same source, FMA contraction on (perturbed build)


def build(force: bool = False) -> str:
    """Compile the C++ restatement with g++ (see oracle/Makefile)."""
    if force or not (os.path.exists(_LIB_PATH) and os.path.exists(_LIB_FMA_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "all"] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return _LIB_PATH


class NdtParams(C.Structure):
    _fields_ = [("resolution", C.c_double), ("step_size", C.c_double), ("outlier_ratio", C.c_double),
                ("transformation_epsilon", C.c_double), ("min_covar_eigvalue_mult", C.c_double),
                ("max_iterations", C.c_int32), ("search_method", C.c_int32), ("min_points_per_voxel", C.c_int32),
                ("line_search", C.c_int32), ("mt_max_step_iterations", C.c_int32), ("num_threads", C.c_int32),
                ("fix_hessian_d1", C.c_int32), ("exp_libm", C.c_int32),
                ("newton_solver", C.c_int32), ("hessian_recompute_double", C.c_int32), ("guess_rotation_polar", C.c_int32),
                ("cov_eigensolver", C.c_int32), ("pad0", C.c_int32)]


class GicpParams(C.Structure):
    _fields_ = [("transformation_epsilon", C.c_double), ("rotation_epsilon", C.c_double),
                ("max_correspondence_distance", C.c_double), ("lm_init_lambda_factor", C.c_double),
                ("max_iterations", C.c_int32), ("k_correspondences", C.c_int32), ("regularization", C.c_int32),
                ("optimizer", C.c_int32), ("lm_max_iterations", C.c_int32), ("num_threads", C.c_int32), ("cov_svd", C.c_int32), ("pad0", C.c_int32)]


class Result(C.Structure):
    _fields_ = [("T", C.c_float * 16), ("converged", C.c_int32), ("iterations", C.c_int32),
                ("evaluations", C.c_int32), ("pad", C.c_int32), ("score", C.c_double)]


_libs = {}


def lib(perturbed: bool = False):
    """The restatement (perturbed=False) or its FMA-contracted twin (perturbed=True)."""
    if perturbed not in _libs:
        build()
        L = C.CDLL(_LIB_FMA_PATH if perturbed else _LIB_PATH)
        L.orc_ndt_create.restype = C.c_void_p
        L.orc_ndt_derivatives.restype = C.c_double
        L.orc_ndt_voxels.restype = C.c_int64
        if hasattr(L, "orc_gicp_create"):
            L.orc_gicp_create.restype = C.c_void_p
            if hasattr(L, "orc_vgicp_create"):
                L.orc_vgicp_create.restype = C.c_void_p
                L.orc_vgicp_voxels.restype = C.c_int64
            L.orc_gicp_linearize.restype = C.c_double
            L.orc_gicp_compute_error.restype = C.c_double
        if hasattr(L, "orc_fitness_score"):
            L.orc_fitness_score.restype = C.c_double
        _libs[perturbed] = L
    return _libs[perturbed]


SEARCH = {"KDTREE": 0, "DIRECT26": 1, "DIRECT7": 2, "DIRECT1": 3}


def _f32c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def _f64c(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _colmajor16(T) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T.reshape(16))


def _from_colmajor16(t16) -> np.ndarray:
    return np.array(t16, dtype=np.float32).reshape(4, 4).T.copy()


class NdtOracle:
    """CPU restatement of pclomp::NormalDistributionsTransform (registrations.cpp:101-120)."""

    def __init__(self, resolution=1.0, transformation_epsilon=0.01, max_iterations=64, search_method="DIRECT7",
                 step_size=0.1, outlier_ratio=0.55, line_search=1, num_threads=0, min_points_per_voxel=6,
                 min_covar_eigvalue_mult=0.01, mt_max_step_iterations=10, fix_hessian_d1=0, perturbed=False, exp_libm=1,
                 newton_solver=1, hessian_recompute_double=1, guess_rotation_polar=1, cov_eigensolver=1):
        L = lib(perturbed)
        self._L = L
        p = NdtParams()
        L.orc_ndt_default_params(C.byref(p))
        p.resolution = resolution
        p.transformation_epsilon = transformation_epsilon
        p.max_iterations = max_iterations
        p.search_method = SEARCH[search_method] if isinstance(search_method, str) else int(search_method)
        p.step_size = step_size
        p.outlier_ratio = outlier_ratio
        p.line_search = line_search
        p.num_threads = num_threads
        p.min_points_per_voxel = min_points_per_voxel
        p.min_covar_eigvalue_mult = min_covar_eigvalue_mult
        p.mt_max_step_iterations = mt_max_step_iterations
        p.fix_hessian_d1 = fix_hessian_d1
        p.exp_libm = exp_libm   # std::exp(float): 1 = glibc's expf restated (equal to the image's libm on every float in [-104, 0]), 0 = det_expf (rounds 1-3), 2 = the host libm itself
        # round 4 (ndt_cpu.hpp): Eigen's two-sided JacobiSVD sequence / PCL's double computeHessian / Affine3f::rotation() -- 0 = rounds 1-3
        p.newton_solver = newton_solver
        p.hessian_recompute_double = hessian_recompute_double
        p.guess_rotation_polar = guess_rotation_polar
        p.cov_eigensolver = cov_eigensolver   # voxel covariances: Eigen's SelfAdjointEigenSolver restated (1) / cyclic Jacobi (0)
        self.params = p
        self.max_iterations = max_iterations
        self._h = C.c_void_p(L.orc_ndt_create(C.byref(p)))

    def __del__(self):
        try:
            self._L.orc_ndt_destroy(self._h)
        except Exception:
            pass

    def set_target(self, cloud):
        a, pa = _f32c(cloud)
        assert a.ndim == 2 and a.shape[1] == 4
        self._L.orc_ndt_set_target(self._h, pa, C.c_int64(a.shape[0]))

    def set_source(self, cloud):
        a, pa = _f32c(cloud)
        assert a.ndim == 2 and a.shape[1] == 4
        self._L.orc_ndt_set_source(self._h, pa, C.c_int64(a.shape[0]))

    def align(self, guess=None):
        g = _colmajor16(np.eye(4) if guess is None else guess)
        res = Result()
        traj = np.zeros((self.max_iterations + 4, 6))
        tl = C.c_int32(0)
        self._L.orc_ndt_align(self._h, g.ctypes.data_as(C.POINTER(C.c_float)), C.byref(res),
                            traj.ctypes.data_as(C.POINTER(C.c_double)), C.byref(tl))
        return dict(T=_from_colmajor16(res.T), converged=bool(res.converged), iterations=res.iterations,
                    evaluations=res.evaluations, score=res.score, trajectory=traj[:tl.value].copy(),
                    hessian_recomputes=res.pad)

    def derivatives(self, p, T=None, compute_hessian=True):
        p, pp = _f64c(p)
        g = np.zeros(6)
        H = np.zeros((6, 6))
        Tp = None
        if T is not None:
            t16 = _colmajor16(T)
            Tp = t16.ctypes.data_as(C.POINTER(C.c_float))
        s = self._L.orc_ndt_derivatives(self._h, pp, Tp, g.ctypes.data_as(C.POINTER(C.c_double)),
                                      H.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(1 if compute_hessian else 0))
        return s, g, H

    def hessian_double(self, p):
        """computeHessian in PCL's double form at pose p (ndt_cpu.cpp hessian_double_with)."""
        p, pp = _f64c(p)
        H = np.zeros((6, 6))
        self._L.orc_ndt_hessian_double(self._h, pp, H.ctypes.data_as(C.POINTER(C.c_double)))
        return H

    def voxels(self):
        L = self._L
        n = L.orc_ndt_voxels(self._h, None, None, None, None, None, None)
        keys = np.zeros(n, np.int64)
        counts = np.zeros(n, np.int32)
        valid = np.zeros(n, np.int32)
        mean = np.zeros((n, 3))
        cov = np.zeros((n, 3, 3))
        icov = np.zeros((n, 3, 3))
        L.orc_ndt_voxels(self._h, keys.ctypes.data_as(C.POINTER(C.c_int64)), counts.ctypes.data_as(C.POINTER(C.c_int32)),
                         valid.ctypes.data_as(C.POINTER(C.c_int32)), mean.ctypes.data_as(C.POINTER(C.c_double)),
                         cov.ctypes.data_as(C.POINTER(C.c_double)), icov.ctypes.data_as(C.POINTER(C.c_double)))
        o = np.argsort(keys)
        return dict(keys=keys[o], counts=counts[o], valid=valid[o].astype(bool), mean=mean[o], cov=cov[o], icov=icov[o])

    def grid(self):
        mn = (C.c_int32 * 3)()
        mx = (C.c_int32 * 3)()
        dv = (C.c_int32 * 3)()
        self._L.orc_ndt_grid(self._h, mn, mx, dv)
        return np.array(mn), np.array(mx), np.array(dv)


def euler_angles_012(T) -> np.ndarray:
    out = np.zeros(3, np.float32)
    t16 = _colmajor16(T)
    lib().orc_euler_angles_012(t16.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def pose_to_matrix_f32(p) -> np.ndarray:
    p, pp = _f64c(p)
    out = np.zeros(16, np.float32)
    lib().orc_pose_to_matrix_f32(pp, out.ctypes.data_as(C.POINTER(C.c_float)))
    return _from_colmajor16(out)


def svd_solve6(A, b):
    A, pa = _f64c(A)
    b, pb = _f64c(b)
    x = np.zeros(6)
    lib().orc_svd_solve6(pa, pb, x.ctypes.data_as(C.POINTER(C.c_double)))
    return x


def jsvd_solve6(A, b, stats=False):
    """Eigen::JacobiSVD<Matrix6d>(A, FullU | FullV).solve(b) restated (linalg.hpp jsvd_solve6); stats -> (x, sweeps, rotations)."""
    a, pa = _f64c(np.asarray(A, np.float64).reshape(36))
    bb, pb = _f64c(b)
    x = np.zeros(6)
    sr = (C.c_int32 * 2)()
    lib().orc_jsvd_solve6(pa, pb, x.ctypes.data_as(C.POINTER(C.c_double)), sr)
    return (x, int(sr[0]), int(sr[1])) if stats else x


def affine_rotation_f32(T) -> np.ndarray:
    """Eigen::Affine3f::rotation() of a 4x4 (polar factor through a float JacobiSVD), 3x3 float32."""
    t = _colmajor16(T)
    R = np.zeros(9, np.float32)
    lib().orc_affine_rotation_f32(t.ctypes.data_as(C.POINTER(C.c_float)), R.ctypes.data_as(C.POINTER(C.c_float)))
    return R.reshape(3, 3)


def det_exp(x: float) -> float:
    L = lib()
    L.orc_det_exp.restype = C.c_double
    L.orc_det_exp.argtypes = [C.c_double]
    return float(L.orc_det_exp(float(x)))


def glibc_expf(x) -> float:
    """linalg.hpp glibc_expf: std::exp(float) as glibc computes it (the restatement's default for updateDerivatives' exponential)."""
    L = lib()
    L.orc_glibc_expf.restype = C.c_float
    L.orc_glibc_expf.argtypes = [C.c_float]
    return float(L.orc_glibc_expf(float(np.float32(x))))


def glibc_exp(x: float) -> float:
    """linalg.hpp glibc_exp: std::exp(double) as glibc computes it (the double computeHessian pass's exponential)."""
    L = lib()
    L.orc_glibc_exp.restype = C.c_double
    L.orc_glibc_exp.argtypes = [C.c_double]
    return float(L.orc_glibc_exp(float(x)))


def glibc_exp_mismatches(n: int, seed: int = 88172645463325252):
    """glibc_exp against the host libm's exp on ~n doubles of a fixed pseudo-random stream over [-760, 720]; (count, one differing argument)."""
    L = lib()
    L.orc_glibc_exp_mismatches.restype = C.c_longlong
    L.orc_glibc_exp_mismatches.argtypes = [C.c_longlong, C.c_uint64, C.POINTER(C.c_double)]
    fb = C.c_double(0)
    m = int(L.orc_glibc_exp_mismatches(int(n), int(seed), C.byref(fb)))
    return m, (None if m == 0 else fb.value)


def glibc_expf_mismatches(first: float, last: float):
    """Floats between `first` and `last` (same sign, |first| <= |last|: bit patterns ascending) on which glibc_expf and the host libm's
    expf differ; returns (count, first differing float or None)."""
    L = lib()
    L.orc_glibc_expf_mismatches.restype = C.c_longlong
    L.orc_glibc_expf_mismatches.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    a = int(np.float32(first).view(np.uint32))
    b = int(np.float32(last).view(np.uint32))
    fb = C.c_uint32(0)
    n = int(L.orc_glibc_expf_mismatches(a, b, C.byref(fb)))
    return n, (None if n == 0 else float(np.uint32(fb.value).view(np.float32)))


def ldlt_solve6(A, b):
    A, pa = _f64c(A)
    b, pb = _f64c(b)
    x = np.zeros(6)
    lib().orc_ldlt_solve6(pa, pb, x.ctypes.data_as(C.POINTER(C.c_double)))
    return x


def sym_eig3(A):
    A, pa = _f64c(A)
    ev = np.zeros(3)
    V = np.zeros((3, 3))
    lib().orc_sym_eig3(pa, ev.ctypes.data_as(C.POINTER(C.c_double)), V.ctypes.data_as(C.POINTER(C.c_double)))
    return ev, V


def eigen_selfadjoint3(A):
    """linalg.hpp eigen_selfadjoint3: Eigen::SelfAdjointEigenSolver<Matrix3d>::compute restated; (evals ascending, V, QR steps)."""
    A, pa = _f64c(A)
    ev = np.zeros(3)
    V = np.zeros((3, 3))
    L = lib()
    L.orc_eigen_selfadjoint3.restype = C.c_int
    it = int(L.orc_eigen_selfadjoint3(pa, ev.ctypes.data_as(C.POINTER(C.c_double)), V.ctypes.data_as(C.POINTER(C.c_double))))
    return ev, V, it


def max_threads() -> int:
    return int(lib().orc_max_threads())


GICP_REG = {"NONE": 0, "MIN_EIG": 1, "NORMALIZED_MIN_EIG": 2, "PLANE": 3, "FROBENIUS": 4}


class GicpOracle:
    """CPU restatement of fast_gicp::FastGICP (registrations.cpp:27-36)."""

    def __init__(self, transformation_epsilon=0.01, max_iterations=64, max_correspondence_distance=2.5, k_correspondences=20,
                 rotation_epsilon=2e-3, regularization="PLANE", optimizer=1, lm_max_iterations=10, lm_init_lambda_factor=1e-9,
                 num_threads=0, perturbed=False, cov_svd=0):
        L = lib(perturbed)
        self._L = L
        p = GicpParams()
        L.orc_gicp_default_params(C.byref(p))
        p.transformation_epsilon = transformation_epsilon
        p.max_iterations = max_iterations
        p.max_correspondence_distance = max_correspondence_distance
        p.k_correspondences = k_correspondences
        p.rotation_epsilon = rotation_epsilon
        p.regularization = GICP_REG[regularization] if isinstance(regularization, str) else int(regularization)
        p.optimizer = optimizer
        p.lm_max_iterations = lm_max_iterations
        p.lm_init_lambda_factor = lm_init_lambda_factor
        p.num_threads = num_threads
        p.cov_svd = cov_svd   # covariance regularisation through Eigen's JacobiSVD restated (1) / the symmetric eigen-decomposition (0)
        self.params = p
        self._h = self._create(L, p)
        self.ns = self.nt = 0

    def _create(self, L, p):
        return C.c_void_p(L.orc_gicp_create(C.byref(p)))

    def __del__(self):
        try:
            self._L.orc_gicp_destroy(self._h)
        except Exception:
            pass

    def set_target(self, cloud):
        a, pa = _f32c(cloud)
        self.nt = a.shape[0]
        self._L.orc_gicp_set_target(self._h, pa, C.c_int64(a.shape[0]))

    def set_source(self, cloud):
        a, pa = _f32c(cloud)
        self.ns = a.shape[0]
        self._L.orc_gicp_set_source(self._h, pa, C.c_int64(a.shape[0]))

    def align(self, guess=None):
        g = _colmajor16(np.eye(4) if guess is None else guess)
        res = Result()
        self._L.orc_gicp_align(self._h, g.ctypes.data_as(C.POINTER(C.c_float)), C.byref(res))
        return dict(T=_from_colmajor16(res.T), converged=bool(res.converged), iterations=res.iterations,
                    evaluations=res.evaluations, score=res.score)

    def linearize(self, T):
        T, pt = _f64c(np.asarray(T, np.float64))
        H = np.zeros((6, 6))
        b = np.zeros(6)
        e = self._L.orc_gicp_linearize(self._h, pt, H.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)))
        return e, H, b

    def compute_error(self, T):
        T, pt = _f64c(np.asarray(T, np.float64))
        return self._L.orc_gicp_compute_error(self._h, pt)

    def covariances(self, which="source"):
        n = self.ns if which == "source" else self.nt
        out = np.zeros((n, 3, 3))
        self._L.orc_gicp_covariances(self._h, C.c_int32(0 if which == "source" else 1), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def correspondences(self):
        corr = np.zeros(self.ns, np.int32)
        sq = np.zeros(self.ns, np.float32)
        self._L.orc_gicp_correspondences(self._h, corr.ctypes.data_as(C.POINTER(C.c_int32)), sq.ctypes.data_as(C.POINTER(C.c_float)))
        return corr, sq


VGICP_SEARCH = {"DIRECT1": 0, "DIRECT7": 1, "DIRECT27": 2}


class VgicpOracle(GicpOracle):
    """CPU restatement of fast_gicp::FastVGICP (registrations.cpp:48-56); same driver surface as GicpOracle."""

    def __init__(self, resolution=1.0, search_method="DIRECT1", **kw):
        self.resolution = float(resolution)
        self.search_method = VGICP_SEARCH[search_method] if isinstance(search_method, str) else int(search_method)
        super().__init__(**kw)

    def _create(self, L, p):
        return C.c_void_p(L.orc_vgicp_create(C.byref(p), C.c_double(self.resolution), C.c_int32(self.search_method)))

    def voxels(self):
        """-> (coords [V,3] int32, counts [V], means [V,3], covs [V,3,3]) in ascending (z, y, x) coordinate order"""
        n = self._L.orc_vgicp_voxels(self._h, None, None, None, None)
        coords = np.zeros((n, 3), np.int32)
        counts = np.zeros(n, np.int32)
        means = np.zeros((n, 3))
        covs = np.zeros((n, 3, 3))
        self._L.orc_vgicp_voxels(self._h, coords.ctypes.data_as(C.POINTER(C.c_int32)), counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                 means.ctypes.data_as(C.POINTER(C.c_double)), covs.ctypes.data_as(C.POINTER(C.c_double)))
        return coords, counts, means, covs


def fitness_score(target, source, T, max_range=1.7976931348623157e308, inlier_sq=0.25):
    """pcl::Registration::getFitnessScore + inlier count -> (score, n_used, n_inliers)"""
    t, pt = _f32c(target)
    s, ps = _f32c(source)
    t16 = _colmajor16(T)
    nu = C.c_int64(0)
    ni = C.c_int64(0)
    sc = lib().orc_fitness_score(pt, C.c_int64(t.shape[0]), ps, C.c_int64(s.shape[0]), t16.ctypes.data_as(C.POINTER(C.c_float)),
                                 C.c_double(max_range), C.c_double(inlier_sq), C.byref(nu), C.byref(ni))
    return sc, nu.value, ni.value


def knn(cloud, queries, k):
    c, pc = _f32c(cloud)
    q, pq = _f32c(queries)
    idx = np.zeros((q.shape[0], k), np.int32)
    d2 = np.zeros((q.shape[0], k), np.float32)
    lib().orc_knn(pc, C.c_int64(c.shape[0]), pq, C.c_int64(q.shape[0]), C.c_int32(k), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                  d2.ctypes.data_as(C.POINTER(C.c_float)))
    return idx, d2


def se3_exp(a):
    a, pa = _f64c(a)
    T = np.zeros((4, 4))
    lib().orc_se3_exp(pa, T.ctypes.data_as(C.POINTER(C.c_double)))
    return T


def voxel_grid(cloud, leaf: float) -> np.ndarray:
    """pcl::VoxelGrid centroid filter (restatement in oracle/cpu/voxelgrid_cpu.cpp)."""
    c, pc = _f32c(cloud)
    out = np.empty((max(c.shape[0], 1), 4), np.float32)
    L = lib()
    L.orc_voxel_grid.restype = C.c_int64
    m = L.orc_voxel_grid(pc, C.c_int64(c.shape[0]), C.c_float(leaf), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:m].copy()


def approx_voxel_grid(cloud, leaf: float) -> np.ndarray:
    """pcl::ApproximateVoxelGrid (restatement in oracle/cpu/voxelgrid_cpu.cpp); the input must be finite."""
    c, pc = _f32c(cloud)
    out = np.empty((max(c.shape[0], 1), 4), np.float32)
    L = lib()
    L.orc_approx_voxel_grid.restype = C.c_int64
    m = L.orc_approx_voxel_grid(pc, C.c_int64(c.shape[0]), C.c_float(leaf), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:m].copy()


def pose_error(Ta, Tb):
    """(translation error [m], rotation angle error [rad]) between two 4x4 transforms."""
    Ta = np.asarray(Ta, np.float64)
    Tb = np.asarray(Tb, np.float64)
    dt = np.linalg.norm(Ta[:3, 3] - Tb[:3, 3])
    R = Ta[:3, :3].T @ Tb[:3, :3]
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])   # skew part: accurate for tiny angles
    return float(dt), float(np.arctan2(np.linalg.norm(w), 0.5 * (np.trace(R) - 1.0)))


def _ulp_shift(G, k):
    """Move the guess translation by k float32 ulps (x up, y down): a perturbation below the input's own resolution."""
    Gp = np.asarray(G, np.float32).copy()
    for _ in range(abs(k)):
        Gp[0, 3] = np.nextafter(Gp[0, 3], np.float32(np.inf if k > 0 else -np.inf))
        Gp[1, 3] = np.nextafter(Gp[1, 3], np.float32(-np.inf if k > 0 else np.inf))
    return Gp


def ndt_band(tgt, src, guess=None, twins=None, **kw):
    """The restated algorithm's own reproducibility on one pair: the largest deviation of its answer under perturbations that
    carry no information -- the same source compiled with FMA contraction, the OTHER exp(float) (det_expf where the run uses glibc's
    expf and the other way round: a <= 1 ulp perturbation), and the float32 initial guess moved by +-1 and +-2 ulps.  `twins` selects a
    subset (tuples (perturbed build, other exp, ulps)).  Returns (result of the unperturbed run, band_translation [m], band_rotation [rad])."""
    G = np.eye(4, dtype=np.float32) if guess is None else np.asarray(guess, np.float32)
    if twins is None:
        twins = ((True, 0, 0), (False, 1, 0), (False, 0, 1), (False, 0, -1), (False, 0, 2), (False, 0, -2))
    kw = dict(kw)
    base_exp = int(kw.pop("exp_libm", 1))
    runs = []
    for perturbed, other_exp, k in ((False, 0, 0),) + tuple(twins):
        o = NdtOracle(perturbed=perturbed, exp_libm=(0 if base_exp else 1) if other_exp else base_exp, **kw)
        o.set_target(tgt)
        o.set_source(src)
        runs.append(o.align(_ulp_shift(G, k)))
    bt = max(pose_error(r["T"], runs[0]["T"])[0] for r in runs[1:])
    br = max(pose_error(r["T"], runs[0]["T"])[1] for r in runs[1:])
    return runs[0], bt, br
