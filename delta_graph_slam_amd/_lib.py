"""ctypes binding of libdgs_reg.so (include/dgs_reg.h).  Fails loudly when the HIP library is missing --
there is no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DGS_REG_LIB") or os.path.join(_HERE, "libdgs_reg.so")   # DGS_REG_LIB: an alternative build (A/B runs)

DGS_OK = 0
STATUS = {0: "DGS_OK", 1: "DGS_ERR_INVALID_ARGUMENT", 2: "DGS_ERR_HIP", 3: "DGS_ERR_NO_TARGET", 4: "DGS_ERR_NO_SOURCE",
          5: "DGS_ERR_GRID_TOO_LARGE", 6: "DGS_ERR_UNSUPPORTED"}
METHOD_NDT, METHOD_GICP, METHOD_VGICP = 0, 1, 2
VGICP_SEARCH = {"DIRECT1": 0, "DIRECT7": 1, "DIRECT27": 2}
NDT_SEARCH = {"KDTREE": 0, "DIRECT26": 1, "DIRECT7": 2, "DIRECT1": 3}
NDT_ORDER = {"FAST": 0, "UPSTREAM": 1, "UPSTREAM_SEQUENTIAL": 2}
GICP_REG = {"NONE": 0, "MIN_EIG": 1, "NORMALIZED_MIN_EIG": 2, "PLANE": 3, "FROBENIUS": 4}
K_NDT_DERIVATIVES, K_NDT_SOLVE, K_NDT_VOXEL_BUILD, K_NN_SEARCH, K_GICP_LINEARIZE, K_GICP_COVARIANCE, K_TRANSFORM = range(7)


class DgsError(RuntimeError):
    def __init__(self, status: int, msg: str = ""):
        self.status = status
        super().__init__(f"{STATUS.get(status, status)}: {msg}")


class Params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("method", C.c_int32), ("device", C.c_int32), ("num_threads", C.c_int32),
        ("transformation_epsilon", C.c_double), ("maximum_iterations", C.c_int32), ("ndt_search_method", C.c_int32),
        ("ndt_resolution", C.c_double), ("ndt_step_size", C.c_double), ("ndt_outlier_ratio", C.c_double),
        ("ndt_min_covar_eigvalue_mult", C.c_double), ("ndt_min_points_per_voxel", C.c_int32),
        ("ndt_line_search", C.c_int32), ("ndt_mt_max_step_iterations", C.c_int32), ("ndt_fix_hessian_d1", C.c_int32),
        ("ndt_strict_order", C.c_int32),
        ("gicp_max_correspondence_distance", C.c_double), ("gicp_rotation_epsilon", C.c_double),
        ("gicp_lm_init_lambda_factor", C.c_double), ("gicp_correspondence_randomness", C.c_int32),
        ("gicp_regularization", C.c_int32), ("gicp_optimizer", C.c_int32), ("gicp_lm_max_iterations", C.c_int32),
        ("vgicp_search_method", C.c_int32), ("vgicp_resolution", C.c_double),
        ("ndt_newton_solver", C.c_int32), ("ndt_hessian_recompute_double", C.c_int32), ("ndt_guess_rotation_polar", C.c_int32),
        ("ndt_exp_glibc", C.c_int32),
        ("ndt_cov_eigensolver", C.c_int32), ("gicp_cov_jacobi_svd", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [("final_transformation", C.c_float * 16), ("converged", C.c_int32), ("iterations", C.c_int32),
                ("evaluations", C.c_int32), ("status", C.c_int32), ("score", C.c_double), ("fitness", C.c_double)]


# every symbol include/dgs_reg.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "dgs_params_init", "dgs_create", "dgs_destroy", "dgs_last_error", "dgs_abi_version", "dgs_set_stream",
    "dgs_synchronize", "dgs_set_input_target", "dgs_set_input_source", "dgs_align", "dgs_get_fitness_score",
    "dgs_get_inlier_fraction", "dgs_nearest_search_target", "dgs_nn_fitness_distances", "dgs_align_batch", "dgs_find_loop_candidates", "dgs_calc_fitness_score", "dgs_voxel_grid_filter", "dgs_approx_voxel_grid_filter", "dgs_cloud_create", "dgs_cloud_destroy", "dgs_cloud_size",
    "dgs_set_input_target_cloud", "dgs_set_input_source_cloud", "dgs_align_batch_clouds", "dgs_profile_enable",
    "dgs_profile_get", "dgs_profile_reset", "dgs_get_counts", "dgs_ndt_derivatives", "dgs_ndt_hessian_double", "dgs_ndt_get_voxels",
    "dgs_ndt_get_trajectory", "dgs_gicp_get_covariances", "dgs_gicp_linearize", "dgs_vgicp_get_voxels",
    "dgs_group_create", "dgs_group_destroy", "dgs_group_last_error", "dgs_group_size", "dgs_group_uses_rccl", "dgs_group_rccl_ranks", "dgs_group_last_gather_used_rccl",
    "dgs_group_member", "dgs_group_set_input_target", "dgs_group_align_batch",
    "dgs_group_cloud_create", "dgs_group_cloud_destroy", "dgs_group_cloud_size", "dgs_group_cloud_copies", "dgs_group_cloud_trim", "dgs_group_set_input_target_cloud",
    "dgs_group_align_batch_clouds",
]

_libs = {}
EXPERIMENTS_LIB_PATH = os.path.join(_HERE, "libdgs_reg_exp.so")   # `make experiments`: + nn_grid.hip and the packed-FP32 kernel (measured losers)


def load(path=None):
    """Load libdgs_reg.so (or another build of it, e.g. EXPERIMENTS_LIB_PATH).  Raises ImportError (never falls back) when it has
    not been built."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C delta_graph_slam_amd/csrc).  delta_graph_slam_amd has no CPU fallback.")
    lib = C.CDLL(path)
    P = C.POINTER
    lib.dgs_last_error.restype = C.c_char_p
    lib.dgs_last_error.argtypes = [C.c_void_p]
    lib.dgs_params_init.argtypes = [P(Params), C.c_int32]
    lib.dgs_create.argtypes = [P(Params), P(C.c_void_p)]
    lib.dgs_destroy.argtypes = [C.c_void_p]
    lib.dgs_destroy.restype = None
    lib.dgs_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.dgs_synchronize.argtypes = [C.c_void_p]
    lib.dgs_set_input_target.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    lib.dgs_set_input_source.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
    lib.dgs_align.argtypes = [C.c_void_p, C.c_void_p, P(Result), C.c_void_p, C.c_int32]
    lib.dgs_get_fitness_score.argtypes = [C.c_void_p, C.c_double, P(C.c_double)]
    lib.dgs_get_inlier_fraction.argtypes = [C.c_void_p, C.c_double, P(C.c_double)]
    lib.dgs_nearest_search_target.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    lib.dgs_nn_fitness_distances.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    lib.dgs_align_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                    C.c_double, P(Result)]
    lib.dgs_find_loop_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_void_p,
                                             C.c_int64, P(C.c_int64)]
    lib.dgs_calc_fitness_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_double, P(C.c_double)]
    lib.dgs_voxel_grid_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_int32, P(C.c_int64)]
    lib.dgs_approx_voxel_grid_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_int32, P(C.c_int64)]
    lib.dgs_cloud_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, P(C.c_void_p)]
    lib.dgs_cloud_destroy.argtypes = [C.c_void_p]
    lib.dgs_cloud_destroy.restype = None
    lib.dgs_cloud_size.argtypes = [C.c_void_p]
    lib.dgs_cloud_size.restype = C.c_int64
    lib.dgs_set_input_target_cloud.argtypes = [C.c_void_p, C.c_void_p]
    lib.dgs_set_input_source_cloud.argtypes = [C.c_void_p, C.c_void_p]
    lib.dgs_align_batch_clouds.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, P(Result)]
    lib.dgs_profile_enable.argtypes = [C.c_void_p, C.c_int32]
    lib.dgs_profile_get.argtypes = [C.c_void_p, C.c_int32, P(C.c_double), P(C.c_int64)]
    lib.dgs_profile_reset.argtypes = [C.c_void_p]
    lib.dgs_get_counts.argtypes = [C.c_void_p, P(C.c_int64)]
    lib.dgs_ndt_derivatives.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_double), C.c_void_p, C.c_void_p]
    lib.dgs_ndt_hessian_double.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dgs_ndt_get_voxels.argtypes = [C.c_void_p, P(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dgs_ndt_get_trajectory.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, P(C.c_int32)]
    lib.dgs_gicp_get_covariances.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.dgs_gicp_linearize.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, P(C.c_double), C.c_void_p, C.c_void_p]
    lib.dgs_vgicp_get_voxels.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_int64)]
    lib.dgs_group_create.argtypes = [P(Params), C.c_void_p, C.c_int32, P(C.c_void_p)]
    lib.dgs_group_destroy.argtypes = [C.c_void_p]
    lib.dgs_group_destroy.restype = None
    lib.dgs_group_last_error.argtypes = [C.c_void_p]
    lib.dgs_group_last_error.restype = C.c_char_p
    lib.dgs_group_size.argtypes = [C.c_void_p]
    lib.dgs_group_uses_rccl.argtypes = [C.c_void_p]
    lib.dgs_group_rccl_ranks.argtypes = [C.c_void_p]
    lib.dgs_group_last_gather_used_rccl.argtypes = [C.c_void_p]
    lib.dgs_group_member.argtypes = [C.c_void_p, C.c_int32]
    lib.dgs_group_member.restype = C.c_void_p
    lib.dgs_group_set_input_target.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.dgs_group_align_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, P(Result), P(C.c_int32),
                                          P(C.c_double)]
    lib.dgs_group_cloud_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, P(C.c_void_p)]
    lib.dgs_group_cloud_destroy.argtypes = [C.c_void_p]
    lib.dgs_group_cloud_destroy.restype = None
    lib.dgs_group_cloud_size.argtypes = [C.c_void_p]
    lib.dgs_group_cloud_size.restype = C.c_int64
    lib.dgs_group_cloud_copies.argtypes = [C.c_void_p]
    lib.dgs_group_cloud_trim.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.dgs_group_set_input_target_cloud.argtypes = [C.c_void_p, C.c_void_p]
    lib.dgs_group_align_batch_clouds.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, P(Result), P(C.c_int32),
                                                 P(C.c_double)]
    _libs[path] = lib
    return lib
