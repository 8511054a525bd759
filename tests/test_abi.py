"""CPU tests of the drop-in boundary: libdgs_reg.so loads, exports every symbol include/dgs_reg.h declares, the ctypes
mirrors match the C structs, defaults equal the reference factory's, and the product path fails loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dgs_reg.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dgs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from delta_graph_slam_amd import _lib as L
    lib = L.load()
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in dgs_reg.h but not exported"
    assert sorted(L.SYMBOLS) == declared
    assert lib.dgs_abi_version() == 5


def test_struct_layouts_match_the_header():
    from delta_graph_slam_amd import _lib as L
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "dgs_reg.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(dgs_params), sizeof(dgs_result), offsetof(dgs_params, transformation_epsilon),
         offsetof(dgs_params, ndt_resolution), offsetof(dgs_params, gicp_max_correspondence_distance),
         offsetof(dgs_params, gicp_lm_max_iterations), offsetof(dgs_result, score), offsetof(dgs_params, vgicp_search_method),
         offsetof(dgs_params, vgicp_resolution));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        cfile = os.path.join(d, "t.c")
        open(cfile, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), cfile, "-o", exe])   # the header is plain C
        vals = [int(x) for x in subprocess.check_output([exe]).split()]
    P, R = L.Params, L.Result
    assert vals == [C.sizeof(P), C.sizeof(R), P.transformation_epsilon.offset, P.ndt_resolution.offset,
                    P.gicp_max_correspondence_distance.offset, P.gicp_lm_max_iterations.offset, R.score.offset,
                    P.vgicp_search_method.offset, P.vgicp_resolution.offset]


def test_defaults_are_the_reference_factory_defaults():
    """registrations.cpp:26-36,93-119: eps 0.01, 64 iterations, NDT resolution 0.5, DIRECT7, GICP dmax 2.5, k 20."""
    from delta_graph_slam_amd import _lib as L
    lib = L.load()
    for method in (L.METHOD_NDT, L.METHOD_GICP, L.METHOD_VGICP):
        p = L.Params()
        assert lib.dgs_params_init(C.byref(p), method) == 0
        assert p.struct_size == C.sizeof(L.Params) and p.method == method
        assert p.transformation_epsilon == 0.01 and p.maximum_iterations == 64
        assert p.ndt_resolution == 0.5 and p.ndt_search_method == L.NDT_SEARCH["DIRECT7"]
        assert p.ndt_step_size == 0.1 and p.ndt_outlier_ratio == 0.55 and p.ndt_min_points_per_voxel == 6
        assert p.gicp_max_correspondence_distance == 2.5 and p.gicp_correspondence_randomness == 20
        assert p.gicp_regularization == L.GICP_REG["PLANE"] and p.gicp_rotation_epsilon == 2e-3
        assert p.vgicp_resolution == 1.0 and p.vgicp_search_method == L.VGICP_SEARCH["DIRECT1"]      # registrations.cpp:52, FastVGICP ctor
        # the evaluation order that reproduces a CPU run of upstream is the default (ABI 5); the re-associated fast order is opt-in
        assert p.ndt_strict_order == 1 and p.ndt_newton_solver == 1 and p.ndt_hessian_recompute_double == 1 and p.ndt_guess_rotation_polar == 1
        assert p.ndt_exp_glibc == 1 and p.ndt_cov_eigensolver == 1 and p.gicp_cov_jacobi_svd == 0
    assert lib.dgs_params_init(C.byref(L.Params()), 7) == 1      # unknown method: DGS_ERR_INVALID_ARGUMENT
    assert lib.dgs_params_init(None, 0) == 1


def test_invalid_arguments_are_rejected_without_touching_a_device():
    from delta_graph_slam_amd import _lib as L
    lib = L.load()
    h = C.c_void_p()
    p = L.Params()
    lib.dgs_params_init(C.byref(p), L.METHOD_NDT)
    p.struct_size = 12
    assert lib.dgs_create(C.byref(p), C.byref(h)) == 1
    assert lib.dgs_set_input_target(None, None, 0, 0) == 1
    assert lib.dgs_align(None, None, None, None, 0) == 1
    assert lib.dgs_get_fitness_score(None, 1.0, None) == 1
    assert lib.dgs_last_error(None) is not None


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from delta_graph_slam_amd.registration import DgsError, Registration, select_registration_method
    with pytest.raises(DgsError) as e:
        Registration("NDT_OMP")
    assert e.value.status == 2          # DGS_ERR_HIP: the product path never routes to a CPU implementation
    with pytest.raises(DgsError):
        select_registration_method({"registration_method": "FAST_GICP"})


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "delta_graph_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f in (), f"{f} mentions the oracle"


def test_group_has_no_cpu_fallback_either():
    """dgs_group_create fails with DGS_ERR_HIP on a box without a usable device (no member handle can be created)."""
    import ctypes as C
    import torch
    from delta_graph_slam_amd import _lib as L
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    lib = L.load()
    p = L.Params()
    assert lib.dgs_params_init(C.byref(p), L.METHOD_NDT) == 0
    g = C.c_void_p()
    dv = (C.c_int32 * 2)(0, 1)
    assert lib.dgs_group_create(C.byref(p), dv, 2, C.byref(g)) == 2 and not g.value
    assert lib.dgs_group_size(None) == 0 and lib.dgs_group_uses_rccl(None) == 0


def test_product_library_carries_only_the_measured_winners():
    """the voxel-hierarchy fitness index (nn_grid.hip) and the packed-FP32 derivative kernel are measured losers; they
    live in the experiments build (`make experiments`), not in the library a nodelet links."""
    from delta_graph_slam_amd import _lib as L
    prod = open(L.LIB_PATH, "rb").read()
    pack2 = b"ndt_derivatives_kernelILi2ELb1ELb1EE"     # <DIRECT7, fused, PACK2>
    assert b"nn_grid" not in prod and pack2 not in prod
    if os.path.exists(L.EXPERIMENTS_LIB_PATH):
        exp = open(L.EXPERIMENTS_LIB_PATH, "rb").read()
        assert b"nn_grid" in exp and pack2 in exp
        assert len(prod) < len(exp)
