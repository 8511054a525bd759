"""Generates tests/golden/*.npz from the CPU oracle on fixed-seed synthetic clouds.

The reference repository holds no tests, fixtures or golden vectors for this path, and its
registration arithmetic cannot be built or imported here, so these are SELF-goldens: they pin the oracle (and through
it the HIP path) against regressions, they do not pin it to upstream.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from delta_graph_slam_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    tgt, src, Tgt = synth.planar_pair(n=2048, seed_target=101, seed_source=102)
    out = dict(tgt=tgt, src=src, T_gt=Tgt)
    # NDT: derivatives at two poses, final transform + trajectory, both line-search modes and three searches
    poses = np.array([[0, 0, 0, 0, 0, 0], [0.2, -0.05, 0.03, 0.02, -0.03, 0.04]], float)
    for search in ("DIRECT7", "DIRECT1", "KDTREE"):
        o = orc.NdtOracle(resolution=2.0, search_method=search)
        o.set_target(tgt)
        o.set_source(src)
        for k, p in enumerate(poses):
            s, g, H = o.derivatives(p)
            out[f"ndt_{search}_p{k}_score"] = s
            out[f"ndt_{search}_p{k}_grad"] = g
            out[f"ndt_{search}_p{k}_hess"] = H
        r = o.align()
        out[f"ndt_{search}_T"] = r["T"]
        out[f"ndt_{search}_traj"] = r["trajectory"]
        out[f"ndt_{search}_iters"] = np.array([r["iterations"], r["evaluations"], int(r["converged"])])
    o = orc.NdtOracle(resolution=2.0, line_search=0)
    o.set_target(tgt)
    o.set_source(src)
    r = o.align()
    out["ndt_fixedstep_T"] = r["T"]
    out["ndt_fixedstep_iters"] = np.array([r["iterations"], r["evaluations"], int(r["converged"])])
    v = o.voxels()
    for k in ("keys", "counts", "valid", "mean", "icov"):
        out[f"ndt_vox_{k}"] = v[k]
    # GICP
    for reg in ("PLANE", "FROBENIUS"):
        g = orc.GicpOracle(regularization=reg, max_correspondence_distance=2.0)
        g.set_target(tgt)
        g.set_source(src)
        out[f"gicp_{reg}_cov_source"] = g.covariances("source")
        e, H, b = g.linearize(np.eye(4))
        out[f"gicp_{reg}_lin_err"] = e
        out[f"gicp_{reg}_lin_H"] = H
        out[f"gicp_{reg}_lin_b"] = b
        r = g.align()
        out[f"gicp_{reg}_T"] = r["T"]
        out[f"gicp_{reg}_iters"] = np.array([r["iterations"], r["evaluations"], int(r["converged"])])
    out["fitness"] = np.array(orc.fitness_score(tgt, src, out["gicp_PLANE_T"]))
    np.savez_compressed(os.path.join(HERE, "registration_small.npz"), **out)
    print("wrote", os.path.join(HERE, "registration_small.npz"))


def main_vgicp():
    """FAST_VGICP self-goldens, kept in their own file so that the first file never needs regenerating."""
    tgt, src, Tgt = synth.planar_pair(n=2048, seed_target=101, seed_source=102)
    out = dict(tgt=tgt, src=src, T_gt=Tgt)
    T1 = synth.make_transform((0.12, -0.07, 0.03), (0.01, -0.02, 0.03))
    for search in ("DIRECT1", "DIRECT7", "DIRECT27"):
        o = orc.VgicpOracle(resolution=1.0, search_method=search)
        o.set_target(tgt)
        o.set_source(src)
        if search == "DIRECT1":
            coords, counts, means, covs = o.voxels()
            out["vox_coords"], out["vox_counts"], out["vox_means"], out["vox_covs"] = coords, counts, means, covs
        e, H, b = o.linearize(T1)
        out[f"{search}_lin_err"], out[f"{search}_lin_H"], out[f"{search}_lin_b"] = e, H, b
        r = o.align()
        out[f"{search}_T"] = r["T"]
        out[f"{search}_iters"] = np.array([r["iterations"], r["evaluations"], int(r["converged"])])
    out["T1"] = T1
    np.savez_compressed(os.path.join(HERE, "vgicp_small.npz"), **out)
    print("wrote", os.path.join(HERE, "vgicp_small.npz"))


if __name__ == "__main__":
    if "--vgicp-only" not in sys.argv:
        main()
    if "--registration-only" not in sys.argv:
        main_vgicp()
