import os
import numpy as np
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
from oracle import oracle as O

tgt, src, _ = synth.planar_pair(n=2048)
o = O.NdtOracle(resolution=0.5)
o.set_target(tgt); o.set_source(src)
ro = o.align()
s, g, H = o.derivatives(np.zeros(6))
print("oracle", ro["converged"], ro["iterations"], "sv(H)", np.linalg.svd(H, compute_uv=False))
print("traj1", ro["trajectory"][:2])
def run(env, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        r = Registration("NDT_OMP", ndt_strict_order=1, **kw)
    finally:
        for k, v in old.items():
            if v is None: del os.environ[k]
            else: os.environ[k] = v
    r.setInputTarget(tgt); r.setInputSource(src); r.align()
    lr = r.last_result
    print(env, kw, r.hasConverged(), lr.iterations, lr.evaluations, np.abs(r.getFinalTransformation() - ro["T"]).max(), flush=True)
    return r
r = run({})
sg, gg, Hg = r.ndt_derivatives(np.zeros(6))
print("dH", np.abs(Hg - H).max(), "dg", np.abs(gg - g).max(), "finite", np.isfinite(Hg).all())
run({"DGS_NDT_SPECULATE": "0"})
run({"DGS_NDT_STRICT_KERNEL": "2"})
run({"DGS_NDT_FUSED": "0"})
run({}, ndt_newton_solver=0)
run({}, ndt_hessian_recompute_double=0)
run({}, ndt_resolution=1.0)
b = r.align_batch([src, src[:2000], src[:1024]], None)
print([(x["converged"], x["iterations"]) for x in b])
