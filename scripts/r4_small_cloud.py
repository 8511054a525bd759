"""Round-4 check: a 2,048-point planar pair at the factory resolution (0.5 m), upstream order -- host source and device-resident source against
the oracle."""
import numpy as np
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration
from oracle import oracle as O

for n in (2048, 4096, 16384):
    tgt, src, _ = synth.planar_pair(n=n)
    o = O.NdtOracle(resolution=0.5)
    o.set_target(tgt); o.set_source(src)
    ro = o.align()
    print(n, "oracle", ro["converged"], ro["iterations"], ro["evaluations"])
    for order in (1, 2, 0):
        for cloud in (False, True):
            r = Registration("NDT_OMP", ndt_strict_order=order)
            r.setInputTarget(tgt)
            r.setInputSource(r.make_cloud(src) if cloud else src)
            r.align()
            lr = r.last_result
            print("  order", order, "cloud", cloud, r.hasConverged(), lr.iterations, lr.evaluations, np.abs(r.getFinalTransformation() - ro["T"]).max())
            if order == 1 and not cloud:
                tr = r.ndt_trajectory()
                k = min(len(tr), len(ro["trajectory"]))
                d = np.abs(tr[:k] - ro["trajectory"][:k]).max(1)
                print("   traj diffs", np.array2string(d[:8], precision=3), "first >1e-12 at", int(np.argmax(d > 1e-12)) if (d > 1e-12).any() else None)
