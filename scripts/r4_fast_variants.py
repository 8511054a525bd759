#!/usr/bin/env python3
"""Which of round 3's changes to the DEFAULT (fast) NDT evaluation order moves which pair: the hardware exp against the library expf and
the upstream orders' det_expf, N = A - M accumulated directly against A and M apart -- A/B builds (make -C delta_graph_slam_amd/csrc
variants) on the three 32 x 65,536 shards ranks 0-2 of bench.py register (seeds 40 / 1040 / 2040), against the oracle with the round-4
switches on (its default) and off (rounds 1-3's oracle).  For every pair outside the gate in ANY variant the oracle's own 34-twin band."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from delta_graph_slam_amd import synth  # noqa: E402
from delta_graph_slam_amd import _lib as L  # noqa: E402
from delta_graph_slam_amd.registration import Registration  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.helpers import pose_error  # noqa: E402

PKG = os.path.join(ROOT, "delta_graph_slam_amd")
VARIANTS = [("library expf, N direct (the product since round 4)", L.LIB_PATH), ("hw exp, N direct (round 3)", os.path.join(PKG, "libdgs_reg_v_hw.so")),
            ("det_expf, N direct", os.path.join(PKG, "libdgs_reg_v_det.so")), ("library expf, A and M apart (round 2)", os.path.join(PKG, "libdgs_reg_v_am.so")),
            ("hw exp, A and M apart", os.path.join(PKG, "libdgs_reg_v_hw_am.so"))]
TOL_M, TOL_RAD = 1e-4, 1e-5


def main():
    seeds = [int(x) for x in sys.argv[1:]] or [40, 1040, 2040]
    for oname, okw, polar in (("round-4 oracle (jsvd, double computeHessian, polar guess)", {}, 1),
                              ("rounds 1-3 oracle (switches off)", dict(newton_solver=0, hessian_recompute_double=0, guess_rotation_polar=0), 0)):
        for seed in seeds:
            tgt, sources, guesses, _ = synth.loop_batch(n_candidates=32, n_points=65536, seed=seed, distinct_scans=32)
            o = orc.NdtOracle(resolution=1.0, **okw)
            o.set_target(tgt)
            To = []
            for c in range(32):
                o.set_source(sources[c])
                To.append(o.align(guesses[c])["T"])
            rows, union = [], set()
            for name, path in VARIANTS:
                if not os.path.exists(path):
                    continue
                r = Registration("NDT_OMP", lib_path=path, ndt_resolution=1.0, ndt_strict_order=0, ndt_guess_rotation_polar=polar)
                r.setInputTarget(tgt)
                res = r.align_batch(sources, guesses, compute_fitness=False)
                e = np.array([pose_error(res[c]["T"], To[c]) for c in range(32)])
                ins = (e[:, 0] <= TOL_M) & (e[:, 1] <= TOL_RAD)
                out = [int(c) for c in np.nonzero(~ins)[0]]
                union.update(out)
                rows.append({"variant": name, "pairs_inside": int(ins.sum()), "max_m": float(e[:, 0].max()), "max_rad": float(e[:, 1].max()),
                             "outside": {str(c): [float(e[c, 0]), float(e[c, 1])] for c in out}})
                r.close()
            bands = {}
            twins = ((True, 0, 0), (False, 1, 0)) + tuple((False, 0, k) for k in range(-16, 17) if k)
            for c in sorted(union):
                _, bt, br = orc.ndt_band(tgt, sources[c], guesses[c], twins=twins, resolution=1.0, **okw)
                bands[str(c)] = [float(bt), float(br)]
            for row in rows:
                row["outside_all_inside_oracle_band"] = all(v[0] <= bands[c][0] + TOL_M and v[1] <= bands[c][1] + TOL_RAD for c, v in row["outside"].items())
                print(json.dumps(dict(oracle=oname, seed=seed, **row)), flush=True)
            print(json.dumps({"oracle": oname, "seed": seed, "oracle_band_34_twins_m_rad": bands}), flush=True)


if __name__ == "__main__":
    main()
