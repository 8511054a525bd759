#!/usr/bin/env python3
"""How much do independent launch sequences on separate streams overlap?  G handles (one host thread each), each aligning 32 / G of the bench
step's candidates against the same target, against ONE handle aligning all 32.  align_batch only (target set once, fitness included)."""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--groups", type=int, nargs="*", default=[1, 2, 4])
    ap.add_argument("--no-fitness", action="store_true")
    a = ap.parse_args()
    import torch
    from delta_graph_slam_amd import synth
    from delta_graph_slam_amd.registration import Registration
    dev = torch.device("cuda", 0)
    P = 32
    tgt, sources, guesses, gts = synth.loop_batch(n_candidates=P, n_points=65536, seed=40, distinct_scans=P)
    tgt_d = torch.from_numpy(tgt).to(dev)
    src_d = [torch.from_numpy(s).to(dev) for s in sources]
    for G in a.groups:
        regs = [Registration("NDT_OMP", device=0, ndt_resolution=1.0, ndt_strict_order=a.order) for _ in range(G)]
        for r in regs:
            r.setInputTarget(tgt_d)
        parts = [list(range(g, P, G)) for g in range(G)]
        results = [None] * G

        def work(g):
            idx = parts[g]
            results[g] = regs[g].align_batch([src_d[i] for i in idx], [guesses[i] for i in idx], compute_fitness=not a.no_fitness)

        def step():
            if G == 1:
                work(0)
                return
            th = [threading.Thread(target=work, args=(g,)) for g in range(G)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"order": a.order, "groups": G, "ms_per_step": 1e3 * dt / a.steps, "fitness": not a.no_fitness}), flush=True)
        for r in regs:
            r.close()


if __name__ == "__main__":
    main()
