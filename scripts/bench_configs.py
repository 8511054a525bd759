#!/usr/bin/env python3
"""Measures every BASELINE.json config on one MI355X beside the CPU oracle (the numbers quoted in BASELINE.md / DESIGN.md).

bench.py stays the driver's contract (configs[1] pairs in the configs[3] per-GPU batch); this script adds the other rows:
  cfg1  16k planar pair, NDT 1.0 m            single align latency
  cfg2  65,536-pt HDL-64E pair, NDT 1.0 m     single align latency (identity guess and odometry-like guess), ms per evaluation
  cfg3  VLP-16 stream, FAST_GICP              per-frame latency p50 / p99 through the odometry driver (100 ms budget at 10 Hz)
  cfg5  200k indoor pair, NDT 0.5 m           single align latency + ndt_derivatives GB/s (the HBM-stress row)
Prints one JSON object per config.  usage: python scripts/bench_configs.py [--frames 100] [--cpu]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pose_error(Ta, Tb):
    Ta = np.asarray(Ta, np.float64)
    Tb = np.asarray(Tb, np.float64)
    dt = np.linalg.norm(Ta[:3, 3] - Tb[:3, 3])
    R = Ta[:3, :3].T @ Tb[:3, :3]
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return float(dt), float(np.arctan2(np.linalg.norm(w), 0.5 * (np.trace(R) - 1.0)))


def timed(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


ORDER_NAME = {0: "fast order", 1: "upstream order (the timed mode)"}
KERNEL = {0: "ndt_derivatives_kernel<DIRECT7, fused>", 1: "ndt_strict3_kernel<DIRECT7, fused>"}
REC_BYTES = {0: 48, 1: 64}


def ndt_pair(name, tgt, src, res, guess, reps, cpu, orc, threads, order=1):
    import torch
    from delta_graph_slam_amd import _lib as L
    from delta_graph_slam_amd.registration import Registration
    name = name + ", " + ORDER_NAME[order]
    reg = Registration("NDT_OMP", ndt_resolution=res, ndt_strict_order=order)
    dt_, ds_ = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()
    t_set, _ = timed(lambda: reg.setInputTarget(dt_), reps)
    reg.setInputSource(ds_)
    t_align, _ = timed(lambda: reg.align(guess), reps)
    t_fit, _ = timed(lambda: reg.getFitnessScore(), reps)
    ev = reg.last_result.evaluations
    reg.profile_enable(True)
    reg.profile_reset()
    for _ in range(reps):
        reg.align(guess)
    ms, n = reg.profile_get(L.K_NDT_DERIVATIVES)
    ms_s, n_s = reg.profile_get(L.K_NDT_SOLVE)
    reg.profile_enable(False)
    c = reg.counts()
    bytes_eval = 16 * src.shape[0] + REC_BYTES[order] * c["valid_voxels"] + 344
    out = {"config": name, "ndt_strict_order": order, "points": int(src.shape[0]), "resolution": res, "gpu_set_target_ms": 1e3 * t_set, "gpu_align_ms": 1e3 * t_align,
           "gpu_fitness_ms": 1e3 * t_fit, "iterations": reg.last_result.iterations, "evaluations": ev, "converged": bool(reg.hasConverged()),
           "gpu_ms_per_evaluation": 1e3 * t_align / max(ev, 1), "ndt_derivatives_us": 1e3 * ms / max(n, 1), "ndt_solve_us": 1e3 * ms_s / max(n_s, 1),
           "ndt_derivatives_GBps": bytes_eval * ev * reps / (ms * 1e-3) / 1e9 if ms > 0 else None, "valid_voxels": c["valid_voxels"],
           "gpu_registrations_per_s": 1.0 / t_align}
    # the per-kernel roofline case "in isolation" -- one pair, every launch is one evaluation of it
    ach = bytes_eval * n / (ms * 1e-3) / 1e9 if ms > 0 else None
    out["roofline"] = {"bound": "hbm", "kernel": KERNEL[order] + " (one pair: derivatives + the optimiser step in the closing workgroup)",
                       "algorithmic_bytes_per_launch": bytes_eval, "launches": n, "avg_launch_us": 1e3 * ms / max(n, 1), "achieved": ach, "peak": 8000.0,
                       "unit": "GB/s", "frac": ach / 8000.0 if ach else None, "traffic": None,
                       "points_per_s": src.shape[0] * n / (ms * 1e-3) if ms > 0 else None}
    if cpu:
        o = orc.NdtOracle(resolution=res, num_threads=threads)
        o.set_target(tgt)
        o.set_source(src)
        ro = o.align(guess)
        t_cpu, _ = timed(lambda: o.align(guess), 3, warm=1)
        o1 = orc.NdtOracle(resolution=res, num_threads=1)
        o1.set_target(tgt)
        o1.set_source(src)
        t_cpu1, _ = timed(lambda: o1.align(guess), 1, warm=0)
        et, er = pose_error(reg.getFinalTransformation(), ro["T"])
        out.update({"cpu_align_ms": 1e3 * t_cpu, "cpu_threads": threads, "cpu_align_ms_1thread": 1e3 * t_cpu1, "speedup": t_cpu / t_align,
                    "vs_oracle_translation_m": et, "vs_oracle_rotation_rad": er, "oracle_iterations": ro["iterations"]})
    return out


def gicp_pair(name, method, tgt, src, guess, reps, cpu, orc, threads, **kw):
    """Single align of a FAST_GICP / FAST_VGICP pair: first align (index + covariances + model build) and repeat aligns."""
    import torch
    from delta_graph_slam_amd.registration import Registration
    reg = Registration(method, **kw)
    dt_, ds_ = torch.from_numpy(tgt).cuda(), torch.from_numpy(src).cuda()

    def cold():
        reg.setInputTarget(dt_)
        reg.setInputSource(ds_)
        reg.align(guess)
    t_cold, _ = timed(cold, reps)
    t_warm, _ = timed(lambda: reg.align(guess), reps)
    out = {"config": name, "points": int(src.shape[0]), "gpu_first_align_ms": 1e3 * t_cold, "gpu_repeat_align_ms": 1e3 * t_warm,
           "iterations": reg.last_result.iterations, "evaluations": reg.last_result.evaluations, "converged": bool(reg.hasConverged())}
    # ---- roofline objects (SURVEY.md 8d byte formulas; HIP-event kernel times on the handle's stream)
    from delta_graph_slam_amd import _lib as L
    Ns, Nt = int(src.shape[0]), int(tgt.shape[0])
    reg.profile_enable(True)
    reg.profile_reset()
    cold()
    ms_cov, n_cov = reg.profile_get(L.K_GICP_COVARIANCE)
    reg.profile_reset()
    for _ in range(reps):
        reg.align(guess)
    ms_lin, n_lin = reg.profile_get(L.K_GICP_LINEARIZE)
    ms_nn, n_nn = reg.profile_get(L.K_NN_SEARCH)
    reg.profile_enable(False)
    cov_bytes = (Ns + Nt) * (16 + 24)                                      # read the points, write a symmetric 3x3 per point
    lin_bytes = Ns * 40 + Ns * 40 + 16 * Nt + 344                          # per linearisation, Nc taken as Ns (upper bound)
    out["roofline"] = {
        "covariance": {"bound": "hbm", "kernel": "gicp_knn_leaf_kernel + gicp_cov_from_knn_kernel (both clouds)", "algorithmic_bytes": cov_bytes, "kernel_ms": ms_cov,
                       "launches": n_cov, "achieved": cov_bytes / (ms_cov * 1e-3) / 1e9 if ms_cov > 0 else None, "peak": 8000.0, "unit": "GB/s",
                       "frac": cov_bytes / (ms_cov * 1e-3) / 1e9 / 8000.0 if ms_cov > 0 else None},
        "linearize": {"bound": "hbm", "kernel": ("vgicp_linearize_kernel" if method == "FAST_VGICP" else "gicp_correspond_kernel + gicp_linearize_kernel"),
                      "algorithmic_bytes_per_evaluation": lin_bytes, "launches": n_lin, "avg_launch_us": 1e3 * (ms_lin + ms_nn) / max(n_lin, 1),
                      "achieved": lin_bytes * n_lin / ((ms_lin + ms_nn) * 1e-3) / 1e9 if ms_lin + ms_nn > 0 else None, "peak": 8000.0, "unit": "GB/s",
                      "frac": lin_bytes * n_lin / ((ms_lin + ms_nn) * 1e-3) / 1e9 / 8000.0 if ms_lin + ms_nn > 0 else None,
                      "correspond_ms_total": ms_nn, "linearize_ms_total": ms_lin}}
    if cpu:
        if method == "FAST_VGICP":
            o = orc.VgicpOracle(resolution=kw.get("vgicp_resolution", 1.0), num_threads=threads)
        else:
            o = orc.GicpOracle(max_correspondence_distance=kw.get("gicp_max_correspondence_distance", 2.5), num_threads=threads)

        def cpu_cold():
            o.set_target(tgt)
            o.set_source(src)
            return o.align(guess)
        ro = cpu_cold()
        t_cpu, _ = timed(cpu_cold, 2, warm=0)
        et, er = pose_error(reg.getFinalTransformation(), ro["T"])
        out.update({"cpu_first_align_ms": 1e3 * t_cpu, "cpu_threads": threads, "speedup_first_align": t_cpu / t_cold,
                    "vs_oracle_translation_m": et, "vs_oracle_rotation_rad": er, "oracle_iterations": ro["iterations"]})
    return out


def pmc_traffic(extra_args, kernel_substring):
    """HBM bytes per launch of one kernel from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate
    rocprofv3 --pmc passes (kernel trace only), KiB units, the read side doubled on gfx950 (an upper bound for gathers)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None or any(k.startswith(("ROCP_", "ROCPROF", "HSA_TOOLS_LIB")) for k in os.environ):
        return None, {"error": "no rocprofv3, or already under a profiler"}
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="dgs_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__)] + extra_args
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        except subprocess.TimeoutExpired:
            return None, {"error": "rocprofv3 child timed out"}
        tot, n = 0.0, 0
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = {}
            for r in csv.DictReader(open(f)):
                if kernel_substring in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            tot += sum(per.values())
            n += len(per)
        shutil.rmtree(d, ignore_errors=True)
        res[counter] = (tot * 1024.0 / n) if n else None
        res[counter + "_launches"] = n
    if res.get("FETCH_SIZE") is None or res.get("WRITE_SIZE") is None:
        return None, res
    return 2.0 * res["FETCH_SIZE"] + res["WRITE_SIZE"], res


def cfg5_batched(order, reps, traffic=True, n_pairs=8):
    """BASELINE configs[4] as the HBM stress it is named as: 8 pairs of 200,000-point indoor scans against one target, NDT
    0.5 m, every derivative launch covers all pairs still iterating -- the roofline of the iteration kernel with the chip full."""
    import torch
    from delta_graph_slam_amd import _lib as L
    from delta_graph_slam_amd import synth
    from delta_graph_slam_amd.registration import Registration
    tgt, _, _ = synth.indoor_pair()
    srcs = [synth.indoor_pair(seed_source=51 + 7 * k, t_gt=(0.10 - 0.02 * k, 0.05, 0.0), r_gt=(0.0, 0.0, 0.03 - 0.005 * k))[1] for k in range(n_pairs)]
    reg = Registration("NDT_OMP", ndt_resolution=0.5, ndt_strict_order=order)
    reg.setInputTarget(torch.from_numpy(tgt).cuda())
    ds = [torch.from_numpy(x).cuda() for x in srcs]
    t_batch, _ = timed(lambda: reg.align_batch(ds, None, compute_fitness=False), reps)
    reg.profile_enable(True)
    reg.profile_reset()
    ev = 0
    for _ in range(reps):
        res = reg.align_batch(ds, None, compute_fitness=False)
        ev += reg.counts()["evaluations"]
    ms, n = reg.profile_get(L.K_NDT_DERIVATIVES)
    reg.profile_enable(False)
    c = reg.counts()
    bytes_eval = 16 * srcs[0].shape[0] + REC_BYTES[order] * c["valid_voxels"] + 344
    ach = bytes_eval * ev / (ms * 1e-3) / 1e9 if ms > 0 else None
    out = {"config": "cfg5_batched: %d pairs of 200,000-point indoor scans per launch, NDT 0.5 m DIRECT7, %s" % (n_pairs, ORDER_NAME[order]), "ndt_strict_order": order,
           "pairs": n_pairs, "points": int(srcs[0].shape[0]), "gpu_batch_ms": 1e3 * t_batch, "gpu_registrations_per_s": n_pairs / t_batch,
           "evaluations_per_batch": ev / reps, "converged": [bool(r["converged"]) for r in res], "valid_voxels": c["valid_voxels"],
           "roofline": {"bound": "hbm", "kernel": KERNEL[order], "bytes_per_evaluation": bytes_eval, "launches": n, "avg_launch_us": 1e3 * ms / max(n, 1),
                        "algorithmic_bytes_per_launch": bytes_eval * ev / max(n, 1), "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0 if ach else None,
                        "traffic": None, "points_per_s": srcs[0].shape[0] * ev / (ms * 1e-3) if ms > 0 else None}}
    if traffic:
        t, detail = pmc_traffic(["--only-cfg5-batched", "--orders", str(order), "--reps", "4", "--no-cpu"], "ndt_strict3_kernel" if order == 1 else "ndt_derivatives_kernel")
        out["roofline"]["traffic"] = t
        out["roofline"]["traffic_detail"] = detail
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    ap.add_argument("--only-ndt", action="store_true", help="the three single-pair NDT rows only (launch-shape sweeps: DGS_NDT_PPT / DGS_NDT_CAP / DGS_NDT_BLOCKS)")
    ap.add_argument("--orders", type=int, nargs="*", default=[1, 0], help="NDT evaluation orders of the NDT rows (dgs_params.ndt_strict_order)")
    ap.add_argument("--only-cfg5-batched", action="store_true", help="the cfg5_batched row only (the child runs of its PMC traffic passes)")
    ap.add_argument("--no-traffic", action="store_true")
    args = ap.parse_args()
    if args.only_cfg5_batched:
        for order in args.orders:
            print(json.dumps(cfg5_batched(order, max(2, args.reps // 2), traffic=False)), flush=True)
        return
    import torch
    from delta_graph_slam_amd import synth
    from delta_graph_slam_amd.odometry import ScanMatchingOdometry
    from delta_graph_slam_amd.registration import Registration
    from oracle import oracle as orc
    cpu = not args.no_cpu
    th = min(args.cpu_threads, orc.max_threads())

    tgt, src, Tgt = synth.planar_pair()
    for order in args.orders:
        print(json.dumps(ndt_pair("cfg1 planar 16k, NDT 1.0 m, identity guess", tgt, src, 1.0, None, args.reps, cpu, orc, th, order=order)), flush=True)

    tgt, src, Tgt = synth.kitti_pair()
    for order in args.orders:
        print(json.dumps(ndt_pair("cfg2 HDL-64E 65,536 pair, NDT 1.0 m, identity guess", tgt, src, 1.0, None, args.reps, cpu, orc, th, order=order)), flush=True)
    g = Tgt.copy()
    g[0, 3] -= 0.25
    g[1, 3] += 0.10
    for order in args.orders:
        print(json.dumps(ndt_pair("cfg2 HDL-64E 65,536 pair, NDT 1.0 m, odometry-like guess (0.27 m off)", tgt, src, 1.0, g.astype(np.float32), args.reps, cpu, orc, th, order=order)), flush=True)

    if not args.only_ndt:
      print(json.dumps(gicp_pair("cfg2 HDL-64E 65,536 pair, FAST_GICP dmax 2.5, odometry-like guess", "FAST_GICP", tgt, src, g.astype(np.float32), args.reps, cpu, orc, th)), flush=True)
      print(json.dumps(gicp_pair("cfg2 HDL-64E 65,536 pair, FAST_VGICP res 1.0 DIRECT1, odometry-like guess", "FAST_VGICP", tgt, src, g.astype(np.float32), args.reps, cpu, orc, th, vgicp_resolution=1.0)), flush=True)

    tgt, src, Tgt = synth.indoor_pair()
    for order in args.orders:
        print(json.dumps(ndt_pair("cfg5 indoor 200k pair, NDT 0.5 m, identity guess", tgt, src, 0.5, None, max(3, args.reps // 2), cpu, orc, th, order=order)), flush=True)

    for order in args.orders:   # cfg5 as the stress row BASELINE names: 8 pairs of 200,000 points per launch, the chip full
        print(json.dumps(cfg5_batched(order, max(3, args.reps // 2), traffic=not args.no_traffic)), flush=True)
    if args.only_ndt:
        return
    # ---- cfg3: VLP-16 stream through the odometry driver, FAST_GICP with the launch-file values
    clouds, poses = synth.vlp16_stream(n_frames=args.frames)
    dclouds = [torch.from_numpy(c).cuda() for c in clouds]
    kw = dict(keyframe_delta_trans=1.0, keyframe_delta_angle=1.0, keyframe_delta_time=1e9)
    odo = ScanMatchingOdometry(Registration("FAST_GICP", gicp_max_correspondence_distance=2.0, transformation_epsilon=0.1), kw)
    lat = []
    traj = []
    for k, c in enumerate(dclouds):
        t0 = time.perf_counter()
        traj.append(odo.matching(0.1 * k, c))
        lat.append(time.perf_counter() - t0)
    lat_ms = 1e3 * np.array(lat[1:])
    gt = [np.linalg.inv(poses[0]) @ p for p in poses]
    drift = float(np.linalg.norm(traj[-1][:3, 3] - gt[-1][:3, 3]))
    out = {"config": "cfg3 VLP-16 stream (%d frames, ~26k pts), FAST_GICP k=20 dmax 2.0 eps 0.1, odometry driver" % args.frames,
           "gpu_frame_ms_p50": float(np.percentile(lat_ms, 50)), "gpu_frame_ms_p99": float(np.percentile(lat_ms, 99)), "gpu_frame_ms_max": float(lat_ms.max()),
           "budget_ms": 100.0, "keyframes": odo.n_keyframes, "end_drift_m": drift, "path_length_m": float(np.linalg.norm(gt[-1][:3, 3]))}
    if cpu:
        from tests.oracle_engine import OracleRegistration
        n_cpu = min(args.frames, 12)
        oc = ScanMatchingOdometry(OracleRegistration("FAST_GICP", max_correspondence_distance=2.0, transformation_epsilon=0.1, num_threads=th), kw)
        lc = []
        errs = []
        for k in range(n_cpu):
            t0 = time.perf_counter()
            T = oc.matching(0.1 * k, clouds[k])
            lc.append(time.perf_counter() - t0)
            errs.append(pose_error(traj[k], T))
        out.update({"cpu_frame_ms_p50": float(1e3 * np.median(lc[1:])), "cpu_threads": th, "cpu_frames": n_cpu,
                    "vs_oracle_max_translation_m": float(max(e[0] for e in errs)), "vs_oracle_max_rotation_rad": float(max(e[1] for e in errs)),
                    "speedup_p50": float(np.median(lc[1:]) / (np.percentile(lat_ms, 50) * 1e-3))})
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
