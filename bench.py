#!/usr/bin/env python3
"""bench.py -- scan registrations/sec on 64k-pt KITTI-shaped pairs (BASELINE.json metric), MI355X.

One "step" = one LoopDetector.matching() pass (/root/reference/include/hdl_graph_slam/loop_detector.hpp:119-173)
over a batch of candidate registrations per GPU: setInputTarget(new keyframe) once, then for each of P candidate
65,536-point HDL-64E-shaped source scans: setInputSource, align(yaw/xy guess), getFitnessScore; then the arg-min.
Every pair is the BASELINE configs[1] workload (NDT, 1.0 m resolution, DIRECT7, 64 max iterations); P pairs per GPU
is the per-GPU shard of configs[3] (256 candidates over 8 GPUs = 32).  All clouds are resident in HBM before the
timed region.  Multi-GPU, two forms with the same dealing and the same single exchange step (an all-gather of result records):
  * one process per GPU (torch.distributed, backend nccl = RCCL) -- the default, what the driver launches;
  * --group: ONE process driving N devices through the C ABI's dgs_group (one handle + host thread + stream per device,
    ncclCommInitAll / ncclAllGather) -- the form a nodelet links (apps/delta_graph_slam_nodelet.cpp:797,816).
Weak scaling either way (P candidates per GPU fixed).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python bench.py --gpus N ...          # starts N ranks itself (torch.distributed.run child, RCCL) when WORLD_SIZE is unset
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
  python bench.py --gpus N --group ...  # one process, N devices

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`, `cpu_baseline` and `parity_gate` objects.
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 GB/s achievable
TOL_M, TOL_RAD = 1e-4, 1e-5   # BASELINE.json north_star: "final pose RMSE within 1e-4 m / 1e-5 rad of reference"
METRIC = "scan registrations/sec (64k-pt pairs)"
DTYPE = "f32 per-point, f64 accumulate"


def pose_error(Ta, Tb):
    Ta = np.asarray(Ta, np.float64)
    Tb = np.asarray(Tb, np.float64)
    dt = np.linalg.norm(Ta[:3, 3] - Tb[:3, 3])
    R = Ta[:3, :3].T @ Tb[:3, :3]
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return dt, float(np.arctan2(np.linalg.norm(w), 0.5 * (np.trace(R) - 1.0)))


def sequential_best(converged, fitness):
    """loop_detector.hpp:126-156: skip a candidate iff it did not converge or its score is greater than the best so far."""
    best, best_score = -1, 1.7976931348623157e308
    for c, (ok, s) in enumerate(zip(converged, fitness)):
        if (not ok) or s > best_score:
            continue
        best, best_score = c, s
    return best, best_score


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pairs", type=int, default=32, help="candidate registrations per GPU per step")
    ap.add_argument("--points", type=int, default=65536)
    ap.add_argument("--distinct-scans", type=int, default=0, help="distinct ray-cast source scans per GPU (re-used round-robin); 0 = one per candidate")
    ap.add_argument("--group", action="store_true", help="one process, --gpus devices behind the C ABI (dgs_group) instead of one process per GPU")
    ap.add_argument("--dry-run", action="store_true", help="launcher / collective plumbing only (no HIP work, runs without a GPU): the step "
                                                           "is the all_gather of empty records; value is null")
    ap.add_argument("--resolution", type=float, default=1.0)
    ap.add_argument("--order", type=int, default=1, choices=(0, 1),
                    help="NDT evaluation order of the TIMED mode (dgs_params.ndt_strict_order): 1 = upstream's operation order, the mode that "
                         "reproduces the reference's transforms (default); 0 = the re-associated fast order, reported beside it as fast_value")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU budget of the cpu_baseline sample (the child process's timed work)")
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)   # internal: the cpu_baseline leg's own process
    ap.add_argument("--traffic", dest="traffic", action="store_true", default=True,
                    help="measure HBM traffic of the dominant kernel: two short child runs under rocprofv3 (--pmc FETCH_SIZE, then "
                         "WRITE_SIZE); default at N=1 when rocprofv3 is on PATH")
    ap.add_argument("--no-traffic", dest="traffic", action="store_false", help="leave roofline.traffic null (no rocprofv3 child runs)")
    ap.add_argument("--traffic-dir", default=os.path.join(ROOT, "gpurun_out", "traffic"))
    args = ap.parse_args(argv)
    if args.distinct_scans <= 0:
        args.distinct_scans = args.pairs
    return args


_LINE_FD = None


def claim_stdout():
    global _LINE_FD
    if _LINE_FD is None:
        sys.stdout.flush()
        _LINE_FD = os.dup(1)
        os.dup2(2, 1)


def emit_line(line: str):
    sys.stdout.flush()
    os.write(_LINE_FD if _LINE_FD is not None else 1, (line + "\n").encode())


def main():
    args = parse_args()
    if args.cpu_baseline_child:
        return cpu_baseline_child(args)

    # ---- `bench.py --gpus N` on its own: start the N ranks here, BEFORE anything touches the GPU (a process that has
    # initialised HIP must never be replaced or forked), as a child `torch.distributed.run`, and leave with its status.
    if args.gpus > 1 and not args.group and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))

    # From here on file descriptor 1 belongs to the ONE JSON line: libraries write to it too (RCCL prints its version banner there on some
    # boxes), so everything else -- Python's prints and C's alike -- is sent to stderr and the line goes out through the saved descriptor.
    claim_stdout()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0")) if not args.group else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if not args.group else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if not args.group else 1
    if not args.group and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    # one process per GPU; DGS_BENCH_BACKEND=gloo lets several ranks share one card (or none: --dry-run) to rehearse the path
    backend = os.environ.get("DGS_BENCH_BACKEND", "nccl")
    if args.dry_run:
        return dry_run(args, rank, world, backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from delta_graph_slam_amd import _lib as L
    from delta_graph_slam_amd import synth
    from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
    from delta_graph_slam_amd.registration import Registration, RegistrationGroup

    P = args.pairs
    reg_kw = dict(ndt_resolution=args.resolution, ndt_search_method=L.NDT_SEARCH["DIRECT7"], transformation_epsilon=0.01, maximum_iterations=64)
    timed_kw = dict(reg_kw, ndt_strict_order=args.order)   # the timed mode; reg_kw stays order-free for the other legs
    ORDER_NAME = {0: "fast (dgs_params.ndt_strict_order = 0: the per-point float math re-associated, 1/3 of the flops)",
                  1: "upstream_order (dgs_params.ndt_strict_order = 1: every float operation of upstream's update in upstream's order, double sums in the GPU's fixed order)"}
    KERNEL = {0: "ndt_derivatives_kernel<DIRECT7, fused>", 1: "ndt_strict3_kernel<DIRECT7, fused>"}
    dev = torch.device("cuda", local_rank)

    def keyframes(tgt, sources, guesses, on_device, first_id=1):
        put = (lambda a: torch.from_numpy(a).to(dev)) if on_device else (lambda a: a)
        uploaded = {}
        new_kf = KeyFrame(cloud=put(tgt), estimate=np.eye(3), accum_distance=100.0, id=0)
        out = []
        for c in range(len(sources)):
            G = guesses[c]
            est = np.eye(3)
            est[:2, :2] = G[:2, :2]
            est[:2, 2] = G[:2, 3]
            if id(sources[c]) not in uploaded:
                uploaded[id(sources[c])] = put(sources[c])
            out.append(KeyFrame(cloud=uploaded[id(sources[c])], estimate=est, accum_distance=0.0, id=first_id + c))
        return new_kf, out

    if args.group:
        # ---- one process, G devices: ONE target keyframe, G * P candidate keyframes resident on their owners (candidate c -> member
        # c mod G); DGS_BENCH_GROUP_DEVICES="0,0" rehearses several members on one card
        devices = [int(x) for x in os.environ["DGS_BENCH_GROUP_DEVICES"].split(",")] if os.environ.get("DGS_BENCH_GROUP_DEVICES") else list(range(args.gpus))
        if len(devices) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but DGS_BENCH_GROUP_DEVICES lists %d devices" % (args.gpus, len(devices)))
        G = len(devices)
        n_total = G * P
        distinct = min(args.distinct_scans * G, n_total) if G == 1 else min(args.distinct_scans, n_total)
        tgt, sources, guesses, gts = synth.loop_batch(n_candidates=n_total, n_points=args.points, seed=40, distinct_scans=distinct)
        new_kf, cands = keyframes(tgt, sources, guesses, on_device=False)
        new_kf.cloud = torch.from_numpy(tgt).pin_memory()   # the new keyframe waits in pinned host memory: its upload inside the step is one DMA per member
        # the group keeps keyframes by id: scan s is shared by the candidates s, s + distinct, ... -> give those the id of the scan so
        # that each distinct scan is resident once (on member s mod G); the guesses stay per candidate
        for c, k in enumerate(cands):
            k.cache_id = 1 + (c % distinct)
        reg = RegistrationGroup("NDT_OMP", devices=devices, **timed_kw)
        det = _GroupDetector({"fitness_score_thresh": 1e9}, registration=reg, cache_clouds=True)
        det.matching(cands, new_kf)   # uploads every CANDIDATE keyframe once (KeyFrame::cloud is immutable, keyframe.hpp:51): not a timed step
        prof_reg = reg.member(0)
        collective = "rccl (ncclCommInitAll)" if reg.uses_rccl else "host gather (a device listed twice, or RCCL unavailable)"
        n_dev = G

        def step():
            return det.matching(cands, new_kf)

        def barrier():
            for d in sorted(set(devices)):
                torch.cuda.synchronize(d)
    else:
        # ---- synthetic workload (seeded; rank-specific scans), uploaded to HBM before timing
        tgt, sources, guesses, gts = synth.loop_batch(n_candidates=P, n_points=args.points, seed=40 + 1000 * rank,
                                                     distinct_scans=min(args.distinct_scans, P))
        distinct = min(args.distinct_scans, P)
        new_kf, cands = keyframes(tgt, sources, guesses, on_device=True)
        reg = Registration("NDT_OMP", device=local_rank, **timed_kw)
        det = LoopDetector({"fitness_score_thresh": 1e9}, registration=reg)
        prof_reg = reg
        collective = dist.get_backend() if world > 1 else None
        n_dev = world

        # the detector shards candidates[rank::world]; give every rank its own P candidates by offering a world*P list
        # whose rank-th stride is this rank's data (other entries are never touched by this rank)
        def step():
            if world == 1:
                return det.matching(cands, new_kf)
            full = [None] * (world * P)
            full[rank::world] = cands
            # only this rank's entries are dereferenced by register_shard
            return det.matching(_Sparse(full, cands[0]), new_kf)

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()            # nothing but the detector's own work inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    dt_own = dt
    ex_calls = max(getattr(det, "exchange_calls", 0), 1)
    ex_us_own = 1e6 * getattr(det, "exchange_seconds", 0.0) / ex_calls
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = torch.tensor([1e3 * dt_own / args.steps, ex_us_own], dtype=torch.float64, device=t.device)
        allr = torch.empty(world * 2, dtype=torch.float64, device=t.device)   # flat: gloo (the CPU rehearsal) chunks the output and wants chunks of the input's shape
        dist.all_gather_into_tensor(allr, mine)
        per_rank = allr.cpu().numpy().reshape(world, 2)
    regs = n_dev * P * args.steps
    value = regs / dt
    records = det.last_records

    # ---- the other evaluation order on the same step, timed the same way (rank-local: no collective inside it)
    other = 1 - args.order
    other_line = None
    if not args.group:
        reg_o = Registration("NDT_OMP", device=local_rank, **dict(reg_kw, ndt_strict_order=other))
        det_o = LoopDetector({"fitness_score_thresh": 1e9}, registration=reg_o, local_only=True)
        for _ in range(min(args.warmup, 3)):
            det_o.matching(cands, new_kf)
        torch.cuda.synchronize()
        to0 = time.perf_counter()
        for _ in range(args.steps):
            det_o.matching(cands, new_kf)
        torch.cuda.synchronize()
        dto = time.perf_counter() - to0
        other_line = {"ndt_order": ORDER_NAME[other], "value": P * args.steps / dto, "ms_per_step": 1e3 * dto / args.steps,
                      "note": "registrations/s of ONE GPU's step in the other evaluation order (no exchange step), same candidates"}
        records_other = det_o.last_records.copy()
        reg_o.close()

    out = {
        "metric": METRIC, "value": value, "unit": "registrations/s", "n_gpus": n_dev,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "cfg2 KITTI HDL-64E-shaped pairs (65,536 pts after voxel filter), NDT res %.1f m DIRECT7, eps 0.01, "
                               "max 64 iterations; %d candidate pairs per GPU per step against one target (LoopDetector::matching, "
                               "cfg4 shard), %d distinct source scans, fitness score per candidate; inputs resident on the device before "
                               "timing -- the %.0f MB working set of a step fits the 256 MB Infinity Cache, so re-reads between "
                               "evaluations are served on-die and the HBM roofline is an upper bound of what the kernels could use"
                               % (args.resolution, P, distinct, (min(distinct, P) + 1) * args.points * 16 / 1e6),
                   "pairs_per_gpu": P, "points_per_scan": args.points, "distinct_scans": distinct,
                   "ndt_order": ORDER_NAME[args.order] + "; see parity_gate for what that means against the reference",
                   "parallelism": ("one process, %d devices behind the C ABI (dgs_group): candidate c -> member c mod G, candidate keyframes resident on "
                                   "their owners, the new keyframe uploaded to every member inside the step (1 MB per member over PCIe, counted in "
                                   "the timed region), records written on the device and all-gathered" % n_dev) if args.group else
                                  "candidates sharded one process per GPU, all_gather of result records",
                   "collective_backend": collective, "collective_world_size": n_dev},
    }

    # ---- roofline leg: the same steps with every ndt_derivatives launch bracketed by HIP events on its stream.  Every rank
    # runs it (the step contains the all_gather), rank 0 reports its own kernel timings (group mode: member 0's).
    prof_reg.profile_enable(True)
    prof_reg.profile_reset()
    ev2 = 0
    for _ in range(args.steps):
        step()
        ev2 += prof_reg.counts()["evaluations"]
    ms, launches = prof_reg.profile_get(L.K_NDT_DERIVATIVES)
    ms_solve, l_solve = prof_reg.profile_get(L.K_NDT_SOLVE)
    ms_nn, l_nn = prof_reg.profile_get(L.K_NN_SEARCH)
    ms_vox, l_vox = prof_reg.profile_get(L.K_NDT_VOXEL_BUILD)
    prof_reg.profile_enable(False)

    alone = None
    if not args.group:
        # the same workload through (derivatives, solve) launch pairs: the derivative phase alone, for continuity with round 1's figure
        os.environ["DGS_NDT_FUSED"] = "0"
        reg_u = Registration("NDT_OMP", device=local_rank, **timed_kw)
        del os.environ["DGS_NDT_FUSED"]
        det_u = LoopDetector({"fitness_score_thresh": 1e9}, registration=reg_u, local_only=True)
        det_u.matching(cands, new_kf)
        reg_u.profile_enable(True)
        reg_u.profile_reset()
        ev_u = 0
        for _ in range(args.steps):
            det_u.matching(cands, new_kf)
            ev_u += reg_u.counts()["evaluations"]
        ms_u, launches_u = reg_u.profile_get(L.K_NDT_DERIVATIVES)
        ms_su, launches_su = reg_u.profile_get(L.K_NDT_SOLVE)
        reg_u.close()
        alone = (ev_u, ms_u, launches_u, ms_su, launches_su)

    if rank == 0:
        cnt = prof_reg.counts()
        Ns, Nt, V = args.points, cnt["target_points"], cnt["valid_voxels"]
        shard_regs = P if not args.group else len(range(0, n_dev * P, n_dev))   # registrations behind ev2 (rank 0's / member 0's share)
        out["ms_per_iter"] = 1e3 * dt / max(ev2, 1) * shard_regs   # wall ms per derivative evaluation of one pair stream (P run concurrently)
        out["evaluations_per_registration"] = ev2 / (shard_regs * args.steps)
        out["converged_fraction"] = float(np.mean(records[:, 1] > 0.5)) if records is not None else None
        rec_bytes = 64 if args.order == 1 else 48          # voxel record the kernel reads: 3 double means + 9 floats (upstream order) / 6 floats (fast order)
        bytes_per_eval = 16 * Ns + rec_bytes * V + 344     # stream source once, table once, 43 doubles out
        total_bytes = ev2 * bytes_per_eval
        achieved = total_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out["roofline"] = {"bound": "hbm", "kernel": KERNEL[args.order] + " (derivatives of every active pair + the optimiser step of each pair in "
                                                     "its closing workgroup)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                           "avg_launch_us": 1e3 * ms / max(launches, 1), "launches": launches,
                           "algorithmic_bytes_per_launch": total_bytes / max(launches, 1),
                           "bytes_per_evaluation": bytes_per_eval, "valid_voxels": V,
                           "other_kernels_ms_per_step": {"ndt_solve": ms_solve / args.steps, "nn_fitness": ms_nn / args.steps,
                                                         "voxel_build": ms_vox / args.steps, "ndt_derivatives": ms / args.steps}}
        if alone is not None:
            ev_u, ms_u, launches_u, ms_su, launches_su = alone
            ach_u = ev_u * bytes_per_eval / (ms_u * 1e-3) / 1e9 if ms_u > 0 else 0.0
            out["roofline"]["derivative_phase_alone"] = {
                "kernel": KERNEL[args.order].replace(", fused", "") + " as its own launch (DGS_NDT_FUSED=0, followed by ndt_solve_kernel)",
                "avg_launch_us": 1e3 * ms_u / max(launches_u, 1), "achieved": ach_u, "frac": ach_u / HBM_PEAK_GBS,
                "ndt_solve_avg_launch_us": 1e3 * ms_su / max(launches_su, 1)}

        # ---- CPU baseline (a child process with pinned OpenMP threads) + the parity gate against its poses
        if other_line is not None:
            out["fast_value" if other == 0 else "upstream_order_value"] = other_line["value"]
            out["other_order"] = other_line
        if per_rank is not None:
            out["per_rank"] = {"ms_per_step": [float(v) for v in per_rank[:, 0]], "exchange_step_us": [float(v) for v in per_rank[:, 1]],
                               "note": "every rank's own wall time per step and its host time in the exchange step (copy in, all_gather_into_tensor over "
                                       "RCCL, copy out, one synchronisation); `value` uses the slowest rank"}
        elif getattr(det, "exchange_calls", 0):
            out["exchange_step_us"] = ex_us_own
        if args.group:
            out["config"]["rccl_communicator_ranks"] = getattr(reg, "rccl_ranks", None)
        if not args.no_cpu_baseline and not args.group:
            # N = 1: the cpu_baseline leg (the contract's) + the parity gate on all its pairs.  N > 1: the contract reports cpu_baseline at
            # N = 1 only, but a multi-GPU line must still say how its transforms compare with the reference: rank 0 runs the oracle on ITS
            # shard (same child process) and states the gate for it.
            cpu = run_cpu_baseline(args, tgt, sources, LoopDetector.guesses_for(new_kf, cands))
            if "error" in cpu:
                if n_dev == 1:
                    out["cpu_baseline"] = {"value": None, "unit": "registrations/s", "cores": None, "kind": "port", "sample": cpu["error"]}
                else:
                    out["parity_gate"] = {"error": cpu["error"]}
            else:
                if n_dev == 1:
                    out["cpu_baseline"] = cpu["cpu_baseline"]
                    out["speedup_vs_cpu_baseline"] = value / cpu["cpu_baseline"]["value"]
                rec_mine = records[0::n_dev] if n_dev > 1 else records     # rank 0's candidates are 0, world, 2 world, ... of the gathered records
                out["parity_gate"], out["pose_rmse_vs_oracle"] = parity_legs(args, cpu, rec_mine, value, cands, new_kf, tgt, sources, reg_kw, local_rank,
                                                                              records_other if other_line is not None else None)
                if n_dev > 1:
                    out["parity_gate"]["scope"] = "rank 0's shard of the step (%d of %d candidates)" % (len(cands), n_dev * P)
        if n_dev == 1 and args.traffic and not args.group:
            try:
                out["roofline"]["traffic"], out["roofline"]["traffic_detail"] = measure_traffic(args, bytes_per_eval, "ndt_strict3_kernel" if args.order == 1 else "ndt_derivatives_kernel")
            except Exception as e:  # profiler missing / refused: the counter stays null, the bench line is still valid
                out["roofline"]["traffic_detail"] = {"error": repr(e)[:200]}
        emit_line(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ================================================================================================ CPU baseline (child process)
def run_cpu_baseline(args, tgt, sources, guesses_used):
    """Times the oracle (oracle/cpu/*.cpp, the C++/OpenMP restatement of NDT_OMP: kind "port") in a CHILD process whose OpenMP
    threads are pinned (OMP_PROC_BIND=close, OMP_PLACES=cores) -- in this process the binding would also pin the HIP runtime's
    threads.  The workload travels as a temporary .npz (the very float32 matrices the detector handed to align(): loop_detector.hpp:
    139-143 rebuilds the guesses from the 2-D keyframe estimates, which moves some entries by an ulp)."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory(prefix="dgs_cpu_") as d:
        path = os.path.join(d, "work.npz")
        uniq, index = [], []
        seen = {}
        for s in sources:
            if id(s) not in seen:
                seen[id(s)] = len(uniq)
                uniq.append(s)
            index.append(seen[id(s)])
        np.savez(path, tgt=tgt, scans=np.stack(uniq), index=np.array(index), guesses=np.asarray(guesses_used, np.float32))
        env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
        env.pop("OMP_NUM_THREADS", None)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", path, "--cpu-seconds", str(args.cpu_seconds),
               "--resolution", str(args.resolution)]
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=max(300.0, 20 * args.cpu_seconds))
        except subprocess.TimeoutExpired:
            return {"error": "cpu_baseline child timed out"}
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return {"error": "cpu_baseline child failed: " + (p.stderr or "")[-300:]}
        res = json.loads(lines[-1])
        res["T"] = np.load(os.path.join(d, "poses.npy"))
        return res


def cpu_baseline_child(args):
    """The cpu_baseline leg proper (no GPU, no torch): thread sweep on >= 8 pairs, three repeats of the bounded sample at the best
    thread count, the single-threaded getFitnessScore the reference runs per candidate on the same pairs."""
    from oracle import oracle as orc
    w = np.load(args.cpu_baseline_child)
    tgt, scans, index, guesses = w["tgt"], w["scans"], w["index"], w["guesses"]
    P = len(index)
    ncpu = orc.max_threads()

    def run(threads, limit, fitness=False):
        o = orc.NdtOracle(resolution=args.resolution, transformation_epsilon=0.01, max_iterations=64, search_method="DIRECT7", num_threads=threads)
        t0 = time.perf_counter()
        o.set_target(tgt)
        t_target = time.perf_counter() - t0
        res, per_pair, fits, t_fit = [], [], [], []
        for c in range(limit):
            tc = time.perf_counter()
            o.set_source(scans[index[c]])
            res.append(o.align(guesses[c]))
            per_pair.append(time.perf_counter() - tc)
            if fitness:   # pcl::Registration::getFitnessScore: one thread, one kd-tree query per source point
                tf = time.perf_counter()
                fits.append(orc.fitness_score(tgt, scans[index[c]], res[-1]["T"])[0])
                t_fit.append(time.perf_counter() - tf)
        return res, t_target, per_pair, fits, t_fit

    # thread sweep on 8 pairs each (the reference's reg_num_threads = 0 means "all cores"; on a many-core host fewer threads are
    # faster for 64k points)
    n_sweep = min(8, P)
    sweep = {}
    for th in sorted({ncpu, 64, 32, 16, 8}):
        if th <= ncpu:
            _, tt, pp, _, _ = run(th, n_sweep if not sweep or th <= 64 else 2)   # (all of a many-core host's threads: 2 pairs are enough to see it lose)
            sweep[th] = (n_sweep if not sweep or th <= 64 else 2) / (tt + sum(pp))
            if sweep[th] < 0.5 * max(sweep.values()):
                break   # past the knee: more threads only get slower
    best_th = max(sweep, key=sweep.get)
    per_pair_est = 1.0 / sweep[best_th]
    n = int(min(P, max(2, args.cpu_seconds / (3.0 * per_pair_est))))   # bounded sample: three repeats inside the budget
    reps = []
    for _ in range(3):
        res, tt, pp, _, _ = run(best_th, n)
        reps.append((tt, pp))
    totals = sorted(tt + sum(pp) for tt, pp in reps)
    rates = [n / t for t in totals]
    med_pair_ms = np.median(np.array([pp for _, pp in reps]), axis=0) * 1e3
    evals = np.array([r["evaluations"] for r in res], float)
    res_f, tt_f, pp_f, fits, t_fit = run(best_th, n, fitness=True)
    np.save(os.path.join(os.path.dirname(args.cpu_baseline_child), "poses.npy"), np.stack([r["T"] for r in res]))
    out = {"cpu_baseline": {
        "value": float(np.median(rates)), "unit": "registrations/s", "cores": int(best_th), "kind": "port",
        "value_spread_3_repeats": [float(min(rates)), float(max(rates))],
        "value_with_fitness_score": float(n / (totals[1] + sum(t_fit))),
        "fitness_score_ms_per_candidate_1_thread": float(1e3 * np.mean(t_fit)),
        "ms_per_pair": {"mean": float(med_pair_ms.mean()), "median": float(np.median(med_pair_ms)), "max": float(med_pair_ms.max())},
        "ms_per_evaluation": float(med_pair_ms.sum() / max(evals.sum(), 1.0)),
        "evaluations_per_registration": float(evals.mean()),
        "threads_available": int(ncpu), "thread_binding": "OMP_PROC_BIND=close OMP_PLACES=cores",
        "sweep_8_pairs_reg_per_s_by_threads": {str(k): round(v, 2) for k, v in sweep.items()},
        "sample": "%d of the %d candidate pairs of one step (setInputTarget once, then setInputSource + align per candidate), oracle C++/OpenMP "
                  "restatement in a child process with pinned threads, median of 3 repeats at %d threads (the best of a sweep over 8 pairs "
                  "per thread count); value_with_fitness_score adds the single-threaded getFitnessScore the reference runs per candidate "
                  "(loop_detector.hpp:148; kd-tree build included) on the SAME pairs" % (n, P, best_th)},
        "pairs": n, "converged": [bool(r["converged"]) for r in res], "fitness": [float(f) for f in fits]}
    print(json.dumps(out), flush=True)


# ================================================================================================ parity legs
def parity_legs(args, cpu, records, value, cands, new_kf, tgt, sources, reg_kw, device, records_other=None):
    """Final poses of the TIMED run (dgs_params.ndt_strict_order = args.order) and of the other evaluation orders against the oracle's, and
    the top-level gate object: does the timed mode meet north_star's tolerance on this workload."""
    import torch
    from delta_graph_slam_amd.loop_detector import LoopDetector
    from delta_graph_slam_amd.registration import Registration
    from oracle import oracle as orc
    T_cpu = cpu["T"]
    n_cmp = min(len(T_cpu), len(records))
    NAME = {0: "fast", 1: "upstream_order", 2: "upstream_order_sequential_sum"}

    def rms(a):
        return float(np.sqrt(np.mean(np.square(a)))) if len(a) else None

    def errors(T_list):
        return np.array([pose_error(T_list[c], T_cpu[c]) for c in range(n_cmp)])

    def parity(T_list):
        e = errors(T_list)
        return {"pairs": int(n_cmp), "pairs_within_1e-4m_1e-5rad": int(((e[:, 0] <= TOL_M) & (e[:, 1] <= TOL_RAD)).sum()),
                "bit_equal_transforms": int(sum(np.array_equal(np.asarray(T_list[c], np.float32), T_cpu[c]) for c in range(n_cmp))),
                "translation_m": rms(e[:, 0]), "rotation_rad": rms(e[:, 1]), "max_translation_m": float(e[:, 0].max()),
                "max_rotation_rad": float(e[:, 1].max())}

    def transforms(rec):
        return [rec[c, 4:20].reshape(4, 4) for c in range(n_cmp)]

    def bands(e):
        """the oracle's own band on every pair outside the gate: the largest move of ITS answer under perturbations that carry no information --
        the FMA build, libm's expf, the float32 guess moved by +-1 .. +-16 ulps (34 twins; scripts/dbg_gate_bands.py)"""
        inside = (e[:, 0] <= TOL_M) & (e[:, 1] <= TOL_RAD)
        outside = []
        for c in np.nonzero(~inside)[0][:6]:
            twins = ((True, 0, 0), (False, 1, 0)) + tuple((False, 0, k) for k in range(-16, 17) if k)
            _, bt, br = orc.ndt_band(tgt, sources[c], LoopDetector.guess_for(new_kf, cands[c]), twins=twins, resolution=args.resolution)
            outside.append({"pair": int(c), "translation_m": float(e[c, 0]), "rotation_rad": float(e[c, 1]), "oracle_band_m": float(bt),
                            "oracle_band_rad": float(br), "oracle_band_twins": 34, "inside_oracle_band": bool(e[c, 0] <= bt + TOL_M and e[c, 1] <= br + TOL_RAD)})
        return inside, outside

    timed = NAME[args.order]
    rec_by_mode = {args.order: records}
    par = {timed: parity(transforms(records))}
    if records_other is not None:
        rec_by_mode[1 - args.order] = records_other
        par[NAME[1 - args.order]] = parity(transforms(records_other))
    # the order that also sums in point-index order (bit-identical evaluations): one step, for the record
    for mode, reps in ((2, 1),) + (((1, 3),) if 1 not in rec_by_mode else ()):
        rs = Registration("NDT_OMP", device=device, ndt_strict_order=mode, **reg_kw)
        ds = LoopDetector({"fitness_score_thresh": 1e9}, registration=rs, local_only=True)   # rank 0 alone runs this leg: no collective may be inside
        ds.matching(cands, new_kf)
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(reps):
            ds.matching(cands, new_kf)
        torch.cuda.synchronize()
        t_step = (time.perf_counter() - ts0) / reps
        par[NAME[mode]] = parity(transforms(ds.last_records))
        par[NAME[mode]]["ms_per_step"] = 1e3 * t_step
        par[NAME[mode]]["registrations_per_s"] = len(cands) / t_step
        rec_by_mode[mode] = ds.last_records.copy()
        rs.close()
    par["note"] = ("fast = dgs_params.ndt_strict_order 0 (per-point float math re-associated); upstream_order = 1 (every float operation in upstream's "
                   "order; GPU-ordered double sums); upstream_order_sequential_sum = 2 (index-order sums too: bit-identical evaluations); the timed "
                   "value is " + timed + ".")

    # ---- the gate, on the timed mode
    e = errors(transforms(records))
    inside, outside = bands(e)
    conv_cpu, fit_cpu = cpu["converged"][:n_cmp], cpu["fitness"][:n_cmp]
    b_ref, s_ref = sequential_best(conv_cpu, fit_cpu)
    b_gpu, s_gpu = sequential_best(records[:n_cmp, 1] > 0.5, records[:n_cmp, 2])
    gate = {
        "gate": "final pose within %g m / %g rad of the CPU reference path on every pair (BASELINE.json north_star)" % (TOL_M, TOL_RAD),
        "timed_mode": timed + " (dgs_params.ndt_strict_order = %d)" % args.order,
        "pairs": int(n_cmp), "pairs_inside": int(inside.sum()), "rms_m": rms(e[:, 0]), "rms_rad": rms(e[:, 1]),
        "max_m": float(e[:, 0].max()), "max_rad": float(e[:, 1].max()), "bit_equal_transforms": par[timed]["bit_equal_transforms"],
        "passes_north_star_gate": bool(inside.all()),
        "timed_value": value,
        "pairs_outside": outside,
        "pairs_outside_all_inside_oracle_band": bool(all(o["inside_oracle_band"] for o in outside)),
        "same_best_candidate_as_sequential_reference": bool(b_gpu == b_ref), "best_candidate": int(b_gpu), "reference_best_candidate": int(b_ref),
        "best_fitness_relative_difference": float(abs(s_gpu - s_ref) / s_ref) if b_ref >= 0 and s_ref > 0 else None,
        "oracle": "CPU restatement of ndt_omp with Eigen's two-sided JacobiSVD and SelfAdjointEigenSolver sequences, PCL's double computeHessian, the polar-factor guess and glibc's expf / exp (those two verified against the image's libm) "
                  "(oracle/cpu; parity unpinned: the reference holds no fixtures, DESIGN.md 2)",
    }
    if args.order == 1 and 0 in rec_by_mode:
        # the fast order beside it: what re-associating the float math costs in agreement with the reference
        ef = errors(transforms(rec_by_mode[0]))
        inside_f, outside_f = bands(ef)
        b_f, s_f = sequential_best(rec_by_mode[0][:n_cmp, 1] > 0.5, rec_by_mode[0][:n_cmp, 2])
        gate["fast_order"] = {"pairs_inside": int(inside_f.sum()), "rms_m": rms(ef[:, 0]), "rms_rad": rms(ef[:, 1]), "max_m": float(ef[:, 0].max()),
                              "max_rad": float(ef[:, 1].max()), "passes_north_star_gate": bool(inside_f.all()), "pairs_outside": outside_f,
                              "pairs_outside_all_inside_oracle_band": bool(all(o["inside_oracle_band"] for o in outside_f)),
                              "same_best_candidate_as_sequential_reference": bool(b_f == b_ref),
                              "note": "secondary mode (fast_value): it ignores dgs_params.ndt_hessian_recompute_double (its closing computeHessian is its own "
                                      "float pass) and takes a Gauss-Jordan Newton step; NOT the mode `value` is measured in"}
    if args.order == 0:
        up = par["upstream_order"]
        b_str, _ = sequential_best(rec_by_mode[1][:n_cmp, 1] > 0.5, rec_by_mode[1][:n_cmp, 2])
        gate.update({"gate_passing_mode": "upstream_order (dgs_params.ndt_strict_order = 1)",
                     "gate_passing_pairs_inside": up["pairs_within_1e-4m_1e-5rad"], "gate_passing_bit_equal_transforms": up["bit_equal_transforms"],
                     "gate_passing_mode_passes": bool(up["pairs_within_1e-4m_1e-5rad"] == n_cmp), "gate_passing_same_best_candidate": bool(b_str == b_ref)})
    return gate, par


PROFILER_ENV_PREFIXES = ("ROCP_", "ROCPROF", "ROCPROFILER_", "ROCTRACER_", "HSA_TOOLS_LIB", "ROCTX_")


def under_profiler() -> bool:
    if any(k.startswith(PROFILER_ENV_PREFIXES) for k in os.environ):
        return True
    pre = os.environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or "roctracer" in pre


def spawn_ranks(n: int) -> int:
    """`bench.py --gpus N` without a launcher: run N ranks as a child `python -m torch.distributed.run` (one process per GPU,
    rendezvous on 127.0.0.1) and return its exit status.  Nothing in this process has touched the GPU yet."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    return subprocess.call(cmd, env=env)


def dry_run(args, rank, world, backend):
    """Launcher / collective plumbing without HIP work: init_process_group, barrier-bracketed timing, all_gather of the result
    records, MAX over ranks, one JSON line from rank 0.  `value` is null: nothing was registered.  With --group nothing is spawned:
    the line reports the single-process shape (members, dealing) that the GPU run would use."""
    import torch
    import torch.distributed as dist
    P = args.pairs
    if args.group:
        G = args.gpus
        shares = [len(range(k, G * P, G)) for k in range(G)]
        emit_line(json.dumps({"metric": METRIC, "value": None, "unit": "registrations/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
                          "dry_run": True,
                          "config": {"workload": "dry run: single-process group shape only", "pairs_per_gpu": P, "group": True, "members": G,
                                     "candidates_per_step": G * P, "shares": shares, "processes": 1,
                                     "collective_backend": "rccl (ncclCommInitAll)", "collective_world_size": G}}))
        return
    if backend == "nccl" and not torch.cuda.is_available():
        backend = "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=120))
        assert dist.get_world_size() == args.gpus
    rec = torch.full((P, 20), float(rank), dtype=torch.float64)
    for _ in range(args.warmup):
        if world > 1:
            dist.barrier()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    seen = None
    for _ in range(args.steps):
        if world > 1:
            out = torch.empty((world * P, 20), dtype=torch.float64)
            dist.all_gather_into_tensor(out, rec)
            seen = sorted(set(int(v) for v in out[:, 0].tolist()))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        emit_line(json.dumps({"metric": METRIC, "value": None, "unit": "registrations/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic", "dry_run": True,
                          "config": {"workload": "dry run: launcher + collective plumbing only", "pairs_per_gpu": P,
                                     "collective_backend": (dist.get_backend() if world > 1 else None), "collective_world_size": world,
                                     "ranks_seen_in_all_gather": seen}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measure_traffic(args, bytes_per_eval, kernel_substring="ndt_derivatives_kernel"):
    """HBM bytes per launch of ndt_derivatives from the PMC counters, as MI355X_MICROARCH.md (HBM / rocprofv3) prescribes:
    FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), values in KiB, and on gfx950 FETCH_SIZE counts a wide coalesced
    read at half its bytes, so the read side is doubled (an upper bound here: the kernel's gathers are not wide streams)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    if shutil.which("rocprofv3") is None:
        return None, {"error": "rocprofv3 not on PATH"}
    if under_profiler():   # never nest a profiler inside a profiled process (the launcher would exec with the tool preloaded)
        return None, {"error": "already under a profiler"}
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(args.traffic_dir, counter)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d, exist_ok=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-traffic", "--pairs", str(args.pairs),
               "--points", str(args.points), "--distinct-scans", str(args.distinct_scans), "--order", str(args.order)]
        env = {k: v for k, v in os.environ.items() if not k.startswith(PROFILER_ENV_PREFIXES)}
        if "rocprof" in env.get("LD_PRELOAD", "") or "roctracer" in env.get("LD_PRELOAD", ""):
            env.pop("LD_PRELOAD")
        env["TMPDIR"] = "/tmp"
        child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        try:
            child.wait(timeout=240)
        except subprocess.TimeoutExpired:  # end exactly the process group started here
            os.killpg(child.pid, signal.SIGKILL)
            child.wait()
            return None, {"error": "rocprofv3 --pmc %s child timed out" % counter}
        tot, n = 0.0, 0
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = {}
            for r in csv.DictReader(open(f)):
                if kernel_substring in r["Kernel_Name"] and r["Counter_Name"] == counter:
                    per_dispatch[r["Dispatch_Id"]] = per_dispatch.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            tot += sum(per_dispatch.values())
            n += len(per_dispatch)
        res[counter] = (tot * 1024.0 / n) if n else None
        res[counter + "_launches"] = n
    if res["FETCH_SIZE"] is None or res["WRITE_SIZE"] is None:
        return None, res
    traffic = 2.0 * res["FETCH_SIZE"] + res["WRITE_SIZE"]
    res["note"] = "bytes per launch averaged over all launches of the profiled child run (incl. launches whose pairs had finished)"
    return traffic, res


class _Sparse(list):
    """A candidate list in which only this rank's stride is populated (len() is the global candidate count)."""

    def __init__(self, items, filler):
        super().__init__(items)
        self._filler = filler

    def __getitem__(self, i):
        v = super().__getitem__(i)
        return self._filler if v is None else v


def _GroupDetector(params, registration, cache_clouds):
    """LoopDetector whose keyframe cache keys on KeyFrame.cache_id (scans shared by several synthetic candidates are resident once)."""
    from delta_graph_slam_amd.loop_detector import LoopDetector

    class D(LoopDetector):
        def resident(self, keyframe, as_target=False):
            if as_target:
                # The new keyframe arrives from the host and is registered against ONCE (one tick): it is uploaded inside the step and its
                # voxel model / NN index are built inside the step, as in the one-process-per-GPU form -- nothing derived from the target
                # is carried over from the previous step.  Only the candidate keyframes (immutable, re-used tick after tick) are resident.
                return keyframe.cloud
            cid = getattr(keyframe, "cache_id", keyframe.id)
            c = self._cloud_cache.get(cid)
            if c is None:
                c = self.registration.make_cloud(keyframe.cloud, owner=cid - 1)
                self._cloud_cache[cid] = c
            return c

    return D(params, registration=registration, cache_clouds=cache_clouds)


if __name__ == "__main__":
    main()
