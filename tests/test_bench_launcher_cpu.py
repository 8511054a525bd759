"""`python bench.py --gpus N` must produce an N-rank line by itself (the driver calls it that way): the launcher half is
exercised here on the CPU box with --dry-run (no HIP work) over gloo; the GPU half is tests/test_bench_contract_gpu.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout          # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_reports_them():
    out = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"], {"DGS_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["config"]["collective_world_size"] == 2
    assert out["config"]["ranks_seen_in_all_gather"] == [0, 1]
    assert out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak" and out["dry_run"] is True


def test_gpus_1_stays_in_process():
    out = _run(["--gpus", "1", "--dry-run", "--steps", "1", "--warmup", "0"])
    assert out["n_gpus"] == 1 and out["config"]["collective_world_size"] == 1


def test_world_size_must_match_gpus_flag():
    env = {k: v for k, v in os.environ.items()}
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_traffic_leg_refuses_to_nest_profilers(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("ROCPROFILER_LIBRARY_CTOR", "1")
    assert bench.under_profiler()
    monkeypatch.delenv("ROCPROFILER_LIBRARY_CTOR")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.under_profiler()
    monkeypatch.delenv("LD_PRELOAD")
    for k in list(os.environ):
        if k.startswith(bench.PROFILER_ENV_PREFIXES):
            monkeypatch.delenv(k)
    assert not bench.under_profiler()


def test_group_mode_stays_in_one_process_and_reports_the_group_shape():
    """`bench.py --gpus N --group`: ONE process, N devices behind dgs_group -- nothing is spawned, the dealing is c -> c mod N."""
    out = _run(["--gpus", "4", "--group", "--dry-run", "--steps", "3", "--warmup", "1", "--pairs", "5"])
    cfg = out["config"]
    assert out["n_gpus"] == 4 and out["dry_run"] is True and out["value"] is None and out["scaling"] == "weak"
    assert cfg["group"] is True and cfg["processes"] == 1 and cfg["members"] == 4 and cfg["candidates_per_step"] == 20 and cfg["shares"] == [5, 5, 5, 5]
    assert cfg["collective_backend"].startswith("rccl")


def test_cpu_baseline_child_times_the_oracle_with_pinned_threads(tmp_path):
    """The cpu_baseline leg runs in a child process (OMP_PROC_BIND / OMP_PLACES must not reach the process that drives the GPU):
    thread sweep, three repeats, per-pair and per-evaluation times, the fitness loop on the same pairs, the poses for the parity gate."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from delta_graph_slam_amd import synth
    tgt, sources, guesses, _ = synth.loop_batch(n_candidates=3, n_points=4096, seed=5, distinct_scans=2)
    args = bench.parse_args(["--cpu-seconds", "2", "--pairs", "3"])
    res = bench.run_cpu_baseline(args, tgt, sources, guesses)
    assert "error" not in res, res
    cb = res["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "registrations/s" and cb["value"] > 0 and cb["cores"] >= 1
    lo, hi = cb["value_spread_3_repeats"]
    assert lo <= cb["value"] <= hi and cb["value_with_fitness_score"] < cb["value"]      # the fitness loop only adds work, on the same pairs
    assert cb["ms_per_pair"]["max"] >= cb["ms_per_pair"]["median"] > 0 and cb["ms_per_evaluation"] > 0
    assert cb["thread_binding"] == "OMP_PROC_BIND=close OMP_PLACES=cores" and "OMP_PROC_BIND" not in os.environ
    assert res["T"].shape == (res["pairs"], 4, 4) and len(res["fitness"]) == res["pairs"] == len(res["converged"])
    from oracle import oracle as orc
    o = orc.NdtOracle(resolution=1.0)
    o.set_target(tgt)
    o.set_source(sources[0])
    assert np.array_equal(o.align(guesses[0])["T"], res["T"][0])                          # same poses as an in-process run
