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
