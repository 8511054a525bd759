"""world_size-2 gloo test of the loop-closure sharding: candidates dealt round-robin, one all_gather of result records,
arg-min in original candidate order on every rank (loop_detector.hpp:137-162).  The registration engine is the CPU
oracle (tests/oracle_engine.py) because this container has no GPU; the sharding / gather / selection code under test is
the product's delta_graph_slam_amd/loop_detector.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from delta_graph_slam_amd import synth
from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene():
    tgt, src, Tgt = synth.planar_pair(n=2048, seed_target=7, seed_source=8)
    rng = np.random.default_rng(11)
    cands = []
    for c in range(5):
        n = int(rng.integers(1200, 2048))
        est = np.eye(3)
        yaw = 0.05 + rng.uniform(-0.02, 0.02)
        est[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
        est[:2, 2] = [0.3 + rng.uniform(-0.1, 0.1), -0.1 + rng.uniform(-0.1, 0.1)]
        cands.append(KeyFrame(cloud=src[:n].copy(), estimate=est, accum_distance=0.0, id=c))
    cands.append(KeyFrame(cloud=np.zeros((0, 4), np.float32), estimate=np.eye(3), accum_distance=0.0, id=5))   # ragged: empty
    new = KeyFrame(cloud=tgt, estimate=np.eye(3), accum_distance=100.0, id=99)
    return cands, new


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from tests.oracle_engine import OracleEngine
    cands, new = _scene()
    det = LoopDetector({"fitness_score_thresh": 10.0}, registration=OracleEngine("NDT_OMP", resolution=2.0, num_threads=2))
    loop = det.matching(cands, new)
    np.save(os.path.join(out_dir, f"rec{rank}.npy"), det.last_records)
    np.save(os.path.join(out_dir, f"best{rank}.npy"), np.array([-1 if loop is None else loop.key2.id]))
    if rank == 0:
        # a side measurement ONE rank makes while a process group exists (bench.py's parity legs): local_only keeps it out of every collective --
        # without it this call waits for rank 1's all_gather for ever (found by a two-rank rehearsal of bench.py)
        side = LoopDetector({"fitness_score_thresh": 10.0}, registration=OracleEngine("NDT_OMP", resolution=2.0, num_threads=2), local_only=True)
        side.matching(cands, new)
        np.save(os.path.join(out_dir, "side0.npy"), side.last_records)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_matching_equals_sequential(tmp_path):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ["PYTHONPATH"] = root + os.pathsep + os.environ.get("PYTHONPATH", "")
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    recs = [np.load(tmp_path / f"rec{r}.npy") for r in range(world)]
    bests = [int(np.load(tmp_path / f"best{r}.npy")[0]) for r in range(world)]
    assert np.array_equal(recs[0], recs[1], equal_nan=True) and bests[0] == bests[1]             # every rank reaches the same decision
    # single-process reference: the same detector without a process group
    from tests.oracle_engine import OracleEngine
    cands, new = _scene()
    det = LoopDetector({"fitness_score_thresh": 10.0}, registration=OracleEngine("NDT_OMP", resolution=2.0, num_threads=2))
    loop = det.matching(cands, new)
    assert np.array_equal(det.last_records, recs[0], equal_nan=True)
    assert (-1 if loop is None else loop.key2.id) == bests[0]
    assert recs[0][5, 3] == 4 and recs[0][5, 1] == 0                             # the empty candidate reported DGS_ERR_NO_SOURCE
    assert list(recs[0][:, 0]) == [0, 1, 2, 3, 4, 5]
    assert np.array_equal(np.load(tmp_path / "side0.npy"), det.last_records, equal_nan=True)   # rank 0's local_only detector registered all six itself
