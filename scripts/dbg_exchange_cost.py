"""Cost of the loop detector's exchange step (LoopDetector._exchange) on a one-rank RCCL group: pageable copies either side of the
all_gather (rounds 1-2) against pinned staging buffers and one synchronisation (round 3).  The collective itself grows with the ranks;
the copies and synchronisations around it are what a rank pays at any size.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 scripts/dbg_exchange_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch, torch.distributed as dist
from delta_graph_slam_amd.loop_detector import LoopDetector, RECORD_WIDTH
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
dist.init_process_group("nccl")
world = dist.get_world_size()
dev = torch.device("cuda", torch.cuda.current_device())
rec = np.random.default_rng(0).normal(size=(32, RECORD_WIDTH))


def old(rec):
    local = torch.from_numpy(rec).to(dev)
    gathered = torch.empty((world * 32, RECORD_WIDTH), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(gathered, local)
    return gathered.cpu().numpy()


class _Reg:   # the detector only needs an object to hold
    pass


det = LoopDetector({}, registration=_Reg())
for name, fn in (("pageable copies (rounds 1-2)", old), ("pinned staging, one synchronisation", lambda r: det._exchange(r, 32, world))):
    for _ in range(50):
        out = fn(rec)
    assert np.array_equal(out[:32], rec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000):
        fn(rec)
    print("%-40s %.1f us per exchange" % (name, (time.perf_counter() - t0) * 1e3))
dist.destroy_process_group()
