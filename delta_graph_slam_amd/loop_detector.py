"""Loop-closure candidate matching: the reference's LoopDetector over the HIP registration, sharded across GPUs.

Mirrors /root/reference/include/hdl_graph_slam/loop_detector.hpp:
  LoopDetector(pnh)          :41-51   parameter names and defaults
  detect                     :59-70
  find_candidates            :83-111
  matching                   :119-173  setInputTarget once, then per candidate setInputSource / align(guess) /
                                       getFitnessScore(fitness_score_max_range); keep arg-min over converged candidates
                                       (a candidate is skipped iff !converged or score > best, so on an exact tie the
                                       LATER candidate wins, :149); accept iff best <= fitness_score_thresh (:162).

MI355X mapping: the candidate loop has no cross-iteration dependency, so candidates are dealt
round-robin to the ranks of a torch.distributed group (one process per GPU; backend "nccl" is RCCL over xGMI), each
rank registers its shard as ONE batched launch sequence (Registration.align_batch), and the only exchange step is an
all_gather of fixed-size result records; the arg-min then runs in ORIGINAL candidate order on every rank, which
reproduces the reference's sequential tie-breaking exactly.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, List, Optional, Sequence

import time

from collections import OrderedDict

import numpy as np

from .transforms import transform2Dto3D_batch, transform2Dto3D, transform3Dto2D

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None

__all__ = ["KeyFrame", "Loop", "LoopDetector", "RECORD_WIDTH"]

DBL_MAX = 1.7976931348623157e308
RECORD_WIDTH = 20  # candidate index, converged, fitness, status, 16 x transform (row-major)


@dataclass
class KeyFrame:
    """The fields of hdl_graph_slam::KeyFrame the loop detector reads (keyframe.hpp:25-59)."""
    cloud: Any                      # float32 [N,4] (numpy or HBM-resident torch tensor)
    estimate: np.ndarray            # node->estimate(): 3x3 SE2 matrix (Eigen::Isometry2d)
    accum_distance: float = 0.0
    id: int = -1


@dataclass
class Loop:
    """loop_detector.hpp:16-28"""
    key1: KeyFrame
    key2: KeyFrame
    relative_pose: np.ndarray       # 4x4 float32
    relative_pose2D: np.ndarray     # 3x3 float32
    score: float = field(default=0.0)


class LoopDetector:
    def __init__(self, params: Optional[dict] = None, registration=None, group=None, device: Optional[int] = None,
                 cache_clouds: bool = False, filter_on_device: bool = False, cache_capacity: Optional[int] = None, local_only: bool = False):
        pr = dict(params or {})
        self.distance_thresh = float(pr.get("distance_thresh", 5.0))
        self.accum_distance_thresh = float(pr.get("accum_distance_thresh", 8.0))
        self.distance_from_last_edge_thresh = float(pr.get("min_edge_interval", 5.0))
        self.fitness_score_max_range = float(pr.get("fitness_score_max_range", DBL_MAX))
        self.fitness_score_thresh = float(pr.get("fitness_score_thresh", 0.5))
        if registration is None:
            from .registration import select_registration_method
            registration = select_registration_method(pr, device=device)
        self.registration = registration
        self.last_edge_accum_distance = 0.0
        self.group = group
        # True: this detector registers every candidate it is given on its own device and never enters a collective, whatever process group
        # exists (side measurements of one rank of a multi-rank job: bench.py's other-order / parity legs)
        self.local_only = bool(local_only)
        self.last_records = None
        # keyframe id -> DeviceCloud: a keyframe that is a candidate tick after tick (delta_graph_slam_nodelet.cpp:816 calls
        # detect() every graph_update_interval) is uploaded and indexed once; needs KeyFrame.id to be unique
        self.cache_clouds = bool(cache_clouds)
        # find_candidates through dgs_find_loop_candidates (second half).  OFF by default: the call costs ~40 us whatever the
        # graph holds (two small uploads, one workgroup, one synchronisation; scripts/r4_find_candidates_time.py on an MI355X: 39 us at 100-1,000
        # keyframes, 74 us at 10,000, 0.33 ms at 100,000) -- the same two tests vectorised on the host take 5 / 17 / 129 us / 1.3 ms, and the
        # reference's C++ loop a few ns per keyframe: it pays beyond ~5,000 keyframes here, or when the keyframe table already lives in HBM
        self.filter_on_device = bool(filter_on_device)
        self._cloud_cache = OrderedDict()         # least recently used first
        self.cache_capacity = None if cache_capacity is None else max(1, int(cache_capacity))   # keyframes kept resident at most (None: all)
        self._exchange_buffers = {}
        self.exchange_seconds = 0.0   # host time spent in the exchange step (all-gather of the result records) since construction
        self.exchange_calls = 0
        self.force_exchange = False   # measurement: run the exchange step even when the group has one rank (its floor: two copies + one collective)

    def resident(self, keyframe: "KeyFrame", as_target: bool = False):
        """KeyFrame::cloud as the registration should see it: the cached HBM-resident object when caching is on."""
        if not self.cache_clouds or keyframe.id < 0 or not hasattr(self.registration, "make_cloud"):
            return keyframe.cloud
        c = self._cloud_cache.get(keyframe.id)
        if c is not None:
            self._cloud_cache.move_to_end(keyframe.id)
        if c is None:
            if as_target or not hasattr(self.registration, "devices"):
                c = self.registration.make_cloud(keyframe.cloud)
            else:   # a group (several devices, one process): a candidate keyframe lives on ONE member, owner = its id (even shares)
                c = self.registration.make_cloud(keyframe.cloud, owner=keyframe.id)
            self._cloud_cache[keyframe.id] = c
        return c

    def _after_tick(self, new_keyframe: "KeyFrame", used_ids):
        """Bounds what a tick leaves in HBM.  On a group the new keyframe was every member's target (one copy per member, cloned device to
        device by dgs_group_set_input_target_cloud): as a candidate of later ticks it needs the copy on its owner only, so the others go
        now.  And with a capacity set, the least recently used keyframes beyond it are released -- never one this tick used."""
        c = self._cloud_cache.get(new_keyframe.id) if new_keyframe.id >= 0 else None
        if c is not None and hasattr(c, "trim"):
            c.trim(owner=new_keyframe.id)
        if self.cache_capacity is not None and len(self._cloud_cache) > self.cache_capacity:
            keep = set(used_ids)
            for kid in [k for k in self._cloud_cache if k not in keep]:
                if len(self._cloud_cache) <= self.cache_capacity:
                    break
                self.evict(kid)

    def evict(self, keyframe_id: int):
        c = self._cloud_cache.pop(keyframe_id, None)
        if c is not None:
            c.close()

    # ---------------------------------------------------------------------------------------------- reference logic
    def detect(self, keyframes: Sequence[KeyFrame], new_keyframes: Sequence[KeyFrame]) -> List[Loop]:
        loops = []
        for nk in new_keyframes:
            cands = self.find_candidates(keyframes, nk)
            loop = self.matching(cands, nk)
            if loop is not None:
                loops.append(loop)
        return loops

    def find_candidates(self, keyframes: Sequence[KeyFrame], new_keyframe: KeyFrame) -> List[KeyFrame]:
        if new_keyframe.accum_distance - self.last_edge_accum_distance < self.distance_from_last_edge_thresh:
            return []
        if self.filter_on_device and len(keyframes) and hasattr(self.registration, "find_loop_candidates"):
            # the same two tests over all keyframes in one device call (dgs_find_loop_candidates); keyframe order is kept
            acc = np.array([k.accum_distance for k in keyframes], np.float64)
            xy = np.array([np.asarray(k.estimate, np.float64)[:2, 2] for k in keyframes], np.float64)
            idx = self.registration.find_loop_candidates(acc, xy, new_keyframe.accum_distance, np.asarray(new_keyframe.estimate, np.float64)[:2, 2],
                                                         self.accum_distance_thresh, self.distance_thresh)
            return [keyframes[i] for i in idx]
        out = []
        p2 = np.asarray(new_keyframe.estimate, np.float64)[:2, 2]
        for k in keyframes:
            if new_keyframe.accum_distance - k.accum_distance < self.accum_distance_thresh:
                continue
            p1 = np.asarray(k.estimate, np.float64)[:2, 2]
            if np.linalg.norm(p1 - p2) > self.distance_thresh:
                continue
            out.append(k)
        return out

    @staticmethod
    def guess_for(new_keyframe: KeyFrame, candidate: KeyFrame) -> np.ndarray:
        """loop_detector.hpp:139-143"""
        rel = np.linalg.inv(np.asarray(new_keyframe.estimate, np.float64)) @ np.asarray(candidate.estimate, np.float64)
        return transform2Dto3D(rel.astype(np.float32))

    @staticmethod
    def guesses_for(new_keyframe: KeyFrame, candidates: Sequence[KeyFrame]) -> np.ndarray:
        """guess_for over many candidates in one numpy pass -> [n,4,4] float32 (bit-identical to the per-candidate form)."""
        if len(candidates) == 0:
            return np.zeros((0, 4, 4), np.float32)
        inv_new = np.linalg.inv(np.asarray(new_keyframe.estimate, np.float64))
        est = np.array([k.estimate for k in candidates], dtype=np.float64)      # [n,3,3] in one conversion
        return transform2Dto3D_batch((inv_new @ est).astype(np.float32))

    # ---------------------------------------------------------------------------------------------- sharding
    def _world(self):
        if not self.local_only and dist is not None and dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def register_shard(self, candidates: Sequence[KeyFrame], new_keyframe: KeyFrame) -> np.ndarray:
        """Registers this rank's share (candidates[rank::world]) against the new keyframe and returns the
        gathered [n_candidates, RECORD_WIDTH] float64 records in original candidate order."""
        rank, world = self._world()
        n = len(candidates)
        # the host-side preparation of the candidate loop first: setInputTarget leaves the device building the target's voxel model, and
        # nothing should stand between that and the batch's first launch
        mine = list(range(rank, n, world))
        sources = [self.resident(candidates[c]) for c in mine]
        guesses = self.guesses_for(new_keyframe, [candidates[c] for c in mine])
        self.registration.setInputTarget(self.resident(new_keyframe, as_target=True))
        per_rank = (n + world - 1) // world
        rec = np.full((per_rank, RECORD_WIDTH), -1.0, dtype=np.float64)
        if mine:
            if hasattr(self.registration, "align_batch_records"):
                # product path: the C ABI's result array goes straight into the record rows
                rec[:len(mine)] = self.registration.align_batch_records(sources, guesses, compute_fitness=True,
                                                                        fitness_max_range=self.fitness_score_max_range)
                rec[:len(mine), 0] = mine
            else:
                results = self.registration.align_batch(sources, list(guesses), compute_fitness=True, fitness_max_range=self.fitness_score_max_range)
                for j, (c, r) in enumerate(zip(mine, results)):
                    rec[j, 0] = c
                    rec[j, 1] = 1.0 if r["converged"] else 0.0
                    rec[j, 2] = r["fitness"]
                    rec[j, 3] = r.get("status", 0)
                    rec[j, 4:20] = np.asarray(r["T"], np.float64).reshape(16)
        if self.cache_clouds:
            self._after_tick(new_keyframe, [new_keyframe.id] + [candidates[c].id for c in mine])
        if world == 1 and not self.force_exchange:
            allrec = rec
        else:
            t_ex = time.perf_counter()
            allrec = self._exchange(rec, per_rank, world)
            self.exchange_seconds += time.perf_counter() - t_ex   # the path's one exchange step, as the host sees it (bench.py reports it per rank)
            self.exchange_calls += 1
        out = np.full((n, RECORD_WIDTH), -1.0, dtype=np.float64)
        idx = allrec[:, 0].astype(np.int64)
        ok = (idx >= 0) & (idx < n)          # padding rows of the gather carry -1
        out[idx[ok]] = allrec[ok]
        self.last_records = out
        return out

    def _exchange(self, rec: np.ndarray, per_rank: int, world: int) -> np.ndarray:
        """The path's one exchange step: every rank's [per_rank, RECORD_WIDTH] records to every rank (ncclAllGather over RCCL, or gloo).
        On the device path the staging buffers are made once per shape -- pinned host memory both ways, so that the two copies are
        asynchronous and the step pays ONE synchronisation after the gather instead of a pageable copy either side of it."""
        backend = dist.get_backend(self.group)
        if backend != "nccl":
            local = torch.from_numpy(rec)
            gathered = torch.empty((world * per_rank, RECORD_WIDTH), dtype=torch.float64)
            dist.all_gather_into_tensor(gathered, local, group=self.group)
            return gathered.numpy()
        dev = torch.device("cuda", torch.cuda.current_device())
        key = (per_rank, world, dev.index)
        buf = self._exchange_buffers.get(key)
        if buf is None:
            buf = (torch.empty((per_rank, RECORD_WIDTH), dtype=torch.float64).pin_memory(),
                   torch.empty((per_rank, RECORD_WIDTH), dtype=torch.float64, device=dev),
                   torch.empty((world * per_rank, RECORD_WIDTH), dtype=torch.float64, device=dev),
                   torch.empty((world * per_rank, RECORD_WIDTH), dtype=torch.float64).pin_memory())
            self._exchange_buffers[key] = buf
        host_in, dev_in, dev_out, host_out = buf
        host_in.numpy()[...] = rec
        dev_in.copy_(host_in, non_blocking=True)
        dist.all_gather_into_tensor(dev_out, dev_in, group=self.group)
        host_out.copy_(dev_out, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        return host_out.numpy().copy()

    @staticmethod
    def select_best(records: np.ndarray):
        """The arg-min of loop_detector.hpp:126-156 over gathered records, in candidate order."""
        best_score = DBL_MAX
        best = -1
        for c in range(records.shape[0]):
            converged = records[c, 1] > 0.5
            score = records[c, 2]
            if (not converged) or score > best_score:
                continue
            best_score = score
            best = c
        return best, best_score

    def matching(self, candidates: Sequence[KeyFrame], new_keyframe: KeyFrame) -> Optional[Loop]:
        if len(candidates) == 0:
            return None
        records = self.register_shard(candidates, new_keyframe)
        best, best_score = self.select_best(records)
        if best < 0 or best_score > self.fitness_score_thresh:
            return None   # "loop not found..."
        rel = records[best, 4:20].reshape(4, 4).astype(np.float32)
        self.last_edge_accum_distance = new_keyframe.accum_distance
        return Loop(new_keyframe, candidates[best], rel, transform3Dto2D(rel), best_score)
