"""-m gpu: dgs_group (single process, several devices behind the C ABI; include/dgs_reg.h).  On the one-GPU box: a group over
{0} runs the RCCL path (ncclCommInitAll + ncclAllGather of the records, one rank) and must equal dgs_align_batch bit for bit; a
group over {0, 0} (the one-GPU rehearsal, host gather) exercises the dealing c -> c mod G, the re-ordering of the records and the
arg-min with the reference's tie rule."""
import numpy as np
import pytest

from delta_graph_slam_amd import synth

# test_cfg4_at_its_full_candidate_count...: measured on an MI355X with the round-4 upstream-order kernel (item-compacted, one launch per round)
CFG4_OTHER_EVALUATION_COUNT = []   # candidates of the 256 whose evaluation count (or transform) differs from the oracle's: none since repeated trial points take their earlier value (NdtSolver::trial_x; before that [2, 24, 40, 225, 237])
CFG4_ONE_ULP_TRANSFORMS = []              # ... of which: one float of the transform one ulp off
CFG4_ONE_ULP_TRANSFORMS_HOST_DEALING = [] # the same batch dealt candidate c -> member c mod 8 (another partition of the sums)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def batch():
    return synth.loop_batch(n_candidates=7, n_points=16384, seed=91, distinct_scans=4)


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert np.array_equal(x["T"], y["T"]) and x["converged"] == y["converged"] and x["iterations"] == y["iterations"]
        assert x["fitness"] == y["fitness"] and x["status"] == y["status"] == 0


@pytest.mark.parametrize("method,kw", [("NDT_OMP", dict(ndt_resolution=1.0)), ("NDT_OMP", dict(ndt_resolution=1.0, ndt_strict_order=0)),
                                       ("FAST_GICP", dict(gicp_max_correspondence_distance=2.0))])
def test_group_of_one_device_equals_align_batch_and_gathers_with_rccl(batch, method, kw):
    from delta_graph_slam_amd.registration import Registration, RegistrationGroup
    tgt, sources, guesses, _ = batch
    r = Registration(method, **kw)
    r.setInputTarget(tgt)
    ref = r.align_batch(sources, guesses)
    g = RegistrationGroup(method, devices=[0], **kw)
    assert g.uses_rccl                                   # communicators exist: the image ships RCCL
    g.setInputTarget(tgt)
    got = g.align_batch(sources, guesses)
    assert g.last_gather_used_rccl                       # the records came back through ncclAllGather
    _same(got, ref)
    fit = [x["fitness"] if x["converged"] else np.inf for x in ref]
    assert g.best_index == int(len(fit) - 1 - np.argmin(fit[::-1])) and g.best_score == min(fit)


def test_group_deals_round_robin_and_keeps_candidate_order(batch):
    from delta_graph_slam_amd.registration import Registration, RegistrationGroup
    tgt, sources, guesses, _ = batch
    r = Registration("NDT_OMP", ndt_resolution=1.0)
    r.setInputTarget(tgt)
    ref = r.align_batch(sources, guesses)
    for devs in ([0, 0], [0, 0, 0]):                     # 7 candidates over 2 / 3 members: ragged shares
        g = RegistrationGroup("NDT_OMP", devices=devs, ndt_resolution=1.0)
        assert not g.uses_rccl                           # a device listed twice: host gather
        g.setInputTarget(tgt)
        _same(g.align_batch(sources, guesses), ref)
        # the reference's tie rule (loop_detector.hpp:149): on equal scores the LATER candidate wins -- in candidate order
        twice = list(sources) + [sources[2]]
        gg = np.concatenate([guesses, guesses[2:3]])
        res = g.align_batch(twice, gg)
        assert np.array_equal(res[7]["T"], res[2]["T"]) and res[7]["fitness"] == res[2]["fitness"]
        fit = [x["fitness"] if x["converged"] else np.inf for x in res]
        if np.argmin(fit) == 2:
            assert g.best_index == 7
        g.close()


def test_loop_detector_over_a_group_matches_the_single_handle_detector(batch):
    from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
    from delta_graph_slam_amd.registration import Registration, RegistrationGroup
    tgt, sources, guesses, _ = batch
    new = KeyFrame(tgt, np.eye(3), 100.0, 0)
    cands = []
    for c, G in enumerate(guesses):
        est = np.eye(3)
        est[:2, :2] = G[:2, :2]
        est[:2, 2] = G[:2, 3]
        cands.append(KeyFrame(sources[c], est, 0.0, c + 1))
    d1 = LoopDetector({"fitness_score_thresh": 1e9}, registration=Registration("NDT_OMP", ndt_resolution=1.0))
    d2 = LoopDetector({"fitness_score_thresh": 1e9}, registration=RegistrationGroup("NDT_OMP", devices=[0, 0], ndt_resolution=1.0))
    l1, l2 = d1.matching(cands, new), d2.matching(cands, new)
    assert np.array_equal(d1.last_records[:, 1:], d2.last_records[:, 1:])
    assert (l1 is None) == (l2 is None) and (l1 is None or (l1.key2.id == l2.key2.id and np.array_equal(l1.relative_pose, l2.relative_pose)))


def test_group_failures_stay_per_candidate(batch):
    from delta_graph_slam_amd.registration import RegistrationGroup
    tgt, sources, guesses, _ = batch
    g = RegistrationGroup("NDT_OMP", devices=[0, 0], ndt_resolution=1.0)
    g.setInputTarget(tgt)
    srcs = [sources[0], np.zeros((0, 4), np.float32), sources[1]]     # an empty candidate: not converged, transform = guess
    res = g.align_batch(srcs, guesses[:3])
    assert res[0]["converged"] and res[2]["converged"] and not res[1]["converged"] and res[1]["status"] == 4
    assert np.array_equal(res[1]["T"], guesses[1])


def test_group_of_eight_members_with_more_and_fewer_candidates_than_members():
    """The 8-GPU node's shape rehearsed on one card: 8 members (device 0 listed 8 times: 8 handles, 8 host threads, 8 streams, host
    gather), 21 candidates (ragged shares 3,3,3,3,3,2,2,2) and 5 candidates (three members idle): records in candidate order,
    equal to the single handle's."""
    from delta_graph_slam_amd.registration import Registration, RegistrationGroup
    tgt, sources, guesses, _ = synth.loop_batch(n_candidates=21, n_points=8192, seed=17, distinct_scans=5)
    r = Registration("NDT_OMP", ndt_resolution=1.0)
    r.setInputTarget(tgt)
    ref = r.align_batch(sources, guesses)
    g = RegistrationGroup("NDT_OMP", devices=[0] * 8, ndt_resolution=1.0)
    assert len(g.devices) == 8
    g.setInputTarget(tgt)
    _same(g.align_batch(sources, guesses), ref)
    _same(g.align_batch(sources[:5], guesses[:5]), ref[:5])
    fit = [x["fitness"] if x["converged"] else np.inf for x in ref[:5]]
    assert g.best_index == int(4 - np.argmin(fit[::-1])) and g.best_score == min(fit)


# ---- keyframe clouds resident on the group's devices (dgs_group_cloud_*, include/dgs_reg.h) ------------------------------------
@pytest.mark.parametrize("method,kw", [("NDT_OMP", dict(ndt_resolution=1.0)), ("NDT_OMP", dict(ndt_resolution=1.0, ndt_strict_order=0)),
                                       ("FAST_GICP", dict(gicp_max_correspondence_distance=2.0))])
def test_group_of_one_device_with_resident_clouds_equals_align_batch_clouds(batch, method, kw):
    """group{0}: resident keyframes, records written on the device and exchanged with ncclAllGather == dgs_align_batch_clouds"""
    from delta_graph_slam_amd.registration import Registration, RegistrationGroup
    tgt, sources, guesses, _ = batch
    r = Registration(method, **kw)
    r.setInputTarget(r.make_cloud(tgt))
    ref = r.align_batch([r.make_cloud(s) for s in sources], guesses)
    g = RegistrationGroup(method, devices=[0], **kw)
    gt = g.make_cloud(tgt)
    gs = [g.make_cloud(s, owner=i) for i, s in enumerate(sources)]
    assert gt.copies == 1 and all(c.copies == 1 for c in gs) and len(gs[0]) == sources[0].shape[0]
    g.setInputTarget(gt)
    for _ in range(2):                                   # a second tick over the same keyframes: nothing is uploaded again
        got = g.align_batch(gs, guesses)
        assert g.last_gather_used_rccl
        _same(got, ref)
    fit = [x["fitness"] if x["converged"] else np.inf for x in ref]
    assert g.best_index == int(len(fit) - 1 - np.argmin(fit[::-1])) and g.best_score == min(fit)


def test_group_resident_clouds_owners_replicas_and_promotion(batch):
    """3 members: candidate keyframes live on their owner (id mod 3), the new keyframe on every member; a keyframe created as a
    candidate and used as the target later is copied to the other members device-to-device and stays there."""
    import torch
    from delta_graph_slam_amd.registration import Registration, RegistrationGroup
    tgt, sources, guesses, _ = batch
    r = Registration("NDT_OMP", ndt_resolution=1.0)
    r.setInputTarget(tgt)
    ref = r.align_batch(sources, guesses)
    dev_before = torch.cuda.current_device()
    g = RegistrationGroup("NDT_OMP", devices=[0, 0, 0], ndt_resolution=1.0)
    gt = g.make_cloud(tgt)                               # owner None: every member
    assert gt.copies == 3
    owned = [g.make_cloud(s, owner=i + 1) for i, s in enumerate(sources)]   # ids 1..7 -> members 1, 2, 0, 1, 2, 0, 1
    assert all(c.copies == 1 for c in owned)
    g.setInputTarget(gt)
    _same(g.align_batch(owned, guesses), ref)
    empty = g.make_cloud(np.zeros((0, 4), np.float32), owner=0)
    res = g.align_batch([owned[0], empty, owned[1]], guesses[:3])           # an empty keyframe: per-candidate failure, written on the device
    assert res[0]["converged"] and res[2]["converged"] and not res[1]["converged"] and res[1]["status"] == 4
    assert np.array_equal(res[1]["T"], guesses[1]) and np.isnan(res[1]["fitness"])
    # a candidate keyframe becomes the target: promoted to every member, results equal the single handle's
    g.setInputTarget(owned[3])
    assert owned[3].copies == 3
    r.setInputTarget(sources[3])
    _same(g.align_batch([owned[0], owned[5]], guesses[[0, 5]]), r.align_batch([sources[0], sources[5]], guesses[[0, 5]]))
    assert torch.cuda.current_device() == dev_before     # the group leaves the caller's current device alone
    g.close()


def test_loop_detector_caches_keyframes_on_a_group(batch):
    from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
    from delta_graph_slam_amd.registration import GroupCloud, Registration, RegistrationGroup
    tgt, sources, guesses, _ = batch
    new = KeyFrame(tgt, np.eye(3), 100.0, 100)
    cands = []
    for c, G in enumerate(guesses):
        est = np.eye(3)
        est[:2, :2] = G[:2, :2]
        est[:2, 2] = G[:2, 3]
        cands.append(KeyFrame(sources[c], est, 0.0, c + 1))
    d1 = LoopDetector({"fitness_score_thresh": 1e9}, registration=Registration("NDT_OMP", ndt_resolution=1.0))
    d2 = LoopDetector({"fitness_score_thresh": 1e9}, registration=RegistrationGroup("NDT_OMP", devices=[0, 0], ndt_resolution=1.0), cache_clouds=True)
    l1 = d1.matching(cands, new)
    for _ in range(2):
        l2 = d2.matching(cands, new)
        assert np.array_equal(d1.last_records[:, 1:], d2.last_records[:, 1:])
        assert (l1 is None) == (l2 is None) and (l1 is None or (l1.key2.id == l2.key2.id and np.array_equal(l1.relative_pose, l2.relative_pose)))
    assert len(d2._cloud_cache) == 8 and all(isinstance(c, GroupCloud) for c in d2._cloud_cache.values())
    assert d2._cloud_cache[100].copies == 1 and d2._cloud_cache[3].copies == 1   # the target was on both members during the tick, trimmed to its owner's after it


def test_group_cloud_trim_and_mixed_candidate_lists(batch):
    """dgs_group_cloud_trim: a replicated keyframe goes back to ONE copy (its owner's) and still serves as a candidate and, promoted again, as
    a target; a candidate list that mixes resident keyframes with raw clouds (a keyframe without an id) is served, the raw ones uploaded for the
    call -- all equal to the single handle's records."""
    from delta_graph_slam_amd.registration import Registration, RegistrationGroup
    tgt, sources, guesses, _ = batch
    r = Registration("NDT_OMP", ndt_resolution=1.0)
    r.setInputTarget(tgt)
    ref = r.align_batch(sources, guesses)
    g = RegistrationGroup("NDT_OMP", devices=[0, 0, 0], ndt_resolution=1.0)
    gt = g.make_cloud(tgt)
    assert gt.copies == 3
    g.setInputTarget(gt)
    kf = [g.make_cloud(s, owner=i) for i, s in enumerate(sources)]
    _same(g.align_batch(kf, guesses), ref)
    gt.trim(owner=7)                                      # member 7 mod 3 = 1 keeps its copy; the members it was bound to as target let go of it
    assert gt.copies == 1
    gt.trim(owner=7)                                      # idempotent
    assert gt.copies == 1
    g.setInputTarget(gt)                                  # promoted again: cloned device to device from the one holder
    assert gt.copies == 3
    mixed = [kf[0], sources[1], kf[2], sources[3], sources[4], kf[5], kf[6]]
    _same(g.align_batch(mixed, guesses), ref)
    r.setInputTarget(sources[2])
    g.setInputTarget(kf[2])
    kf[2].trim(owner=2)
    assert kf[2].copies == 1
    g.setInputTarget(kf[2])
    _same(g.align_batch([gt, kf[0]], guesses[[0, 1]]), r.align_batch([tgt, sources[0]], guesses[[0, 1]]))   # the trimmed ex-target as a candidate
    g.close()


def test_cfg4_at_its_full_candidate_count_256_candidates_over_8_members(oracle_lib):
    """BASELINE configs[3] at its stated shape on one card: 256 candidate keyframes x 65,536 points against one target, through a
    dgs_group of 8 members (device 0 listed 8 times: the 8-GPU node's 8 handles / host threads / streams; records gathered on the
    host because one device cannot hold 8 RCCL ranks), upstream operation order.  Every final transform EQUALS the one the
    reference's sequential candidate loop (loop_detector.hpp:137-156) produces on the oracle, the chosen candidate is the same, and a
    257th candidate that repeats the best one wins the tie (loop_detector.hpp:149: later candidate wins).  32 distinct ray-cast
    scans are re-used round-robin, each use with its own guess (synth.loop_batch)."""
    from concurrent.futures import ThreadPoolExecutor
    from delta_graph_slam_amd.registration import RegistrationGroup
    from tests.helpers import sequential_best
    N = 256
    tgt, sources, guesses, _ = synth.loop_batch(n_candidates=N, n_points=65536, seed=40, distinct_scans=32)
    o = oracle_lib.NdtOracle(resolution=1.0)
    o.set_target(tgt)
    ref = []
    for c in range(N):                                    # the reference's loop: one candidate after the other
        o.set_source(sources[c])
        ref.append(o.align(guesses[c]))
    with ThreadPoolExecutor(max_workers=8) as ex:
        fit_ref = list(ex.map(lambda c: oracle_lib.fitness_score(tgt, sources[c], ref[c]["T"])[0], range(N)))
    b_ref, s_ref = sequential_best([x["converged"] for x in ref], fit_ref)
    assert b_ref >= 0
    g = RegistrationGroup("NDT_OMP", devices=[0] * 8, ndt_resolution=1.0, ndt_strict_order=1)
    # resident keyframes (the deployable form): the 32 distinct scans uploaded once, owner = keyframe id; candidate c -> its owner
    gt = g.make_cloud(tgt)
    kf = [g.make_cloud(sources[k], owner=k) for k in range(32)]
    g.setInputTarget(gt)
    res = g.align_batch([kf[c % 32] for c in range(N)], guesses)
    off = []
    for c in range(N):
        assert res[c]["status"] == 0 and res[c]["converged"] == ref[c]["converged"] and res[c]["iterations"] == ref[c]["iterations"], c
        if not np.array_equal(res[c]["T"], ref[c]["T"]) or res[c]["evaluations"] != ref[c]["evaluations"]:
            off.append(c)
        # ndt_strict_order = 1 sums the points' double totals in the GPU's own fixed order: an evaluation differs from the CPU's by
        # ~1e-14 relative.  Where a More-Thuente line search sits at that noise level its sufficient-decrease test can take a few
        # more (or fewer) trials -- they refine the step by < 1e-9, far below a float of the transform -- and in rare cases ONE float
        # of a final transform moves by one ulp (measured on these 256 pairs: 9 with another evaluation count and the same
        # transform, 1 with a one-ulp float; 0 of either kind on the 96 pairs of the three bench shards) -- never more
        assert np.abs(res[c]["T"].astype(np.float64) - ref[c]["T"]).max() <= 1.2e-7 * max(1.0, np.abs(ref[c]["T"]).max()), c
        assert abs(res[c]["fitness"] - fit_ref[c]) <= 1e-9 * fit_ref[c], c
    # Round 3 asserted "at most 24 / at most 3"; the measurement itself is the assertion now (round-4 kernels, 8 members of 32 pairs): WHICH
    # candidates took another number of evaluations, and which of them ended one float ulp off.  A kernel change that moves a pair shows here.
    unequal = [c for c in off if not np.array_equal(res[c]["T"], ref[c]["T"])]
    assert (off, unequal) == (CFG4_OTHER_EVALUATION_COUNT, CFG4_ONE_ULP_TRANSFORMS), (off, unequal)
    assert g.best_index == b_ref and abs(g.best_score - s_ref) <= 1e-9 * s_ref
    # ... and on exactly those pairs index-order sums (ndt_strict_order = 2: every evaluation bit-identical to the CPU's) give the very
    # transform, iteration and evaluation counts of the reference loop
    if off:
        from delta_graph_slam_amd.registration import Registration
        r2 = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=2)
        r2.setInputTarget(tgt)
        seq = r2.align_batch([sources[c] for c in off], guesses[off])
        for c, x in zip(off, seq):
            assert np.array_equal(x["T"], ref[c]["T"]) and (x["iterations"], x["evaluations"]) == (ref[c]["iterations"], ref[c]["evaluations"]), c
    # host keyframes (dgs_group_align_batch, candidate c -> member c mod 8) + the tie: the best candidate once more at the end, on the
    # member that registers the original (so both sit in one batch and are summed alike: an exact tie), behind a few fillers
    g.setInputTarget(tgt)
    pad = (b_ref - N) % 8
    extra = [0] * pad + [b_ref]
    res2 = g.align_batch(list(sources) + [sources[c] for c in extra], np.concatenate([guesses, guesses[extra]]))
    for c in range(N):   # another batch size per member: another (fixed) partition of the sums -- the same statement as above holds
        assert res2[c]["iterations"] == ref[c]["iterations"] and res2[c]["converged"] == ref[c]["converged"], c
        assert np.abs(res2[c]["T"].astype(np.float64) - ref[c]["T"]).max() <= 1.2e-7 * max(1.0, np.abs(ref[c]["T"]).max()), c
        assert abs(res2[c]["fitness"] - fit_ref[c]) <= 1e-9 * fit_ref[c], c
    assert [c for c in range(N) if not np.array_equal(res2[c]["T"], ref[c]["T"])] == CFG4_ONE_ULP_TRANSFORMS_HOST_DEALING
    last = N + pad
    assert last % 8 == b_ref % 8
    assert np.array_equal(res2[last]["T"], res2[b_ref]["T"]) and res2[last]["fitness"] == res2[b_ref]["fitness"]
    assert g.best_index == last
    g.close()


def test_two_distinct_devices_rccl_gather_peer_clone_and_owner_dealing(oracle_lib):
    """The G > 1 paths that one-GPU boxes cannot reach (skipped there): ncclAllGather over two communicators, a candidate keyframe promoted
    to a target on the OTHER device (hipMemcpyPeerAsync in cloud_clone_to), and dealing by owner when G does not divide the candidate
    count -- against dgs_align_batch on one device, bit for bit.  Runs on the first multi-GPU node this suite meets."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from delta_graph_slam_amd.registration import Registration
    tgt, sources, guesses, _ = synth.loop_batch(n_candidates=7, n_points=16384, seed=40, distinct_scans=5)
    one = Registration("NDT_OMP", device=0, ndt_resolution=1.0, ndt_strict_order=1)
    one.setInputTarget(tgt)
    ref = one.align_batch(sources, guesses)
    g = RegistrationGroup("NDT_OMP", devices=[0, 1], ndt_resolution=1.0, ndt_strict_order=1)
    assert g.uses_rccl and g.rccl_ranks == 2
    kf = [g.make_cloud(sources[k], owner=k) for k in range(5)]       # owners 0,1,0,1,0: candidates 5, 6 re-use keyframes 0, 1
    g.setInputTarget(g.make_cloud(tgt))
    res = g.align_batch([kf[c % 5] for c in range(7)], guesses)
    assert g.last_gather_used_rccl
    for c in range(7):
        assert np.array_equal(res[c]["T"], ref[c]["T"]) and res[c]["converged"] == ref[c]["converged"], c
        assert abs(res[c]["fitness"] - ref[c]["fitness"]) <= 1e-12 * ref[c]["fitness"], c
    # a candidate keyframe owned by device 1 becomes the target: every member needs it (peer copy), results equal the single-device run
    one.setInputTarget(sources[1])
    ref2 = one.align_batch([sources[0], sources[2]], None)
    g.setInputTarget(kf[1])
    res2 = g.align_batch([kf[0], kf[2]], None)
    for c in range(2):
        assert np.array_equal(res2[c]["T"], ref2[c]["T"]), c
    assert kf[1].copies == 2
    g.close()
    one.close()
