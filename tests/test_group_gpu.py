"""-m gpu: dgs_group (single process, several devices behind the C ABI; include/dgs_reg.h).  On the one-GPU box: a group over
{0} runs the RCCL path (ncclCommInitAll + ncclAllGather of the records, one rank) and must equal dgs_align_batch bit for bit; a
group over {0, 0} (the one-GPU rehearsal, host gather) exercises the dealing c -> c mod G, the re-ordering of the records and the
arg-min with the reference's tie rule."""
import numpy as np
import pytest

from delta_graph_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def batch():
    return synth.loop_batch(n_candidates=7, n_points=16384, seed=91, distinct_scans=4)


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert np.array_equal(x["T"], y["T"]) and x["converged"] == y["converged"] and x["iterations"] == y["iterations"]
        assert x["fitness"] == y["fitness"] and x["status"] == y["status"] == 0


@pytest.mark.parametrize("method,kw", [("NDT_OMP", dict(ndt_resolution=1.0)), ("FAST_GICP", dict(gicp_max_correspondence_distance=2.0))])
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
