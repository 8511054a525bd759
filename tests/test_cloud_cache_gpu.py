"""-m gpu tests for the device-resident keyframe clouds (dgs_cloud): a resident cloud must give the results of
the copying calls bit for bit, and its derived index / covariances must be built once."""
import numpy as np
import pytest

from delta_graph_slam_amd import synth

pytestmark = pytest.mark.gpu


def _keyframes(n_cand=4, n=8192, seed=3):
    from delta_graph_slam_amd.loop_detector import KeyFrame
    from delta_graph_slam_amd.transforms import transform3Dto2D
    new_cloud, cands, guesses, _ = synth.loop_batch(n_candidates=n_cand, n_points=n, seed=seed)
    new = KeyFrame(new_cloud, np.eye(3), accum_distance=100.0, id=1000)
    kfs = []
    for i, (c, g) in enumerate(zip(cands, guesses)):
        kfs.append(KeyFrame(c, transform3Dto2D(np.asarray(g, np.float32)).astype(np.float64), accum_distance=float(i), id=i))
    return new, kfs


@pytest.mark.parametrize("method", ["NDT_OMP", "FAST_GICP"])
def test_resident_batch_equals_copying_batch(method):
    from delta_graph_slam_amd.loop_detector import LoopDetector
    from delta_graph_slam_amd.registration import Registration
    new, kfs = _keyframes()
    kw = dict(gicp_max_correspondence_distance=2.0) if method == "FAST_GICP" else {}
    plain = LoopDetector({"fitness_score_thresh": 10.0}, Registration(method, **kw))
    cached = LoopDetector({"fitness_score_thresh": 10.0}, Registration(method, **kw), cache_clouds=True)
    a = plain.register_shard(kfs, new)
    b = cached.register_shard(kfs, new)
    assert np.array_equal(a, b, equal_nan=True)
    # a second tick over the same keyframes uses the cache and still gives the same records
    c = cached.register_shard(kfs, new)
    assert np.array_equal(a, c, equal_nan=True)
    assert len(cached._cloud_cache) == len(kfs) + 1
    cached.evict(0)
    assert len(cached._cloud_cache) == len(kfs)


def test_gicp_resident_cloud_builds_covariances_once():
    from delta_graph_slam_amd import _lib as L
    from delta_graph_slam_amd.registration import Registration
    tgt, src, _ = synth.planar_pair(n=8192)
    ref = Registration("FAST_GICP", gicp_max_correspondence_distance=2.0)
    ref.setInputTarget(tgt)
    ref.setInputSource(src)
    ref.align()
    T_ref = ref.getFinalTransformation()
    f_ref = ref.getFitnessScore()

    r = Registration("FAST_GICP", gicp_max_correspondence_distance=2.0)
    ct, cs = r.make_cloud(tgt), r.make_cloud(src)
    assert len(ct) == tgt.shape[0] and len(cs) == src.shape[0]
    r.profile_enable(True)
    r.profile_reset()
    for _ in range(3):
        r.setInputTarget(ct)
        r.setInputSource(cs)
        r.align()
        assert np.array_equal(r.getFinalTransformation(), T_ref)
        assert r.getFitnessScore() == f_ref
    _, n_cov = r.profile_get(L.K_GICP_COVARIANCE)
    assert n_cov == 2          # one k-NN covariance pass per cloud, not per setInput call
    # swapped roles reuse the same cached covariances (fast_gicp swapSourceAndTarget semantics)
    r.setInputTarget(cs)
    r.setInputSource(ct)
    r.align()
    _, n_cov = r.profile_get(L.K_GICP_COVARIANCE)
    assert n_cov == 2
    assert r.hasConverged()
    # a second handle with another k must not reuse covariances built for k = 20
    r2 = Registration("FAST_GICP", gicp_max_correspondence_distance=2.0, gicp_correspondence_randomness=10)
    r2.profile_enable(True)
    r2.setInputTarget(ct)
    r2.setInputSource(cs)
    r2.align()
    _, n2 = r2.profile_get(L.K_GICP_COVARIANCE)
    assert n2 == 2
    ref2 = Registration("FAST_GICP", gicp_max_correspondence_distance=2.0, gicp_correspondence_randomness=10)
    ref2.setInputTarget(tgt)
    ref2.setInputSource(src)
    ref2.align()
    assert np.array_equal(r2.getFinalTransformation(), ref2.getFinalTransformation())


def test_ndt_resident_target_and_source():
    import torch
    from delta_graph_slam_amd.registration import Registration
    tgt, src, _ = synth.planar_pair(n=8192)
    ref = Registration("NDT_OMP")
    ref.setInputTarget(tgt)
    ref.setInputSource(src)
    ref.align()
    r = Registration("NDT_OMP")
    ct = r.make_cloud(torch.from_numpy(tgt).cuda())     # created from a device tensor
    cs = r.make_cloud(src)                               # created from a host array
    r.setInputTarget(ct)
    r.setInputSource(cs)
    r.align()
    assert np.array_equal(r.getFinalTransformation(), ref.getFinalTransformation())
    assert r.getFitnessScore() == ref.getFitnessScore()
    # going back to copying inputs after a resident one must not touch the resident object
    r.setInputSource(tgt)
    r.setInputTarget(src)
    r.align()
    r.setInputTarget(ct)
    r.setInputSource(cs)
    r.align()
    assert np.array_equal(r.getFinalTransformation(), ref.getFinalTransformation())


def test_empty_resident_cloud_is_reported():
    from delta_graph_slam_amd.registration import Registration
    tgt, _, _ = synth.planar_pair(n=2048)
    r = Registration("NDT_OMP")
    r.setInputTarget(tgt)
    empty = r.make_cloud(np.zeros((0, 4), np.float32))
    assert len(empty) == 0
    out = r.align_batch([empty, r.make_cloud(tgt)], [np.eye(4, dtype=np.float32)] * 2)
    assert not out[0]["converged"]
    assert out[1]["converged"]


def test_destroying_a_bound_cloud_detaches_it():
    from delta_graph_slam_amd.registration import Registration
    tgt, src, _ = synth.planar_pair(n=2048)
    r = Registration("NDT_OMP")
    r.setInputTarget(tgt)
    cs = r.make_cloud(src)
    r.setInputSource(cs)
    r.align()
    assert r.hasConverged()
    cs.close()                          # the handle must not keep a dangling pointer
    r.align()                           # "no source set": PCL's contract, not converged and no crash
    assert not r.hasConverged()
    r.setInputSource(src)
    r.align()
    assert r.hasConverged()


def test_find_candidates_on_the_device_equals_the_reference_loop():
    """dgs_find_loop_candidates == LoopDetector::find_candidates (loop_detector.hpp:83-111): the same keyframes in the same order,
    including keyframes that sit exactly on either threshold, for a few hundred thousand keyframes as well as for none."""
    from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
    from delta_graph_slam_amd.registration import Registration
    rng = np.random.default_rng(5)
    reg = Registration("NDT_OMP")
    for n in (0, 1, 63, 64, 65, 1023, 1025, 5000):
        kfs = []
        for i in range(n):
            e = np.eye(3)
            e[:2, 2] = rng.uniform(-12, 12, 2)
            kfs.append(KeyFrame(None, e, float(rng.uniform(0, 60)), i))
        new = KeyFrame(None, np.eye(3), 40.0, n)
        new.estimate[:2, 2] = [1.5, -0.5]
        if n >= 64:   # exactly on the thresholds: accum difference == 8 is kept (`<` skips), distance == 5 is kept (`>` skips)
            kfs[10].accum_distance = 32.0
            kfs[10].estimate[:2, 2] = [1.5 + 3.0, -0.5 + 4.0]
            kfs[11].accum_distance = np.nextafter(32.0, 100.0)
            kfs[11].estimate[:2, 2] = [1.5, -0.5]
        host = LoopDetector({}, registration=reg)
        dev = LoopDetector({}, registration=reg, filter_on_device=True)
        a, b = host.find_candidates(kfs, new), dev.find_candidates(kfs, new)
        assert [k.id for k in a] == [k.id for k in b], n
        if n >= 64:
            assert 10 in [k.id for k in b] and 11 not in [k.id for k in b]
    big = 300000
    acc = rng.uniform(0, 60, big)
    xy = rng.uniform(-12, 12, (big, 2))
    idx = reg.find_loop_candidates(acc, xy, 40.0, [1.5, -0.5], 8.0, 5.0)
    want = np.nonzero(~(40.0 - acc < 8.0) & ~(np.sqrt((xy[:, 0] - 1.5) ** 2 + (xy[:, 1] + 0.5) ** 2) > 5.0))[0]
    assert np.array_equal(idx, want)
