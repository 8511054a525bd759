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
