"""CPU tests of the host-side mirror: 2-D/3-D guess flattening, candidate filtering, arg-min tie-breaking, generators."""
import hashlib

import pytest

import numpy as np

from delta_graph_slam_amd import synth
from delta_graph_slam_amd.loop_detector import RECORD_WIDTH, KeyFrame, LoopDetector
from delta_graph_slam_amd.transforms import euler_angles_012, normalize_euler_angs, transform2Dto3D, transform3Dto2D


def _se2(x, y, yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, x], [s, c, y], [0, 0, 1.0]])


def test_transform2d3d_roundtrip():
    """ros_utils.cpp:105-144"""
    for x, y, yaw in [(1.0, -2.0, 0.3), (0.0, 0.0, -2.5), (5.0, 1.0, 3.1)]:
        T3 = transform2Dto3D(_se2(x, y, yaw))
        assert T3.dtype == np.float32 and T3[2, 3] == 0 and T3[2, 2] == 1
        assert np.allclose(T3[:2, 3], [x, y], atol=1e-6)
        T2 = transform3Dto2D(T3)
        assert np.allclose(T2, _se2(x, y, yaw), atol=2e-6)


def test_transform3dto2d_ignores_roll_pitch_like_the_reference():
    T = synth.make_transform((1, 2, 3), (0.02, -0.03, 0.7)).astype(np.float32)
    T2 = transform3Dto2D(T)
    e = normalize_euler_angs(euler_angles_012(T[:3, :3]))
    assert abs(e[2] - 0.7) < 1e-5
    assert np.allclose(T2[:2, 2], [1, 2]) and abs(np.arctan2(T2[1, 0], T2[0, 0]) - e[2]) < 1e-6


def test_euler_angles_match_eigen_convention():
    # Eigen's eulerAngles(0,1,2) returns the first angle in [0, pi]; the triple must still reproduce the rotation
    for r in [(0.01, -0.02, 0.05), (-0.01, 0.02, -0.05), (-2.0, 0.4, 1.0), (3.0, -1.2, -2.0)]:
        R = synth.euler_to_matrix(*r)
        e = euler_angles_012(R.astype(np.float32))
        assert 0.0 <= e[0] <= np.pi + 1e-6
        assert np.allclose(synth.euler_to_matrix(*e.astype(np.float64)), R, atol=2e-6)


def test_find_candidates_follows_the_reference_filters():
    """loop_detector.hpp:83-111"""
    det = LoopDetector({"distance_thresh": 5.0, "accum_distance_thresh": 8.0, "min_edge_interval": 5.0}, registration=object())
    kfs = [KeyFrame(None, _se2(0, 0, 0), 0.0), KeyFrame(None, _se2(3, 0, 0), 3.0), KeyFrame(None, _se2(30, 0, 0), 30.0),
           KeyFrame(None, _se2(1, 1, 0), 45.0)]
    new = KeyFrame(None, _se2(1, 0, 0), 50.0)
    c = det.find_candidates(kfs, new)
    assert [k.accum_distance for k in c] == [0.0, 3.0]          # #2 too far in space, #3 too close in travelled distance
    det.last_edge_accum_distance = 47.0
    assert det.find_candidates(kfs, new) == []                   # too close to the last loop edge


def test_select_best_tie_goes_to_the_later_candidate():
    """loop_detector.hpp:149: a candidate is skipped only if score > best, so an exact tie replaces the earlier one."""
    rec = np.full((5, RECORD_WIDTH), -1.0)
    rec[:, 0] = np.arange(5)
    rec[:, 1] = [1, 1, 0, 1, 1]
    rec[:, 2] = [0.30, 0.20, 0.01, 0.20, 0.25]
    best, score = LoopDetector.select_best(rec)
    assert best == 3 and score == 0.20                           # #2 is better but did not converge; #1 and #3 tie -> #3
    rec[:, 1] = 0
    assert LoopDetector.select_best(rec)[0] == -1


def test_guess_is_relative_pose_flattened():
    new = KeyFrame(None, _se2(10, 5, 0.5), 0)
    cand = KeyFrame(None, _se2(11, 5.5, 0.7), 0)
    G = LoopDetector.guess_for(new, cand)
    rel = np.linalg.inv(_se2(10, 5, 0.5)) @ _se2(11, 5.5, 0.7)
    assert np.allclose(G[:2, 3], rel[:2, 2], atol=1e-6) and abs(np.arctan2(G[1, 0], G[0, 0]) - 0.2) < 1e-6
    assert G[2, 3] == 0 and G[2, 2] == 1


def test_generators_are_deterministic_and_shaped():
    a1, b1, T1 = synth.planar_pair(n=4096)
    a2, b2, T2 = synth.planar_pair(n=4096)
    assert hashlib.sha1(a1.tobytes()).hexdigest() == hashlib.sha1(a2.tobytes()).hexdigest()
    assert a1.shape == (4096, 4) and a1.dtype == np.float32 and np.all(a1[:, 3] == 1)
    s, Tw = synth.hdl64_scan((0.0, 0.0, 0.0), seed=10, n_points=65536)
    assert s.shape == (65536, 4) and np.isfinite(s).all()
    r = np.linalg.norm(s[:, :3], axis=1)
    assert r.min() > 0.05 and r.max() <= 100.5                   # distance filter of delta_graph_slam.launch:31-33
    clouds, poses = synth.vlp16_stream(n_frames=2)
    assert all(20000 < c.shape[0] <= 30000 for c in clouds)      # ragged: sky rays miss


def test_odometry_driver_sequence_on_cpu_engine():
    """scan_matching_odometry_nodelet.cpp:173-270 with the oracle as the registration (call-sequence test, no GPU)."""
    from delta_graph_slam_amd.odometry import ScanMatchingOdometry
    from tests.oracle_engine import OracleRegistration
    clouds, poses = [], []
    for k in range(5):
        xyz, T = synth.street_scan((-20.0 + 0.3 * k, 0.2 * np.sin(k), 0.01 * k), 16, (15.0, -15.0), 1875, 300 + k)
        clouds.append(synth._xyz1(xyz[::3]))
        poses.append(T)
    odo = ScanMatchingOdometry(OracleRegistration("FAST_GICP", max_correspondence_distance=2.0, transformation_epsilon=0.01, num_threads=4),
                               {"keyframe_delta_trans": 0.5, "keyframe_delta_angle": 1.0, "keyframe_delta_time": 1e9})
    out = [odo.matching(0.1 * k, c, want_status=(k == 1)) for k, c in enumerate(clouds)]
    assert np.array_equal(out[0], np.eye(4, dtype=np.float32))
    assert odo.last_status is not None and odo.last_status.has_converged and 0.5 < odo.last_status.inlier_fraction <= 1.0
    gt = [np.linalg.inv(poses[0]) @ p for p in poses]
    for o, g in zip(out[1:], gt[1:]):
        assert o.shape == (4, 4) and o.dtype == np.float32
        assert abs(o[1, 3] - g[1, 3]) < 0.02 and abs(o[0, 3] - g[0, 3]) < 0.15      # thinned 16-beam scans: x is the weak direction
    # the keyframe is replaced as soon as |trans| exceeds keyframe_delta_trans (:249-260) and prev_trans restarts from I
    # 0.3 m per frame against keyframe_delta_trans 0.5: the keyframe is replaced after frames 2 and 4
    assert odo.n_keyframes == 3
    assert np.array_equal(odo.prev_trans, np.eye(4, dtype=np.float32))            # reset at the last switch (:259)


def test_batched_guesses_equal_the_per_candidate_form():
    from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector
    rng = np.random.default_rng(3)

    def se2():
        a = rng.uniform(-np.pi, np.pi)
        return np.array([[np.cos(a), -np.sin(a), rng.uniform(-30, 30)], [np.sin(a), np.cos(a), rng.uniform(-30, 30)], [0, 0, 1.0]])
    new = KeyFrame(None, se2(), 50.0, 0)
    cands = [KeyFrame(None, se2(), 0.0, i) for i in range(17)]
    one_by_one = np.stack([LoopDetector.guess_for(new, k) for k in cands])
    assert np.array_equal(LoopDetector.guesses_for(new, cands), one_by_one)
    assert LoopDetector.guesses_for(new, []).shape == (0, 4, 4)


def test_record_path_equals_dict_path():
    """register_shard fills its exchange records either from align_batch's dicts or from align_batch_records (the product
    path, straight from the C ABI result array): same rows either way."""
    from delta_graph_slam_amd.loop_detector import RECORD_WIDTH, KeyFrame, LoopDetector
    rng = np.random.default_rng(5)
    n = 5
    Ts = rng.normal(size=(n, 4, 4)).astype(np.float32)
    fits = rng.uniform(0.1, 1.0, n)
    conv = [True, False, True, True, False]

    class DictEngine:
        def setInputTarget(self, cloud):
            pass

        def align_batch(self, sources, guesses=None, compute_fitness=True, fitness_max_range=0.0):
            self.guesses = np.asarray(guesses)
            return [dict(T=Ts[i], converged=conv[i], iterations=1, evaluations=2, status=0 if conv[i] else 4, score=0.0, fitness=fits[i])
                    for i in range(len(sources))]

    class RecordEngine(DictEngine):
        def align_batch_records(self, sources, guesses=None, compute_fitness=True, fitness_max_range=0.0):
            self.guesses = np.asarray(guesses)
            out = np.full((len(sources), RECORD_WIDTH), -1.0)
            out[:, 1] = conv
            out[:, 2] = fits
            out[:, 3] = [0 if c else 4 for c in conv]
            out[:, 4:20] = Ts.astype(np.float64).reshape(n, 16)
            return out

    new = KeyFrame(np.zeros((1, 4), np.float32), np.eye(3), 50.0, 0)
    cands = [KeyFrame(np.zeros((1, 4), np.float32), np.eye(3), 0.0, i) for i in range(n)]
    a, b = DictEngine(), RecordEngine()
    ra = LoopDetector({}, registration=a).register_shard(cands, new)
    rb = LoopDetector({}, registration=b).register_shard(cands, new)
    assert np.array_equal(ra, rb) and np.array_equal(a.guesses, b.guesses)
    assert list(ra[:, 0]) == list(range(n))
    assert LoopDetector.select_best(ra)[0] == LoopDetector.select_best(rb)[0]


def test_keyframe_cache_is_bounded_and_a_group_target_goes_back_to_one_copy():
    """LoopDetector(cache_clouds=True): after a tick the new keyframe -- every member's target on a group -- is trimmed to the copy on its owner,
    and with cache_capacity set the least recently used keyframes beyond it are released, never one the tick used (host logic; the device
    side of trim / mixed lists is tests/test_group_gpu.py)."""
    from delta_graph_slam_amd.loop_detector import KeyFrame, LoopDetector, RECORD_WIDTH
    log = []

    class Cloud:
        def __init__(self, owner):
            self.owner, self.closed = owner, False

        def trim(self, owner=-1):
            log.append(("trim", owner))

        def close(self):
            self.closed = True
            log.append(("close", self.owner))

    class Group:
        devices = [0, 1]

        def make_cloud(self, cloud, owner=None):
            return Cloud(owner)

        def setInputTarget(self, cloud):
            self.target = cloud

        def align_batch_records(self, sources, guesses=None, compute_fitness=True, fitness_max_range=0.0):
            out = np.full((len(sources), RECORD_WIDTH), -1.0)
            out[:, 1] = 1.0
            out[:, 2] = np.arange(len(sources)) + 1.0
            out[:, 3] = 0
            out[:, 4:20] = np.eye(4).reshape(16)
            return out

    kf = lambda i, acc: KeyFrame(np.zeros((1, 4), np.float32), np.eye(3), acc, i)
    det = LoopDetector({}, registration=Group(), cache_clouds=True, cache_capacity=4)
    det.register_shard([kf(1, 0.0), kf(2, 0.0), kf(3, 0.0)], kf(10, 50.0))
    assert ("trim", 10) in log and list(det._cloud_cache) == [1, 2, 3, 10]        # 4 resident: at capacity, nothing released
    assert det._cloud_cache[10].owner is None and det._cloud_cache[2].owner == 2   # target on every member, candidates on their owners
    log.clear()
    det.register_shard([kf(3, 0.0), kf(10, 50.0), kf(4, 0.0)], kf(11, 60.0))      # keyframe 10 is a candidate now: served from the cache
    assert ("trim", 11) in log
    assert sorted(det._cloud_cache) == [3, 4, 10, 11] and ("close", 1) in log and ("close", 2) in log   # the two least recently used went
    assert not det._cloud_cache[10].closed
    log.clear()
    det.cache_capacity = 2
    det.register_shard([kf(3, 0.0), kf(10, 50.0), kf(4, 0.0)], kf(12, 70.0))      # a tick that uses 4 keyframes keeps all 4: only 11 can go
    assert sorted(det._cloud_cache) == [3, 4, 10, 12] and log.count(("close", None)) == 1


def test_factory_refuses_the_reference_branches_it_does_not_serve():
    """registrations.cpp:59-100: ICP / GICP / GICP_OMP / plain NDT / FAST_VGICP_CUDA are other algorithms -- never silently replaced."""
    from delta_graph_slam_amd.registration import Registration, select_registration_method
    for name in ("ICP", "GICP", "GICP_OMP", "NDT", "FAST_VGICP_CUDA"):
        with pytest.raises(NotImplementedError):
            select_registration_method({"registration_method": name})
        with pytest.raises(NotImplementedError):
            Registration(name)
    # :88-123 names the reference does not know: it warns and then decides on "OMP" alone -- "FOO" and "NDT_FOO" become
    # pcl::NormalDistributionsTransform (not served), "FOO_OMP" becomes pclomp's NDT (served: reaches dgs_create)
    for name in ("FOO", "NDT_FOO", "MY_NDT"):
        with pytest.raises(NotImplementedError):
            select_registration_method({"registration_method": name})


def test_factory_unknown_name_with_omp_falls_to_ndt_omp_like_the_reference(capsys):
    from delta_graph_slam_amd.registration import DgsError, select_registration_method
    try:
        r = select_registration_method({"registration_method": "FOO_OMP", "reg_resolution": 1.5})
    except DgsError as e:          # no GPU here: the factory got as far as dgs_create for NDT_OMP
        assert e.status == 2
    else:
        assert r.method == "NDT_OMP" and r.params.ndt_resolution == 1.5
        r.close()
    assert "unknown registration type(FOO_OMP)" in capsys.readouterr().err
