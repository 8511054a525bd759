"""-m gpu: the reference's deployment shape -- TWO live registration objects of different methods in one process, used at the
same time from two host threads: the odometry nodelet's FAST_GICP object (apps/scan_matching_odometry_nodelet.cpp:105, one align per
incoming scan) and the loop detector's NDT object (include/hdl_graph_slam/loop_detector.hpp:49, a candidate batch per graph update)
live in one nodelet manager.  include/dgs_reg.h promises: a handle is used by one thread at a time, different
handles are independent.  Every result of the concurrent run must EQUAL the same work run alone."""
import threading

import numpy as np
import pytest

from delta_graph_slam_amd import synth

pytestmark = pytest.mark.gpu


def test_gicp_odometry_stream_and_ndt_loop_batches_run_concurrently_on_one_device():
    from delta_graph_slam_amd.odometry import ScanMatchingOdometry
    from delta_graph_slam_amd.registration import Registration
    clouds, _ = synth.vlp16_stream(n_frames=32)                                                  # cfg3
    tgt, sources, guesses, _ = synth.loop_batch(n_candidates=32, n_points=65536, seed=40, distinct_scans=8)   # the bench step's shape
    kw = dict(keyframe_delta_trans=1.0, keyframe_delta_angle=1.0, keyframe_delta_time=1e9)

    def odometry_run(out):
        odo = ScanMatchingOdometry(Registration("FAST_GICP", gicp_max_correspondence_distance=2.0, transformation_epsilon=0.1), kw)
        for k, c in enumerate(clouds):
            out.append(odo.matching(0.1 * k, c, want_status=True).copy())
            if k:
                out.append(np.array([odo.last_status.matching_error, odo.last_status.inlier_fraction, float(odo.last_status.has_converged)]))
        out.append(np.array([odo.n_keyframes]))

    def loop_run(out, rounds):
        reg = Registration("NDT_OMP", ndt_resolution=1.0)
        for _ in range(rounds):
            reg.setInputTarget(tgt)
            res = reg.align_batch(sources, guesses)
            out.append(np.concatenate([np.concatenate([x["T"].ravel(), [x["fitness"], x["iterations"], float(x["converged"])]]) for x in res]))

    solo_odo, solo_loop = [], []
    odometry_run(solo_odo)
    loop_run(solo_loop, 1)
    both_odo, both_loop, errors = [], [], []

    def guarded(fn, *a):
        try:
            fn(*a)
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    stop = threading.Event()

    def loop_until_stopped(out):
        n = 0
        while n < 3 or (not stop.is_set() and n < 200):   # keeps registering batches for as long as the odometry stream runs
            loop_run(out, 1)
            n += 1

    t1 = threading.Thread(target=guarded, args=(odometry_run, both_odo))
    t2 = threading.Thread(target=guarded, args=(loop_until_stopped, both_loop))
    t2.start()
    t1.start()
    t1.join()
    stop.set()
    t2.join()
    assert not errors, errors
    assert len(both_loop) >= 3                                              # the two did overlap: several batches during the stream
    assert len(both_odo) == len(solo_odo)
    for a, b in zip(both_odo, solo_odo):
        assert np.array_equal(a, b)
    for rec in both_loop:
        assert np.array_equal(rec, solo_loop[0])
