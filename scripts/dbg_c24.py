import sys, numpy as np
sys.path.insert(0, '.')
from delta_graph_slam_amd import synth
from delta_graph_slam_amd.registration import Registration, RegistrationGroup
from oracle import oracle as orc
N = 256
tgt, sources, guesses, _ = synth.loop_batch(n_candidates=256, n_points=65536, seed=40, distinct_scans=32)
o = orc.NdtOracle(resolution=1.0); o.set_target(tgt)
ref = []
for c in range(N):
    o.set_source(sources[c]); ref.append(o.align(guesses[c]))
g = RegistrationGroup("NDT_OMP", devices=[0] * 8, ndt_resolution=1.0, ndt_strict_order=1)
g.setInputTarget(tgt)
res = g.align_batch(sources, guesses)
off = [c for c in range(N) if res[c]["evaluations"] != ref[c]["evaluations"] or not np.array_equal(res[c]["T"], ref[c]["T"])]
print('off', [(c, res[c]["evaluations"], ref[c]["evaluations"]) for c in off])
for mode in (2, 1):
    r2 = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=mode); r2.setInputTarget(tgt)
    seq = r2.align_batch([sources[c] for c in off], guesses[off])
    print('mode', mode, 'batch of off', [(c, x["iterations"], ref[c]["iterations"], x["evaluations"], ref[c]["evaluations"], bool(np.array_equal(x["T"], ref[c]["T"]))) for c, x in zip(off, seq)])
    for k, c in enumerate(off[:3]):
        tg = r2.ndt_trajectory(k); to = ref[c]['trajectory']; n = min(len(tg), len(to))
        print('   ', c, len(tg), len(to), 'traj max diff', np.abs(tg[:n] - to[:n]).max())
# the same in order 2 with ALL 256 in one handle batch (slow mode: 256 x ~8 ms)
r3 = Registration("NDT_OMP", ndt_resolution=1.0, ndt_strict_order=2); r3.setInputTarget(tgt)
rr = r3.align_batch(sources[:64], guesses[:64])
print('order 2, 64 in one batch:', [(c, rr[c]["evaluations"], ref[c]["evaluations"]) for c in range(64) if rr[c]["evaluations"] != ref[c]["evaluations"] or not np.array_equal(rr[c]["T"], ref[c]["T"])])
