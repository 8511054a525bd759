"""TEST-ONLY registration engine backed by the CPU oracle, with the align_batch surface LoopDetector expects.
It exists so that the multi-process (gloo) sharding / gather / arg-min logic can be exercised without a GPU.
Nothing under delta_graph_slam_amd/ imports this."""
import numpy as np

from oracle import oracle as orc


class OracleEngine:
    def __init__(self, method="NDT_OMP", **kw):
        self.method = method
        self.o = orc.GicpOracle(**kw) if "GICP" in method else orc.NdtOracle(**kw)
        self.target = None

    def setInputTarget(self, cloud):
        self.target = np.asarray(cloud, np.float32)
        self.o.set_target(self.target)

    def align_batch(self, sources, guesses=None, compute_fitness=True, fitness_max_range=1.7976931348623157e308):
        out = []
        for i, s in enumerate(sources):
            s = np.asarray(s, np.float32)
            g = None if guesses is None else guesses[i]
            if s.shape[0] == 0:
                out.append(dict(T=np.eye(4, dtype=np.float32) if g is None else np.asarray(g, np.float32), converged=False, iterations=0,
                                evaluations=0, status=4, score=0.0, fitness=float("nan")))
                continue
            self.o.set_source(s)
            r = self.o.align(g)
            fit = orc.fitness_score(self.target, s, r["T"], fitness_max_range)[0] if compute_fitness else float("nan")
            out.append(dict(T=r["T"], converged=r["converged"], iterations=r["iterations"], evaluations=r["evaluations"], status=0,
                            score=r["score"], fitness=fit))
        return out


class OracleRegistration:
    """TEST-ONLY pcl::Registration-shaped object over the CPU oracle (same method names as
    delta_graph_slam_amd.registration.Registration), used to check the odometry driver's call sequence."""

    def __init__(self, method="FAST_GICP", **kw):
        self.o = orc.GicpOracle(**kw) if "GICP" in method else orc.NdtOracle(**kw)
        self.target = self.source = None
        self._r = None

    def setInputTarget(self, cloud):
        self.target = np.asarray(cloud, np.float32)
        self.o.set_target(self.target)

    def setInputSource(self, cloud):
        self.source = np.asarray(cloud, np.float32)
        self.o.set_source(self.source)

    def align(self, guess=None, want_aligned=False):
        self._r = self.o.align(guess)

    def hasConverged(self):
        return self._r["converged"]

    def getFinalTransformation(self):
        return self._r["T"].copy()

    def getFitnessScore(self, max_range=1.7976931348623157e308):
        return orc.fitness_score(self.target, self.source, self._r["T"], max_range)[0]

    def getInlierFraction(self, max_sq=0.25):
        return orc.fitness_score(self.target, self.source, self._r["T"], 1.7976931348623157e308, max_sq)[2] / self.source.shape[0]
