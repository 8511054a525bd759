"""Edge information matrices from the NN fitness score: the reference's InformationMatrixCalculator over the HIP kernel.

Mirrors /root/reference/src/hdl_graph_slam/information_matrix_calculator.cpp:30-75 (parameters, calc_information_matrix) and
include/hdl_graph_slam/information_matrix_calculator.hpp:46-49 (weight); calc_fitness_score (:77-108) runs on the device
through dgs_calc_fitness_score (called per odometry edge and per loop edge,
apps/delta_graph_slam_nodelet.cpp:572,820, each time building a fresh kd-tree on the CPU in the reference).
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

__all__ = ["InformationMatrixCalculator"]
DBL_MAX = 1.7976931348623157e308


class InformationMatrixCalculator:
    def __init__(self, params: Optional[dict] = None, registration=None, device: Optional[int] = None):
        pr = dict(params or {})
        self.use_const_inf_matrix = bool(pr.get("use_const_inf_matrix", False))
        self.const_stddev_x = float(pr.get("const_stddev_x", 0.5))
        self.const_stddev_q = float(pr.get("const_stddev_q", 0.1))
        self.var_gain_a = float(pr.get("var_gain_a", 20.0))
        self.min_stddev_x = float(pr.get("min_stddev_x", 0.1))
        self.max_stddev_x = float(pr.get("max_stddev_x", 5.0))
        self.min_stddev_q = float(pr.get("min_stddev_q", 0.05))
        self.max_stddev_q = float(pr.get("max_stddev_q", 0.2))
        self.fitness_score_thresh = float(pr.get("fitness_score_thresh", 0.5))
        if registration is None and not self.use_const_inf_matrix:
            from .registration import Registration
            registration = Registration("NDT_OMP", device=device)   # any handle: only its NN machinery is used
        self.registration = registration

    @staticmethod
    def weight(a: float, max_x: float, min_y: float, max_y: float, x: float) -> float:
        y = (1.0 - math.exp(-a * x)) / (1.0 - math.exp(-a * max_x))
        return min_y + (max_y - min_y) * y

    def calc_fitness_score(self, cloud1, cloud2, relpose, max_range: float = DBL_MAX) -> float:
        return self.registration.calc_fitness_score(cloud1, cloud2, np.asarray(relpose, np.float64).astype(np.float32), max_range)

    def calc_information_matrix(self, cloud1, cloud2, relpose) -> np.ndarray:
        inf = np.eye(3)
        if self.use_const_inf_matrix:
            inf[:2, :2] /= self.const_stddev_x
            inf[2, 2] /= self.const_stddev_q
            return inf
        fitness = self.calc_fitness_score(cloud1, cloud2, relpose)
        w_x = np.float32(self.weight(self.var_gain_a, self.fitness_score_thresh, self.min_stddev_x ** 2, self.max_stddev_x ** 2, fitness))
        w_q = np.float32(self.weight(self.var_gain_a, self.fitness_score_thresh, self.min_stddev_q ** 2, self.max_stddev_q ** 2, fitness))
        inf[:2, :2] /= float(w_x)
        inf[2, 2] /= float(w_q)
        return inf
