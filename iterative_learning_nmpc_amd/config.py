"""Configuration dataclasses of the MPC hot path.

Field names, defaults and factory keys restate the reference's configuration surface so a
caller written against it keeps working:
  GaitConfig / MPCOptConfig / MPCCostConfig   mpc_controller/config/config_abstract.py:7-146
  trot & friends                              mpc_controller/config/quadruped/mpc_gait.py:15-64
  MPCQuadrupedCyclic                          mpc_controller/config/quadruped/mpc_opt.py:8-27
  Go2TrotCost / Go2SlowTrotCost               mpc_controller/config/quadruped/mpc_cost.py:15-128
  get_quadruped_config                        mpc_controller/config/quadruped/utils.py:8-16
Golden values: tests/golden/config_*.json (generated from the reference, tests/golden/make_golden.py).
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Tuple

import numpy as np


class HPIPM_MODE(enum.Enum):
    """Stand-in for contact_tamp's enum (config_abstract.py:5,64); only `speed` is used."""
    speed_abs = 0
    speed = 1
    balance = 2
    robust = 3


def _arr(values, scale: float = 1.0):
    return field(default_factory=lambda: np.asarray(values, dtype=np.float64) * scale)


# --------------------------------------------------------------------------- gait
@dataclass
class GaitConfig:
    gait_name: str
    nominal_period: float
    stance_ratio: np.ndarray
    phase_offset: np.ndarray
    nom_height: float
    step_height: float
    n_eeff: int = 4

    def __post_init__(self):
        self.stance_ratio = np.asarray(self.stance_ratio, dtype=np.float64)
        self.phase_offset = np.asarray(self.phase_offset, dtype=np.float64)
        if not (np.all(self.stance_ratio >= 0) and np.all(self.stance_ratio <= 1)):
            raise AssertionError("stance_ratio should be in [0,1]")
        if not (np.all(self.phase_offset >= 0) and np.all(self.phase_offset <= 1)):
            raise AssertionError("phase_offset should be in [0,1]")
        if len(self.stance_ratio) != self.n_eeff or len(self.phase_offset) != self.n_eeff:
            raise AssertionError(f"stance_ratio / phase_offset must be of length {self.n_eeff}")


# name -> (period, stance ratio, phase offsets FL FR RL RR, nominal height, step height)
_GAIT_TABLE = {
    "trot": (0.5, 0.5, (0.5, 0.0, 0.0, 0.5), 0.30, 0.05),
    "slow_trot": (1.0, 0.63, (0.5, 0.0, 0.0, 0.5), 0.32, 0.065),
    "jump": (50.0, 0.4, (0.0, 0.0, 0.0, 0.0), 0.3, 0.05),
    "crawl": (1.0, 0.75, (0.0, 0.25, 0.5, 0.75), 0.3, 0.05),
    "pace": (0.5, 0.6, (0.0, 0.5, 0.5, 0.0), 0.05, 0.32),
    "bound": (0.5, 0.6, (0.0, 0.5, 0.5, 0.0), 0.05, 0.32),
}


class GaitConfigFactory:
    AVAILABLE_GAITS = tuple(_GAIT_TABLE)

    @staticmethod
    def get(gait_name: str) -> GaitConfig:
        key = gait_name.lower()
        if key not in _GAIT_TABLE:
            raise ValueError(f"{gait_name} not available.")
        period, ratio, offsets, height, step = _GAIT_TABLE[key]
        return GaitConfig(key, float(period), np.full(4, ratio), np.asarray(offsets, float),
                          height, step)


# --------------------------------------------------------------------------- optimisation
@dataclass
class MPCOptConfig:
    time_horizon: float = 1.0
    n_nodes: int = 25
    replanning_freq: int = 25
    Kp: float = 20.0
    Kd: float = 1.75
    recompile: bool = False
    max_iter: int = 1
    max_qp_iter: int = 6
    real_time_it: bool = False
    enable_time_opt: bool = False
    opt_dt_scale: Tuple[float, float] = (0.5, 1.75)
    enable_impact_dyn: bool = False
    opt_peak: bool = True
    warm_start_sol: bool = True
    warm_start_nlp: bool = True
    warm_start_qp: bool = True
    hpipm_mode: HPIPM_MODE = HPIPM_MODE.speed
    use_cython: bool = False
    torque_limit: bool = True
    mu: float = 0.7
    nlp_tol: float = 1.0e-1
    qp_tol: float = 1.0e-2

    def __post_init__(self):
        if len(self.opt_dt_scale) != 2:
            raise AssertionError("opt_dt_scale must be of shape 2")
        if not self.mu > 0:
            raise AssertionError("Friction coefficient must be positive")

    def get_dt_nodes(self) -> float:
        return round(self.time_horizon / self.n_nodes, 4)

    def get_dt_bounds(self) -> Tuple[float, float]:
        dt = self.get_dt_nodes()
        return round(dt * self.opt_dt_scale[0], 4), round(dt * self.opt_dt_scale[1], 4)


class MPCQuadrupedCyclic(MPCOptConfig):
    """Same defaults as the base (the reference's only concrete optimisation config)."""


# --------------------------------------------------------------------------- cost
_LEG_SCALE = (15.0, 5.0, 1.0)  # hip, shoulder, elbow
_N_FEET = 4


@dataclass
class MPCCostConfig:
    robot_name: str
    gait_name: str
    W_e_base: np.ndarray
    W_base: np.ndarray
    W_joint: np.ndarray
    W_e_joint: np.ndarray
    W_acc: np.ndarray
    W_swing: np.ndarray
    W_eeff_ori: np.ndarray
    W_cnt_f_reg: np.ndarray
    W_foot_pos_constr_stab: np.ndarray
    W_foot_displacement: np.ndarray
    cnt_radius: float
    time_opt: np.ndarray
    reg_eps: float
    reg_eps_e: float

    def __post_init__(self):
        assert len(self.W_e_base) == 12, "W_e_base must be of shape 12"
        assert len(self.W_base) == 12, "W_base must be of shape 12"
        assert len(self.W_acc) == 12, "W_acc must be of shape 12"
        assert len(self.W_swing) == len(self.W_cnt_f_reg) == len(self.W_foot_pos_constr_stab), \
            "W_swing and W_foot should have the same length."
        for i, w in enumerate(self.W_cnt_f_reg):
            assert len(w) == 3, f"W_foot[{i}] must be of shape 3"


def _go2_trot() -> MPCCostConfig:
    legs = np.tile(_LEG_SCALE, _N_FEET)
    return MPCCostConfig(
        robot_name="Go2", gait_name="trot",
        W_base=np.array([1e3, 3e3, 1e2, 5e2, 5e2, 5e2, 5e2, 1e1, 1e0, 1e0, 2e1, 1e1]),
        W_e_base=np.array([1e1, 1e1, 1e3, 1e1, 1e2, 1e2, 5e2, 5e2, 1e3, 1e1, 1e2, 1e2]),
        W_joint=np.concatenate([legs, np.full(12, 0.03)]) * 5.0,
        W_e_joint=np.concatenate([legs, np.full(12, 0.1)]),
        W_acc=legs * 5.0e-4,
        W_swing=np.full(_N_FEET, 2e4),
        W_eeff_ori=np.ones(_N_FEET),
        W_cnt_f_reg=np.tile([0.01, 0.01, 0.05], (_N_FEET, 1)),
        W_foot_pos_constr_stab=np.full(_N_FEET, 5e1),
        W_foot_displacement=np.array([1e3]),
        cnt_radius=0.015, time_opt=np.array([1.0e4]), reg_eps=1.0e-6, reg_eps_e=1.0e-5)


def _go2_slow_trot() -> MPCCostConfig:
    base = np.array([0, 0, 5e3, 0, 3e3, 3e3, 0, 0, 1e1, 1e0, 1e2, 2e2])
    legs = np.tile(_LEG_SCALE, _N_FEET)
    return MPCCostConfig(
        robot_name="Go2", gait_name="slow_trot",
        W_base=base * 7.0, W_e_base=base * 10.0,
        W_joint=np.concatenate([legs, np.zeros(12)]) * 0.1,
        W_e_joint=np.zeros(24),
        W_acc=np.tile([7.0, 3.0, 1.0], _N_FEET) * 1.0e-2,
        W_swing=np.full(_N_FEET, 5e5),
        W_eeff_ori=np.zeros(_N_FEET),
        W_cnt_f_reg=np.tile([1.2, 1.2, 0.9], (_N_FEET, 1)),
        W_foot_pos_constr_stab=np.full(_N_FEET, 5e1),
        W_foot_displacement=np.array([1e6]),
        cnt_radius=0.005, time_opt=np.array([1.0e4]), reg_eps=1.0e-6, reg_eps_e=1.0e-5)


class CostConfigFactory:
    _MAKERS = {("go2", "trot"): _go2_trot, ("go2", "slow_trot"): _go2_slow_trot}

    @staticmethod
    def get(robot_name: str, gait_name: str) -> MPCCostConfig:
        maker = CostConfigFactory._MAKERS.get((robot_name.lower(), gait_name.lower()))
        if maker is None:
            raise ValueError(f"Cost config: {gait_name} for {robot_name} not available.")
        return maker()


def get_quadruped_config(gait_name: str, robot_name: str):
    """(gait, opt, cost) triple, as mpc_controller/config/quadruped/utils.py:8-16."""
    return (GaitConfigFactory.get(gait_name), MPCQuadrupedCyclic(),
            CostConfigFactory.get(robot_name, gait_name))
