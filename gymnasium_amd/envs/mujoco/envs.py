"""HalfCheetah-v5 / Ant-v5 / Humanoid-v5 as lockstep MI355X vector environments.

Each class is the ``vector_entry_point`` creator for one id; its constructor takes the kwargs of the reference's scalar env
(``make_vec`` forwards them verbatim, envs/registration.py:957-963) and describes the same spaces:

  HalfCheetahVectorEnv   envs/mujoco/half_cheetah_v5.py:153-218   obs float64[17],  action float32[6]  in [-1, 1]
  AntVectorEnv           envs/mujoco/ant_v5.py:228-321            obs float64[105], action float32[8]  in [-1, 1]
  HumanoidVectorEnv      envs/mujoco/humanoid_v5.py:307-411       obs float64[348], action float32[17] in [-0.4, 0.4]

Observation space: Box(-inf, inf, (obs_size,), float64); action space: Box(ctrlrange, float32) (mujoco_env.py:113-117).
The robots themselves are gymnasium_amd/envs/mujoco/models.py (a transcription of the reference's MJCF assets); the
physics runs in gymnasium_amd/csrc (HIP).  Nothing here computes a step.  rendering / xml_file overrides are not
supported (GPU-resident lanes): ``xml_file`` other than the stock asset raises.
"""
from __future__ import annotations

import numpy as np

from ...gym_api import error, logger, spaces
from ...vector.hip_vector_env import HipVectorEnv
from . import models as _models

_WARNED_UNPINNED = False


def _fma(a, b, c):
    """fma(a, b, c) = RN(a b + c) elementwise without a hardware instruction: Dekker's exact product (Veltkamp splitting) + Knuth's exact sum; the
    final addition rounds the three-term sum once (exact unless the sum sits within 2^-53 ulp of a rounding boundary)."""
    p = a * b
    ah = 134217729.0 * a
    ah = ah - (ah - a)
    al = a - ah
    bh = 134217729.0 * b
    bh = bh - (bh - b)
    bl = b - bh
    e = ((ah * bh - p) + ah * bl + al * bh) + al * bl
    s = c + p
    bb = s - c
    t = (c - (s - bb)) + (p - bb)
    return s + (t + e)


def _np_norm(*cols):
    """np.linalg.norm of the rows (x, y[, z]) the way the reference gets it for one row: sqrt(x.dot(x)) with the BLAS dot's accumulation -- x0 x0,
    then fused multiply-adds (oracle/mujoco_envs.c orc_np_norm; ant_v5.py:427 `np.linalg.norm(self.data.qpos[0:2], ord=2)`)."""
    acc = cols[0] * cols[0]
    for c in cols[1:]:
        acc = _fma(c, c, acc)
    return np.sqrt(acc)


class _MujocoVectorEnv(HipVectorEnv):
    STOCK_XML = ""
    NQ = NV = NU = NBODY = 0
    CTRL_LOW = CTRL_HIGH = 0.0
    TENDONS: tuple = ()  # ((qpos address, dof address, coefficient), ...) per fixed tendon (humanoid.xml:91-100)

    def _check_common(self, xml_file, frame_skip, kwargs):
        global _WARNED_UNPINNED
        if xml_file != self.STOCK_XML:
            raise error.Error(f"gymnasium_amd runs the stock {self.STOCK_XML} model compiled into the engine; got xml_file={xml_file!r}")
        for k in ("default_camera_config", "width", "height", "camera_id", "camera_name", "max_geom", "visual_options"):
            kwargs.pop(k, None)  # rendering-only arguments of MujocoEnv (mujoco_env.py:38-55)
        self.frame_skip = int(frame_skip)
        if not _WARNED_UNPINNED:  # once per process
            _WARNED_UNPINNED = True
            logger.warn("gymnasium_amd MuJoCo-family environments run a from-scratch restatement of MuJoCo's published pipeline.  It reproduces "
                        "the two `mujoco`-produced known answers the reference holds to every printed digit (HalfCheetah-v5: Euler + frictional "
                        "contact + Newton solver, gymnasium/wrappers/vector/dict_info_to_list.py:49-56; Reacher-v5: RK4 + joint limits, "
                        "gymnasium/wrappers/transform_action.py:223-257); free-joint RK4 contact (Ant), PGS (Humanoid), capsule-capsule collisions, "
                        "tendons and fluid forces have no `mujoco` fixture yet (DESIGN.md section 7), so those trajectories are NOT guaranteed to "
                        "equal the reference's within a tolerance.")

    @property
    def dt(self) -> float:
        """MujocoEnv.dt (mujoco_env.py:189-191): model.opt.timestep * frame_skip."""
        return _models.MODELS[self.KIND]()["option"]["timestep"] * self.frame_skip

    def _tendon_values(self, qpos, qvel):
        """data.ten_length / data.ten_velocity of mj_forward at (qpos, qvel): fixed tendons are linear in both."""
        ln = np.stack([sum(c * qpos[:, qa] for qa, _, c in t) for t in self.TENDONS], axis=1)
        vl = np.stack([sum(c * qvel[:, da] for _, da, c in t) for t in self.TENDONS], axis=1)
        return ln, vl

    def _single_spaces(self):
        obs = spaces.Box(low=-np.inf, high=np.inf, shape=(self._obs_size(),), dtype=np.float64)
        low = np.full(self.NU, self.CTRL_LOW, dtype=np.float32)
        high = np.full(self.NU, self.CTRL_HIGH, dtype=np.float32)
        return obs, spaces.Box(low=low, high=high, dtype=np.float32)

    def _parse_reset_options(self, options):
        return None  # MujocoEnv.reset ignores options (mujoco_env.py:172-187)

    def _reset_infos(self, mask):
        """_get_reset_info (half_cheetah_v5.py:277-280, ant_v5.py:423-428, humanoid_v5.py:534-541): positions only."""
        sel = np.ones(self.num_envs, dtype=np.bool_) if mask is None else mask.view(np.bool_).copy()
        qpos = self.get_state()[0]
        infos = {"x_position": np.where(sel, qpos[:, 0], 0.0), "_x_position": sel}
        if self.N_RESET_INFO_KEYS == 3:
            infos.update({"y_position": np.where(sel, qpos[:, 1], 0.0), "_y_position": sel.copy()})
            infos.update(self._reset_tendon_infos(qpos, sel))  # key order of humanoid_v5.py:534-541
            infos.update({"distance_from_origin": np.where(sel, _np_norm(qpos[:, 0], qpos[:, 1]), 0.0), "_distance_from_origin": sel.copy()})
        return infos

    def _reset_tendon_infos(self, state, sel):
        if not self.TENDONS:
            return {}
        ln, vl = self._tendon_values(state[:, :self.NQ], state[:, self.NQ:self.NQ + self.NV])
        return {"tendon_length": np.where(sel[:, None], ln, 0.0), "_tendon_length": sel.copy(),
                "tendon_velocity": np.where(sel[:, None], vl, 0.0), "_tendon_velocity": sel.copy()}


class HalfCheetahVectorEnv(_MujocoVectorEnv):
    KIND = "half_cheetah"
    DEFAULT_MAX_EPISODE_STEPS = 1000
    STOCK_XML = "half_cheetah.xml"
    NQ, NV, NU, NBODY = 9, 9, 6, 8
    CTRL_LOW, CTRL_HIGH = -1.0, 1.0
    INFO_KEYS = ("x_position", "x_velocity", "reward_forward", "reward_ctrl")
    INFO_DTYPES = {"reward_ctrl": np.float32}  # -ctrl_cost_weight * np.sum(np.square(float32 action)): an np.float32 in the reference's info
    N_RESET_INFO_KEYS = 1

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, xml_file: str = "half_cheetah.xml", frame_skip: int = 5,
                 forward_reward_weight: float = 1.0, ctrl_cost_weight: float = 0.1, reset_noise_scale: float = 0.1,
                 exclude_current_positions_from_observation: bool = True, **kwargs):
        self._check_common(xml_file, frame_skip, kwargs)
        self._params = (forward_reward_weight, ctrl_cost_weight, reset_noise_scale, float(bool(exclude_current_positions_from_observation)),
                        float(frame_skip))
        self._exclude = bool(exclude_current_positions_from_observation)
        self.observation_structure = {"skipped_qpos": 1 * self._exclude, "qpos": self.NQ - 1 * self._exclude, "qvel": self.NV}
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _obs_size(self):
        return self.NQ + self.NV - self._exclude

    def _engine_params(self):
        return self._params


class AntVectorEnv(_MujocoVectorEnv):
    KIND = "ant"
    DEFAULT_MAX_EPISODE_STEPS = 1000
    STOCK_XML = "ant.xml"
    NQ, NV, NU, NBODY = 15, 14, 8, 14
    CTRL_LOW, CTRL_HIGH = -1.0, 1.0
    INFO_KEYS = ("x_position", "y_position", "distance_from_origin", "x_velocity", "y_velocity", "reward_forward", "reward_ctrl",
                 "reward_contact", "reward_survive")
    INFO_DTYPES = {"reward_ctrl": np.float32}  # -ctrl_cost_weight * np.sum(np.square(float32 action)): an np.float32 in the reference's info
    N_RESET_INFO_KEYS = 3

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, xml_file: str = "ant.xml", frame_skip: int = 5,
                 forward_reward_weight: float = 1, ctrl_cost_weight: float = 0.5, contact_cost_weight: float = 5e-4,
                 healthy_reward: float = 1.0, main_body: int | str = 1, terminate_when_unhealthy: bool = True,
                 healthy_z_range=(0.2, 1.0), contact_force_range=(-1.0, 1.0), reset_noise_scale: float = 0.1,
                 exclude_current_positions_from_observation: bool = True, include_cfrc_ext_in_observation: bool = True, **kwargs):
        self._check_common(xml_file, frame_skip, kwargs)
        if main_body not in (1, "torso"):
            raise error.Error("gymnasium_amd Ant tracks the torso (main_body=1) only")
        self._exclude, self._cfrc = bool(exclude_current_positions_from_observation), bool(include_cfrc_ext_in_observation)
        self._params = (forward_reward_weight, ctrl_cost_weight, reset_noise_scale, float(self._exclude), float(frame_skip),
                        contact_cost_weight, healthy_reward, float(bool(terminate_when_unhealthy)), healthy_z_range[0], healthy_z_range[1],
                        contact_force_range[0], contact_force_range[1], float(self._cfrc))
        self.observation_structure = {"skipped_qpos": 2 * self._exclude, "qpos": self.NQ - 2 * self._exclude, "qvel": self.NV,
                                      "cfrc_ext": 6 * (self.NBODY - 1) * self._cfrc}
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _obs_size(self):
        return self.NQ + self.NV - 2 * self._exclude + 6 * (self.NBODY - 1) * self._cfrc

    def _engine_params(self):
        return self._params


class HumanoidVectorEnv(_MujocoVectorEnv):
    KIND = "humanoid"
    DEFAULT_MAX_EPISODE_STEPS = 1000
    STOCK_XML = "humanoid.xml"
    NQ, NV, NU, NBODY = 24, 23, 17, 14
    CTRL_LOW, CTRL_HIGH = -0.4, 0.4
    INFO_KEYS = AntVectorEnv.INFO_KEYS
    INFO_DTYPES = {}  # (control cost from data.ctrl, float64: humanoid_v5.py:418)
    N_RESET_INFO_KEYS = 3
    INFO_VECTOR_KEYS = (("tendon_length", 2), ("tendon_velocity", 2))  # data.ten_length / data.ten_velocity (humanoid_v5.py:486-487)
    # <tendon><fixed name="left_hipknee"> -left_hip_y + left_knee, "right_hipknee" likewise (humanoid.xml:91-100): (qpos adr, dof adr, coef)
    TENDONS = (((16, 15, -1.0), (17, 16, 1.0)), ((12, 11, -1.0), (13, 12, 1.0)))

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, xml_file: str = "humanoid.xml", frame_skip: int = 5,
                 forward_reward_weight: float = 1.25, ctrl_cost_weight: float = 0.1, contact_cost_weight: float = 5e-7,
                 contact_cost_range=(-np.inf, 10.0), healthy_reward: float = 5.0, terminate_when_unhealthy: bool = True,
                 healthy_z_range=(1.0, 2.0), reset_noise_scale: float = 1e-2, exclude_current_positions_from_observation: bool = True,
                 include_cinert_in_observation: bool = True, include_cvel_in_observation: bool = True,
                 include_qfrc_actuator_in_observation: bool = True, include_cfrc_ext_in_observation: bool = True, solver: str = "PGS", **kwargs):
        self._check_common(xml_file, frame_skip, kwargs)
        self._set_solver(solver)
        self._exclude = bool(exclude_current_positions_from_observation)
        inc = [bool(include_cinert_in_observation), bool(include_cvel_in_observation), bool(include_qfrc_actuator_in_observation),
               bool(include_cfrc_ext_in_observation)]
        self._inc = inc
        self._params = (forward_reward_weight, ctrl_cost_weight, reset_noise_scale, float(self._exclude), float(frame_skip),
                        contact_cost_weight, healthy_reward, float(bool(terminate_when_unhealthy)), healthy_z_range[0], healthy_z_range[1],
                        contact_cost_range[0], contact_cost_range[1], *[float(x) for x in inc])
        nb1 = self.NBODY - 1
        self.observation_structure = {"skipped_qpos": 2 * self._exclude, "qpos": self.NQ - 2 * self._exclude, "qvel": self.NV,
                                      "cinert": 10 * nb1 * inc[0], "cvel": 6 * nb1 * inc[1], "qfrc_actuator": (self.NV - 6) * inc[2],
                                      "cfrc_ext": 6 * nb1 * inc[3], "ten_length": 0, "ten_velocity": 0}
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _set_solver(self, solver):
        """``solver="PGS"`` (default): what humanoid.xml:8 asks for, `solver="PGS" iterations="50"`.  ``solver="Newton"`` is an explicit
        opt-in deviation: the same convex problem solved to convergence (qacc differs by ~1e-5 relative), a little faster."""
        if str(solver).lower() not in ("pgs", "newton"):
            raise error.Error(f"solver must be 'PGS' (the model's own) or 'Newton', got {solver!r}")
        self.solver = "Newton" if str(solver).lower() == "newton" else "PGS"

    def _engine_options(self):
        from ... import _native

        return _native.CFG_SOLVER_NEWTON if self.solver == "Newton" else 0

    def _obs_size(self):
        nb1, inc = self.NBODY - 1, self._inc
        return self.NQ + self.NV - 2 * self._exclude + 10 * nb1 * inc[0] + 6 * nb1 * inc[1] + (self.NV - 6) * inc[2] + 6 * nb1 * inc[3]

    def _engine_params(self):
        return self._params


class HumanoidStandupVectorEnv(HumanoidVectorEnv):
    """humanoidstandup_v5.py:266-486: the humanoid lying on its back; same spaces and observation as Humanoid-v5, reward for height."""

    KIND = "humanoid_standup"
    STOCK_XML = "humanoidstandup.xml"
    INFO_KEYS = ("x_position", "y_position", "z_distance_from_origin", "reward_linup", "reward_quadctrl", "reward_impact")
    N_RESET_INFO_KEYS = 3

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, xml_file: str = "humanoidstandup.xml", frame_skip: int = 5,
                 uph_cost_weight: float = 1, ctrl_cost_weight: float = 0.1, impact_cost_weight: float = 0.5e-6,
                 impact_cost_range=(-np.inf, 10.0), reset_noise_scale: float = 1e-2, exclude_current_positions_from_observation: bool = True,
                 include_cinert_in_observation: bool = True, include_cvel_in_observation: bool = True,
                 include_qfrc_actuator_in_observation: bool = True, include_cfrc_ext_in_observation: bool = True, solver: str = "PGS", **kwargs):
        self._check_common(xml_file, frame_skip, kwargs)
        self._set_solver(solver)
        self._exclude = bool(exclude_current_positions_from_observation)
        inc = [bool(include_cinert_in_observation), bool(include_cvel_in_observation), bool(include_qfrc_actuator_in_observation),
               bool(include_cfrc_ext_in_observation)]
        self._inc = inc
        self._params = (uph_cost_weight, ctrl_cost_weight, reset_noise_scale, float(self._exclude), float(frame_skip), impact_cost_weight, 0.0, 0.0,
                        0.0, 0.0, impact_cost_range[0], impact_cost_range[1], *[float(x) for x in inc])
        nb1 = self.NBODY - 1
        self.observation_structure = {"skipped_qpos": 2 * self._exclude, "qpos": self.NQ - 2 * self._exclude, "qvel": self.NV,
                                      "cinert": 10 * nb1 * inc[0], "cvel": 6 * nb1 * inc[1], "qfrc_actuator": (self.NV - 6) * inc[2],
                                      "cfrc_ext": 6 * nb1 * inc[3], "ten_length": 0, "ten_velocity": 0}
        _MujocoVectorEnv.__init__(self, num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _reset_infos(self, mask):
        sel = np.ones(self.num_envs, dtype=np.bool_) if mask is None else mask.view(np.bool_).copy()
        qpos = self.get_state()[0]
        infos = {"x_position": np.where(sel, qpos[:, 0], 0.0), "_x_position": sel, "y_position": np.where(sel, qpos[:, 1], 0.0), "_y_position": sel.copy(),
                 "z_distance_from_origin": np.where(sel, qpos[:, 2] - 0.105, 0.0), "_z_distance_from_origin": sel.copy()}
        infos.update(self._reset_tendon_infos(qpos, sel))  # humanoidstandup_v5.py:479-486
        return infos


class _PlanarWalkerVectorEnv(_MujocoVectorEnv):
    """Hopper-v5 / Walker2d-v5: obs = qpos[1:] + clip(qvel, -10, 10); reward = forward + healthy - ctrl (hopper_v5.py:236-343)."""

    DEFAULT_MAX_EPISODE_STEPS = 1000
    CTRL_LOW, CTRL_HIGH = -1.0, 1.0
    INFO_KEYS = ("x_position", "z_distance_from_origin", "x_velocity", "reward_forward", "reward_ctrl", "reward_survive")
    INFO_DTYPES = {"reward_ctrl": np.float32}  # -ctrl_cost_weight * np.sum(np.square(float32 action)): an np.float32 in the reference's info
    N_RESET_INFO_KEYS = 2

    def _init_walker(self, num_envs, max_episode_steps, xml_file, frame_skip, forward_reward_weight, ctrl_cost_weight, healthy_reward,
                     terminate_when_unhealthy, healthy_state_range, healthy_z_range, healthy_angle_range, reset_noise_scale, exclude, kwargs):
        self._check_common(xml_file, frame_skip, kwargs)
        self._exclude = bool(exclude)
        self._params = (forward_reward_weight, ctrl_cost_weight, reset_noise_scale, float(self._exclude), float(frame_skip), 0.0, healthy_reward,
                        float(bool(terminate_when_unhealthy)), healthy_z_range[0], healthy_z_range[1], healthy_angle_range[0],
                        healthy_angle_range[1], healthy_state_range[0], healthy_state_range[1])
        self.observation_structure = {"skipped_qpos": 1 * self._exclude, "qpos": self.NQ - 1 * self._exclude, "qvel": self.NV}
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _obs_size(self):
        return self.NQ + self.NV - self._exclude

    def _engine_params(self):
        return self._params

    def _reset_infos(self, mask):
        sel = np.ones(self.num_envs, dtype=np.bool_) if mask is None else mask.view(np.bool_).copy()
        qpos = self.get_state()[0]
        return {"x_position": np.where(sel, qpos[:, 0], 0.0), "_x_position": sel,
                "z_distance_from_origin": np.where(sel, qpos[:, 1] - self.INIT_Z, 0.0), "_z_distance_from_origin": sel.copy()}


class HopperVectorEnv(_PlanarWalkerVectorEnv):
    KIND = "hopper"
    STOCK_XML = "hopper.xml"
    NQ, NV, NU, NBODY = 6, 6, 3, 5
    INIT_Z = 1.25

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, xml_file: str = "hopper.xml", frame_skip: int = 4,
                 forward_reward_weight: float = 1.0, ctrl_cost_weight: float = 1e-3, healthy_reward: float = 1.0,
                 terminate_when_unhealthy: bool = True, healthy_state_range=(-100.0, 100.0), healthy_z_range=(0.7, float("inf")),
                 healthy_angle_range=(-0.2, 0.2), reset_noise_scale: float = 5e-3, exclude_current_positions_from_observation: bool = True, **kwargs):
        self._init_walker(num_envs, max_episode_steps, xml_file, frame_skip, forward_reward_weight, ctrl_cost_weight, healthy_reward,
                          terminate_when_unhealthy, healthy_state_range, healthy_z_range, healthy_angle_range, reset_noise_scale,
                          exclude_current_positions_from_observation, kwargs)


class Walker2dVectorEnv(_PlanarWalkerVectorEnv):
    KIND = "walker2d"
    STOCK_XML = "walker2d_v5.xml"
    NQ, NV, NU, NBODY = 9, 9, 6, 8
    INIT_Z = 1.25

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, xml_file: str = "walker2d_v5.xml", frame_skip: int = 4,
                 forward_reward_weight: float = 1.0, ctrl_cost_weight: float = 1e-3, healthy_reward: float = 1.0,
                 terminate_when_unhealthy: bool = True, healthy_z_range=(0.8, 2.0), healthy_angle_range=(-1.0, 1.0),
                 reset_noise_scale: float = 5e-3, exclude_current_positions_from_observation: bool = True, **kwargs):
        self._init_walker(num_envs, max_episode_steps, xml_file, frame_skip, forward_reward_weight, ctrl_cost_weight, healthy_reward,
                          terminate_when_unhealthy, (-np.inf, np.inf), healthy_z_range, healthy_angle_range, reset_noise_scale,
                          exclude_current_positions_from_observation, kwargs)


class _PendulumVectorEnv(_MujocoVectorEnv):
    DEFAULT_MAX_EPISODE_STEPS = 1000
    N_RESET_INFO_KEYS = 0

    def _engine_params(self):
        return self._params

    def _reset_infos(self, mask):
        return {}  # _get_reset_info is empty (inverted_pendulum_v5.py:198-199)


class InvertedPendulumVectorEnv(_PendulumVectorEnv):
    """inverted_pendulum_v5.py:100-199: obs = qpos + qvel (float64[4]), action float32[1] in [-3, 3]."""

    KIND = "inverted_pendulum"
    STOCK_XML = "inverted_pendulum.xml"
    NQ, NV, NU, NBODY = 2, 2, 1, 3
    CTRL_LOW, CTRL_HIGH = -3.0, 3.0
    INFO_KEYS = ("reward_survive",)
    INFO_DTYPES = {"reward_survive": np.int64}  # inverted_pendulum_v5.py:188: `reward = int(not terminated)`

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, xml_file: str = "inverted_pendulum.xml", frame_skip: int = 2,
                 reset_noise_scale: float = 0.01, **kwargs):
        self._check_common(xml_file, frame_skip, kwargs)
        self._params = (0.0, 0.0, reset_noise_scale, 0.0, float(frame_skip))
        self.observation_structure = {"qpos": self.NQ, "qvel": self.NV}
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _obs_size(self):
        return self.NQ + self.NV


class InvertedDoublePendulumVectorEnv(_PendulumVectorEnv):
    """inverted_double_pendulum_v5.py:125-246: obs float64[9], action float32[1] in [-1, 1]."""

    KIND = "inverted_double_pendulum"
    STOCK_XML = "inverted_double_pendulum.xml"
    NQ, NV, NU, NBODY = 3, 3, 1, 4
    CTRL_LOW, CTRL_HIGH = -1.0, 1.0
    INFO_KEYS = ("reward_survive", "distance_penalty", "velocity_penalty")

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, xml_file: str = "inverted_double_pendulum.xml", frame_skip: int = 5,
                 healthy_reward: float = 10.0, reset_noise_scale: float = 0.1, **kwargs):
        self._check_common(xml_file, frame_skip, kwargs)
        self._params = (0.0, 0.0, reset_noise_scale, 0.0, float(frame_skip), 0.0, healthy_reward)
        self.observation_structure = {"qpos": 1, "sinqpos": 2, "cosqpos": 2, "qvel": self.NV, "qfrc_constraint": 1}
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _obs_size(self):
        return 1 + 2 * (self.NQ - 1) + self.NV + 1


class ReacherVectorEnv(_PendulumVectorEnv):
    """reacher_v5.py:127-245: obs float64[10] (cos / sin of the arm angles, target, arm velocities, fingertip - target), action
    float32[2] in [-1, 1]; never terminates (TimeLimit 50)."""

    KIND = "reacher"
    DEFAULT_MAX_EPISODE_STEPS = 50
    STOCK_XML = "reacher.xml"
    NQ, NV, NU, NBODY = 4, 4, 2, 5
    CTRL_LOW, CTRL_HIGH = -1.0, 1.0
    INFO_KEYS = ("reward_dist", "reward_ctrl")
    INFO_DTYPES = {"reward_ctrl": np.float32}  # -ctrl_cost_weight * np.sum(np.square(float32 action)): an np.float32 in the reference's info

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, xml_file: str = "reacher.xml", frame_skip: int = 2,
                 reward_dist_weight: float = 1, reward_control_weight: float = 1, **kwargs):
        self._check_common(xml_file, frame_skip, kwargs)
        self._params = (reward_dist_weight, reward_control_weight, 0.0, 0.0, float(frame_skip))
        self.observation_structure = {"cos": 2, "sin": 2, "target": 2, "qvel": 2, "fingertip_dist": 2}
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _obs_size(self):
        return 10


class PusherVectorEnv(_PendulumVectorEnv):
    """pusher_v5.py:165-326: a 7-joint arm pushes a cylinder to a goal on a table (pusher_v5.xml: no gravity, Euler, frictionless
    contacts); obs float64[23], action float32[7] in [-2, 2]; never terminates (TimeLimit 100)."""

    KIND = "pusher"
    DEFAULT_MAX_EPISODE_STEPS = 100
    STOCK_XML = "pusher_v5.xml"
    NQ, NV, NU, NBODY = 11, 11, 7, 13
    CTRL_LOW, CTRL_HIGH = -2.0, 2.0
    INFO_KEYS = ("reward_dist", "reward_ctrl", "reward_near")
    INFO_DTYPES = {"reward_ctrl": np.float32}  # -ctrl_cost_weight * np.sum(np.square(float32 action)): an np.float32 in the reference's info

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, xml_file: str = "pusher_v5.xml", frame_skip: int = 5,
                 reward_near_weight: float = 0.5, reward_dist_weight: float = 1, reward_control_weight: float = 0.1, **kwargs):
        self._check_common(xml_file, frame_skip, kwargs)
        self._params = (reward_near_weight, reward_control_weight, 0.0, 0.0, float(frame_skip), reward_dist_weight)
        self.observation_structure = {"qpos": 7, "qvel": 7, "tips_arm": 3, "object": 3, "goal": 3}
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _obs_size(self):
        return 23


class SwimmerVectorEnv(_MujocoVectorEnv):
    """swimmer_v5.py:153-301: three links in a viscous medium (swimmer.xml: option density 4000, viscosity 0.1, RK4); obs = qpos[2:] + qvel
    (float64[8]), action float32[2] in [-1, 1]; never terminates."""

    KIND = "swimmer"
    DEFAULT_MAX_EPISODE_STEPS = 1000
    STOCK_XML = "swimmer.xml"
    NQ, NV, NU, NBODY = 5, 5, 2, 4
    CTRL_LOW, CTRL_HIGH = -1.0, 1.0
    INFO_KEYS = ("x_position", "y_position", "distance_from_origin", "x_velocity", "y_velocity", "reward_forward", "reward_ctrl")
    INFO_DTYPES = {"reward_ctrl": np.float32}  # -ctrl_cost_weight * np.sum(np.square(float32 action)): an np.float32 in the reference's info
    N_RESET_INFO_KEYS = 3

    def __init__(self, num_envs: int = 1, max_episode_steps: int | None = None, xml_file: str = "swimmer.xml", frame_skip: int = 4,
                 forward_reward_weight: float = 1.0, ctrl_cost_weight: float = 1e-4, reset_noise_scale: float = 0.1,
                 exclude_current_positions_from_observation: bool = True, **kwargs):
        self._check_common(xml_file, frame_skip, kwargs)
        self._exclude = bool(exclude_current_positions_from_observation)
        self._params = (forward_reward_weight, ctrl_cost_weight, reset_noise_scale, float(self._exclude), float(frame_skip))
        self.observation_structure = {"skipped_qpos": 2 * self._exclude, "qpos": self.NQ - 2 * self._exclude, "qvel": self.NV}
        super().__init__(num_envs=num_envs, max_episode_steps=max_episode_steps, **kwargs)

    def _obs_size(self):
        return self.NQ + self.NV - 2 * self._exclude

    def _engine_params(self):
        return self._params


# id -> (creator, max_episode_steps, reward_threshold): gymnasium/envs/__init__.py:246-374
ENV_TABLE = {
    "HalfCheetah-v5": (HalfCheetahVectorEnv, 1000, 4800.0),
    "Ant-v5": (AntVectorEnv, 1000, 6000.0),
    "Humanoid-v5": (HumanoidVectorEnv, 1000, None),
    "Hopper-v5": (HopperVectorEnv, 1000, 3800.0),
    "Walker2d-v5": (Walker2dVectorEnv, 1000, None),
    "InvertedPendulum-v5": (InvertedPendulumVectorEnv, 1000, 950.0),
    "InvertedDoublePendulum-v5": (InvertedDoublePendulumVectorEnv, 1000, 9100.0),
    "Reacher-v5": (ReacherVectorEnv, 50, -3.75),
    "HumanoidStandup-v5": (HumanoidStandupVectorEnv, 1000, None),
    "Swimmer-v5": (SwimmerVectorEnv, 1000, 360.0),
    "Pusher-v5": (PusherVectorEnv, 100, 0.0),
}
