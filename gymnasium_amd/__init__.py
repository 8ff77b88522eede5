"""gymnasium_amd -- MI355X-native lockstep vector environments behind gymnasium's ``make_vec`` plug-in boundary.

    import gymnasium_amd as gym_amd
    envs = gym_amd.make_vec("MI355X/CartPole-v1", num_envs=65536)          # NumPy in / NumPy out
    envs = gym_amd.make_vec("MI355X/CartPole-v1", num_envs=65536, output="torch")  # tensors stay in HBM

With Farama gymnasium installed the same ids are registered in ITS registry
(``gymnasium.make_vec("MI355X/CartPole-v1", ...)``, or ``gymnasium.make_vec("gymnasium_amd:MI355X/CartPole-v1")``),
see INTEGRATION.md.
"""
from . import gym_api
from .envs import ENV_TABLE
from .envs.mujoco.envs import ENV_TABLE as _MUJOCO_TABLE
from .gym_api import AutoresetMode, VectorEnv, make_vec, register, registry, spaces  # noqa: F401
from .vector import HipVectorEnv  # noqa: F401

__version__ = "0.1.0"
NAMESPACE = "MI355X"
MUJOCO_IDS = frozenset(_MUJOCO_TABLE)


def register_envs(override_stock_ids: bool = False) -> None:
    """Register ``MI355X/<id>`` for every supported id through ``register(id, vector_entry_point=...)``
    (envs/registration.py:564-638).  ``override_stock_ids=True`` additionally attaches the engine to the stock ids
    (``CartPole-v1`` ...), so that plain ``make_vec("CartPole-v1", n)`` picks it up (a spec with a
    ``vector_entry_point`` makes that the default mode, registration.py:887-891)."""
    for env_id, entry in ENV_TABLE.items():
        creator, max_steps, threshold = entry[:3]
        kw = dict(entry[3]) if len(entry) > 3 else {}
        name = f"{NAMESPACE}/{env_id}"
        if name not in registry:
            register(id=name, vector_entry_point=creator, max_episode_steps=max_steps, reward_threshold=threshold, kwargs=kw)
        if override_stock_ids or not gym_api.HAVE_GYMNASIUM:
            if env_id in registry:
                if override_stock_ids and env_id in MUJOCO_IDS:
                    # the stock id keeps Farama's `mujoco`-backed env: ours is not pinned against it
                    gym_api.logger.warn(f"{env_id}: not overriding the stock id -- the MI355X restatement of MuJoCo is parity-unpinned; use {name}")
                elif override_stock_ids:
                    registry[env_id].vector_entry_point = creator
            else:
                register(id=env_id, vector_entry_point=creator, max_episode_steps=max_steps, reward_threshold=threshold, kwargs=kw)


register_envs()
