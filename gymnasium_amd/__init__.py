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
from .gym_api import AutoresetMode, VectorEnv, register, registry, spaces  # noqa: F401
from .vector import HipVectorEnv  # noqa: F401

from .envs.classic_control import StockCartPoleVectorEnv

__version__ = "0.1.0"
STOCK_CREATORS = {"CartPole-v1": StockCartPoleVectorEnv}
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
                    # CartPole-v1 already HAS a vector_entry_point in gymnasium (the NumPy CartPoleVectorEnv: one shared generator, float32
                    # rewards): the stock id gets the engine in THAT class's semantics, so no seeded trajectory of a make_vec("CartPole-v1") user changes
                    registry[env_id].vector_entry_point = STOCK_CREATORS.get(env_id, creator)
            else:
                register(id=env_id, vector_entry_point=creator, max_episode_steps=max_steps, reward_threshold=threshold, kwargs=kw)


def _resolve_id(env_id):
    """``[module:][namespace/]name-vK`` -> the id of this package's registration.  A bare stock id (``CartPole-v1``) maps to
    ``MI355X/CartPole-v1``; an id that names another namespace is refused (this function never hands out somebody else's env)."""
    if not isinstance(env_id, str):
        return env_id
    module, _, name = env_id.rpartition(":")
    ns, _, bare = name.rpartition("/")
    if ns not in ("", NAMESPACE):
        raise gym_api.error.Error(f"gymnasium_amd.make_vec creates MI355X engines only: `{env_id}` names the namespace `{ns}`. "
                                  f"Use `{NAMESPACE}/{bare}` (or the bare id), or gymnasium.make_vec for other environments.")
    return (module + ":" if module else "") + f"{NAMESPACE}/{bare}"


def make_vec(id, num_envs: int = 1, vectorization_mode=None, vector_kwargs=None, wrappers=None, **kwargs):
    """``gymnasium.make_vec`` (envs/registration.py:829-988) restricted to THIS package's engines.

    The id always resolves into the ``MI355X/`` namespace -- ``make_vec("CartPole-v1", n)`` IS
    ``make_vec("MI355X/CartPole-v1", n)`` -- whether or not Farama gymnasium is installed, so a stock id can never come
    back as the reference's CPU implementation (there is no CPU fallback in this package; tests/test_abi.py).  The only
    vectorisation mode is ``"vector_entry_point"``; ``"sync"`` / ``"async"`` wrap scalar Python envs on the CPU and are
    gymnasium's own (``gymnasium.make_vec(id, n, "sync")``).  Everything else -- kwargs forwarding, ``max_episode_steps``
    from the spec, ``env.spec`` -- is the registry's ``make_vec`` (gymnasium's when it is importable, the mirror otherwise).
    """
    if isinstance(id, gym_api.EnvSpec):
        if id.namespace != NAMESPACE:
            raise gym_api.error.Error(f"gymnasium_amd.make_vec creates MI355X engines only, got the spec of `{id.id}`")
    else:
        id = _resolve_id(id)
    if vectorization_mode is not None:
        mode = vectorization_mode.value if isinstance(vectorization_mode, gym_api.VectorizeMode) else vectorization_mode
        if mode != gym_api.VectorizeMode.VECTOR_ENTRY_POINT.value:
            if mode not in [m.value for m in gym_api.VectorizeMode]:
                raise ValueError(f"Invalid vectorization mode: {vectorization_mode!r}, valid modes: {[m.value for m in gym_api.VectorizeMode]}")
            raise gym_api.error.Error(f"vectorization_mode={mode!r} wraps scalar Python environments on the CPU; that is gymnasium's own path "
                                      "(gymnasium.vector.SyncVectorEnv / AsyncVectorEnv) and is not provided by gymnasium_amd.")
    return gym_api.make_vec(id, num_envs=num_envs, vectorization_mode=vectorization_mode, vector_kwargs=vector_kwargs, wrappers=wrappers, **kwargs)


register_envs()
