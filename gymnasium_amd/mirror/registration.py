"""register()/make_vec() for ``vector_entry_point`` environments (mirror of gymnasium/envs/registration.py).

Only the plug-in route the engine uses is mirrored: EnvSpec (:72-115), register (:564-638), id parsing
(:25-27,259-280), "module:Env-v0" auto-import (:494-502) and make_vec's VECTOR_ENTRY_POINT branch (:933-988).
"sync"/"async" vectorisation of scalar Python envs is the reference's CPU path and is not provided here.
"""
from __future__ import annotations

import copy
import importlib
import re
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Callable

from . import error, logger
from .vector_env import AutoresetMode

ENV_ID_RE = re.compile(r"^(?:(?P<namespace>[\w:-]+)\/)?(?:(?P<name>[\w:.-]+?))(?:-v(?P<version>\d+))?$")


class VectorizeMode(Enum):
    ASYNC = "async"
    SYNC = "sync"
    VECTOR_ENTRY_POINT = "vector_entry_point"


def parse_env_id(env_id: str):
    match = ENV_ID_RE.fullmatch(env_id)
    if not match:
        raise error.Error(f"Malformed environment ID: {env_id}. (Currently all IDs must be of the form [namespace/](env-name)-v(version). (namespace is optional))")
    ns, name, version = match.group("namespace", "name", "version")
    return ns, name, (int(version) if version is not None else None)


@dataclass
class EnvSpec:
    id: str
    entry_point: Callable | str | None = None
    reward_threshold: float | None = None
    nondeterministic: bool = False
    max_episode_steps: int | None = None
    order_enforce: bool = True
    disable_env_checker: bool = False
    kwargs: dict = field(default_factory=dict)
    additional_wrappers: tuple = field(default_factory=tuple)
    vector_entry_point: Callable | str | None = None
    namespace: str | None = field(init=False)
    name: str = field(init=False)
    version: int | None = field(init=False)

    def __post_init__(self):
        self.namespace, self.name, self.version = parse_env_id(self.id)

    def make(self, **kwargs):
        raise error.Error("scalar env creation is not part of the mirrored path; use make_vec")


registry: dict[str, EnvSpec] = {}


def register(id: str, entry_point=None, reward_threshold=None, nondeterministic=False, max_episode_steps=None,
             order_enforce=True, disable_env_checker=False, additional_wrappers=(), vector_entry_point=None,
             kwargs=None):
    assert entry_point is not None or vector_entry_point is not None, "Either `entry_point` or `vector_entry_point` (or both) must be provided"
    new_spec = EnvSpec(id=id, entry_point=entry_point, reward_threshold=reward_threshold, nondeterministic=nondeterministic,
                       max_episode_steps=max_episode_steps, order_enforce=order_enforce, disable_env_checker=disable_env_checker,
                       kwargs=dict(kwargs or {}), additional_wrappers=tuple(additional_wrappers), vector_entry_point=vector_entry_point)
    if new_spec.id in registry:
        logger.warn(f"Overriding environment {new_spec.id} already in registry.")
    registry[new_spec.id] = new_spec


def load_env_creator(name: str):
    mod_name, attr_name = name.split(":")
    return getattr(importlib.import_module(mod_name), attr_name)


def _find_spec(env_id: str) -> EnvSpec:
    module, env_name = (None, env_id) if ":" not in env_id else env_id.split(":")
    if module is not None:
        try:
            importlib.import_module(module)
        except ModuleNotFoundError as e:
            raise ModuleNotFoundError(f"{e}. Environment registration via importing a module failed. Check whether '{module}' contains env registration and can be imported.") from e
    env_spec = registry.get(env_name)
    if env_spec is None:
        ns, name, version = parse_env_id(env_name)
        known = [s for s in registry.values() if s.namespace == ns and s.name == name]
        if not known:
            raise error.NameNotFound(f"Environment `{name}` doesn't exist{'' if ns is None else f' in namespace {ns}'}.")
        raise error.VersionNotFound(f"Environment version `v{version}` for environment `{name}` doesn't exist. It provides versioned environments: [ {', '.join(f'`v{s.version}`' for s in known)} ].")
    return env_spec


def spec(env_id: str) -> EnvSpec:
    return _find_spec(env_id)


def make_vec(id, num_envs: int = 1, vectorization_mode=None, vector_kwargs: dict[str, Any] | None = None, wrappers=None, **kwargs):
    vector_kwargs = {} if vector_kwargs is None else vector_kwargs
    wrappers = [] if wrappers is None else wrappers
    if isinstance(id, EnvSpec):
        env_spec = id
    elif isinstance(id, str):
        env_spec = _find_spec(id)
    else:
        raise error.Error(f"Invalid id type: {type(id)}. Expected `str` or `EnvSpec`")
    env_spec = copy.deepcopy(env_spec)
    env_spec_kwargs = env_spec.kwargs
    env_spec.kwargs = dict()
    num_envs = env_spec_kwargs.pop("num_envs", num_envs)
    vectorization_mode = env_spec_kwargs.pop("vectorization_mode", vectorization_mode)
    vector_kwargs = env_spec_kwargs.pop("vector_kwargs", vector_kwargs)
    wrappers = env_spec_kwargs.pop("wrappers", wrappers)
    env_spec_kwargs.update(kwargs)

    if vectorization_mode is None:
        vectorization_mode = VectorizeMode.VECTOR_ENTRY_POINT if env_spec.vector_entry_point is not None else VectorizeMode.SYNC
    else:
        try:
            vectorization_mode = VectorizeMode(vectorization_mode)
        except ValueError as e:
            raise ValueError(f"Invalid vectorization mode: {vectorization_mode!r}, valid modes: {[m.value for m in VectorizeMode]}") from e

    if vectorization_mode != VectorizeMode.VECTOR_ENTRY_POINT:
        raise error.Error(f"vectorization_mode={vectorization_mode.value!r} wraps scalar Python environments on the CPU; that is the reference's "
                          "own path (gymnasium.vector.SyncVectorEnv/AsyncVectorEnv) and is not provided by gymnasium_amd. Install gymnasium for it.")
    if len(vector_kwargs) > 0:
        raise error.Error(f"Custom vector environment can be passed arguments only through kwargs and `vector_kwargs` is not empty ({vector_kwargs})")
    if len(wrappers) > 0:
        raise error.Error(f"Cannot use `vector_entry_point` vectorization mode with the wrappers argument ({wrappers}).")
    if len(env_spec.additional_wrappers) > 0:
        raise error.Error(f"Cannot use `vector_entry_point` vectorization mode with the additional_wrappers parameter in spec being not empty ({env_spec.additional_wrappers}).")
    entry_point = env_spec.vector_entry_point
    if entry_point is None:
        raise error.Error(f"Cannot create vectorized environment for {id} because it doesn't have a vector entry point defined.")
    env_creator = entry_point if callable(entry_point) else load_env_creator(entry_point)
    if env_spec.max_episode_steps is not None and "max_episode_steps" not in env_spec_kwargs:
        env_spec_kwargs["max_episode_steps"] = env_spec.max_episode_steps
    env = env_creator(num_envs=num_envs, **env_spec_kwargs)

    copied = copy.deepcopy(env_spec)
    copied.kwargs = env_spec_kwargs.copy()
    if num_envs != 1:
        copied.kwargs["num_envs"] = num_envs
    copied.kwargs["vectorization_mode"] = vectorization_mode.value
    env.unwrapped.spec = copied
    if "autoreset_mode" not in env.metadata:
        logger.warn(f"The VectorEnv ({env}) is missing AutoresetMode metadata, metadata={env.metadata}")
    elif not isinstance(env.metadata["autoreset_mode"], AutoresetMode):
        logger.warn(f"The VectorEnv ({env}) metadata['autoreset_mode'] is not an instance of AutoresetMode, {type(env.metadata['autoreset_mode'])}.")
    return env
