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


# ---- make_vec: the plug-in route only -------------------------------------------------------------------------------------------------------
# Written from the contract of SURVEY.md section 8(b), not from the reference's function body:
#   * `id` is a registered id (optionally "module:id", which imports the module first) or an EnvSpec;
#   * the four vectorisation arguments may also live in the spec's own kwargs (a spec recorded by an earlier make_vec carries them) -- the spec wins;
#   * every remaining keyword goes to the creator, on top of the spec's kwargs; the spec's max_episode_steps is the creator's default;
#   * a vector_entry_point creator takes neither `vector_kwargs` nor `wrappers` nor spec-level additional wrappers;
#   * the returned env remembers how it was made: `env.unwrapped.spec` = the spec with the creator's kwargs plus num_envs (when not 1) and the mode,
#     so that `make_vec(env.spec)` builds the same env again;
#   * the env is expected to advertise its AutoresetMode in `metadata` (a warning otherwise).
_VECTOR_ARGUMENTS = ("num_envs", "vectorization_mode", "vector_kwargs", "wrappers")


def _as_mode(value, has_vector_entry_point: bool) -> VectorizeMode:
    if value is None:
        return VectorizeMode.VECTOR_ENTRY_POINT if has_vector_entry_point else VectorizeMode.SYNC
    if isinstance(value, VectorizeMode):
        return value
    for mode in VectorizeMode:
        if mode.value == value:
            return mode
    raise ValueError(f"Invalid vectorization mode: {value!r}, valid modes: {[m.value for m in VectorizeMode]}")


def _plugin_route_only(spec_: EnvSpec, shown_id, mode: VectorizeMode, vector_kwargs, wrappers):
    """Everything that cannot be honoured by a vector_entry_point creator is refused before the creator runs."""
    if mode is not VectorizeMode.VECTOR_ENTRY_POINT:
        raise error.Error(f"vectorization_mode={mode.value!r} wraps scalar Python environments on the CPU; that is the reference's own path "
                          "(gymnasium.vector.SyncVectorEnv/AsyncVectorEnv) and is not provided by gymnasium_amd. Install gymnasium for it.")
    problems = {
        "`vector_kwargs`": vector_kwargs,  # the creator is configured through plain keyword arguments
        "the `wrappers` argument": wrappers,
        "the spec's `additional_wrappers`": spec_.additional_wrappers,
    }
    for what, value in problems.items():
        if value:
            raise error.Error(f"A `vector_entry_point` environment cannot be combined with {what} (got {value}): pass constructor arguments as keywords of make_vec.")
    if spec_.vector_entry_point is None:
        raise error.Error(f"{shown_id} has no `vector_entry_point`: there is nothing for vectorization_mode='vector_entry_point' to call.")


def make_vec(id, num_envs: int = 1, vectorization_mode=None, vector_kwargs: dict[str, Any] | None = None, wrappers=None, **kwargs):
    if isinstance(id, EnvSpec):
        found = id
    elif isinstance(id, str):
        found = _find_spec(id)
    else:
        raise error.Error(f"Invalid id type: {type(id)}. Expected `str` or `EnvSpec`")
    recorded = copy.deepcopy(found)  # the caller's / the registry's spec is never modified
    creator_kwargs = dict(recorded.kwargs)
    given = {"num_envs": num_envs, "vectorization_mode": vectorization_mode, "vector_kwargs": vector_kwargs or {}, "wrappers": wrappers or []}
    for name in _VECTOR_ARGUMENTS:  # vectorisation arguments stored in the spec take precedence over the call's
        if name in creator_kwargs:
            given[name] = creator_kwargs.pop(name)
    creator_kwargs.update(kwargs)
    mode = _as_mode(given["vectorization_mode"], recorded.vector_entry_point is not None)
    _plugin_route_only(recorded, id, mode, given["vector_kwargs"], given["wrappers"])

    creator = recorded.vector_entry_point if callable(recorded.vector_entry_point) else load_env_creator(recorded.vector_entry_point)
    if recorded.max_episode_steps is not None:
        creator_kwargs.setdefault("max_episode_steps", recorded.max_episode_steps)
    env = creator(num_envs=given["num_envs"], **creator_kwargs)

    recorded.kwargs = dict(creator_kwargs, vectorization_mode=mode.value)
    if given["num_envs"] != 1:
        recorded.kwargs["num_envs"] = given["num_envs"]
    env.unwrapped.spec = recorded
    advertised = env.metadata.get("autoreset_mode", None)
    if not isinstance(advertised, AutoresetMode):
        logger.warn(f"The VectorEnv ({env}) does not advertise an AutoresetMode in metadata['autoreset_mode'] (found {advertised!r}, metadata={env.metadata})")
    return env
