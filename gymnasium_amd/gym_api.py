"""The gymnasium API objects the engine plugs into.

If Farama gymnasium is installed, these ARE its classes (so ``isinstance(env, gymnasium.vector.VectorEnv)`` holds,
``gymnasium.make_vec("MI355X/CartPole-v1")`` works and every gymnasium wrapper composes).  Otherwise they come
from :mod:`gymnasium_amd.mirror`, a from-scratch mirror of the same slice (same names / arguments / errors).
Set ``GYMNASIUM_AMD_FORCE_MIRROR=1`` to use the mirror even when gymnasium is importable.
"""
import os

HAVE_GYMNASIUM = False
if os.environ.get("GYMNASIUM_AMD_FORCE_MIRROR", "0") != "1":
    try:
        import gymnasium as _gym  # noqa: F401

        HAVE_GYMNASIUM = True
    except ImportError:
        HAVE_GYMNASIUM = False

if HAVE_GYMNASIUM:
    from gymnasium import error, logger, spaces  # noqa: F401
    from gymnasium.envs.registration import EnvSpec, VectorizeMode, make_vec, register, registry, spec  # noqa: F401
    from gymnasium.utils import seeding  # noqa: F401
    from gymnasium.vector import AutoresetMode, VectorEnv  # noqa: F401
    from gymnasium.vector.utils import batch_space  # noqa: F401
else:
    from .mirror import error, logger, seeding, spaces  # noqa: F401
    from .mirror.registration import EnvSpec, VectorizeMode, make_vec, register, registry, spec  # noqa: F401
    from .mirror.spaces import batch_space  # noqa: F401
    from .mirror.vector_env import AutoresetMode, VectorEnv  # noqa: F401
