from .classic_control import (  # noqa: F401
    AcrobotVectorEnv,
    CartPoleVectorEnv,
    MountainCarContinuousVectorEnv,
    MountainCarVectorEnv,
    PendulumVectorEnv,
)
from .classic_control import ENV_TABLE as _CLASSIC
from .mujoco.envs import AntVectorEnv, HalfCheetahVectorEnv, HumanoidVectorEnv  # noqa: F401
from .mujoco.envs import ENV_TABLE as _MUJOCO

ENV_TABLE = {**_CLASSIC, **_MUJOCO}
