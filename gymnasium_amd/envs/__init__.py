from .classic_control import (  # noqa: F401
    AcrobotVectorEnv,
    CartPoleVectorEnv,
    MountainCarContinuousVectorEnv,
    MountainCarVectorEnv,
    PendulumVectorEnv,
)
from .classic_control import ENV_TABLE as _CLASSIC
from .mujoco.envs import (  # noqa: F401
    AntVectorEnv,
    HalfCheetahVectorEnv,
    HopperVectorEnv,
    HumanoidStandupVectorEnv,
    HumanoidVectorEnv,
    InvertedDoublePendulumVectorEnv,
    InvertedPendulumVectorEnv,
    PusherVectorEnv,
    ReacherVectorEnv,
    SwimmerVectorEnv,
    Walker2dVectorEnv,
)
from .mujoco.envs import ENV_TABLE as _MUJOCO

from .toy_text import BlackjackVectorEnv, CliffWalkingVectorEnv, FrozenLakeVectorEnv, TaxiVectorEnv  # noqa: F401
from .toy_text import ENV_TABLE as _TOY

ENV_TABLE = {**_CLASSIC, **_MUJOCO, **_TOY}
