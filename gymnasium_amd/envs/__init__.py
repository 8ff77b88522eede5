from .classic_control import (  # noqa: F401
    ENV_TABLE,
    AcrobotVectorEnv,
    CartPoleVectorEnv,
    MountainCarContinuousVectorEnv,
    MountainCarVectorEnv,
    PendulumVectorEnv,
)
