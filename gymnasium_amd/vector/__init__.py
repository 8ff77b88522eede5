from .hip_vector_env import HipVectorEnv  # noqa: F401
