"""Device-side counterparts of gymnasium.wrappers.vector for HipVectorEnv."""
from .vector import ClipReward, NormalizeObservation, NumpyToTorch, RecordEpisodeStatistics, NormalizeReward, RunningMeanStd, VectorWrapper  # noqa: F401
