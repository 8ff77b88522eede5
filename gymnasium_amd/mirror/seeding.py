"""np_random(seed) -> (Generator(PCG64(SeedSequence(seed))), seed)  (mirror of gymnasium/utils/seeding.py:10-42)."""
import numpy as np

from . import error


def np_random(seed=None):
    if seed is not None and not (isinstance(seed, int) and 0 <= seed):
        if isinstance(seed, int) is False:
            raise error.Error(f"Seed must be a python integer, actual type: {type(seed)}")
        raise error.Error(f"Seed must be greater or equal to zero, actual value: {seed}")
    seed_seq = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(seed_seq)), seed_seq.entropy


RNG = RandomNumberGenerator = np.random.Generator
