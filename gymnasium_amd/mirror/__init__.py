"""Stand-alone mirror of the slice of the gymnasium API that the hot path touches.

The engine is a plug-in for Farama gymnasium (``register(..., vector_entry_point=...)`` + ``make_vec``).  When
gymnasium is importable, :mod:`gymnasium_amd.gym_api` re-exports the real classes and this package is unused.
When it is not (e.g. the bare MI355X box), these classes provide the same names, argument meaning and error
behaviour for exactly the pieces the path needs -- nothing else of gymnasium is rebuilt:

  error.py         gymnasium/error.py (the exception types raised on this path)
  logger.py        gymnasium/logger.py:17-47 (warn)
  seeding.py       gymnasium/utils/seeding.py:10-42 (np_random)
  spaces.py        gymnasium/spaces/{space,box,discrete,multi_discrete}.py (sample/contains/seed)
                   + gymnasium/vector/utils/space_utils.py:51-100 (batch_space for Box/Discrete)
  vector_env.py    gymnasium/vector/vector_env.py:34-351 (AutoresetMode, VectorEnv)
  registration.py  gymnasium/envs/registration.py:72-115,564-638,833-988 (EnvSpec, register, make_vec)
"""
from . import error, logger, seeding, spaces  # noqa: F401
from .registration import EnvSpec, VectorizeMode, make_vec, register, registry, spec  # noqa: F401
from .vector_env import AutoresetMode, VectorEnv  # noqa: F401
