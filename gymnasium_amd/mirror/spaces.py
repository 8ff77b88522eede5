"""Box / Discrete / MultiDiscrete and batch_space -- the three spaces the classic-control path uses.

Mirror of gymnasium/spaces/space.py:20-160, box.py:52-560, discrete.py:16-260, multi_discrete.py:18-330 and
gymnasium/vector/utils/space_utils.py:51-100, restricted to unmasked sampling.  ``sample()`` draws the same
NumPy generator calls in the same order as the reference, so a seeded space yields identical actions
(pinned by tests/golden/action_samples.npz).
"""
from __future__ import annotations

from copy import deepcopy

import numpy as np

from . import error, seeding


class Space:
    """Base class: shape, dtype and a lazily created NumPy generator (space.py:20-160)."""

    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._np_random = None
        if seed is not None:
            if isinstance(seed, np.random.Generator):
                self._np_random = seed
            else:
                self.seed(seed)

    @property
    def np_random(self) -> np.random.Generator:
        if self._np_random is None:
            self.seed()
        return self._np_random

    @property
    def shape(self):
        return self._shape

    @property
    def is_np_flattenable(self):
        return True

    def seed(self, seed=None):
        self._np_random, np_random_seed = seeding.np_random(seed)
        return np_random_seed

    def sample(self, mask=None, probability=None):
        raise NotImplementedError

    def contains(self, x) -> bool:
        raise NotImplementedError

    def __contains__(self, x) -> bool:
        return self.contains(x)


def _as_bound(value, shape, dtype, name):
    if np.isscalar(value):
        return np.full(shape, value, dtype=dtype)
    arr = np.asarray(value)
    if shape is not None and arr.shape != tuple(shape):
        raise ValueError(f"Box {name}.shape and shape are expected to match, actual {name}.shape={arr.shape}, shape={shape}")
    return arr.astype(dtype)


class Box(Space):
    """Closed box in R^n (box.py:52-560)."""

    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        if dtype is None:
            raise ValueError("Box dtype must be explicitly provided, cannot be None.")
        dtype = np.dtype(dtype)
        if shape is not None:
            shape = tuple(int(d) for d in shape)
        elif isinstance(low, np.ndarray):
            shape = low.shape
        elif isinstance(high, np.ndarray):
            shape = high.shape
        elif np.isscalar(low) and np.isscalar(high):
            shape = (1,)
        else:
            raise ValueError(f"Box shape is not specified, therefore inferred from low and high, low={low}, high={high}")
        self.low = _as_bound(low, shape, dtype, "low")
        self.high = _as_bound(high, shape, dtype, "high")
        if np.any(self.low > self.high):
            raise ValueError(f"Box all low values must be less than or equal to high (some values break this), low={self.low}, high={self.high}")
        self.bounded_below = -np.inf < self.low
        self.bounded_above = np.inf > self.high
        super().__init__(shape, dtype, seed)

    def is_bounded(self, manner="both"):
        below, above = bool(np.all(self.bounded_below)), bool(np.all(self.bounded_above))
        if manner == "both":
            return below and above
        if manner == "below":
            return below
        if manner == "above":
            return above
        raise ValueError(f"manner is not in {{'below', 'above', 'both'}}, actual value: {manner}")

    def sample(self, mask=None, probability=None):
        if mask is not None:
            raise error.Error(f"Box.sample cannot be provided a mask, actual value: {mask}")
        if probability is not None:
            raise error.Error(f"Box.sample cannot be provided a probability mask, actual value: {probability}")
        high = self.high if self.dtype.kind == "f" else self.high.astype("int64") + 1
        out = np.empty(self.shape)
        unbounded = ~self.bounded_below & ~self.bounded_above
        upp_bounded = ~self.bounded_below & self.bounded_above
        low_bounded = self.bounded_below & ~self.bounded_above
        bounded = self.bounded_below & self.bounded_above
        # generator calls in the reference's order (box.py:443-465); empty draws consume nothing
        out[unbounded] = self.np_random.normal(size=unbounded[unbounded].shape)
        out[low_bounded] = self.np_random.exponential(size=low_bounded[low_bounded].shape) + self.low[low_bounded]
        out[upp_bounded] = -self.np_random.exponential(size=upp_bounded[upp_bounded].shape) + high[upp_bounded]
        out[bounded] = self.np_random.uniform(low=self.low[bounded], high=high[bounded], size=bounded[bounded].shape)
        if self.dtype.kind in ("i", "u", "b"):
            out = np.floor(out)
        if np.issubdtype(self.dtype, np.integer):
            info = np.iinfo(self.dtype)
            lo, hi = info.min, info.max
            if self.dtype == np.int64:
                lo, hi = lo + 2, hi - 2
            out = out.clip(min=lo, max=hi)
        out = out.astype(self.dtype)
        if self.dtype == np.int64:
            out = out.clip(min=self.low, max=self.high)
        return out

    def contains(self, x) -> bool:
        if not isinstance(x, np.ndarray):
            try:
                x = np.asarray(x, dtype=self.dtype)
            except (ValueError, TypeError):
                return False
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape and np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape and self.dtype == other.dtype
                and np.allclose(self.low, other.low) and np.allclose(self.high, other.high))


class Discrete(Space):
    """{start, ..., start + n - 1} (discrete.py:16-260)."""

    def __init__(self, n, seed=None, start=0, dtype=np.int64):
        if not np.issubdtype(type(n), np.integer):
            raise TypeError(f"Expects `n` to be an integer, actual dtype: {type(n)}")
        if n <= 0:
            raise ValueError("n (counts) have to be positive")
        if not np.issubdtype(type(start), np.integer):
            raise TypeError(f"Expects `start` to be an integer, actual type: {type(start)}")
        dtype = np.dtype(dtype)
        self.n = dtype.type(n)
        self.start = dtype.type(start)
        super().__init__((), dtype, seed)

    def sample(self, mask=None, probability=None):
        if mask is not None or probability is not None:
            raise error.Error("masked sampling is outside the mirrored path; install gymnasium for it")
        return self.start + self.np_random.integers(self.n, dtype=self.dtype.type)

    def contains(self, x) -> bool:
        if isinstance(x, int):
            as_int = np.int64(x)
        elif isinstance(x, (np.generic, np.ndarray)) and np.issubdtype(x.dtype, np.integer) and x.shape == ():
            as_int = np.int64(x)
        else:
            return False
        return bool(self.start <= as_int < self.start + self.n)

    def __repr__(self):
        return f"Discrete({self.n}, start={self.start})" if self.start != 0 else f"Discrete({self.n})"

    def __eq__(self, other):
        return isinstance(other, Discrete) and self.n == other.n and self.start == other.start


class MultiDiscrete(Space):
    """Cartesian product of Discrete spaces (multi_discrete.py:18-330)."""

    def __init__(self, nvec, dtype=np.int64, seed=None, start=None):
        dtype = np.dtype(dtype)
        self.nvec = np.array(nvec, dtype=dtype, copy=True)
        if start is not None:
            self.start = np.array(start, dtype=dtype, copy=True)
        else:
            self.start = np.zeros(self.nvec.shape, dtype=dtype)
        assert self.start.shape == self.nvec.shape, "start and nvec (counts) should have the same shape"
        assert (self.nvec > 0).all(), "nvec (counts) have to be positive"
        super().__init__(self.nvec.shape, dtype, seed)

    def sample(self, mask=None, probability=None):
        if mask is not None or probability is not None:
            raise error.Error("masked sampling is outside the mirrored path; install gymnasium for it")
        return (self.np_random.random(self.nvec.shape) * self.nvec).astype(self.dtype) + self.start

    def contains(self, x) -> bool:
        if isinstance(x, (list, tuple)):
            x = np.array(x)
        return bool(isinstance(x, np.ndarray) and x.shape == self.shape and np.issubdtype(x.dtype, np.integer)
                    and np.all(self.start <= x) and np.all(x - self.start < self.nvec))

    def __repr__(self):
        if np.any(self.start != 0):
            return f"MultiDiscrete({self.nvec}, start={self.start})"
        return f"MultiDiscrete({self.nvec})"

    def __eq__(self, other):
        return (isinstance(other, MultiDiscrete) and self.dtype == other.dtype and self.shape == other.shape
                and np.all(self.nvec == other.nvec) and np.all(self.start == other.start))


class Tuple(Space):
    """A product of spaces (spaces/tuple.py:17-150): samples are tuples, one element per sub-space."""

    def __init__(self, spaces, seed=None):
        self.spaces = tuple(spaces)
        for sp in self.spaces:
            assert isinstance(sp, Space), f"{sp} does not inherit from `gymnasium.Space`. Actual Type: {type(sp)}"
        super().__init__(None, None, None)
        if seed is not None:
            self.seed(seed)

    def seed(self, seed=None):
        """spaces/tuple.py:58-97: None -> every sub-space seeds itself; int -> sub-seeds drawn from one generator."""
        if seed is None:
            return tuple(sp.seed(None) for sp in self.spaces)
        if isinstance(seed, (int, np.integer)):
            super().seed(int(seed))
            subseeds = self.np_random.integers(np.iinfo(np.int32).max, size=len(self.spaces))
            return tuple(sp.seed(int(ss)) for sp, ss in zip(self.spaces, subseeds))
        return tuple(sp.seed(ss) for sp, ss in zip(self.spaces, seed))

    def sample(self, mask=None, probability=None):
        return tuple(sp.sample() for sp in self.spaces)

    def contains(self, x) -> bool:
        if isinstance(x, (list, np.ndarray)):
            x = tuple(x)
        return isinstance(x, tuple) and len(x) == len(self.spaces) and all(sp.contains(p) for sp, p in zip(self.spaces, x))

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]

    def __repr__(self):
        return "Tuple(" + ", ".join(str(sp) for sp in self.spaces) + ")"

    def __eq__(self, other):
        return isinstance(other, Tuple) and self.spaces == other.spaces


def batch_space(space, n=1):
    """n independent copies of ``space`` as one batched space (space_utils.py:51-100; Box, Discrete and Tuple only)."""
    if isinstance(space, Tuple):
        return Tuple(tuple(batch_space(sp, n) for sp in space.spaces), seed=deepcopy(space._np_random))
    if isinstance(space, Box):
        repeats = tuple([n] + [1] * space.low.ndim)
        return Box(low=np.tile(space.low, repeats), high=np.tile(space.high, repeats), dtype=space.dtype,
                   seed=deepcopy(space.np_random))
    if isinstance(space, Discrete):
        return MultiDiscrete(np.full((n,), space.n, dtype=space.dtype), dtype=space.dtype, seed=deepcopy(space.np_random),
                             start=np.full((n,), space.start, dtype=space.dtype))
    raise TypeError(f"The space provided to `batch_space` is not a supported Space instance, type: {type(space)}, {space}")
