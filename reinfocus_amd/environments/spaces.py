"""gymnasium spaces if gymnasium is installed, else a minimal local stand-in.

The reference builds its observation/action spaces from gymnasium (~=0.29).  This image
has no gymnasium, so the few pieces the hot path's callers touch (Box, Discrete,
batch_space) are provided locally with the same attributes (low, high, shape, dtype, n).
"""

import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium import spaces as _gym_spaces
    from gymnasium.vector.utils import batch_space as _gym_batch_space

    Box = _gym_spaces.Box
    Discrete = _gym_spaces.Discrete
    MultiDiscrete = _gym_spaces.MultiDiscrete
    batch_space = _gym_batch_space
    HAVE_GYMNASIUM = True
except ImportError:
    HAVE_GYMNASIUM = False

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            if shape is None:
                if np.isscalar(low) and np.isscalar(high):
                    shape = (1,)
                else:
                    shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            self.shape = tuple(shape)
            self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Discrete:
        def __init__(self, n, seed=None, start=0):
            self.n = int(n)
            self.start = int(start)
            self.shape = ()
            self.dtype = np.dtype(np.int64)
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return int(self._rng.integers(self.start, self.start + self.n))

        def contains(self, x):
            return self.start <= int(x) < self.start + self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class MultiDiscrete:
        def __init__(self, nvec, seed=None):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            self.shape = self.nvec.shape
            self.dtype = np.dtype(np.int64)
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return self._rng.integers(0, self.nvec)

        def __repr__(self):
            return f"MultiDiscrete({self.nvec.tolist()})"

    def batch_space(space, n=1):
        if isinstance(space, Box):
            reps = (n,) + (1,) * len(space.shape)
            return Box(np.tile(space.low, reps), np.tile(space.high, reps), dtype=space.dtype)
        if isinstance(space, Discrete):
            return MultiDiscrete(np.full((n,), space.n, dtype=np.int64))
        raise TypeError(f"cannot batch {space!r}")
