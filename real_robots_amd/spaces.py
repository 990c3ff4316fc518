"""Minimal gym.spaces work-alikes (Box, Dict, MultiBinary).

`gym` is not a dependency of this package: when it is importable its classes are used, so that policies written
against gym.spaces keep working (isinstance checks included); otherwise these stand-ins provide the attributes the
reference's code and example policies touch: .low/.high/.shape/.dtype/.sample()/.contains() and Dict.spaces
(real_robots/envs/env.py:42-81, robot.py:69-112, README "Usage" RandomPolicy).
"""
import numpy as np

try:  # pragma: no cover - gym is absent in the build image
    from gym.spaces import Box, Dict, MultiBinary  # noqa: F401
    HAVE_GYM = True
except Exception:
    HAVE_GYM = False

    class Space:
        _rng = np.random.default_rng()

        @classmethod
        def seed_all(cls, seed):
            Space._rng = np.random.default_rng(seed)

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=float):
            if shape is not None:
                low = np.full(shape, low, dtype=np.float64)
                high = np.full(shape, high, dtype=np.float64)
            self.low = np.asarray(low, dtype=np.float64)
            self.high = np.asarray(high, dtype=np.float64)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1e6)
            hi = np.where(np.isfinite(self.high), self.high, 1e6)
            x = Space._rng.uniform(lo, hi)
            if np.issubdtype(self.dtype, np.integer):
                x = np.floor(x)
            return x.astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return "Box%s" % (self.shape,)

    class MultiBinary(Space):
        def __init__(self, n):
            self.n = n
            self.shape = (n,)
            self.dtype = np.dtype(np.int8)

        def sample(self):
            return Space._rng.integers(0, 2, size=self.n).astype(np.int8)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all((x == 0) | (x == 1)))

    class Dict(Space):
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def contains(self, x):
            return isinstance(x, dict) and all(k in x and s.contains(x[k]) for k, s in self.spaces.items())

        def __getitem__(self, k):
            return self.spaces[k]

        def keys(self):
            return self.spaces.keys()
