"""Minimal action / observation space objects (gym is not a dependency of the device path).
Shapes and dtypes follow gridworld/env.py:56-95."""
import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = int(n)

    def sample(self):
        return int(np.random.randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __repr__(self):
        return f'Discrete({self.n})'


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()

    def sample(self):
        if np.issubdtype(self.dtype, np.integer):
            return np.random.randint(self.low, self.high + 1).astype(self.dtype)
        return np.random.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f'Box{self.shape}'


class String(Space):
    def __init__(self):
        super().__init__((), np.object_)

    def sample(self):
        return ''

    def contains(self, x):
        return isinstance(x, str)


class Dict(Space):
    def __init__(self, spaces):
        super().__init__(None, None)
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

    def contains(self, x):
        return isinstance(x, dict) and all(k in x for k in self.spaces)

    def __repr__(self):
        return 'Dict(' + ', '.join(f'{k}: {v!r}' for k, v in self.spaces.items()) + ')'
