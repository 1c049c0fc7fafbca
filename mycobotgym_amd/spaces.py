"""Minimal observation/action space descriptors with the Gymnasium ``Box`` / ``Dict`` surface.

gymnasium is not installed in the build image; callers that have it can convert with
``to_gymnasium()``.  Shapes/dtypes follow the reference: Dict{observation, achieved_goal,
desired_goal} of float64 Boxes (mycobot.py:117-130) and Box(-1, 1, (A,), float32) actions (:108-110).
"""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, shape, dtype):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
        self._rng = np.random.default_rng()

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    def to_gymnasium(self):
        from gymnasium import spaces
        return spaces.Box(self.low, self.high, self.shape, self.dtype)


class Dict:
    def __init__(self, spaces: dict):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

    def __repr__(self):
        return "Dict(" + ", ".join(f"{k}: {v}" for k, v in self.spaces.items()) + ")"

    def to_gymnasium(self):
        from gymnasium import spaces
        return spaces.Dict({k: v.to_gymnasium() for k, v in self.spaces.items()})


def batch_box(box: Box, n: int) -> Box:
    b = Box(0, 0, (n,) + box.shape, box.dtype)
    b.low = np.broadcast_to(box.low, b.shape).copy()
    b.high = np.broadcast_to(box.high, b.shape).copy()
    return b
