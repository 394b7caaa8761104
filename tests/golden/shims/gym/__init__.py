"""Minimal stand-in for the `gym` package (absent in this image).

Only used by tests/golden/gen_golden.py so that /root/reference (pure Python)
can be imported to generate golden vectors.  It carries class shells only --
no arithmetic of the env.step() path lives in gym.
"""
from . import spaces, envs  # noqa: F401
from .envs import register, make  # noqa: F401


class Env:
    action_space = None
    observation_space = None

    @property
    def unwrapped(self):
        return self

    def reset(self):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.action_space = env.action_space
        self.observation_space = env.observation_space

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(action)
