"""Wrappers of the reference that sit on the step path (gridworld/wrappers.py)."""
from . import spaces


class Wrapper:
    """Attribute pass-through to the wrapped env (gridworld/env.py:306-314)."""

    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name == 'env':
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self):
        return self.env.reset()

    def step(self, action):
        return self.env.step(action)


class Actions(Wrapper):
    """Discrete(17) action set without `place` (gridworld/wrappers.py:11-32): new index -> Discrete(18) index.
    With select_and_place the hotbar actions place blocks, so index 17 (use) is redundant."""

    def __init__(self, env):
        super().__init__(env)
        self.action_map = list(range(17))  # 0 noop, 1-4 move, 5 jump, 6-11 hotbar, 12-15 camera, 16 break
        self.action_space = spaces.Discrete(len(self.action_map))

    def step(self, action):
        return self.env.step(self.action_map[action])
