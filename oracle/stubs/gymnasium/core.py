from typing import Any

ObsType = Any
ActType = Any


class Env:
    metadata = {}

    def reset(self, seed=None, options=None):
        return None

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, action):
        return self.env.step(action)


class ActionWrapper(Wrapper):
    def step(self, action):
        return self.env.step(self.action(action))


class ObservationWrapper(Wrapper):
    def reset(self, **kw):
        obs, info = self.env.reset(**kw)
        return self.observation(obs), info

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        return self.observation(obs), reward, terminated, truncated, info
