"""Observation/action space descriptors.  Uses `gymnasium.spaces` when gymnasium is installed (it is an
optional dependency: absent from the build image and the GPU box), otherwise a minimal stand-in that
supports what the reference's examples use: `.sample()`, `.n`, `.spaces`, `[]`."""
import numpy as np

try:  # (gymnasium is optional: tests/test_gym_registration.py covers this branch with a stand-in on the path)
    import gymnasium as gym
    from gymnasium import spaces as _sp
    Box, Discrete, MultiBinary, Tuple, Dict = _sp.Box, _sp.Discrete, _sp.MultiBinary, _sp.Tuple, _sp.Dict
    Env, Wrapper, ActionWrapper = gym.Env, gym.Wrapper, gym.ActionWrapper
    HAVE_GYMNASIUM = True
except ImportError:
    HAVE_GYMNASIUM = False

    class _Space:
        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        @property
        def rng(self):
            if not hasattr(self, "_rng"):
                self._rng = np.random.default_rng()
            return self._rng

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.dtype = np.asarray(low), np.asarray(high), np.dtype(dtype)
            self.shape = tuple(shape) if shape is not None else self.low.shape

        def sample(self):
            lo = np.broadcast_to(self.low, self.shape).astype(np.int64)
            hi = np.broadcast_to(self.high, self.shape).astype(np.int64)
            return self.rng.integers(lo, hi + 1).astype(self.dtype)

    class Discrete(_Space):
        def __init__(self, n):
            self.n = int(n)

        def sample(self, mask=None):
            if mask is not None:
                return int(self.rng.choice(np.nonzero(np.asarray(mask))[0]))
            return int(self.rng.integers(0, self.n))

    class MultiBinary(_Space):
        def __init__(self, n):
            self.n = n
            self.shape = (n,) if np.isscalar(n) else tuple(n)

        def sample(self):
            return self.rng.integers(0, 2, self.shape).astype(np.int8)

    class Tuple(_Space):
        def __init__(self, spaces):
            self.spaces = tuple(spaces)

        def sample(self):
            return tuple(s.sample() for s in self.spaces)

        def __getitem__(self, i):
            return self.spaces[i]

    class Dict(_Space):
        def __init__(self, spaces=None):
            self.spaces = dict(spaces or {})

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def __getitem__(self, k):
            return self.spaces[k]

    class Env:
        metadata = {}
        render_mode = None

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
            self.observation_space = getattr(env, "observation_space", None)
            self.action_space = getattr(env, "action_space", None)

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def reset(self, **kw):
            return self.env.reset(**kw)

        def step(self, action):
            return self.env.step(action)

        def close(self):
            return self.env.close()

    class ActionWrapper(Wrapper):
        def step(self, action):
            return self.env.step(self.action(action))
