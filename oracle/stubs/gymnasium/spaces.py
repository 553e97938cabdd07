class Space:
    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k


class Box(Space):
    pass


class MultiBinary(Space):
    pass


class Discrete(Space):
    def __init__(self, n, *a, **k):
        super().__init__(n, *a, **k)
        self.n = n

    def sample(self):
        import random
        return random.randrange(int(self.n))


class Tuple(Space):
    def __init__(self, spaces, *a, **k):
        super().__init__(spaces, *a, **k)
        self.spaces = tuple(spaces)

    def sample(self):
        return tuple(s.sample() for s in self.spaces)


class Dict(Space):
    """Key order as gymnasium 0.29.1 (the reference's pin, poetry.lock:26-27) defines it, spaces/dict.py: a plain
    mapping is SORTED by key, an OrderedDict keeps its order.  FlattenObservation concatenates in this order."""

    def __init__(self, spaces=None, **k):
        super().__init__(spaces, **k)
        import collections
        spaces = spaces or {}
        if isinstance(spaces, collections.OrderedDict):
            self.spaces = collections.OrderedDict(spaces)
        else:
            self.spaces = collections.OrderedDict(sorted(dict(spaces).items()))

    def __getitem__(self, key):
        return self.spaces[key]


def flatten(space, x):
    """gymnasium.spaces.utils.flatten for the space kinds the reference uses (0.29.1 spaces/utils.py): Dict -> the
    flattened sub-observations concatenated in the SPACE's key order; Box / MultiBinary -> ravel."""
    import numpy as np
    if isinstance(space, Dict):
        return np.concatenate([flatten(s, x[k]) for k, s in space.spaces.items()])
    return np.asarray(x).ravel()
