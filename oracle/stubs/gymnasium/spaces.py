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


class Tuple(Space):
    def __init__(self, spaces, *a, **k):
        super().__init__(spaces, *a, **k)
        self.spaces = tuple(spaces)


class Dict(Space):
    def __init__(self, spaces=None, **k):
        super().__init__(spaces, **k)
        self.spaces = dict(spaces or {})

    def __getitem__(self, key):
        return self.spaces[key]
