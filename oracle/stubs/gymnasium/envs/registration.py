import gymnasium

registry = gymnasium._registry  # id -> entry point (the real package maps id -> EnvSpec; `id in registry` works on both)


def register(id, entry_point, **kwargs):
    gymnasium._registry[id] = entry_point
