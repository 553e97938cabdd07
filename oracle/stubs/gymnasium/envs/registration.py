def register(id, entry_point, **kwargs):
    import gymnasium
    gymnasium._registry[id] = entry_point
