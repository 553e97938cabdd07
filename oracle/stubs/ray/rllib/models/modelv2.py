class ModelV2:
    pass
