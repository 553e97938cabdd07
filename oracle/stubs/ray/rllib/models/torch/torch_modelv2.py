class TorchModelV2:
    pass
