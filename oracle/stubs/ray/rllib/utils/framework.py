from typing import Any

TensorType = Any
