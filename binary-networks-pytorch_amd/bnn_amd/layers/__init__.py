from .conv import Conv1d, Conv2d
from .linear import Linear

__all__ = ["Linear", "Conv2d", "Conv1d"]
