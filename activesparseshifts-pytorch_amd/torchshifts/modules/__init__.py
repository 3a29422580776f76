from .shifts import Shift1d, Shift2d, Shift3d  # noqa: F401
