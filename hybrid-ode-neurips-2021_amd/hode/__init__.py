"""hode -- host side of the MI355X-native hybrid-ODE hot path.

``hode.odeint`` is the drop-in for ``torchdiffeq.odeint`` at the reference's call sites
(``model.py:1116``, ``:837``); it launches the hand-written gfx950 kernels of ``libhode.so``
through the C ABI declared in ``include/hode.h``.  There is no CPU or PyTorch fallback:
calling it without the library or with CPU tensors raises.
"""

from ._lib import HodeConfigError, HodeError, lib, library_path  # noqa: F401
from .solver import odeint, roche_solve  # noqa: F401

__all__ = ["odeint", "roche_solve", "lib", "library_path", "HodeError", "HodeConfigError"]
