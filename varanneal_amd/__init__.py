"""varanneal_amd -- MI355X-native variational-annealing action minimiser.

Drop-in for the hot path of paulrozdeba/varanneal (`va_ode.Annealer`): the
action A(X,p), its gradient and the L-BFGS / RF-ladder loop run as hand-written
HIP kernels for gfx950 behind a C-ABI (include/varanneal_amd.h).
"""
__version__ = "0.1.0"
from . import va_ode  # noqa: F401  (reference: varanneal/__init__.py:1-2)
