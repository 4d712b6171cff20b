"""Load the reference's Python-2 modules under Python 3, in memory.

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/gen_golden.py`` in the build
container (where ``/root/reference`` is mounted) to produce the data-only
fixtures under ``tests/golden/``.  Nothing here is imported by the product
package, by ``-m gpu`` tests, ``smoke()`` or ``bench.py``; the reference never
travels to the GPU box and no reference source text is written to the repo.

Recipe: the reference (``varanneal/_autodiffmin.py``,
``varanneal/va_ode.py``) is Python 2 only (``exec '…'`` statements, ``xrange``,
bound-method ``.im_func``) and imports PYADOLC, which is not installed.  The
action arithmetic (``va_ode.py:130-234, 341-454``) is pure NumPy and
type-polymorphic, so we

  1. register an inert module named ``adolc`` (only the names the reference
     touches at import time / in isinstance checks, ``va_ode.py:752,763``);
  2. read the two source files as text, apply four mechanical py2->py3
     substitutions, and ``exec`` them into fresh module objects.

The reference's ``A``/``me_gaussian``/``fe_gaussian``/``disc_*``/
``anneal_init``/``anneal_step`` then run unmodified.  ADOL-C itself is absent,
so *gradient* goldens are complex-step derivatives through the reference's own
``A`` (it accepts complex128), not ADOL-C output.
"""
import os
import re
import sys
import types

REF_ROOT = os.environ.get("VARANNEAL_REFERENCE", "/root/reference")


def _py3(src):
    src = re.sub(r"exec ('self\.[A-Za-z]+ = self\.[a-z_%]+'%\([a-z,]+\))", r"exec(\1)", src)
    src = src.replace("xrange", "range")
    src = src.replace(".im_func.", ".__func__.")
    # py2 integer division in the one loop bound that relies on it (va_ode.py:215, full RF matrices)
    src = src.replace("range((self.N_model - 1) / 2)", "range((self.N_model - 1) // 2)")
    return src


class _Inert(object):
    """Placeholder type for adolc._adolc.adouble / adub isinstance checks."""


def make_adolc_stub():
    m = types.ModuleType("adolc")
    m._adolc = types.SimpleNamespace(adouble=_Inert, adub=_Inert)
    m.trace_on = lambda *a, **k: None
    m.trace_off = lambda *a, **k: None
    m.independent = lambda *a, **k: None
    m.dependent = lambda *a, **k: None
    m.adouble = lambda x: x
    return m


def load_reference(which="va_ode"):
    """Return the reference module ``va_ode`` (or ``va_nnet``) running on py3."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not mounted at %s" % REF_ROOT)
    if "adolc" not in sys.modules:
        sys.modules["adolc"] = make_adolc_stub()
    mods = {}
    for name in ("_autodiffmin", which):
        path = os.path.join(REF_ROOT, "varanneal", name + ".py")
        with open(path) as fh:
            src = _py3(fh.read())
        m = types.ModuleType(name)
        m.__file__ = path
        sys.modules[name] = m
        exec(compile(src, path, "exec"), m.__dict__)
        mods[name] = m
    return mods[which]


def complex_step_grad(A, XP, h=1e-30):
    """dA/dXP_i = Im A(XP + i h e_i) / h through the reference's own ``A``."""
    import numpy as np
    n = XP.shape[0]
    g = np.empty(n)
    z = XP.astype(np.complex128)
    for i in range(n):
        z[i] = complex(XP[i], h)
        g[i] = np.imag(A(z)) / h
        z[i] = XP[i]
    return g
