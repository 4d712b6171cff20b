"""Build + ctypes front-end of tests/cpu_emul/va_emul.cpp (TEST INFRASTRUCTURE)."""
import ctypes as C
import os
import subprocess

import numpy as np

from varanneal_amd import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libva_emul.so")
_SRCS = [os.path.join(_HERE, "va_emul.cpp"),
         os.path.join(_HERE, "..", "..", "varanneal_amd", "csrc", "va_core.h"),
         os.path.join(_HERE, "..", "..", "varanneal_amd", "csrc", "va_tile2.h"),
         os.path.join(_HERE, "..", "..", "varanneal_amd", "csrc", "va_tile3.h"),
         os.path.join(_HERE, "..", "..", "include", "varanneal_amd.h")]
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in _SRCS):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                                   "-Wall", "-o", _SO, _SRCS[0], "-lm"])
        _lib = C.CDLL(_SO)
        _lib.emul_action_grad.restype = C.c_int
        _lib.emul_anneal.restype = C.c_int
    return _lib


def action_grad(desc, T, XP, rf_scale):
    B, nv = XP.shape
    A = np.empty(B); me = np.empty(B); fe = np.empty(B); g = np.empty((B, nv))
    XP = np.ascontiguousarray(XP, dtype=np.float64)
    rc = lib().emul_action_grad(C.byref(desc), C.c_int(T), XP.ctypes.data_as(_capi.c_dp),
                                C.c_double(rf_scale), A.ctypes.data_as(_capi.c_dp),
                                me.ctypes.data_as(_capi.c_dp), fe.ctypes.data_as(_capi.c_dp),
                                g.ctypes.data_as(_capi.c_dp))
    if rc:
        raise ValueError("emul_action_grad rc=%d" % rc)
    return A, me, fe, g


def anneal(desc, T, XP, rf_scale, opt_args, want_paths=True):
    B, nv = XP.shape
    rf = np.ascontiguousarray(rf_scale, dtype=np.float64)
    nb = len(rf)
    XP = np.array(XP, dtype=np.float64)
    o = _capi.make_opts(opt_args)
    wide = desc.N_model * desc.D + desc.NP
    ame = np.zeros((B, nb, 3)); pest = np.zeros((B, nb, desc.NPest))
    st = np.zeros((B, nb), np.int32); nit = np.zeros((B, nb), np.int32); nfev = np.zeros((B, nb), np.int64)
    mp = np.zeros((B, nb, wide)) if want_paths else None
    cyc = C.c_longlong()
    rc = lib().emul_anneal(C.byref(desc), C.c_int(T), XP.ctypes.data_as(_capi.c_dp),
                           rf.ctypes.data_as(_capi.c_dp), C.c_int(nb), C.byref(o),
                           ame.ctypes.data_as(_capi.c_dp), pest.ctypes.data_as(_capi.c_dp),
                           st.ctypes.data_as(_capi.c_ip), nit.ctypes.data_as(_capi.c_ip),
                           nfev.ctypes.data_as(_capi.c_lp),
                           mp.ctypes.data_as(_capi.c_dp) if want_paths else None, C.byref(cyc))
    if rc:
        raise ValueError("emul_anneal rc=%d" % rc)
    return dict(x=XP, A=ame[:, :, 0], me=ame[:, :, 1], fe=ame[:, :, 2], pest=pest, status=st, nit=nit,
                nfev=nfev, minpaths=mp, cycles=cyc.value)
