"""ctypes binding of include/varanneal_amd.h (libvaranneal_amd.so).

This is the thin shim north_star asks for: host code stays Python, the HIP
kernels are reached through a plain C-ABI.  There is NO CPU fallback: if the
shared library is missing or fails to load, importing callers get a loud
`VaLibraryError`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VARANNEAL_AMD_LIB", os.path.join(_HERE, "libvaranneal_amd.so"))   # (env: diagnostic builds)

VA_OK = 0
ABI_VERSION = 12         # VA_ABI_VERSION of include/varanneal_amd.h
ERRNAMES = {-1: "VA_EINVAL", -2: "VA_ENOMEM", -3: "VA_EHIP", -4: "VA_EUNSUPPORTED", -5: "VA_ESTATE"}
DISC = {"euler": 0, "trapezoid": 1, "SimpsonHermite": 2, "forwardmap": 3}
RHS = {"lorenz96": 0}
MEM_HOST, MEM_DEVICE = 0, 1

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_lp = C.POINTER(C.c_int64)


class VaLibraryError(RuntimeError):
    pass


class VaError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "%s (%d): %s" % (ERRNAMES.get(code, "VA_E?"), code, msg))
        self.code = code


class ProblemDesc(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32),
                ("D", C.c_int32), ("N_model", C.c_int32), ("N_data", C.c_int32),
                ("merr_nskip", C.c_int32), ("L", C.c_int32),
                ("Lidx", c_ip), ("Y", c_dp), ("dt_model", C.c_double),
                ("rm_kind", C.c_int32), ("rm", C.c_double), ("rm_array", c_dp),
                ("rf_kind", C.c_int32), ("rf0", C.c_double), ("rf0_array", c_dp),
                ("NP", C.c_int32), ("NPest", C.c_int32), ("Pidx", c_ip), ("P", c_dp),
                ("disc", C.c_int32), ("rhs", C.c_int32), ("lbfgs_m", C.c_int32),
                ("max_beta", C.c_int32), ("keep_paths", C.c_int32), ("tile_rows", C.c_int32),
                ("eval_kernel", C.c_int32), ("t_model", c_dp), ("stim", c_dp), ("n_stim", C.c_int32),
                ("p_time_dependent", C.c_int32), ("stream", C.c_void_p),
                ("lower", c_dp), ("upper", c_dp)]


class NnetDesc(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("batch", C.c_int32),
                ("n_layers", C.c_int32), ("structure", c_ip), ("M", C.c_int32),
                ("L_in", C.c_int32), ("L_out", C.c_int32), ("Lidx_in", c_ip), ("Lidx_out", c_ip),
                ("data_in", c_dp), ("data_out", c_dp), ("rm_in", C.c_double), ("rm_out", C.c_double),
                ("rf0", C.c_double), ("NP", C.c_int32), ("NPest", C.c_int32), ("Pidx", c_ip), ("P", c_dp),
                ("activation", C.c_int32), ("lbfgs_m", C.c_int32), ("max_beta", C.c_int32),
                ("keep_paths", C.c_int32), ("stream", C.c_void_p),
                ("rm_in_matrix", c_dp), ("rm_out_matrix", c_dp)]


ACTIVATION = {"sigmoid": 0, "tanh": 1, "linear": 2, "relu": 3, "softplus": 4}


class LbfgsOpts(C.Structure):
    _fields_ = [("maxcor", C.c_int32), ("ftol", C.c_double), ("gtol", C.c_double),
                ("maxiter", C.c_int32), ("maxfun", C.c_int64), ("maxls", C.c_int32)]


def make_opts(opt_args=None):
    """SciPy L-BFGS-B option names and defaults (scipy/optimize/_lbfgsb_py.py)."""
    o = dict(opt_args or {})
    return LbfgsOpts(int(o.get("maxcor", 10)), float(o.get("ftol", 2.2204460492503131e-09)),
                     float(o.get("gtol", 1e-5)), int(min(int(o.get("maxiter", 15000)), 2 ** 31 - 1)),
                     int(o.get("maxfun", 15000)), int(o.get("maxls", 20)))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def make_desc(batch, D, N_model, Y, Lidx, dt_model, RM, RF0, P, Pidx, disc="trapezoid",
              rhs="lorenz96", merr_nskip=1, lbfgs_m=10, max_beta=1, keep_paths=0, tile_rows=0,
              eval_kernel=0, device=0, stream=None, t_model=None, stim=None, p_time_dependent=False, bounds=None):
    """Build a ProblemDesc plus the list of arrays that must outlive it.  With
    p_time_dependent, P has shape (batch, N_model, NP) (or (N_model, NP), shared by the seeds)."""
    Y = _f64(Y)
    N_data, L = Y.shape
    Lidx = np.ascontiguousarray(Lidx, dtype=np.int32)
    Pidx = np.ascontiguousarray(Pidx, dtype=np.int32)
    P = _f64(P)
    if p_time_dependent:
        if P.ndim == 2:
            P = np.ascontiguousarray(np.broadcast_to(P, (batch,) + P.shape))
        if P.ndim != 3 or P.shape[:2] != (batch, N_model):
            raise ValueError("time-dependent P must have shape (batch, N_model, NP)")
    elif P.ndim == 1:
        P = np.ascontiguousarray(np.broadcast_to(P, (batch, P.shape[0])))
    keep = [Y, Lidx, Pidx, P]
    d = ProblemDesc()
    d.struct_size = C.sizeof(ProblemDesc)
    d.device, d.batch, d.D, d.N_model, d.N_data = device, batch, D, N_model, N_data
    d.merr_nskip, d.L = merr_nskip, L
    d.Lidx = Lidx.ctypes.data_as(c_ip)
    d.Y = Y.ctypes.data_as(c_dp)
    d.dt_model = float(dt_model)
    if isinstance(RM, np.ndarray):
        rm = _f64(RM)
        if rm.shape not in [(N_data, L), (N_data, L, L)]:
            raise ValueError("RM array must have shape (N_data, L) or (N_data, L, L)")
        keep.append(rm)
        d.rm_kind, d.rm, d.rm_array = rm.ndim - 1, 0.0, rm.ctypes.data_as(c_dp)
    else:
        d.rm_kind, d.rm, d.rm_array = 0, float(RM), None
    if isinstance(RF0, np.ndarray):
        rf = _f64(RF0)
        if rf.shape not in [(N_model - 1, D), (N_model - 1, D, D)]:
            raise ValueError("RF0 array must have shape (N_model-1, D) or (N_model-1, D, D)")
        keep.append(rf)
        d.rf_kind, d.rf0, d.rf0_array = rf.ndim - 1, 0.0, rf.ctypes.data_as(c_dp)
    else:
        d.rf_kind, d.rf0, d.rf0_array = 0, float(RF0), None
    d.NP, d.NPest = P.shape[-1], len(Pidx)
    d.p_time_dependent = 1 if p_time_dependent else 0
    d.Pidx = Pidx.ctypes.data_as(c_ip)
    d.P = P.ctypes.data_as(c_dp)
    d.disc = DISC[disc] if isinstance(disc, str) else int(disc)
    d.rhs = RHS[rhs] if isinstance(rhs, str) else int(rhs)      # int: a module id from load_rhs_module
    d.lbfgs_m, d.max_beta, d.keep_paths, d.tile_rows = lbfgs_m, max_beta, keep_paths, tile_rows
    d.eval_kernel = eval_kernel
    if t_model is not None:
        tm = _f64(t_model)
        if tm.shape != (N_model,):
            raise ValueError("t_model must have shape (N_model,)")
        keep.append(tm)
        d.t_model = tm.ctypes.data_as(c_dp)
    else:
        d.t_model = None
    if stim is not None:
        st = _f64(stim)
        st = st.reshape(N_model, -1)
        keep.append(st)
        d.stim, d.n_stim = st.ctypes.data_as(c_dp), st.shape[1]
    else:
        d.stim, d.n_stim = None, 0
    d.stream = stream
    d.lower = d.upper = None
    if bounds is not None:
        # one (lo, hi) per entry of the path vector [X | p_est] (va_ode.py:582-605); None = no bound
        n_var = N_model * (D + len(Pidx)) if p_time_dependent else N_model * D + len(Pidx)
        if len(bounds) != n_var:
            raise ValueError("bounds must have one (lo, hi) pair per entry of the path vector (%d), got %d" % (n_var, len(bounds)))
        lo = np.array([-np.inf if b[0] is None else b[0] for b in bounds], dtype=np.float64)
        hi = np.array([np.inf if b[1] is None else b[1] for b in bounds], dtype=np.float64)
        keep += [lo, hi]
        d.lower, d.upper = lo.ctypes.data_as(c_dp), hi.ctypes.data_as(c_dp)
    return d, keep


_lib = None


def lib():
    """Load libvaranneal_amd.so (built by varanneal_amd._build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VaLibraryError("%s not found: build it with `python -m varanneal_amd._build` "
                             "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise VaLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    h = C.c_void_p
    L.va_abi_version.restype = C.c_int32
    if L.va_abi_version() != ABI_VERSION:
        raise VaLibraryError("%s has ABI version %d, this binding expects %d: rebuild it with "
                             "`python -m varanneal_amd._build --force`" % (LIB_PATH, L.va_abi_version(), ABI_VERSION))
    L.va_last_error.restype = C.c_char_p
    L.va_device_count.argtypes = [c_ip]
    L.va_problem_create.argtypes = [C.POINTER(ProblemDesc), C.POINTER(h)]
    L.va_nnet_problem_create.argtypes = [C.POINTER(NnetDesc), C.POINTER(h)]
    L.va_rhs_load_module.argtypes = [C.c_char_p, c_ip]
    L.va_act_load_module.argtypes = [C.c_char_p, c_ip]
    L.va_eval_plan.argtypes = [C.POINTER(ProblemDesc), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    L.va_eval_plan_reach.argtypes = [C.POINTER(ProblemDesc), C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.va_problem_destroy.argtypes = [h]
    L.va_problem_destroy.restype = None
    L.va_problem_info.argtypes = [h, c_lp, c_lp, c_ip, c_ip]
    L.va_action_grad.argtypes = [h, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.va_minimize_lbfgs.argtypes = [h, C.c_void_p, C.c_int64, C.c_int32, C.c_double,
                                    C.POINTER(LbfgsOpts), c_dp, c_dp, c_dp, c_ip, c_ip, c_lp]
    L.va_anneal.argtypes = [h, C.c_void_p, C.c_int64, C.c_int32, c_dp, C.c_int32,
                            C.POINTER(LbfgsOpts), c_dp, c_dp, c_ip, c_ip, c_lp, c_dp]
    L.va_get_minpath.argtypes = [h, C.c_int32, C.c_int32, c_dp]
    L.va_eval_timed.argtypes = [h, C.c_double, C.c_int32, C.POINTER(C.c_float)]
    L.va_eval_timed_prepare.argtypes = [h, C.c_double, C.c_int32]
    L.va_get_counters.argtypes = [h, c_lp, c_lp, c_lp]
    L.va_comm_unique_id.argtypes = [C.c_char_p]
    L.va_comm_create.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(h)]
    L.va_comm_destroy.argtypes = [h]
    L.va_comm_destroy.restype = None
    L.va_gather_results.argtypes = [h, h, C.c_int32, c_dp, c_ip]
    L.va_lbfgs_timed.argtypes = [h, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.va_eval_ls_timed.argtypes = [h, C.c_double, C.c_int32, C.POINTER(C.c_float)]
    L.va_read_eval_outputs.argtypes = [h, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.va_debug_read_partials.argtypes = [h, c_dp, C.c_int64]
    L.va_problem_eval_kernel.argtypes = [h, c_ip, c_ip]
    L.va_problem_tune.argtypes = [h, C.c_int32, C.c_int32]
    L.va_debug_read_persist.argtypes = [h, c_dp, C.c_int64]
    L.va_debug_read_persist.restype = C.c_int
    L.va_problem_persistent.argtypes = [h, c_ip, c_ip]
    L.va_problem_persistent.restype = C.c_int
    for fn in ("va_device_count", "va_rhs_load_module", "va_act_load_module", "va_eval_plan", "va_eval_plan_reach", "va_problem_eval_kernel", "va_problem_tune", "va_problem_tune", "va_problem_create", "va_nnet_problem_create",
               "va_problem_info", "va_action_grad",
               "va_minimize_lbfgs", "va_anneal", "va_get_minpath", "va_eval_timed", "va_eval_timed_prepare",
               "va_get_counters", "va_debug_read_partials", "va_read_eval_outputs", "va_lbfgs_timed", "va_eval_ls_timed",
               "va_comm_unique_id", "va_comm_create", "va_gather_results"):
        getattr(L, fn).restype = C.c_int
    _lib = L
    return L


EXPORTS = ["va_abi_version", "va_last_error", "va_device_count", "va_rhs_load_module", "va_act_load_module", "va_eval_plan", "va_eval_plan_reach", "va_problem_eval_kernel", "va_problem_tune", "va_problem_create",
           "va_problem_destroy", "va_problem_info", "va_action_grad", "va_minimize_lbfgs",
           "va_anneal", "va_get_minpath", "va_eval_timed", "va_eval_timed_prepare", "va_problem_persistent", "va_debug_read_persist", "va_get_counters", "va_nnet_problem_create", "va_debug_read_partials",
           "va_read_eval_outputs", "va_lbfgs_timed", "va_eval_ls_timed", "va_comm_unique_id", "va_comm_create", "va_comm_destroy",
           "va_gather_results"]


def check(rc):
    if rc != VA_OK:
        raise VaError(rc, lib().va_last_error().decode("utf-8", "replace"))


def eval_plan(batch, D, N_model, disc, ne, ghost=0, rm_array=False, rm_full=False, rf_array=False, rf_full=False,
              merr_nskip=1, tile_rows=0, eval_kernel=0, bounded=False, p_time_dependent=False, reach=None, Lidx=None,
              builtin=False):
    """(eval kernel 3 | 4 | 5, disc, K, w) of the column-run kernel instantiation a problem of this shape would run
    for a model with a column form of `ne` products per element and / or a ghosted form of `ghost` ghost
    columns (0 = the model has no such form), or None (flat kernel).  w: kernel 4 -- 1 for scalar weights;
    kernel 3 -- threads per workgroup.  reach = (xl, xr, gl, gr) of the column form and Lidx (the observed columns)
    let wide states pick the streaming kernel 5.  builtin: the plan of the built-in Lorenz-96 (a few instantiations exist
    for it alone: runs of 12 rows, the row-mask variant of merr_nskip); the default is a generated model's plan -- what
    va_ode.py asks for before it writes a module.  No GPU call (va_eval_plan / va_eval_plan_reach)."""
    d = ProblemDesc()
    d.struct_size = C.sizeof(ProblemDesc)
    d.rhs = 0 if builtin else 1000                        # VA_RHS_LORENZ96 / VA_RHS_USER_BASE
    d.batch, d.D, d.N_model, d.merr_nskip = batch, D, N_model, merr_nskip
    d.N_data = (N_model - 1) // merr_nskip + 1
    d.rm_kind = (2 if rm_full else 1) if rm_array else 0
    d.rf_kind = (2 if rf_full else 1) if rf_array else 0
    d.disc = DISC[disc] if isinstance(disc, str) else int(disc)
    d.tile_rows, d.eval_kernel = tile_rows, eval_kernel
    d.p_time_dependent = 1 if p_time_dependent else 0
    dummy = (C.c_double * 1)()
    if bounded:
        d.lower = C.cast(dummy, c_dp); d.upper = C.cast(dummy, c_dp)
    out = (C.c_int32 * 4)()
    if reach is not None and Lidx is not None and len(Lidx) > 0:
        li = np.ascontiguousarray(Lidx, dtype=np.int32)
        d.L = len(li)
        d.Lidx = li.ctypes.data_as(c_ip)
        r = (C.c_int32 * 4)(*[int(v) for v in reach])
        check(lib().va_eval_plan_reach(C.byref(d), int(ne), int(ghost), r, out))
    else:
        check(lib().va_eval_plan(C.byref(d), int(ne), int(ghost), out))
    return (out[0], out[1], out[2], out[3]) if out[0] else None


_modules = {}


def load_rhs_module(path):
    """Register a generated right-hand-side module; returns its rhs id (cached per path)."""
    path = os.path.abspath(path)
    if path not in _modules:
        rid = C.c_int32()
        check(lib().va_rhs_load_module(path.encode(), C.byref(rid)))
        _modules[path] = rid.value
    return _modules[path]


_act_modules = {}


def load_act_module(path):
    """Register a generated activation module; returns its activation id (cached per path)."""
    path = os.path.abspath(path)
    if path not in _act_modules:
        aid = C.c_int32()
        check(lib().va_act_load_module(path.encode(), C.byref(aid)))
        _act_modules[path] = aid.value
    return _act_modules[path]


class Comm(object):
    """RCCL communicator of the job's one collective (va_comm_*): rank 0 makes the id with
    Comm.unique_id() and hands the 128 bytes to the other ranks."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(lib().va_comm_unique_id(buf))
        return buf.raw

    def __init__(self, uid, world, rank, device=0):
        self._L = lib()
        self.world, self.rank = world, rank
        self._c = C.c_void_p()
        check(self._L.va_comm_create(uid, world, rank, device, C.byref(self._c)))

    def close(self):
        if getattr(self, "_c", None) is not None and self._c.value:
            self._L.va_comm_destroy(self._c)
            self._c = C.c_void_p()

    __del__ = close


class Problem(object):
    """RAII wrapper of a va_handle; arrays are NumPy (host) or raw device pointers."""

    def __init__(self, batch, D, N_model, Y, Lidx, dt_model, RM, RF0, P, Pidx, **kw):
        self._L = lib()
        self.desc, self._keep = make_desc(batch, D, N_model, Y, Lidx, dt_model, RM, RF0, P, Pidx, **kw)
        self.B, self.D, self.N = batch, D, N_model
        self.NP, self.NPest = self.desc.NP, self.desc.NPest
        self.n_var = N_model * D + self.NPest
        if self.desc.p_time_dependent:
            # [X | p_est time-major]: to the shared solver one flat run without a parameter tail
            self.n_var = N_model * (D + self.NPest)
            self.NPe_t, self.NPt = self.NPest, self.NP
            self.N, self.D, self.NP, self.NPest = 1, self.n_var, 0, 0
        self.max_beta = self.desc.max_beta
        self._h = C.c_void_p()
        check(self._L.va_problem_create(C.byref(self.desc), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.va_problem_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def info(self):
        nv, ld = C.c_int64(), C.c_int64()
        T, nt = C.c_int32(), C.c_int32()
        check(self._L.va_problem_info(self._h, C.byref(nv), C.byref(ld), C.byref(T), C.byref(nt)))
        ek, rr = C.c_int32(), C.c_int32()
        check(self._L.va_problem_eval_kernel(self._h, C.byref(ek), C.byref(rr)))
        return dict(n_var=nv.value, ld=ld.value, tile_rows=T.value, ntiles=nt.value, eval_kernel=ek.value,
                    run_rows=rr.value)

    TUNE = {"fold": 1, "grad_sc1": 2, "prio": 3, "graph": 4, "persist": 5, "persist_rows": 6, "nnet_fused": 7}      # VA_TUNE_* of include/varanneal_amd.h

    def debug_read_persist(self, n):
        out = np.empty(n)
        check(self._L.va_debug_read_persist(self._h, out.ctypes.data_as(c_dp), n))
        return out

    def persistent(self):
        """(workgroups per seed, time rows per workgroup) when ladders on this handle run the persistent per-seed
        kernel (csrc/va_persist.h), else None"""
        g, t = C.c_int32(), C.c_int32()
        on = self._L.va_problem_persistent(self._h, C.byref(g), C.byref(t))
        return (g.value, t.value) if on else None

    def tune(self, **knobs):
        """performance knobs that change no result: fold= (tail inside the evaluation kernel), grad_sc1=, prio=, graph="""
        for k, v in knobs.items():
            check(self._L.va_problem_tune(self._h, self.TUNE[k], int(v)))

    def _xp(self, XP):
        XP = _f64(XP)
        if XP.shape != (self.B, self.n_var):
            raise ValueError("XP must have shape (B, N*D+NPest) = (%d, %d), got %s" %
                             (self.B, self.n_var, XP.shape))
        return XP

    def action_grad(self, XP, rf_scale=1.0, want_grad=True):
        XP = self._xp(XP)
        A = np.empty(self.B); me = np.empty(self.B); fe = np.empty(self.B)
        g = np.empty((self.B, self.n_var)) if want_grad else None
        check(self._L.va_action_grad(self._h, XP.ctypes.data, self.n_var, MEM_HOST, float(rf_scale),
                                     A.ctypes.data, me.ctypes.data, fe.ctypes.data,
                                     g.ctypes.data if want_grad else None, self.n_var))
        return A, me, fe, g

    def action_grad_device(self, xp_ptr, ld, rf_scale, A_ptr, me_ptr, fe_ptr, g_ptr, ldg):
        check(self._L.va_action_grad(self._h, xp_ptr, ld, MEM_DEVICE, float(rf_scale), A_ptr, me_ptr,
                                     fe_ptr, g_ptr, ldg))

    def minimize_lbfgs(self, XP, rf_scale, opt_args=None):
        XP = self._xp(XP).copy()
        o = make_opts(opt_args)
        A = np.empty(self.B); me = np.empty(self.B); fe = np.empty(self.B)
        st = np.empty(self.B, np.int32); nit = np.empty(self.B, np.int32); nfev = np.empty(self.B, np.int64)
        check(self._L.va_minimize_lbfgs(self._h, XP.ctypes.data, self.n_var, MEM_HOST, float(rf_scale),
                                        C.byref(o), A.ctypes.data_as(c_dp), me.ctypes.data_as(c_dp),
                                        fe.ctypes.data_as(c_dp), st.ctypes.data_as(c_ip),
                                        nit.ctypes.data_as(c_ip), nfev.ctypes.data_as(c_lp)))
        return dict(x=XP, A=A, me=me, fe=fe, status=st, nit=nit, nfev=nfev)

    def anneal(self, XP, rf_scale, opt_args=None, want_paths=False, xp_device=None, ld=None):
        rf = _f64(rf_scale)
        nb = rf.shape[0]
        o = make_opts(opt_args)
        B = self.B
        ame = np.empty((B, nb, 3)); pest = np.empty((B, nb, self.NPest))
        st = np.empty((B, nb), np.int32); nit = np.empty((B, nb), np.int32); nfev = np.empty((B, nb), np.int64)
        mp = np.empty((B, nb, self.N * self.D + self.NP)) if want_paths else None
        if xp_device is None:
            XP = self._xp(XP).copy()
            ptr, stride, mem = XP.ctypes.data, self.n_var, MEM_HOST
        else:
            ptr, stride, mem = xp_device, ld, MEM_DEVICE
        check(self._L.va_anneal(self._h, ptr, stride, mem, rf.ctypes.data_as(c_dp), nb, C.byref(o),
                                ame.ctypes.data_as(c_dp), pest.ctypes.data_as(c_dp),
                                st.ctypes.data_as(c_ip), nit.ctypes.data_as(c_ip),
                                nfev.ctypes.data_as(c_lp), mp.ctypes.data_as(c_dp) if want_paths else None))
        return dict(x=XP if xp_device is None else None, A=ame[:, :, 0], me=ame[:, :, 1],
                    fe=ame[:, :, 2], pest=pest, status=st, nit=nit, nfev=nfev, minpaths=mp)

    def eval_timed_prepare(self, rf_scale, iters):
        """arm the seeds and build / upload the hipGraph eval_timed(rf_scale, iters) replays (nothing timed)"""
        check(self._L.va_eval_timed_prepare(self._h, float(rf_scale), int(iters)))

    def eval_timed(self, rf_scale, iters):
        ms = C.c_float()
        check(self._L.va_eval_timed(self._h, float(rf_scale), int(iters), C.byref(ms)))
        return ms.value

    def gather_results(self, comm, nbeta):
        """va_gather_results: (table [world*B, nbeta, 3+NPest], status [world*B, nbeta]) on every rank"""
        n = comm.world * self.B
        table = np.empty((n, nbeta, 3 + self.NPest)); st = np.empty((n, nbeta), np.int32)
        check(self._L.va_gather_results(self._h, comm._c, nbeta, table.ctypes.data_as(c_dp), st.ctypes.data_as(c_ip)))
        return table, st

    def lbfgs_timed(self, iters):
        """(ms of `iters` k_update launches, ms of `iters` k_direction launches) with full histories;
        destroys the resident state."""
        a, b = C.c_float(), C.c_float()
        check(self._L.va_lbfgs_timed(self._h, int(iters), C.byref(a), C.byref(b)))
        return a.value, b.value

    def eval_ls_timed(self, rf_scale, iters):
        """ms of `iters` evaluation launches as a ladder cycle makes them (line-search trial points, line-search step
        in the tail); destroys the resident line-search state."""
        a = C.c_float()
        check(self._L.va_eval_ls_timed(self._h, float(rf_scale), int(iters), C.byref(a)))
        return a.value

    def read_eval_outputs(self, want_grad=True):
        """(A, me, fe, grad) as the last S1 evaluation (action_grad / eval_timed) left them on the device."""
        A = np.empty(self.B); me = np.empty(self.B); fe = np.empty(self.B)
        g = np.empty((self.B, self.n_var)) if want_grad else None
        check(self._L.va_read_eval_outputs(self._h, A.ctypes.data, me.ctypes.data, fe.ctypes.data,
                                           g.ctypes.data if want_grad else None, self.n_var))
        return A, me, fe, g

    def debug_partials(self, n):
        out = np.empty(n)
        check(self._L.va_debug_read_partials(self._h, out.ctypes.data_as(c_dp), n))
        return out

    def counters(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(self._L.va_get_counters(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(eval_launches=a.value, seed_evals=b.value, cycles=c.value)


def make_nnet_desc(batch, structure, data_in, data_out, Lidx, RM, RF0, P, Pidx, act="sigmoid",
                   lbfgs_m=10, max_beta=1, keep_paths=0, device=0, stream=None):
    """Fill a va_nnet_desc from what va_nnet.Annealer.anneal_init holds (va_nnet.py:288-450)."""
    st = np.ascontiguousarray(structure, dtype=np.int32)
    din = _f64(np.atleast_2d(data_in)); dout = _f64(np.atleast_2d(data_out))
    M = din.shape[0]
    lin = np.ascontiguousarray(Lidx[0], dtype=np.int32); lout = np.ascontiguousarray(Lidx[1], dtype=np.int32)
    if din.shape != (M, lin.size) or dout.shape != (M, lout.size):
        raise ValueError("data_in/data_out must have shapes (M, len(Lidx[0])) / (M, len(Lidx[1])), got %s / %s"
                         % (din.shape, dout.shape))
    rmm = None
    if isinstance(RM, (list, tuple, np.ndarray)) and np.ndim(RM[0] if len(RM) else 0) == 2:
        # [RMin, RMout] with full matrices (va_nnet.py:136-139)
        rmm = (_f64(RM[0]), _f64(RM[1]))
        if len(RM) != 2 or rmm[0].shape != (lin.size, lin.size) or rmm[1].shape != (lout.size, lout.size):
            raise ValueError("matrix RM must be [RMin (Lin x Lin), RMout (Lout x Lout)]")
        rm_in = rm_out = 0.0
    elif isinstance(RM, (list, tuple, np.ndarray)) and np.ndim(RM) > 0:
        RM = np.asarray(RM, dtype=np.float64)
        if RM.shape != (2,):
            raise ValueError("RM must be a scalar, [RM_in, RM_out] or two matrices (va_nnet.py:132-139)")
        rm_in, rm_out = float(RM[0]), float(RM[1])
    else:
        rm_in = rm_out = float(RM)
    if not isinstance(act, (int, np.integer)) and act not in ACTIVATION:      # int: an id from load_act_module
        raise NotImplementedError("activation %r is not built in (have %s)" % (act, sorted(ACTIVATION)))
    P = _f64(P)
    if P.ndim == 1:
        P = np.tile(P, (batch, 1))
    pidx = np.ascontiguousarray(Pidx, dtype=np.int32)
    d = NnetDesc()
    d.struct_size = C.sizeof(NnetDesc); d.device = device; d.batch = batch
    d.n_layers = st.size; d.structure = st.ctypes.data_as(c_ip); d.M = M
    d.L_in, d.L_out = lin.size, lout.size
    d.Lidx_in = lin.ctypes.data_as(c_ip); d.Lidx_out = lout.ctypes.data_as(c_ip)
    d.data_in = din.ctypes.data_as(c_dp); d.data_out = dout.ctypes.data_as(c_dp)
    d.rm_in, d.rm_out, d.rf0 = rm_in, rm_out, float(RF0)
    d.NP, d.NPest = P.shape[1], pidx.size
    d.Pidx = pidx.ctypes.data_as(c_ip); d.P = P.ctypes.data_as(c_dp)
    d.activation = int(act) if isinstance(act, (int, np.integer)) else ACTIVATION[act]; d.lbfgs_m = lbfgs_m; d.max_beta = max_beta; d.keep_paths = keep_paths
    d.stream = stream
    if rmm is not None:
        d.rm_in_matrix = rmm[0].ctypes.data_as(c_dp); d.rm_out_matrix = rmm[1].ctypes.data_as(c_dp)
    return d, (st, din, dout, lin, lout, P, pidx, rmm)


class NnetProblem(Problem):
    """va_handle of a feed-forward-network action; same S1/S2/S3 methods as Problem.  The
    path vector is [X (M*NDnet) | p_est]; `minpaths` rows from anneal() have that width."""

    def __init__(self, batch, structure, data_in, data_out, Lidx, RM, RF0, P, Pidx, **kw):
        self._L = lib()
        self.desc, self._keep = make_nnet_desc(batch, structure, data_in, data_out, Lidx, RM, RF0, P, Pidx, **kw)
        self.B = batch
        self.M, self.NDnet = self.desc.M, int(np.sum(structure))
        self.NDens = self.M * self.NDnet
        self.NP_net, self.NPest_net = self.desc.NP, self.desc.NPest
        self.n_var = self.NDens + self.NPest_net
        # to the shared solver the vector has no separate parameter tail
        self.N, self.D, self.NP, self.NPest = 1, self.n_var, 0, 0
        self.max_beta = self.desc.max_beta
        self._h = C.c_void_p()
        check(self._L.va_nnet_problem_create(C.byref(self.desc), C.byref(self._h)))
