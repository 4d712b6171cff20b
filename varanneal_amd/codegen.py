"""User right-hand sides on the device: trace -> differentiate -> HIP -> module.

The reference accepts an arbitrary Python callable `f(t, x, p)` (or `f(t, x, (p, stim))`
with a stimulus) acting on whole time slices (varanneal/va_ode.py:56-67, 345-375) and
obtains derivatives by replaying an ADOL-C tape of it (_autodiffmin.py:32-58).  The tape
works because `f` is written with type-polymorphic NumPy operations; the same property
lets us run `f` ONCE on an object array of symbolic scalars, differentiate symbolically
(SymPy) and emit the three device functions the tile kernel needs,

    f_i(x, p, t, stim)            (J^T s)_j = sum_i s_i df_i/dx_j
    acc_k += s_i df_i/dp_k

as a generated header `struct RhsUser`, compiled with hipcc for gfx950 together with
csrc/va_user_rhs.hip into a shared object that va_rhs_load_module() registers.  No tape,
no interpreter at evaluation time.

Limits: `f` must be traceable (no data-dependent Python branching on x/p; NumPy ufuncs
exp/log/tanh/... and arithmetic are fine); NP <= 24.  Before use the generated
expressions are checked numerically against `f` itself on random inputs.
"""
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
CACHE = os.environ.get("VARANNEAL_AMD_RHS_CACHE", os.path.join(_HERE, "_rhs_cache"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
MAX_NP = 128           # RHS_BIG_NP of csrc/va_core.h (the flat kernel); the tuned column / ghosted forms take up to MAX_NP_TUNED
MAX_NP_TUNED = 24      # RHS_MAX_NP


def _sympy():
    import sympy
    return sympy


class Sym(object):
    """Scalar wrapper so that NumPy ufuncs on object arrays dispatch to SymPy."""
    def __init__(self, e):
        self.e = e

    @staticmethod
    def _u(o):
        return o.e if isinstance(o, Sym) else o

    # binary operators: with an ndarray on the other side return NotImplemented so that
    # NumPy broadcasts element by element (each element then comes back here as a scalar)
    def _bin(self, o, fn):
        if isinstance(o, np.ndarray):
            return NotImplemented
        return Sym(fn(self.e, self._u(o)))

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: b + a)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: b * a)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: b / a)
    def __pow__(self, o): return self._bin(o, lambda a, b: a ** b)
    def __rpow__(self, o): return self._bin(o, lambda a, b: b ** a)
    def __neg__(self): return Sym(-self.e)
    def __pos__(self): return self

    # comparisons give symbolic conditions (for np.where); only USING one as a Python truth value cannot be traced
    def __lt__(self, o): return self._bin(o, lambda a, b: _sympy().Lt(a, b))
    def __le__(self, o): return self._bin(o, lambda a, b: _sympy().Le(a, b))
    def __gt__(self, o): return self._bin(o, lambda a, b: _sympy().Gt(a, b))
    def __ge__(self, o): return self._bin(o, lambda a, b: _sympy().Ge(a, b))

    def __bool__(self):
        raise TypeError("the model function branches on the value of a state/parameter (if / and / or / max() on traced "
                        "values); write the branch as np.where / np.maximum / np.minimum / np.abs, which are traced to selects")

    def __abs__(self):
        return Sym(_pw_abs(self.e))

    def __getattr__(self, name):          # np.tanh(obj_array) calls elem.tanh(), etc.
        sp = _sympy()
        table = {"exp": sp.exp, "log": sp.log, "sin": sp.sin, "cos": sp.cos, "tan": sp.tan,
                 "tanh": sp.tanh, "sinh": sp.sinh, "cosh": sp.cosh, "sqrt": sp.sqrt,
                 "arctan": sp.atan, "arcsin": sp.asin, "arccos": sp.acos,
                 "arctanh": sp.atanh, "arcsinh": sp.asinh, "log1p": lambda z: sp.log(1 + z),
                 "expm1": lambda z: sp.exp(z) - 1, "square": lambda z: z * z,
                 "reciprocal": lambda z: 1 / z}
        if name in table:
            return lambda: Sym(table[name](self.e))
        raise AttributeError(name)


# ---- piecewise models (the reference tapes whatever executes, _autodiffmin.py:41-44): np.where / maximum / minimum /
# abs / sign / clip and comparisons on traced arrays become SymPy Piecewise expressions, printed as C selects; the
# derivative of a Piecewise is the Piecewise of its branches' derivatives -- the taken branch's, as a tape would give.
def _pw_max(a, b):
    sp = _sympy()
    return sp.Piecewise((a, sp.Ge(a, b)), (b, True))


def _pw_min(a, b):
    sp = _sympy()
    return sp.Piecewise((a, sp.Le(a, b)), (b, True))


def _pw_abs(a):
    sp = _sympy()
    return sp.Piecewise((a, sp.Ge(a, 0)), (-a, True))


def _pw_sign(a):
    sp = _sympy()
    return sp.Piecewise((1.0, sp.Gt(a, 0)), (-1.0, sp.Lt(a, 0)), (0.0, True))


def _pw_where(c, a, b):
    sp = _sympy()
    if c is True or c is sp.true or (isinstance(c, (bool, np.bool_)) and c):
        return a
    if c is False or c is sp.false or (isinstance(c, (bool, np.bool_)) and not c):
        return b
    return sp.Piecewise((a, c), (b, True))


def _elementwise(fn, *args):
    """fn over the broadcast of object arrays / scalars of Sym or numbers -> SymArray of Sym"""
    arrs = [np.asarray(a, dtype=object).view(np.ndarray) for a in args]
    bc = np.broadcast(*arrs)
    res = np.empty(bc.shape, dtype=object)
    for i, vals in enumerate(bc):
        res.flat[i] = Sym(fn(*[Sym._u(v) for v in vals]))
    return res.view(SymArray)


class SymArray(np.ndarray):
    """object array of Sym that intercepts what NumPy would otherwise decide by truth-testing its elements"""
    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        sp = _sympy()
        table = {np.maximum: _pw_max, np.fmax: _pw_max, np.minimum: _pw_min, np.fmin: _pw_min, np.absolute: _pw_abs,
                 np.fabs: _pw_abs, np.sign: _pw_sign, np.greater: sp.Gt, np.greater_equal: sp.Ge, np.less: sp.Lt,
                 np.less_equal: sp.Le,
                 np.heaviside: lambda a, h: sp.Piecewise((1.0, sp.Gt(a, 0)), (0.0, sp.Lt(a, 0)), (h, True))}
        if method == "__call__" and out is None and ufunc in table:
            return _elementwise(table[ufunc], *inputs)
        ins = [i.view(np.ndarray) if isinstance(i, SymArray) else i for i in inputs]
        if out is not None:
            kwargs["out"] = tuple(o.view(np.ndarray) if isinstance(o, SymArray) else o for o in out)
        r = getattr(ufunc, method)(*ins, **kwargs)
        return r.view(SymArray) if isinstance(r, np.ndarray) and r.dtype == object else r

    def __array_function__(self, func, types, args, kwargs):
        if func is np.where and len(args) == 3 and not kwargs:
            return _elementwise(_pw_where, *args)
        if func is np.clip and len(args) >= 3:
            lo, hi = args[1], args[2]
            r = args[0]
            if lo is not None:
                r = _elementwise(_pw_max, r, lo)
            if hi is not None:
                r = _elementwise(_pw_min, r, hi)
            return r
        return super().__array_function__(func, types, args, kwargs)


def trace(f, D, NP, nstim=0, stim_ndim=1, p_rows=False):
    """Run `f` once on symbols.  Returns (exprs[D], symbols dict).  p_rows: the model takes
    time-dependent parameters, i.e. p of shape (rows, NP) indexed p[:, k] (va_ode.py:170-188)."""
    sp = _sympy()
    xs = sp.symbols("x0:%d" % D, real=True)
    ps = sp.symbols("p0:%d" % max(NP, 1), real=True)[:NP]
    ss = sp.symbols("st0:%d" % max(nstim, 1), real=True)[:nstim]
    t = sp.Symbol("t", real=True)
    X = np.empty((1, D), dtype=object).view(SymArray)
    for j in range(D):
        X[0, j] = Sym(xs[j])
    P = np.empty((1, NP) if p_rows else NP, dtype=object).view(SymArray)
    for k in range(NP):
        P[(0, k) if p_rows else k] = Sym(ps[k])
    T = np.empty(1, dtype=object).view(SymArray)
    T[0] = Sym(t)
    if nstim:
        if stim_ndim == 1:
            S = np.empty(1, dtype=object).view(SymArray); S[0] = Sym(ss[0])
        else:
            S = np.empty((1, nstim), dtype=object).view(SymArray)
            for k in range(nstim):
                S[0, k] = Sym(ss[k])
        out = f(T, X, (P, S))
    else:
        out = f(T, X, P)
    out = np.asarray(out, dtype=object)
    if out.shape != (1, D):
        raise ValueError("model function returned shape %s for a (1, %d) state slice" % (out.shape, D))
    exprs = [sp.sympify(Sym._u(out[0, i])) for i in range(D)]
    return exprs, dict(x=xs, p=ps, st=ss, t=t)


def _printer():
    sp = _sympy()
    from sympy.printing.c import C99CodePrinter

    class P(C99CodePrinter):
        def _print_Pow(self, e):
            b, ex = e.as_base_exp()
            if ex.is_Integer and 2 <= int(ex) <= 4:
                s = self.parenthesize(b, 50)
                return "(" + "*".join([s] * int(ex)) + ")"
            if ex.is_Integer and -4 <= int(ex) <= -1:
                s = self.parenthesize(b, 50)
                return "(1.0/(" + "*".join([s] * (-int(ex))) + "))"
            return super()._print_Pow(e)

        def _print_Symbol(self, s):
            n = s.name
            for pre, arr in (("x", "x"), ("p", "p"), ("st", "st"), ("sv", "s")):
                if n.startswith(pre) and n[len(pre):].isdigit():
                    return "%s[%s]" % (arr, n[len(pre):])
            return n
    return P()


def _emit_case(pr, expr, ret):
    """C statements computing `expr` (with CSE temporaries) then `ret % value`."""
    sp = _sympy()
    repl, red = sp.cse([expr], symbols=sp.numbered_symbols("c_"), optimizations="basic")
    lines = ["const double %s = %s;" % (pr.doprint(a), pr.doprint(b)) for a, b in repl]
    lines.append(ret % pr.doprint(red[0]))
    return " ".join(lines)


def _switch_flat(out, exprs, syms, D, NP, pr, sv):
    """f, J^T s and (df/dp)^T s of the flat kernel's struct with one case per state component"""
    sp = _sympy()
    # f
    out.append("    static VA_HD double f(const double *x, int i, int, const double *p, double t, const double *st)")
    out.append("    {")
    out.append("        (void)x; (void)p; (void)t; (void)st;")
    out.append("        switch (i) {")
    for i, e in enumerate(exprs):
        out.append("        case %d: { %s }" % (i, _emit_case(pr, e, "return %s;")))
    out.append("        default: return 0.0;")
    out.append("        }")
    out.append("    }")
    # vjp
    out.append("    static VA_HD double vjp(const double *x, const double *s, int j, int, const double *p, double t, const double *st)")
    out.append("    {")
    out.append("        (void)x; (void)s; (void)p; (void)t; (void)st;")
    out.append("        switch (j) {")
    for j in range(D):
        tot = sum((sv[i] * sp.diff(exprs[i], syms["x"][j]) for i in range(D)), sp.Integer(0))
        out.append("        case %d: { %s }" % (j, _emit_case(pr, tot, "return %s;")))
    out.append("        default: return 0.0;")
    out.append("        }")
    out.append("    }")
    # pgrad
    out.append("    static VA_HD void pgrad(const double *x, const double *s, int i, int, const double *p, double t, const double *st, double *acc)")
    out.append("    {")
    out.append("        (void)x; (void)s; (void)p; (void)t; (void)st; (void)acc;")
    out.append("        switch (i) {")
    for i, e in enumerate(exprs):
        stm = []
        for k in range(NP):
            dk = sp.diff(e, syms["p"][k])
            if dk != 0:
                stm.append("{ %s }" % _emit_case(pr, sv[i] * dk, "acc[%d] += %%s;" % k))
        out.append("        case %d: { %s break; }" % (i, " ".join(stm)))
        out[-1] = out[-1].replace("{  break; }", "{ break; }")
    out.append("        default: break;")
    out.append("        }")
    out.append("    }")


def _uniform_flat(exprs, syms, D, NP, pr0):
    """the same three functions for a TRANSLATION-INVARIANT model (every f_i is f_0 shifted cyclically) without a
    switch: one body, neighbours by cyclic index arithmetic.  A per-component switch diverges 64 ways in a wave of
    a wide state (D = 200: 22 ms per evaluation against 0.35 ms, tools/lin_wide.py).  None for any other model."""
    sp = _sympy()
    xs, ps = list(syms["x"]), list(syms["p"])
    if not _translation_invariant(exprs, xs, D):
        return None
    rel = lambda j: ((j + D // 2) % D) - D // 2
    f0 = exprs[0]
    used = sorted(rel(j) for j in range(D) if f0.has(xs[j]))
    base = pr0.__class__

    def printer(var, shift):
        class P(base):
            def _print_Symbol(self, sym):
                if sym in xs:
                    k = rel(xs.index(sym)) - shift
                    return "x[%s]" % var if k == 0 else "x[w(%s %s %d)]" % (var, "+" if k > 0 else "-", abs(k))
                return base._print_Symbol(self, sym)
        return P()
    out = ["    // translation-invariant: one body for every component, neighbours at cyclic offsets %s" % used,
           "    static VA_HD int w(int k) { return k < 0 ? k + D : (k >= D ? k - D : k); }",
           "    static VA_HD double f(const double *x, int i, int, const double *p, double t, const double *st)",
           "    {",
           "        (void)x; (void)i; (void)p; (void)t; (void)st;",
           "        " + _emit_case(printer("i", 0), f0, "return %s;"),
           "    }",
           "    static VA_HD double vjp(const double *x, const double *s, int j, int, const double *p, double t, const double *st)",
           "    {",
           "        (void)x; (void)s; (void)j; (void)p; (void)t; (void)st; double r = 0.0;"]
    for o in used:
        dk = sp.diff(f0, xs[o % D])
        if dk == 0:
            continue
        # row i = j - o holds the residual whose f reads x_j at offset o; its x_k sits at column j + (k - o)
        sname = "s[j]" if o == 0 else "s[w(j %s %d)]" % ("-" if o > 0 else "+", abs(o))
        out.append("        { %s }" % _emit_case(printer("j", o), sp.Symbol("SADJ") * dk, "r += %s;").replace("SADJ", sname))
    out += ["        return r;", "    }",
            "    static VA_HD void pgrad(const double *x, const double *s, int i, int, const double *p, double t, const double *st, double *acc)",
            "    {",
            "        (void)x; (void)s; (void)i; (void)p; (void)t; (void)st; (void)acc;"]
    for k in range(NP):
        dk = sp.diff(f0, ps[k])
        if dk != 0:
            out.append("        { %s }" % _emit_case(printer("i", 0), sp.Symbol("SADJ") * dk, "acc[%d] += %%s;" % k).replace("SADJ", "s[i]"))
    out.append("    }")
    return out


def generate_header(exprs, syms, D, NP, nstim, name="user", col=None, ghost=None, lin=None):
    """lin = (A0, rest) from linear_split: the struct then describes `rest` and carries the tables of A0"""
    sp = _sympy()
    pr = _printer()
    if lin is not None:
        exprs = lin[1]
    sv = sp.symbols("sv0:%d" % D, real=True)
    out = []
    out.append("// generated by varanneal_amd.codegen from the model function %r -- do not edit" % name)
    out.append("// D=%d NP=%d NSTIM=%d" % (D, NP, nstim))
    out.append("#pragma once")
    out.append("namespace va {")
    if lin is not None:
        DP, tables = _lin_tables(lin[0], D)
        out.append("// dense constant linear part A0 (%d non-zero entries of %d) and its transpose, padded to %d x %d"
                   % (int(np.count_nonzero(lin[0])), D * D, DP, DP))
        out.append(tables)
    out.append("struct RhsUser {")
    out.append("    static constexpr int NP = %d, D = %d, NSTIM = %d;" % (NP, D, nstim))
    if lin is not None:
        out.append("    // f = A0 x + rest: f / vjp / pgrad below are the REST; A0 runs on the matrix cores (va_eval_flat.h lin_gemm)")
        out.append("    static constexpr bool LINEAR = true;")
        out.append("    static constexpr int LIN_DP = %d;" % DP)
        out.append("    static VA_HD const double *lin_A0() { return va_lin_A0; }")
        out.append("    static VA_HD const double *lin_A0T() { return va_lin_A0T; }")
    uni = _uniform_flat(exprs, syms, D, NP, pr) if D >= 8 else None
    if uni is not None:
        out += uni
    else:
        _switch_flat(out, exprs, syms, D, NP, pr, sv)
    out.append("};")
    if col is not None:
        out.append("// the same model in column form (codegen.column_form): %s"
                   % ("translation-invariant stencil, offsets %s" % col["offsets"] if col["uniform"] else "dense, switch on the column"))
        out.append("#define VA_USER_COL 1")
        out.append(col["text"])
    if ghost is not None:
        out.append("// the same stencil in ghosted column form (codegen.ghost_form): offsets %s, %d ghost columns"
                   % (ghost["offsets"], ghost["GHOST"]))
        out.append("#define VA_USER_GHOST 1")
        out.append(ghost["text"])
    out.append("}  // namespace va")
    return "\n".join(out) + "\n"


def linear_split(exprs, syms, D, min_row_fill=8, max_D=512):
    """f = A0 x + rest: the dense CONSTANT linear part of a model (coefficients free of x, p, t and the stimulus),
    which the flat kernel evaluates -- and transposes for the adjoint -- on the matrix cores (csrc/va_eval_flat.h
    lin_gemm; BASELINE north_star: "MFMA only if the user RHS is a dense D x D linear map"), and the rest, which
    stays with the generated element-wise code.  Returns (A0 [D, D] float array, rest expressions), or None when the
    linear part is too sparse to be worth a matrix product (fewer than `min_row_fill` entries per row on average) or
    the state is too narrow (D < 16: one MFMA block) or too wide for the staged rows (D > max_D)."""
    sp = _sympy()
    if D < 16 or D > max_D:
        return None
    xidx = dict((x, j) for j, x in enumerate(syms["x"]))
    A0 = np.zeros((D, D))
    rest = []
    for i, e in enumerate(exprs):
        keep = []
        for term in sp.Add.make_args(sp.expand(e)):
            c, r = term.as_coeff_Mul()
            if r in xidx and c.is_number:
                A0[i, xidx[r]] += float(c)
            else:
                keep.append(term)
        rest.append(sp.Add(*keep) if keep else sp.Integer(0))
    if np.count_nonzero(A0) < min_row_fill * D:
        return None
    return A0, rest


def _lin_tables(A0, D):
    """the two zero-padded row-major tables the kernel reads (DP = D rounded up to 16)"""
    DP = ((D + 15) // 16) * 16
    pad = np.zeros((DP, DP))
    pad[:D, :D] = A0

    def table(name, M):
        rows = [", ".join(repr(float(v)) if v != 0.0 else "0" for v in row) for row in M]
        return "VA_LIN_TABLE double %s[%d] = {\n%s\n};" % (name, DP * DP, ",\n".join(rows))
    text = "\n".join([
        "#ifdef __HIPCC__", "#define VA_LIN_TABLE static __device__ const", "#else", "#define VA_LIN_TABLE static const", "#endif",
        table("va_lin_A0", pad), table("va_lin_A0T", pad.T)])
    return DP, text


def _translation_invariant(exprs, xs, D):
    """every f_i is f_0 with the state indices shifted cyclically by i (structural comparison first: expressions
    traced from one vectorised callable come out in the same form; SymPy's simplify only where that fails)"""
    sp = _sympy()
    if D < 3:
        return False
    for i in range(1, D):
        e = exprs[0].xreplace({xs[j]: xs[(j + i) % D] for j in range(D)})
        if e != exprs[i] and sp.simplify(e - exprs[i]) != 0:
            return False
    return True


def _period(exprs, xs, D, max_period=8):
    """smallest P > 1 dividing D such that every f_{i+P} is f_i with the state indices shifted cyclically by P (a ring of
    D / P identical units of P states each, e.g. coupled neurons), or None.  P = 1 is _translation_invariant's case."""
    sp = _sympy()
    for P in range(2, min(max_period, D // 2) + 1):
        if D % P:
            continue
        ok = True
        for i in range(D):
            e = exprs[i].xreplace({xs[j]: xs[(j + P) % D] for j in range(D)})
            if e != exprs[(i + P) % D] and sp.simplify(e - exprs[(i + P) % D]) != 0:
                ok = False
                break
        if ok:
            return P
    return None


def _local_printer(xmap):
    """C printer that writes state symbols as the column kernel sees them: the lane's own value `x0`
    and the neighbour values `xn[k]` (xmap: symbol -> C text)."""
    base = _printer().__class__

    class P(base):
        def _print_Symbol(self, s):
            if s in xmap:
                return xmap[s]
            return base._print_Symbol(self, s)
    return P()


def column_form(exprs, syms, D, NP, nstim, max_dense=8, uniform=None):
    """The model in the form the wave-private column-run kernel wants (csrc/va_tile4.h), or None.

    A lane owns ONE state column i of a run of time rows; it reads its own value and NB neighbour
    columns, evaluates f_i, and -- for the adjoint -- publishes the products s_i * df_i/dx_j of its own
    element, which the lanes of the columns j gather (J^T s without a second look at x).  Two kinds of
    model fit: (a) translation-invariant stencils (every f_i is f_0 shifted cyclically, e.g. Lorenz-96),
    where the neighbours are fixed column offsets and the code has one path; (b) small systems (D <= 8,
    e.g. the tutorial's NaKL neuron) with any coupling, where every other column is a neighbour and the
    code switches on the column.  Products that are constant multiples of one another are exchanged once."""
    sp = _sympy()
    xs, ps = list(syms["x"]), list(syms["p"])
    uses_t = any(e.has(syms["t"]) for e in exprs)

    uniform = _translation_invariant(exprs, xs, D) if uniform is None else uniform
    period = 1 if uniform else None
    if uniform:
        cols = [j for j in range(1, D) if sp.diff(exprs[0], xs[j]) != 0]
        offs = sorted(((j + D // 2) % D) - D // 2 for j in cols)          # signed cyclic offsets
        if len(set(o % D for o in offs)) != len(offs) or len(offs) > 8 or len(offs) == 0:
            uniform = False
            period = None
    if not uniform and D > max_dense:
        # a ring of identical units (period P): P column classes, each with its own body; the neighbour offsets are
        # the union over the classes, so the exchange of the adjoint products stays the uniform one
        P = _period(exprs, xs, D)
        if P is not None:
            offset_set = set()
            for i in range(P):
                for j in range(D):
                    if j != i and sp.diff(exprs[i], xs[j]) != 0:
                        offset_set.add(((j - i + D // 2) % D) - D // 2)
            offs = sorted(offset_set)
            if 0 < len(offs) <= 8 and len(set(o % D for o in offs)) == len(offs):
                period = P
    if not uniform and period is None:
        if D > max_dense or D < 2:
            return None
        offs = list(range(1, D))
    NB = len(offs)
    out = []
    s_sym = sp.Symbol("s", real=True)

    def local_printer(i):
        xmap = {xs[i]: "x0"}
        for k, o in enumerate(offs):
            xmap[xs[(i + o) % D]] = "xn[%d]" % k
        return _local_printer(xmap)

    rows = [0] if uniform else (list(range(period)) if period else list(range(D)))
    # products: per neighbour k the derivative df_i/dx_{i+off_k}; in the uniform case proportional ones share a slot
    derivs = {i: [sp.diff(exprs[i], xs[(i + o) % D]) for o in offs] for i in rows}
    slot_of, coef_of = list(range(NB)), [sp.Integer(1)] * NB
    if uniform:
        reps = []
        for k in range(NB):
            dk = derivs[0][k]
            hit = None
            for r in reps:
                if dk == 0 or derivs[0][r] == 0:
                    continue
                ratio = sp.simplify(dk / derivs[0][r])
                if ratio.is_number:
                    hit = (r, ratio)
                    break
            if hit is None:
                reps.append(k)
                slot_of[k], coef_of[k] = len(reps) - 1, sp.Integer(1)
            else:
                slot_of[k], coef_of[k] = reps.index(hit[0]), hit[1]
        NE = len(reps)
        e_src = reps
    else:
        NE, e_src = NB, list(range(NB))
    out.append("struct RhsUserCol {")
    out.append("    static constexpr int NP = %d, D = %d, NSTIM = %d, NB = %d, NE = %d, NG = %d;" % (NP, D, nstim, NB, NE, NB))
    # rows of an edge tile that do not exist are evaluated at x = 0 with a zero adjoint: their products are exact
    # zeros on their own when every expression is polynomial in x; otherwise (1/x, log x ...) the kernel selects
    poly = all(e.is_polynomial(*xs) for e in exprs)
    out.append("    static constexpr bool USES_T = %s, UNIFORM = %s, GUARD_EDGE = %s;"
               % ("true" if uses_t else "false", "true" if uniform else "false", "false" if poly else "true"))
    chain = lambda vals: " : ".join(["k == %d ? %s" % (k, v) for k, v in enumerate(vals[:-1])] + [str(vals[-1])]) if len(vals) > 1 else str(vals[0])
    out.append("    static VA_HD constexpr int nb_off(int k) { return %s; }" % chain(offs))
    out.append("    static VA_HD constexpr int g_e(int k) { return %s; }" % chain(slot_of))
    out.append("    static VA_HD constexpr int g_off(int k) { return %s; }" % chain([-o for o in offs]))
    sig = "int col, double x0, const double *xn, const double *p, double t, const double *st"
    unused = "(void)col; (void)x0; (void)xn; (void)p; (void)t; (void)st;"

    def emit_switch(body_for_row, default):
        if uniform:
            return ["        " + body_for_row(0)]
        lines = ["        switch (col %% %d) {" % period] if period else ["        switch (col) {"]
        for i in rows:
            lines.append("        case %d: { %s break; }" % (i, body_for_row(i)))
        lines.append("        default: { %s break; }" % default)
        lines.append("        }")
        return lines
    # f
    out.append("    static VA_HD double f(%s)" % sig)
    out.append("    {")
    out.append("        %s double r = 0.0;" % unused)
    out += emit_switch(lambda i: _emit_case(local_printer(i), exprs[i], "r = %s;"), "r = 0.0;")
    out.append("        return r;")
    out.append("    }")
    # scatter
    out.append("    static VA_HD void scatter(int col, double s, double x0, const double *xn, const double *p, double t, const double *st, double *e, double &diag)")
    out.append("    {")
    out.append("        %s (void)s;" % unused)

    def scat(i):
        pr = local_printer(i)
        parts = []
        for slot, k in enumerate(e_src):
            parts.append("{ %s }" % _emit_case(pr, s_sym * derivs[i][k], "e[%d] = %%s;" % slot))
        parts.append("{ %s }" % _emit_case(pr, s_sym * sp.diff(exprs[i], xs[i]), "diag = %s;"))
        return " ".join(parts)
    out += emit_switch(scat, " ".join("e[%d] = 0.0;" % k for k in range(NE)) + " diag = 0.0;")
    out.append("    }")
    # gather
    def term(k):
        # from the VALUE of the coefficient (text surgery on "1 * " would also hit 2.1, 11, ...)
        c = coef_of[k]
        if c == 1:
            return "r[%d]" % k
        if c == -1:
            return "-r[%d]" % k
        return "(%s) * r[%d]" % (_printer().doprint(sp.Float(c) if c.is_Float else c), k)
    terms = " + ".join(term(k) for k in range(NB))
    out.append("    static VA_HD double gather(const double *r) { return %s; }" % terms)
    # pgrad
    out.append("    static VA_HD void pgrad(int col, double s, double x0, const double *xn, const double *p, double t, const double *st, double *acc)")
    out.append("    {")
    out.append("        %s (void)s; (void)acc;" % unused)

    def pg(i):
        pr = local_printer(i)
        parts = []
        for k in range(NP):
            dk = sp.diff(exprs[i], ps[k])
            if dk != 0:
                parts.append("{ %s }" % _emit_case(pr, s_sym * dk, "acc[%d] += %%s;" % k))
        return " ".join(parts)
    out += emit_switch(pg, "")
    out.append("    }")
    out.append("};")
    xl, xr = max([0] + [-o for o in offs]), max([0] + offs)
    return dict(text="\n".join(out), uniform=uniform, period=period, offsets=offs, NE=NE, NB=NB, reach=(xl, xr, xr, xl),
                autonomous=not uses_t and nstim == 0)


def ghost_form(exprs, syms, D, NP, nstim, uniform=None):
    """A translation-invariant, autonomous stencil in the ghosted column form of the workgroup column-run
    kernel (csrc/va_tile3.h, k_eval3: wide states), or None.  A lane owns one column of rows staged with
    GHOST cyclic ghost columns on each side: f reads xc[off], and the adjoint is the GATHER
    (J^T s)_j = sum_off s_{j-off} (df_0/dx_off shifted to row j-off), which reads s and x up to
    max |k - off| columns away -- that, rounded up to even, is GHOST (at most 4)."""
    sp = _sympy()
    xs, ps = list(syms["x"]), list(syms["p"])
    if nstim > 0 or any(e.has(syms["t"]) for e in exprs) or D < 8:
        return None

    if not (_translation_invariant(exprs, xs, D) if uniform is None else uniform):
        return None
    rel = lambda j: ((j + D // 2) % D) - D // 2
    f0 = exprs[0]
    used = sorted(rel(j) for j in range(D) if f0.has(xs[j]))
    need = max([abs(o) for o in used] + [0])
    terms = []                                  # (off, derivative as an expression in x_k, k relative to column 0)
    for o in used:
        dk = sp.diff(f0, xs[o % D])
        if dk == 0:
            continue
        ks = [rel(j) for j in range(D) if dk.has(xs[j])]
        need = max([need, abs(o)] + [abs(k - o) for k in ks])
        terms.append((o, dk))
    G = need + (need & 1)
    if G < 2:
        G = 2
    if G > 4 or 2 * G >= D:
        return None
    base = _printer().__class__

    def printer(shift, own="xi"):
        class P(base):
            def _print_Symbol(self, sym):
                if sym in xs:
                    k = rel(xs.index(sym)) - shift
                    return own if (k == 0 and shift == 0) else "xc[%d]" % k
                return base._print_Symbol(self, sym)
        return P()
    out = ["struct RhsUserG {",
           "    static constexpr int NP = %d, D = %d, GHOST = %d;" % (NP, D, G),
           "    static VA_HD double f(const double *xc, double xi, const double *p)",
           "    {",
           "        (void)xc; (void)xi; (void)p; double r;",
           "        " + _emit_case(printer(0), f0, "r = %s;"),
           "        return r;",
           "    }",
           "    static VA_HD double vjp(const double *xc, const double *sc, double s_own, const double *p)",
           "    {",
           "        (void)xc; (void)sc; (void)s_own; (void)p; double r = 0.0;"]
    for o, dk in terms:
        sname = "s_own" if o == 0 else "sc[%d]" % (-o)
        # row i = j - off: its x_k sits at column k + i = j + (k - off)
        pr = printer(o, "xc[0]")
        out.append("        { %s }" % _emit_case(pr, sp.Symbol("SADJ") * dk, "r += %s;").replace("SADJ", sname))
    out += ["        return r;", "    }",
            "    static VA_HD void pgrad(const double *xc, double xi, double s_own, const double *p, double *acc)",
            "    {",
            "        (void)xc; (void)xi; (void)s_own; (void)p; (void)acc;"]
    for k in range(NP):
        dk = sp.diff(f0, ps[k])
        if dk != 0:
            out.append("        { %s }" % _emit_case(printer(0), sp.Symbol("SADJ") * dk, "acc[%d] += %%s;" % k).replace("SADJ", "s_own"))
    out += ["    }", "};"]
    return dict(text="\n".join(out), GHOST=G, offsets=used)


def check_against(f, exprs, syms, D, NP, nstim, stim_ndim=1, trials=3, rtol=1e-10, p_rows=False):
    """The traced expressions must reproduce `f` on random numeric slices."""
    sp = _sympy()
    args = list(syms["x"]) + list(syms["p"]) + list(syms["st"]) + [syms["t"]]
    fn = sp.lambdify(args, exprs, modules="numpy")
    rng = np.random.RandomState(20260102)
    for _ in range(trials):
        x = rng.rand(5, D) * 0.8 + 0.1
        p = rng.rand(5, NP) * 0.8 + 0.6 if p_rows else rng.rand(NP) * 0.8 + 0.6
        t = rng.rand(5)
        if nstim:
            st = rng.randn(5) if stim_ndim == 1 else rng.randn(5, nstim)
            want = np.asarray(f(t, x, (p, st)), dtype=np.float64)
        else:
            st = None
            want = np.asarray(f(t, x, p), dtype=np.float64)
        for r in range(5):
            srow = [] if not nstim else (list(np.atleast_1d(st[r])))
            got = np.array(fn(*(list(x[r]) + list(p[r] if p_rows else p) + srow + [t[r]])), dtype=np.float64)
            if not np.allclose(got, want[r], rtol=rtol, atol=1e-12):
                raise ValueError("traced model disagrees with the model function "
                                 "(does it branch on its inputs?): %s vs %s" % (got, want[r]))


BUILD_TAG = "sched=iterative-maxocc"      # part of every module's cache key: a flag change rebuilds them


def _core_fingerprint():
    h = hashlib.sha1(BUILD_TAG.encode())
    for fn in ("va_core.h", "va_device.h", "va_eval_flat.h", "va_eval3.h", "va_eval4.h", "va_epilogue.h", "va_tile2.h", "va_tile3.h",
               "va_tile4.h", "va_tile5.h", "va_eval5.h", "va_persist.h", "va_persist_geo.h", "va_measure.h", "va_user_rhs.hip"):
        with open(os.path.join(CSRC, fn), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def build_module(header_text, verbose=False, col_variant=None, compile=True):
    """Write the header, compile the module for gfx950 (cached by content).  Returns (so, header).
    col_variant = (eval kernel 3 | 4, disc, K, w): the ONE instantiation of a column-run kernel to compile in
    (models with a column / ghosted form; csrc/va_user_rhs.hip)."""
    os.makedirs(CACHE, exist_ok=True)
    defs = []
    if col_variant is not None:
        defs = ["-DVA_USER_EK=%d" % col_variant[0], "-DVA_USER_DISC=%d" % col_variant[1], "-DVA_USER_K=%d" % col_variant[2],
                "-DVA_USER_W=%d" % col_variant[3]]
    # (the header is named by its text alone: the same model built for several kernel variants shares it)
    hkey = hashlib.sha1((header_text + _core_fingerprint()).encode()).hexdigest()[:16]
    key = hashlib.sha1((header_text + _core_fingerprint() + " ".join(defs)).encode()).hexdigest()[:16]
    hdr = os.path.join(CACHE, "rhs_%s.h" % hkey)
    so = os.path.join(CACHE, "libva_rhs_%s.so" % key)
    # Several ranks may build the same module at once (one process per GPU, each calling
    # anneal_init): every process writes to names of its own and publishes with an atomic rename, so
    # nobody ever compiles a half-written header or loads a half-written library.
    tag = ".%d.tmp" % os.getpid()
    if not os.path.exists(hdr):
        with open(hdr + tag, "w") as fh:
            fh.write(header_text)
        os.replace(hdr + tag, hdr)
    if not compile:
        return None, hdr                                   # (header only: the CPU emulator of the tests compiles it itself)
    if not os.path.exists(so):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm",
               "-amdgpu-sched-strategy=iterative-maxocc",            # (as the library: varanneal_amd/_build.py)
               "-Wno-unused-function", "-I", CSRC, '-DVA_USER_RHS_HEADER="%s"' % hdr] + defs + [
               "-o", so + tag, os.path.join(CSRC, "va_user_rhs.hip")]
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            subprocess.check_call(cmd)
            os.replace(so + tag, so)
        finally:
            if os.path.exists(so + tag):
                os.remove(so + tag)
    return so, hdr


def module_for(f, D, NP, nstim=0, stim_ndim=1, verbose=False, p_rows=False, col_variant=None, compile=True, linear=True):
    """trace + check + generate + build.  Returns dict(so=, header=, exprs=, col=).
    col_variant: None, or a callable (NE, GHOST[, reach]) -> (eval kernel 3 | 4 | 5, disc, K, w) or None (as
    _capi.eval_plan returns it) naming the instantiation of a column-run kernel to compile for a model that has
    a column form (NE products per element; 0 = none) and / or a ghosted form (GHOST columns; 0 = none).
    linear=False keeps a dense constant linear part in the element-wise code (linear_split; for comparisons)."""
    if NP > MAX_NP:
        raise NotImplementedError("right-hand sides with more than %d parameters" % MAX_NP)
    exprs, syms = trace(f, D, NP, nstim, stim_ndim, p_rows)
    check_against(f, exprs, syms, D, NP, nstim, stim_ndim, p_rows=p_rows)
    col = ghost = None
    variant = None
    if p_rows and NP > MAX_NP_TUNED:
        raise NotImplementedError("time-dependent parameters: at most %d of them" % MAX_NP_TUNED)
    if col_variant is not None and not p_rows and NP <= MAX_NP_TUNED:      # (many parameters: the flat kernel only)
        uniform = _translation_invariant(exprs, list(syms["x"]), D)
        # column form: k_eval4 (D <= 64: stencils and small dense systems) or, for stencils, the streaming k_eval5
        col = column_form(exprs, syms, D, NP, nstim, uniform=uniform)          # (None when the model has no such form)
        ghost = ghost_form(exprs, syms, D, NP, nstim, uniform=uniform)
        variant = None
        if col or ghost:
            ne, gh = (col["NE"] if col else 0), (ghost["GHOST"] if ghost else 0)
            import inspect
            try:
                three = len(inspect.signature(col_variant).parameters) >= 3
            except (TypeError, ValueError):
                three = False
            # (a callback of three arguments also gets the column form's reaches, or None when the form is not one
            # the streaming kernel can run: non-uniform, explicit time, stimulus)
            reach = col["reach"] if (col and (col["uniform"] or col["period"]) and col["autonomous"]) else None
            variant = col_variant(ne, gh, reach) if three else col_variant(ne, gh)
        if variant is None or variant[0] not in (4, 5):
            col = None
        if variant is None or variant[0] != 3:
            ghost = None
    # no column-run kernel for this model: a dense constant linear part goes to the matrix cores
    lin = linear_split(exprs, syms, D) if (linear and col is None and ghost is None and not p_rows and NP <= MAX_NP_TUNED) else None
    text = generate_header(exprs, syms, D, NP, nstim, getattr(f, "__name__", "f"), col=col, ghost=ghost, lin=lin)
    so, hdr = build_module(text, verbose, variant, compile)
    return dict(so=so, header=hdr, exprs=exprs, text=text, col=col, ghost=ghost, col_variant=variant,
                lin=(None if lin is None else lin[0]))


# ---------------------------------------------------------------------------------------------------------
# Activations of the network action (varanneal/va_nnet.py:71, 260-264: any callable f(x, W, b) is the layer map)
# ---------------------------------------------------------------------------------------------------------
def trace_activation(f):
    """The elementwise g of a layer map f(x, W, b) = g(W.x + b), as a SymPy expression in `z`, with its
    derivative.  Raises TypeError if `f` is not of that form (checked numerically on random W, x, b) or
    branches on values."""
    sp = _sympy()
    z0, z1 = sp.symbols("z0 z1", real=True)
    x = np.empty(2, dtype=object); x[0] = Sym(z0); x[1] = Sym(z1)
    W = np.array([[1, 0], [0, 1]], dtype=object)
    b = np.array([0, 0], dtype=object)
    out = np.asarray(f(x, W, b), dtype=object)
    if out.shape != (2,):
        raise TypeError("activation returned shape %s for a 2-neuron layer" % (out.shape,))
    g0, g1 = sp.sympify(Sym._u(out[0])), sp.sympify(Sym._u(out[1]))
    if (g0.free_symbols - {z0}) or sp.simplify(g0.xreplace({z0: z1}) - g1) != 0:
        raise TypeError("the layer map is not an elementwise function of W.x + b")
    z = sp.Symbol("z", real=True)
    g = g0.xreplace({z0: z})
    dg = sp.diff(g, z)
    gn = sp.lambdify(z, g, "numpy")
    rng = np.random.RandomState(20260103)
    for _ in range(3):
        xv = rng.randn(5); Wv = rng.randn(4, 5); bv = rng.randn(4)
        got = np.asarray(f(xv, Wv, bv), dtype=np.float64)
        want = np.asarray(gn(np.dot(Wv, xv) + bv), dtype=np.float64)
        if got.shape != (4,) or not np.allclose(got, want, rtol=1e-12, atol=1e-12):
            raise TypeError("the layer map is not g(W.x + b) with an elementwise g")
    return g, dg, z


def activation_header(g, dg, z, name="act"):
    base = _printer().__class__

    class P(base):
        def _print_Symbol(self, sym):
            return "z" if sym == z else base._print_Symbol(self, sym)
    pr = P()
    out = ["// generated by varanneal_amd/codegen.py from the user's layer map `%s`: g(z) = %s" % (name, str(g)[:160]),
           "#pragma once", "namespace va {", "struct ActUser {",
           "    static __device__ __forceinline__ double f(double z)", "    {",
           "        double r;", "        " + _emit_case(pr, g, "r = %s;"), "        return r;", "    }",
           "    // f'(z); a = f(z) is at hand but not used by generated code",
           "    static __device__ __forceinline__ double d(double z, double a)", "    {",
           "        (void)a; double r;", "        " + _emit_case(pr, dg, "r = %s;"), "        return r;", "    }",
           "};", "}  // namespace va"]
    return "\n".join(out) + "\n"


def _act_fingerprint():
    h = hashlib.sha1(BUILD_TAG.encode())
    for fn in ("va_core.h", "va_device.h", "va_eval_flat.h", "va_epilogue.h", "va_nnet.h", "va_nnet_kernels.h",
               "va_user_act.hip"):
        with open(os.path.join(CSRC, fn), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def activation_module_for(f, verbose=False):
    """trace + generate + build an activation module.  Returns dict(so=, header=, g=, dg=, z=)."""
    g, dg, z = trace_activation(f)
    text = activation_header(g, dg, z, getattr(f, "__name__", "f"))
    os.makedirs(CACHE, exist_ok=True)
    key = hashlib.sha1((text + _act_fingerprint()).encode()).hexdigest()[:16]
    hdr = os.path.join(CACHE, "act_%s.h" % key)
    so = os.path.join(CACHE, "libva_act_%s.so" % key)
    tag = ".%d.tmp" % os.getpid()
    if not os.path.exists(hdr):
        with open(hdr + tag, "w") as fh:
            fh.write(text)
        os.replace(hdr + tag, hdr)
    if not os.path.exists(so):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function",
               "-mllvm", "-amdgpu-sched-strategy=iterative-maxocc",
               "-I", CSRC, '-DVA_USER_ACT_HEADER="%s"' % hdr, "-o", so + tag, os.path.join(CSRC, "va_user_act.hip")]
        if verbose:
            print(" ".join(cmd), flush=True)
        try:
            subprocess.check_call(cmd)
            os.replace(so + tag, so)
        finally:
            if os.path.exists(so + tag):
                os.remove(so + tag)
    return dict(so=so, header=hdr, g=g, dg=dg, z=z, text=text)
