"""CPU: the C-ABI library loads and exports every symbol include/varanneal_amd.h
declares; argument validation that needs no GPU; the ctypes structs match the header."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "varanneal_amd.h")


@pytest.fixture(scope="module")
def capi():
    from varanneal_amd import _build, _capi
    _build.build(verbose=False)
    return _capi


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(va_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(capi):
    names = declared_functions()
    assert len(names) >= 12
    lib = capi.lib()
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(capi.EXPORTS) == names
    assert lib.va_abi_version() == capi.ABI_VERSION == 12


def test_rhs_module_loader_rejects_bad_paths(capi, tmp_path):
    lib = capi.lib()
    rid = C.c_int(-7)
    assert lib.va_rhs_load_module(None, C.byref(rid)) == -1
    assert lib.va_rhs_load_module(str(tmp_path / "nope.so").encode(), C.byref(rid)) != 0
    assert lib.va_last_error()
    # a shared object without the module entry points is refused, not half-registered
    src = tmp_path / "e.c"
    src.write_text("int unrelated(void){return 0;}\n")
    so = tmp_path / "e.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    assert lib.va_rhs_load_module(str(so).encode(), C.byref(rid)) != 0
    assert b"va_user" in lib.va_last_error()


def test_struct_layout_matches_header(capi, tmp_path):
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "varanneal_amd.h"\n'
                    'int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(va_problem_desc), sizeof(va_lbfgs_opts),'
                    ' offsetof(va_problem_desc, rf0_array), offsetof(va_problem_desc, stream),'
                    ' offsetof(va_lbfgs_opts, maxfun), sizeof(va_nnet_desc), offsetof(va_nnet_desc, rf0),'
                    ' offsetof(va_nnet_desc, stream)); return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(prog)])
    out = subprocess.check_output([str(exe)]).split()
    assert int(out[0]) == C.sizeof(capi.ProblemDesc)
    assert int(out[1]) == C.sizeof(capi.LbfgsOpts)
    assert int(out[2]) == capi.ProblemDesc.rf0_array.offset
    assert int(out[3]) == capi.ProblemDesc.stream.offset
    assert int(out[4]) == capi.LbfgsOpts.maxfun.offset
    assert int(out[5]) == C.sizeof(capi.NnetDesc)
    assert int(out[6]) == capi.NnetDesc.rf0.offset and int(out[7]) == capi.NnetDesc.stream.offset


def test_validation_without_gpu(capi):
    lib = capi.lib()
    h = C.c_void_p()
    d = capi.ProblemDesc()
    d.struct_size = 3
    assert lib.va_problem_create(C.byref(d), C.byref(h)) == -1          # VA_EINVAL
    assert b"struct_size" in lib.va_last_error()
    assert lib.va_problem_create(None, C.byref(h)) == -1
    assert lib.va_problem_info(None, None, None, None, None) == -1
    assert lib.va_device_count(None) == -1
    lib.va_problem_destroy(None)                                         # no-op


def test_default_options_are_scipys(capi):
    o = capi.make_opts(None)
    assert (o.maxcor, o.maxiter, o.maxfun, o.maxls) == (10, 15000, 15000, 20)
    assert o.gtol == 1e-5 and abs(o.ftol - 2.2204460492503131e-09) < 1e-24
    o = capi.make_opts({'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000})
    assert (o.gtol, o.ftol, o.maxfun, o.maxiter) == (1e-8, 1e-8, 1000000, 1000000)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "varanneal_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "va_oracle" not in txt and "cpu_emul" not in txt.replace("tests/cpu_emul", ""), f


def test_nnet_validation_without_gpu(capi):
    """va_nnet_problem_create checks its descriptor before it touches a device."""
    import numpy as np
    lib = capi.lib()
    h = C.c_void_p()
    st = [3, 4, 2]
    NP = 3 * 4 + 4 + 4 * 2 + 2

    def desc(**kw):
        args = dict(batch=1, structure=st, data_in=np.zeros((2, 3)), data_out=np.zeros((2, 2)),
                    Lidx=[np.arange(3), np.arange(2)], RM=1.0, RF0=0.1, P=np.zeros(NP), Pidx=[0, 1])
        args.update(kw)
        return capi.make_nnet_desc(**args)
    d, keep = desc()
    d.struct_size = 8
    assert lib.va_nnet_problem_create(C.byref(d), C.byref(h)) == -1 and b"struct_size" in lib.va_last_error()
    d, keep = desc(P=np.zeros(NP + 1))
    assert lib.va_nnet_problem_create(C.byref(d), C.byref(h)) == -1 and b"weights and biases" in lib.va_last_error()
    d, keep = desc(Pidx=[0, 0])
    assert lib.va_nnet_problem_create(C.byref(d), C.byref(h)) == -1 and b"twice" in lib.va_last_error()
    d, keep = desc(Pidx=[NP])
    assert lib.va_nnet_problem_create(C.byref(d), C.byref(h)) == -1
    d, keep = desc(Lidx=[np.array([0, 1, 5]), np.arange(2)])
    assert lib.va_nnet_problem_create(C.byref(d), C.byref(h)) == -1 and b"input layer" in lib.va_last_error()
    d, keep = desc()
    d.activation = 17
    assert lib.va_nnet_problem_create(C.byref(d), C.byref(h)) == -4          # VA_EUNSUPPORTED
    with pytest.raises(ValueError):
        desc(RM=np.eye(3))                     # neither [RM_in, RM_out] nor two matrices of the observed sizes
    with pytest.raises(NotImplementedError):
        desc(act="sine")
    with pytest.raises(ValueError):
        desc(data_in=np.zeros((2, 2)))


def test_time_dependent_validation_without_gpu(capi):
    import numpy as np
    lib = capi.lib()
    h = C.c_void_p()
    N, D = 6, 5
    d, keep = capi.make_desc(1, D, N, np.zeros((N, 2)), [0, 1], 0.1, 1.0, 1.0, np.ones((1, N, 1)), [0], disc="euler",
                             p_time_dependent=True)
    assert lib.va_problem_create(C.byref(d), C.byref(h)) == -4 and b"trapezoid and SimpsonHermite" in lib.va_last_error()
    with pytest.raises(ValueError):
        capi.make_desc(1, D, N, np.zeros((N, 2)), [0, 1], 0.1, 1.0, 1.0, np.ones((1, N - 1, 1)), [0],
                       p_time_dependent=True)
