"""Shared helpers for the test-suite (golden loading, oracle construction)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz_cases(fname):
    z = np.load(os.path.join(GOLD, fname), allow_pickle=False)
    cases = {}
    for key in z.files:
        c, k = key.split("/", 1)
        v = z[key]
        cases.setdefault(c, {})[k] = v.item() if v.ndim == 0 else v
    return cases


def rm_rf_for(case):
    """Expand the golden case's RM/RF0 (scalar | (L,) | (D,)) the way
    va_ode.py:612-640 does (np.resize over time)."""
    N, D = int(case["N_model"]), int(case["D"])
    Y = case["Y"]
    RM, RF0 = case["RM"], case["RF0"]
    RM = float(RM) if np.ndim(RM) == 0 else np.resize(RM, Y.shape)
    RF0 = float(RF0) if np.ndim(RF0) == 0 else np.resize(RF0, (N - 1, D))
    return RM, RF0


def oracle_problem(case):
    import va_oracle
    N, D = int(case["N_model"]), int(case["D"])
    RM, RF0 = rm_rf_for(case)
    P = case["XP"][N * D:]
    return va_oracle.Problem(D, N, case["Y"], case["Lidx"], case["dt_model"], RM, RF0, P, [0],
                             disc=str(case["disc"]), merr_nskip=int(case["merr_nskip"]))
