"""State and parameter estimation in Lorenz 96 by variational annealing on MI355X.

Counterpart of the reference's examples/Lorenz96_D20/Lorenz96_anneal.py (SURVEY.md
8(a) row H): same model, hyper-parameters, call order and output files; the only
edits are the import (varanneal_amd instead of varanneal), an explicit RNG seed, and
the data file location.  Optional: --seeds B runs B random initial paths as one batch.

    python examples/Lorenz96_D20/Lorenz96_anneal.py [--seeds 1] [--nbeta 101] [--disc SimpsonHermite]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from varanneal_amd import va_ode  # noqa: E402


# Define the model (reference script, line 15-16)
def l96(t, x, k):
    return np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - x + k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=1)
    ap.add_argument("--nbeta", type=int, default=101)
    ap.add_argument("--disc", default="SimpsonHermite")
    ap.add_argument("--rng", type=int, default=12345)
    ap.add_argument("--out", default=".")
    args = ap.parse_args()

    D = 20
    # Measured variable indices, RM, RF0, alpha and the beta ladder (reference lines 22-30)
    Lidx = [0, 2, 4, 6, 8, 10, 14, 16]
    RM = 1.0 / (0.5 ** 2)
    RF0 = 4.0e-6
    alpha = 1.5
    beta_array = np.linspace(0, args.nbeta - 1, args.nbeta)

    # Observed data: the file the reference ships (a copy lives in tests/golden/)
    here = os.path.dirname(os.path.abspath(__file__))
    data = np.load(os.path.join(here, "..", "..", "tests", "golden",
                                "l96_D20_dt0p025_N161_sm0p5_sec1_mem1.npy"))
    times_data = data[:, 0]
    dt_data = times_data[1] - times_data[0]
    N_data = len(times_data)
    data = data[:, 1:][:, Lidx]

    # Initial path/parameter guesses (reference lines 49-68)
    dt_model = dt_data
    N_model = N_data
    rng = np.random.RandomState(args.rng)
    if args.seeds == 1:
        X0 = (20.0 * rng.rand(N_model * D) - 10.0).reshape((N_model, D))
        P0 = np.array([4.0 * rng.rand() + 6.0])
    else:
        X0 = (20.0 * rng.rand(args.seeds, N_model * D) - 10.0).reshape((args.seeds, N_model, D))
        P0 = 4.0 * rng.rand(args.seeds, 1) + 6.0
    Pidx = [0]

    anneal1 = va_ode.Annealer()
    anneal1.set_model(l96, D)
    anneal1.set_data(data, t=times_data)

    BFGS_options = {'gtol': 1.0e-8, 'ftol': 1.0e-8, 'maxfun': 1000000, 'maxiter': 1000000}
    tstart = time.time()
    anneal1.anneal(X0, P0, alpha, beta_array, RM, RF0, Lidx, Pidx, dt_model=dt_model,
                   init_to_data=True, disc=args.disc, method='L-BFGS-B',
                   opt_args=BFGS_options, adolcID=0)
    print("\nHIP annealing completed in %f s." % (time.time() - tstart))
    print("estimated forcing k (true value 8.17): %s" % np.ravel(anneal1.P))

    anneal1.save_paths(os.path.join(args.out, "paths.npy"))
    anneal1.save_params(os.path.join(args.out, "params.npy"))
    anneal1.save_action_errors(os.path.join(args.out, "action_errors.npy"))


if __name__ == "__main__":
    main()
