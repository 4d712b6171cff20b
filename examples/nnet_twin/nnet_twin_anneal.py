""""Twin" neural-network training by variational annealing on MI355X.

Counterpart of the reference's examples/nnet_twin/nnet_twin_anneal.py: same network
(20 layers of 10 sigmoid neurons), M = 2 training pairs, weights estimated / biases fixed at
zero, RM, RF0, alpha = 1.1, beta = 0..435, L-BFGS-B options and output files.  The edits: the
import (varanneal_amd instead of varanneal), explicit RNG seeds, and the training pairs -- the
reference ships only the scripts that generate them (data/gen_params.py, data/gen_io_pairs.py),
so varanneal_amd.twin.make_nnet_twin follows the same recipe.

    python examples/nnet_twin/nnet_twin_anneal.py [--seeds 1] [--nbeta 436] [--M 2]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from varanneal_amd import twin, va_nnet  # noqa: E402


# Define the transfer function (reference script, lines 20-22)
def sigmoid(x, W, b):
    linpart = np.dot(W, x) + b
    return 1.0 / (1.0 + np.exp(-linpart))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=1, help="initial guesses annealed as one batch")
    ap.add_argument("--nbeta", type=int, default=436)
    ap.add_argument("--M", type=int, default=2, help="training examples")
    ap.add_argument("--out", default=".")
    args = ap.parse_args()

    N, D_in, D_out, D_hidden = 20, 10, 10, 10
    structure = twin.nnet_structure(N, D_in, D_out, D_hidden)
    L = 10
    Lidx = [np.linspace(0, L - 1, L, dtype='int'), np.linspace(0, L - 1, L, dtype='int')]

    RM = 1.0 / (0.005 ** 2)
    RF0 = 1.0e-8 * RM * float(np.sum(structure) - structure[0]) / float(structure[0] + structure[-1])
    alpha = 1.1
    beta_array = np.linspace(0, args.nbeta - 1, args.nbeta)

    data_in, data_out, _ = twin.make_nnet_twin(structure, args.M)
    data_in, data_out = data_in[:, Lidx[0]], data_out[:, Lidx[1]]

    guesses = [twin.nnet_initial_guess(structure, args.M, s) for s in range(args.seeds)]
    Pidx = guesses[0][2]
    if args.seeds == 1:
        X0, P0 = guesses[0][0], guesses[0][1]
    else:
        X0 = np.array([g[0] for g in guesses]); P0 = np.array([g[1] for g in guesses])

    anneal1 = va_nnet.Annealer()
    anneal1.set_structure(structure)
    anneal1.set_activation(sigmoid)
    anneal1.set_input_data(data_in)
    anneal1.set_output_data(data_out)

    BFGS_options = {'gtol': 1.0e-12, 'ftol': 1.0e-12, 'maxfun': 1000000, 'maxiter': 1000000}
    tstart = time.time()
    anneal1.anneal(X0, P0, alpha, beta_array, RM, RF0, Pidx, Lidx=Lidx,
                   method='L-BFGS-B', opt_args=BFGS_options, adolcID=0)
    print("\nAnnealing completed in %f s." % (time.time() - tstart))

    if args.seeds == 1:
        anneal1.save_io(os.path.join(args.out, "io.npy"))
        anneal1.save_Wb(os.path.join(args.out, "W.npy"), os.path.join(args.out, "b.npy"))
        anneal1.save_action_errors(os.path.join(args.out, "aerr.npy"))
    else:
        np.save(os.path.join(args.out, "A_array.npy"), anneal1.A_array)
        print("final actions per seed:", anneal1.A_array[:, -1])


if __name__ == "__main__":
    main()
