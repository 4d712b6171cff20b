"""Twin-experiment data for Lorenz-96 (the reference ships only one N=161 file,
examples/Lorenz96_D20/l96_D20_dt0p025_N161_sm0p5_sec1_mem1.npy, and no
generator).  Used by bench.py, the tests and examples to build the BASELINE
configs (D=20 N=200/1000, D=200 N=5000) with identical bytes on every box.

Model (examples/Lorenz96_D20/Lorenz96_anneal.py:15-16):
    dx_i/dt = x_{i-1} (x_{i+1} - x_{i-2}) - x_i + k,  cyclic in i.
"""
import numpy as np

K_TRUE = 8.17       # examples/jupyter-tutorial/Lorenz96/data file names ("k8p17")
DT = 0.025          # dt of the shipped data file
SIGMA = 0.5         # "sm0p5"
GEN_SEED = 20260101
README_LIDX_D20 = [0, 2, 4, 8, 10, 14, 16]      # README.md:105 (L=7)


def l96(t, x, k):
    """The reference example's RHS, verbatim semantics (rows = time points)."""
    return np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - x + k


def _f1(x, k):
    return np.roll(x, 1) * (np.roll(x, -1) - np.roll(x, 2)) - x + k


def integrate_l96(D, N, dt=DT, k=K_TRUE, seed=GEN_SEED, spinup=2000):
    """RK4 trajectory (N, D) sampled every dt after a transient."""
    rng = np.random.RandomState(seed)
    x = k * np.ones(D) + 0.01 * rng.randn(D)
    out = np.empty((N, D))
    for n in range(-spinup, N):
        if n >= 0:
            out[n] = x
        k1 = _f1(x, k); k2 = _f1(x + 0.5 * dt * k1, k)
        k3 = _f1(x + 0.5 * dt * k2, k); k4 = _f1(x + dt * k3, k)
        x = x + dt * (k1 + 2 * k2 + 2 * k3 + k4) / 6.0
    return out


def default_lidx(D):
    """L=7 of D=20 (README.md:105); for other D every index with i%5 in {0,2}
    (L=80 of D=200)."""
    if D == 20:
        return list(README_LIDX_D20)
    return [i for i in range(D) if i % 5 in (0, 2)]


def make_twin(D, N, Lidx=None, dt=DT, k=K_TRUE, sigma=SIGMA, seed=GEN_SEED):
    """Returns (t (N,), Y (N, L) noisy observations, truth (N, D), Lidx)."""
    Lidx = default_lidx(D) if Lidx is None else list(Lidx)
    truth = integrate_l96(D, N, dt, k, seed)
    rng = np.random.RandomState(seed + 1)
    Y = truth[:, Lidx] + sigma * rng.randn(N, len(Lidx))
    t = dt * np.arange(N)
    return t, Y, truth, Lidx


def initial_guess(N, D, seed_index, Y=None, Lidx=None, nskip=1):
    """X0 ~ U(-10,10), P0 ~ U(6,10) (Lorenz96_anneal.py:51,68) from
    RandomState(1000+seed_index); optionally init_to_data (va_ode.py:677-678)."""
    rng = np.random.RandomState(1000 + int(seed_index))
    X0 = (20.0 * rng.rand(N * D) - 10.0).reshape(N, D)
    P0 = np.array([4.0 * rng.rand() + 6.0])
    if Y is not None:
        X0[::nskip, Lidx] = Y
    return X0, P0


# ---------------------------------------------------------------------------------------
# Feed-forward "twin" network data (examples/nnet_twin).  The reference ships the two
# generator scripts (data/gen_params.py, data/gen_io_pairs.py) but none of their output, so
# the BASELINE C5 config needs a generator of our own following the same recipe: weights
# U(-1,1)/fan_in, zero biases, sigmoid layers, inputs standardised N(0,1) draws, Gaussian
# noise of width sigma on input and output, output clipped to (1e-4, 1-1e-4).
NNET_SIGMA = 0.005      # "sm0p005" (nnet_twin_anneal.py:11, gen_io_pairs.py:20)


def sigmoid(x, W, b):
    """examples/nnet_twin/nnet_twin_anneal.py:20-22"""
    return 1.0 / (1.0 + np.exp(-(np.dot(W, x) + b)))


def nnet_structure(N=20, D_in=10, D_out=10, D_hidden=10):
    s = np.full(N, D_hidden, dtype=int)
    s[0], s[-1] = D_in, D_out
    return s


def nnet_param_layout(structure):
    """(woff, boff, NP): offsets of W_n (s[n+1] x s[n], row-major) and b_n in the flat
    parameter vector (va_nnet.py:194-207)."""
    woff, boff, o = [], [], 0
    for n in range(len(structure) - 1):
        woff.append(o); o += int(structure[n + 1]) * int(structure[n])
        boff.append(o); o += int(structure[n + 1])
    return woff, boff, o


def make_nnet_twin(structure, M, sigma=NNET_SIGMA, seed=GEN_SEED):
    """Returns (data_in (M, s0), data_out (M, s_last), P_true (NP,))."""
    structure = np.asarray(structure, dtype=int)
    rng = np.random.RandomState(seed + 7)
    woff, boff, NP = nnet_param_layout(structure)
    P = np.zeros(NP)
    for n in range(len(structure) - 1):
        nw = structure[n + 1] * structure[n]
        P[woff[n]:woff[n] + nw] = (2.0 * rng.rand(nw) - 1.0) / float(structure[n])
    din = np.empty((M, structure[0])); dout = np.empty((M, structure[-1]))
    for m in range(M):
        y = rng.randn(structure[0])
        y = (y - np.average(y)) / np.std(y)
        x = y
        for n in range(len(structure) - 1):
            W = P[woff[n]:boff[n]].reshape(structure[n + 1], structure[n])
            x = sigmoid(x, W, P[boff[n]:boff[n] + structure[n + 1]])
        din[m] = y + sigma * rng.randn(structure[0])
        dout[m] = np.clip(x + sigma * rng.randn(structure[-1]), 0.0001, 0.9999)
    return din, dout, P


def nnet_initial_guess(structure, M, seed_index, weights_only=True):
    """Initial states/parameters drawn as nnet_twin_anneal.py:67-119 draws them (inputs
    standardised N(0,1), other layers U(0.4,0.6), weights U(-1,1)/fan_in, biases 0) from
    RandomState(1000+seed_index).  Returns (X0 (M*NDnet,), P0 (NP,), Pidx)."""
    structure = np.asarray(structure, dtype=int)
    rng = np.random.RandomState(1000 + int(seed_index))
    X0 = []
    for m in range(M):
        xin = rng.randn(structure[0])
        X0.append((xin - np.average(xin)) / np.std(xin))
        for n in range(1, len(structure)):
            X0.append(0.2 * rng.rand(structure[n]) + 0.4)
    X0 = np.concatenate(X0)
    woff, boff, NP = nnet_param_layout(structure)
    P0 = np.zeros(NP); Pidx = []
    for n in range(len(structure) - 1):
        nw = structure[n + 1] * structure[n]
        P0[woff[n]:woff[n] + nw] = (2.0 * rng.rand(nw) - 1.0) / float(structure[n])
        Pidx += list(range(woff[n], woff[n] + nw))
        if not weights_only:
            Pidx += list(range(boff[n], boff[n] + int(structure[n + 1])))
    return X0, P0, Pidx


def dense_coupling_model(D, seed=1):
    """a model with a dense constant linear part: f = C x - p1 x^3 + p0 with a fixed D x D matrix C (NOT from the
    reference: the shape BASELINE north_star names for the matrix cores, "a dense D x D linear map"); bench.py
    extra.lin_d20, tools/lin_wide.py, __graft_entry__.build().  Returns (f, C)."""
    C = np.random.RandomState(seed).randn(D, D) / np.sqrt(D)

    def coupled(t, x, p):
        return x @ C.T - p[1] * x ** 3 + p[0]
    return coupled, C
