#!/usr/bin/env python3
"""bench.py -- (A, grad A) evaluations per second of the variational-annealing action
on MI355X, against the HBM roofline, with the CPU oracle timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c4|c2] [--no-cpu]

A "step" is ONE complete batched evaluation -- the launch va_action_grad makes: the action
A(X,p) (me + fe formed on the device by the last-arriving workgroup of each seed) and its full
gradient for all B resident seeds.  Paths are resident in HBM before the timed region.
The default run (1 GPU) also reports, in `extra`: the C4 shape, the end-to-end C3 ladder with
the two L-BFGS vector kernels priced against the roofline, and the CPU oracle on all host cores.  N>1: one process per GPU (torch.distributed over RCCL), B seeds per
GPU (weak scaling), no data-path collective; the job's one all_gather of the per-seed
actions is issued after the K timed steps and timed separately (config.final_gather_ms).

Workloads (BASELINE.json configs):
  c3 (default)  Lorenz-96 D=20,  N=1000, L=7,  B=64 seeds per GPU, trapezoid
  c4            Lorenz-96 D=200, N=5000, L=80, B=64 seeds per GPU, trapezoid
  c2            Lorenz-96 D=20,  N=1000, L=7,  B=1  (parity config; launch-bound)
  c5            va_nnet twin: structure [10]*20, M=2 examples, 1900 weights estimated, B=64 seeds
  c5x           the same action scaled to where the layer products fill the matrix cores:
                structure [128]*8, M=2048 examples, B=16 seeds (roofline bound: f64 MFMA)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "c3": dict(D=20, N=1000, B=64, name="lorenz96_D20_N1000_L7_B64_trapezoid"),
    "c4": dict(D=200, N=5000, B=64, name="lorenz96_D200_N5000_L80_B64_trapezoid"),
    "c2": dict(D=20, N=1000, B=1, name="lorenz96_D20_N1000_L7_B1_trapezoid"),
    # other state widths, to place the switches between workgroup geometries (not BASELINE configs)
    "w100": dict(D=100, N=5000, B=64, name="lorenz96_D100_N5000_B64_trapezoid"),
    "w300": dict(D=300, N=3000, B=64, name="lorenz96_D300_N3000_B64_trapezoid"),
    "w500": dict(D=500, N=2000, B=64, name="lorenz96_D500_N2000_B64_trapezoid"),
    "w900": dict(D=900, N=1000, B=64, name="lorenz96_D900_N1000_B64_trapezoid"),
    # batch-scaling points of the C3 shape (launch overhead amortised; not BASELINE configs)
    "c3x4": dict(D=20, N=1000, B=256, name="lorenz96_D20_N1000_L7_B256_trapezoid"),
    "c3x16": dict(D=20, N=1000, B=1024, name="lorenz96_D20_N1000_L7_B1024_trapezoid"),
    "c3x64": dict(D=20, N=1000, B=4096, name="lorenz96_D20_N1000_L7_B4096_trapezoid"),
}
NNET_WORKLOADS = {
    "c5": dict(structure=[10] * 20, M=2, B=64, name="nnet_twin_10x20_M2_B64_sigmoid"),
    "c5x": dict(structure=[128] * 8, M=2048, B=16, name="nnet_128x8_M2048_B16_sigmoid"),
}
F64_MFMA_PEAK_TFLOPS = 78.6    # MI355X FP64 matrix = FP64 vector rate (half the 157.3 TF FP32 rows of
                               # MI355X_MICROARCH.md; v_mfma_f64_16x16x4_f64: 2048 flop / 64 cycles / SIMD)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
RF_SCALE = 1.5 ** 15           # mid-ladder RF (value does not change the work)
RAMP_MS = 2.0                  # untimed launches of the same kernel before a clock starts: the GPU's clocks ramp under load


def bytes_alg(B, N, D, NPest, N_data, L):
    """read X once, write grad once, p/grad-p, (A,me,fe); Y once per batch."""
    return 8 * (B * (2 * N * D + 2 * NPest + 3) + N_data * L)


def kernel_name(D, info, generated=False):
    """the evaluation kernel the handle runs (va_problem_eval_kernel; csrc/va_capi.hip: pick_eval_geometry)"""
    ek, K = info["eval_kernel"], info["run_rows"]
    if ek == 4:
        return "k_eval4<%s,trapezoid,K=%d,D=%d>" % ("RhsUserCol" if generated else "RhsL96s", K, D)
    if ek == 5:
        return "k_eval5<%s,trapezoid,D=%d>" % ("RhsUserCol" if generated else "RhsL96s", D)
    if ek == 3:
        return "k_eval3<%s,trapezoid,K=%d,D=%d>" % ("RhsUserG" if generated else "RhsL96g", K, D)
    return "k_eval<%s,trapezoid>" % ("RhsUser" if generated else "RhsL96")


def kernel_source_hash():
    """content hash of what the ODE evaluation kernels are built from (their sources and the build flags):
    stamps a counter measurement with the code it was taken on.  The translation units of generated modules and
    of the network action are not part of the measured kernels and do not invalidate it."""
    import hashlib
    from varanneal_amd import _build
    h = hashlib.sha1(" ".join(_build.FLAGS).encode())
    d = os.path.join(ROOT, "varanneal_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".h", ".hip")) and not fn.startswith(("va_user_", "va_nnet")):
            with open(os.path.join(d, fn), "rb") as fh:
                h.update(fn.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload_name):
    """HBM bytes per k_eval launch from the rocprofv3 --pmc passes recorded under profiles/
    (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE; own
    runs with --kernel-trace only).  Counters cannot be collected inside this process, so this
    is the committed measurement for the same workload -- valid only for the kernel sources it
    was taken on: None when csrc/ has changed since (tools/pmc.sh + tools/pmc_traffic.py redo it)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            rec = json.load(fh).get(workload_name, {})
    except (OSError, ValueError):
        return None
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None
    return rec.get("hbm_bytes_per_launch")


def make_inputs(D, N, B, rank):
    from varanneal_amd import twin
    t, Y, _, Lidx = twin.make_twin(D, N)
    XP = np.empty((B, N * D + 1)); P = np.empty((B, 1))
    for b in range(B):
        X0, P0 = twin.initial_guess(N, D, rank * B + b, Y, Lidx)      # global seed index
        XP[b, :-1] = X0.ravel(); XP[b, -1] = P0[0]; P[b] = P0
    return Y, Lidx, XP, P


def cpu_baseline(D, N, Y, Lidx, XP, P, budget_s=10.0):
    """The oracle's C restatement (oracle/va_oracle.c: vao_action_grad), ONE host core,
    same inputs, bounded sample.  Test infrastructure used only as the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import va_oracle
    from varanneal_amd import twin
    pbs = [va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P[b], [0], disc="trapezoid")
           for b in range(min(len(P), 8))]
    pbs[0].action_grad(XP[0], RF_SCALE)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for b, pb in enumerate(pbs):
            pb.action_grad(XP[b], RF_SCALE)
        n += len(pbs)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": "%d (A,gradA) evaluations of the same D=%d N=%d paths in %.1f s on 1 of %d host cores"
                      % (n, D, N, dt, os.cpu_count() or 0)}


def cpu_baseline_numpy(D, N, Y, Lidx, XP, P, budget_s=3.0):
    """the array-op-for-array-op NumPy twin of the reference's action
    (oracle/va_oracle.py numpy_action follows varanneal/va_ode.py:130-234, 358-380 line by line) -- the VALUE only,
    as the reference's own `A` computes it; the reference then needs 2 forward + 1 reverse ADOL-C tape sweeps per
    (A, grad A) on top (_autodiffmin.py:57-58) and re-records the tape at every beta step (va_ode.py:786): quoted
    cost structure, not re-measured (ADOL-C exists on no box of this pipeline)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import va_oracle
    from varanneal_amd import twin
    pb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P[0], [0], disc="trapezoid")
    pb.numpy_action(XP[0], RF_SCALE)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        pb.numpy_action(XP[n % len(XP)], RF_SCALE)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "A-only evals/s", "cores": 1, "kind": "port (NumPy twin of va_ode.py:130-234)",
            "sample": "%d action VALUES (no gradient) of the same D=%d N=%d paths in %.1f s" % (n, D, N, dt),
            "reference_cost_structure": "quoted, not measured: each (A, grad A) = 2 forward + 1 reverse sweeps of the ADOL-C "
                                        "tape (_autodiffmin.py:57-58); the tape is re-recorded in Python at every beta step "
                                        "(va_ode.py:786, _autodiffmin.py:79-80; BASELINE.md section 1: 74 % of the published run's wall time)"}


def usable_cpus():
    """host CPUs this process may actually use: its affinity mask, capped by the container's CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline_all(D, N, Y, Lidx, XP, P, budget_s=8.0):
    """the same C restatement, OpenMP over the seeds, on every host core this process may
    use (the GPU box gives one GPU's share of the host, not all of its cores)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import va_oracle
    from varanneal_amd import twin
    nt = min(usable_cpus(), len(P))
    va_oracle.set_num_threads(nt)
    nb = len(P) - len(P) % nt if len(P) >= nt else len(P)
    pbs = [va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P[b], [0], disc="trapezoid") for b in range(nb)]
    va_oracle.action_grad_batch(pbs, XP[:nb], RF_SCALE)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        va_oracle.action_grad_batch(pbs, XP[:nb], RF_SCALE)
        n += nb
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "evals/s", "cores": nt, "kind": "port",
            "sample": "%d (A,gradA) evaluations of the same D=%d N=%d paths in %.1f s, OpenMP over seeds on %d "
                      "threads (%d usable of %d host cores)" % (n, D, N, dt, nt, usable_cpus(), os.cpu_count() or 0)}


def event_timed(pb, rf, steps):
    """seconds per launch by HIP events: graph prepared, clocks ramped by the same launches (RAMP_MS), then `steps`"""
    pb.eval_timed_prepare(rf, steps)
    ms = 0.0
    while ms < RAMP_MS:
        ms += pb.eval_timed(rf, steps)
    return pb.eval_timed(rf, steps) * 1e-3 / steps


def extra_c4(device, steps=60):
    """BASELINE config 4 as one GPU's shard: D=200, N=5000, L=80, 64 seeds; complete evaluations."""
    from varanneal_amd import _capi, twin
    w = WORKLOADS["c4"]
    D, N, B = w["D"], w["N"], w["B"]
    Y, Lidx, XP, P = make_inputs(D, N, B, 0)
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", device=device) as pb:
        info = pb.info()
        pb.action_grad(XP, RF_SCALE)
        ks = event_timed(pb, RF_SCALE, steps)
    balg = bytes_alg(B, N, D, 1, N, len(Lidx))
    return {"workload": w["name"], "kernel": kernel_name(D, info) + ("" if B * info["ntiles"] <= 2048 else " + k_finalize_eval"),
            "traffic": pmc_traffic(w["name"]),
            "us_per_eval_launch": ks * 1e6, "evals_per_s": B / ks, "bytes_alg_per_launch": balg,
            "achieved_GBs": balg / ks / 1e9, "frac": balg / ks / 1e9 / HBM_PEAK_GBS}


def extra_variant(device, steps=1000, **kw):
    """One (f)3 variant of the C3 shape (reference: va_ode.py:147-148, 203-209, 404-437, 555-558):
    complete evaluations of 64 seeds, HIP events.  kw: disc=, N=, rf_vec= (RF0 of shape (D,) resized over time),
    rm_vec= (RM of shape (N_data, L)), nskip= (dt_model = dt_data / nskip), D= (200: the same variants at C4's width,
    where the streaming kernel k_eval5 carries them)."""
    from varanneal_amd import _capi, twin
    D, B = kw.get("D", 20), 64
    disc, N, nskip = kw.get("disc", "trapezoid"), kw.get("N", 1000), kw.get("nskip", 1)
    Y, Lidx, XP, P = make_inputs(D, N, B, 0)
    RM, RF0 = 4.0, 4e-6
    if kw.get("rf_vec"):
        RF0 = np.resize(4e-6 * (1.0 + 0.1 * np.arange(D)), (N - 1, D))
    if nskip > 1:
        Y = Y[::nskip]
    if kw.get("rm_vec"):
        RM = np.resize(4.0 * (1.0 + 0.1 * np.arange(len(Lidx))), Y.shape)
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, RM, RF0, P, [0], disc=disc, merr_nskip=nskip, device=device, tile_rows=kw.get("tile_rows", 0)) as pb:
        info = pb.info()
        pb.action_grad(XP, RF_SCALE)
        ks = event_timed(pb, RF_SCALE, steps)
    balg = bytes_alg(B, N, D, 1, Y.shape[0], len(Lidx)) + (8 * (N - 1) * D if kw.get("rf_vec") else 0) + (8 * Y.size if kw.get("rm_vec") else 0)
    return {"workload": "lorenz96_D%d_N%d_B%d_%s%s%s%s" % (D, N, B, disc, "_rfvec" if kw.get("rf_vec") else "", "_rmvec" if kw.get("rm_vec") else "",
                                                          "_nskip%d" % nskip if nskip > 1 else ""),
            "eval_kernel": info["eval_kernel"], "run_rows": info["run_rows"], "us_per_eval_launch": ks * 1e6, "evals_per_s": B / ks,
            "bytes_alg_per_launch": balg, "frac": balg / ks / 1e9 / HBM_PEAK_GBS}


def extra_linear(device, steps=200):
    """A generated model with a dense constant linear part (f = C x - p1 x^3 + p0, C a fixed 20 x 20 matrix) at the C3
    shape: the generator splits C off and the flat kernel forms X C^T and S C on v_mfma_f64_16x16x4_f64
    (codegen.linear_split, csrc/va_eval_flat.h lin_gemm; north_star: MFMA for a dense D x D linear map).  Complete
    evaluations of 64 seeds, HIP events; the two products are 4 N D^2 flop per seed."""
    from varanneal_amd import _capi, codegen, twin
    D, N, B = 20, 1000, 64
    m = codegen.module_for(twin.dense_coupling_model(D, 1)[0], D, 2)
    Y, Lidx, XP, _ = make_inputs(D, N, B, 0)
    P = np.tile(np.array([1.5, 0.3]), (B, 1))
    XP = np.concatenate([XP[:, :N * D], P], axis=1)
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 0.3, P, [0, 1], disc="trapezoid", rhs=_capi.load_rhs_module(m["so"]), device=device) as pb:
        info = pb.info()
        pb.action_grad(XP, 2.0)
        ks = event_timed(pb, 2.0, steps)
    flops = 4.0 * N * D * D * B
    return {"workload": "dense_coupling_D%d_N%d_B%d_trapezoid" % (D, N, B), "linear_part_on_mfma": m["lin"] is not None,
            "eval_kernel": info["eval_kernel"], "tile_rows": info["tile_rows"], "us_per_eval_launch": ks * 1e6, "evals_per_s": B / ks,
            "mfma_TFLOPs": flops / ks / 1e12, "frac": bytes_alg(B, N, D, 2, N, len(Lidx)) / ks / 1e9 / HBM_PEAK_GBS}


def extra_nnet(device, key, steps, fused=None):
    """The network action (BASELINE config 5, va_nnet) as a sub-record: `c5` = the reference's twin example
    (20 layers x 10 neurons, M = 2: one small kernel per evaluation), `c5x` = layers that fill the matrix cores
    (8 x 128 neurons, M = 2048).  Complete evaluations (A formed), HIP events."""
    from varanneal_amd import _capi, twin
    w = NNET_WORKLOADS[key]
    s, M, B = np.array(w["structure"]), w["M"], w["B"]
    din, dout, _ = twin.make_nnet_twin(s, M)
    RM = 1.0 / 0.005 ** 2
    RF0 = 1.0e-8 * RM * float(np.sum(s) - s[0]) / float(s[0] + s[-1])
    g = [twin.nnet_initial_guess(s, M, b) for b in range(B)]
    Pidx = g[0][2]
    P = np.array([x[1] for x in g])
    XP = np.array([np.append(x[0], x[1][Pidx]) for x in g])
    rf = 1.1 ** 100
    with _capi.NnetProblem(B, s, din, dout, [np.arange(s[0]), np.arange(s[-1])], RM, RF0, P, Pidx, device=device) as pb:
        if fused is not None:
            pb.tune(nnet_fused=fused)
        pb.action_grad(XP, rf)
        ks = event_timed(pb, rf, steps)
    flops = B * M * float(np.sum(3 * 2 * s[1:] * s[:-1]))
    return {"workload": w["name"], "us_per_eval_launch": ks * 1e6, "evals_per_s": B / ks, "flops_alg_per_launch": flops,
            "achieved_TFLOPs": flops / ks / 1e12, "frac_of_f64_mfma_peak": flops / ks / 1e12 / F64_MFMA_PEAK_TFLOPS}


def mnist_shape_run(device):
    """The one run the reference publishes a wall time for (VarAnneal_tutorial.ipynb:3413-3415,
    3449-3450, 3469, stored output :3584-10560): va_nnet, structure 784-30-10, M = 2 training pairs, weights estimated,
    RM = 1, RF0 by the tutorial's formula, alpha = 1.1, beta = 0..435, L-BFGS-B with gtol = ftol = 1e-12 -- through the
    drop-in (varanneal_amd.va_nnet.Annealer, device L-BFGS).  MNIST itself is not shipped: SYNTHETIC data of that shape
    (a seeded sigmoid twin network's input / output pairs); the start point follows the tutorial's recipe and seed.
    Context for the published number, not a same-machine comparison."""
    from varanneal_amd import twin, va_nnet
    s = np.array([784, 30, 10])
    M, D_in, D_hidden, D_out = 2, 784, 30, 10
    din, dout, _ = twin.make_nnet_twin(s, M)
    Lidx = [np.arange(D_in), np.arange(D_out)]
    RM = 1.0
    RF0 = 1.0e-8 * RM * float(np.sum(s) - s[0]) / float(s[0] + s[-1])
    alpha, beta = 1.1, np.linspace(0, 435, 436)
    rng = np.random.RandomState(27509436)                 # (the tutorial seeds numpy's global generator with this)
    X0 = np.array([])
    for m in range(M):
        xin = rng.randn(D_in)
        X0 = np.append(X0, (xin - np.average(xin)) / np.std(xin))
        X0 = np.append(X0, 0.2 * rng.rand(D_hidden) + 0.4)
        X0 = np.append(X0, 0.2 * rng.rand(D_out) + 0.4)
    P0, Pidx, off = np.array([]), [], 0
    for n in range(len(s) - 1):
        nw = int(s[n] * s[n + 1])
        Pidx += list(range(off, off + nw))
        P0 = np.append(P0, (2.0 * rng.rand(nw) - 1.0) / (D_in if n == 0 else D_hidden))
        P0 = np.append(P0, np.zeros(s[n + 1]))
        off += nw + int(s[n + 1])
    opts = {'gtol': 1.0e-12, 'ftol': 1.0e-12, 'maxfun': 1000000, 'maxiter': 1000000}

    def run(b):
        a = va_nnet.Annealer()
        a.set_structure(s); a.set_activation(twin.sigmoid)
        a.set_input_data(din); a.set_output_data(dout)
        t0 = time.perf_counter()
        a.anneal(X0.copy(), P0.copy(), alpha, b, RM, RF0, Pidx, Lidx=Lidx, method='L-BFGS-B', opt_args=opts, adolcID=0,
                 device=device, verbose=False)
        dt = time.perf_counter() - t0
        out = (dt, np.array(a.nit_array), np.array(a.nfev_array), np.array(a.A_array), np.array(a.exitflags))
        a.close()
        return out
    run(beta[:3])                                         # warm-up: code load, allocations
    dt, nit, nfev, A, flags = run(beta)
    hist = np.bincount(np.minimum(nit, 10), minlength=11)
    return {"workload": "va_nnet_784_30_10_M2_alpha1.1_beta0..435 (the tutorial's MNIST run, synthetic data of that shape)",
            "seconds": dt, "rungs": int(len(beta)), "lbfgs_iterations": int(nit.sum()), "evaluations": int(nfev.sum()),
            "rungs_with_one_iteration": int((nit == 1).sum()), "iterations_per_rung_histogram_0_to_10plus": hist.tolist(),
            "max_iterations_in_a_rung": int(nit.max()), "exitflag_counts": np.bincount(flags, minlength=3).tolist(),
            "A_first_last": [float(A[0]), float(A[-1])], "n_var": int(X0.size + len(Pidx)),
            "reference_published": {"seconds": 101.516544, "lbfgs_iterations": 1313, "rungs_with_one_iteration": 330,
                                    "taping_seconds": 75.3, "optimisation_seconds": 22.4,
                                    "source": "examples/jupyter-tutorial/VarAnneal_tutorial.ipynb:10560 (author's machine, "
                                              "real MNIST pairs, ADOL-C + SciPy; BASELINE.md section 1)"}}


def extra_ladder(device, D, N, B, Y, Lidx, XP, P, nbeta=30):
    """The C3 ladder end to end (alpha = 1.5, beta = 0..nbeta-1, SciPy-equal stopping rules): every
    kernel of the three-launch L-BFGS cycle counted; then the two vector kernels alone with full
    histories (va_lbfgs_timed), priced by the bytes they must move."""
    from varanneal_amd import _capi, twin
    rf = 1.5 ** np.arange(nbeta)
    opts = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}
    m = 10
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", device=device,
                       max_beta=nbeta) as pb:
        ld = pb.info()["ld"]
        pb.anneal(XP, rf[:2], opts)                                   # warm-up
        c0 = pb.counters()
        t0 = time.perf_counter()
        r = pb.anneal(XP, rf, opts)
        dt = time.perf_counter() - t0
        cyc = pb.counters()["cycles"] - c0["cycles"]
        it = 200
        ms_u, ms_d = pb.lbfgs_timed(it)
        ms_e = pb.eval_ls_timed(RF_SCALE, 1000)
    nfev = int(r["nfev"].sum())
    # per element and launch: k_update reads d, g, gt, x + 2(m-1) old history vectors and writes g, S_new,
    # Y_new (x only on acceptance: counted); k_direction reads g + 2m history vectors and writes d
    b_upd = 8 * B * ld * (4 + 2 * (m - 1) + 4)
    b_dir = 8 * B * ld * (1 + 2 * m + 1)
    us_u, us_d = ms_u * 1e3 / it, ms_d * 1e3 / it
    return {"workload": "lorenz96_D%d_N%d_B%d_trapezoid_ladder%d" % (D, N, B, nbeta),
            "seconds": dt, "seed_evals": nfev, "seed_evals_per_s": nfev / dt, "cycles": cyc,
            "us_per_cycle": dt * 1e6 / max(1, cyc), "launches_per_cycle": 3,
            "eval_us_in_cycle": ms_e,          # (1000 launches: ms total = us each) a line-search evaluation: x + stp*d, g.d, the line-search step
            "A_final_median": float(np.median(r["A"][:, -1])), "k_final_median": float(np.median(r["pest"][:, -1, 0])),
            "k_update": {"us": us_u, "bytes": b_upd, "GBs": b_upd / us_u / 1e3, "frac_of_8TBs": b_upd / us_u / 1e3 / HBM_PEAK_GBS},
            "k_direction": {"us": us_d, "bytes": b_dir, "GBs": b_dir / us_d / 1e3, "frac_of_8TBs": b_dir / us_d / 1e3 / HBM_PEAK_GBS},
            "note": "k_update / k_direction timed alone with full histories (m = 10); the 246 MB they stream per "
                    "seed batch sit in the 256 MiB Infinity Cache, so rates above the HBM figure are possible"}


def extra_ladder_small(device, N, nbeta=30):
    """BASELINE configs 1 (N = 200) and 2 (N = 1000): ONE seed, the whole 30-rung ladder.  Few seeds on a short path run
    the persistent per-seed kernel (csrc/va_persist.h: one cooperative launch, every vector of the minimisation in LDS);
    the same handle with `tune persist=0` runs the three-launch cycle for comparison."""
    from varanneal_amd import _capi, twin
    D, B = 20, 1
    Y, Lidx, XP, P = make_inputs(D, N, B, 0)
    rf = 1.5 ** np.arange(nbeta)
    opts = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}
    out = {"workload": "lorenz96_D%d_N%d_B1_trapezoid_ladder%d" % (D, N, nbeta)}
    with _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", device=device, max_beta=nbeta) as pb:
        geo = pb.persistent()
        out["persistent_workgroups_rows"] = list(geo) if geo else None
        for tag, on in (("persistent", 1), ("three_launch", 0)):
            if on and not geo:
                continue
            pb.tune(persist=on)
            pb.anneal(XP, rf[:2], opts)                               # warm-up
            c0 = pb.counters()["cycles"]
            t0 = time.perf_counter()
            r = pb.anneal(XP, rf, opts)
            dt = time.perf_counter() - t0
            cyc = pb.counters()["cycles"] - c0
            out[tag] = {"seconds": dt, "cycles": int(cyc), "us_per_cycle": dt * 1e6 / max(1, cyc), "seed_evals": int(r["nfev"].sum()),
                        "seed_evals_per_s": int(r["nfev"].sum()) / dt, "A_final": float(r["A"][0, -1]), "k_final": float(r["pest"][0, -1, 0])}
    return out


def timed_steps(pb, rf, steps, warmup, A, dist, world, torch, barrier, graph_fixed=None):
    """EXACTLY `steps` batched evaluations between two (barrier + synchronize) brackets; returns
    (wall seconds, kernel ms by HIP events, gather ms, ramp launches), the times the MAX over ranks.  Before the
    clock starts: the graph of the timed call's chunk of launches is captured, instantiated and uploaded
    (va_eval_timed_prepare), and the same launches run untimed -- at least `warmup` of them and at least RAMP_MS of
    device time -- so that what is timed is `steps` launches at the clocks a running job sees.  The run's one
    collective -- the RCCL all-gather of the per-seed actions that closes a multi-GPU job -- is
    issued once after the timed steps and timed on its own: it is per job, not per step.  Its
    buffers and the communicator are set up (and warmed) before the clock starts."""
    recv = mine = None
    cdev = "cuda"
    if dist is not None:
        cdev = "cuda" if dist.get_backend() == "nccl" else "cpu"      # (gloo: the one-GPU rehearsal of the multi-rank path)
        mine = torch.from_numpy(np.ascontiguousarray(A)).to(cdev)
        recv = torch.empty(world * mine.numel(), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(recv, mine)          # warm-up: communicator, RCCL kernels
    # how the K launches are issued is the caller's choice (va_problem_tune graph=: one hipGraph replay per chunk, or plain
    # launches): a graph replay starts ~15-20 us after the call and then runs back to back, plain launches start at once
    # and stay device-bound above ~4 us per kernel -- for a handful of steps the plain form ends first.  Both are tried
    # untimed and the faster one is timed; either way every launch is the same kernel.
    ramp, ramp_ms, best = 0, 0.0, {}
    for mode in ((1, 0) if graph_fixed is None else (int(graph_fixed),)):
        pb.tune(graph=mode)
        pb.eval_timed_prepare(rf, steps)
        pb.eval_timed(rf, steps); ramp += steps
        w = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ramp_ms += pb.eval_timed(rf, steps)
            torch.cuda.synchronize()
            w.append(time.perf_counter() - t0)
            ramp += steps
        best[mode] = min(w)
    mode = min(best, key=best.get)
    pb.tune(graph=mode)
    pb.eval_timed_prepare(rf, steps)
    while ramp < max(warmup, 1) or ramp_ms < RAMP_MS:
        ramp_ms += pb.eval_timed(rf, steps)              # (same chunk as the timed call: its graph is replayed, not rebuilt)
        ramp += steps
    barrier()
    t0 = time.perf_counter()
    kernel_ms = pb.eval_timed(rf, steps)                 # HIP events on the launch stream
    barrier()
    wall = time.perf_counter() - t0
    gather_ms = 0.0
    if dist is not None:
        t1 = time.perf_counter()
        dist.all_gather_into_tensor(recv, mine)          # the single RCCL gather of the job
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t1) * 1e3
        tmax = torch.tensor([wall, kernel_ms, gather_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall, kernel_ms, gather_ms = (float(v) for v in tmax)
    return wall, kernel_ms, gather_ms, ramp, ("hipGraph replay" if mode else "plain launches")


def nnet_main(args, rank, local_rank, world, dist, torch):
    """C5: (A, grad A) of the feed-forward-network action; one step = one batched evaluation
    (k_nnet_pack, k_nnet_fwd, k_nnet_bwd_x, k_nnet_bwd_w, k_nnet_pred) of B seeds."""
    from varanneal_amd import _capi, twin
    w = NNET_WORKLOADS[args.workload]
    s, M, B = np.array(w["structure"]), w["M"], w["B"]
    din, dout, _ = twin.make_nnet_twin(s, M)
    Lidx = [np.arange(s[0]), np.arange(s[-1])]
    RM = 1.0 / 0.005 ** 2
    RF0 = 1.0e-8 * RM * float(np.sum(s) - s[0]) / float(s[0] + s[-1])       # nnet_twin_anneal.py:45
    g = [twin.nnet_initial_guess(s, M, rank * B + b) for b in range(B)]
    Pidx = g[0][2]
    P = np.array([x[1] for x in g])
    XP = np.array([np.append(x[0], x[1][Pidx]) for x in g])
    pb = _capi.NnetProblem(B, s, din, dout, Lidx, RM, RF0, P, Pidx, device=local_rank)
    rf = 1.1 ** 100
    A, me, fe, gr = pb.action_grad(XP, rf)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    wall, kernel_ms, gather_ms, ramp, launch_mode = timed_steps(pb, rf, args.steps, args.warmup, A, dist, world, torch, barrier)
    if rank != 0:
        return
    # three products per layer transition: Z = X W^T, dX = delta W, dW = delta^T X
    flops = B * M * float(np.sum(3 * 2 * s[1:] * s[:-1]))
    balg = 8 * (B * (2 * (M * int(np.sum(s)) + len(Pidx)) + 3) + din.size + dout.size)
    ks = kernel_ms * 1e-3 / args.steps
    out = {
        "metric": "action+grad evals/sec, va_nnet %s M=%d" % ("x".join(str(v) for v in (s[0], len(s))), M),
        "value": world * B * args.steps / wall, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (sigmoid twin network, sigma=0.005, seeded)",
        "config": {"workload": w["name"], "seeds_per_gpu": B, "structure": [int(v) for v in s], "M": M,
                   "n_var": int(XP.shape[1]), "parallelism": "seeds sharded, %d per GPU" % B,
                   "final_gather_ms": gather_ms, "rccl_ranks": dist.get_world_size() if dist is not None else 1,
                   "ramp_launches": ramp, "launch_mode": launch_mode, "env": bench_env()},
        "roofline": {"bound": "mfma", "achieved": flops / ks / 1e12, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": flops / ks / 1e12 / F64_MFMA_PEAK_TFLOPS,
                     "frac_wall": flops * args.steps / wall / 1e12 / F64_MFMA_PEAK_TFLOPS, "traffic": None,
                     "kernel": "k_nnet_fwd + k_nnet_bwd_x + k_nnet_bwd_w (+ pack, pred): one evaluation",
                     "kernel_us": ks * 1e6, "flops_alg_per_launch": flops, "bytes_alg_per_launch": balg,
                     "hbm_frac_of_8TBs": balg / ks / 1e9 / HBM_PEAK_GBS},
        "cpu_baseline": None,
    }
    if world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import va_nnet_oracle as vno
        from threadpoolctl import threadpool_limits
        pbo = vno.NnetProblem(s, din, dout, Lidx, RM, RF0, P[0], Pidx)
        with threadpool_limits(limits=1):
            pbo.action_grad(XP[0], rf)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 10.0:
                pbo.action_grad(XP[n % B], rf); n += 1
            dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / dt, "unit": "evals/s", "cores": 1, "kind": "port",
                               "sample": "%d (A,gradA) evaluations by oracle/va_nnet_oracle.py (NumPy, BLAS limited "
                                         "to 1 thread) in %.1f s on 1 of %d host cores" % (n, dt, os.cpu_count() or 0)}
    print(json.dumps(out), flush=True)
    pb.close()


def ladder_mode(args, pb0, XP, P, D, N, B, Y, Lidx, device):
    """Whole ladder (alpha=1.5, beta=0..nbeta-1, SciPy-equal stopping rules) for B seeds;
    reports end-to-end seed-evaluations/s including every L-BFGS vector kernel."""
    from varanneal_amd import _capi, twin
    pb0.close()
    nb = args.nbeta
    rf = 1.5 ** np.arange(nb)
    opts = {'gtol': 1e-8, 'ftol': 1e-8, 'maxfun': 1000000, 'maxiter': 1000000}
    pb = _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", device=device,
                       max_beta=nb, tile_rows=args.tile_rows, eval_kernel=args.eval_kernel)
    pb.anneal(XP, rf[:2], opts)                                   # warm-up (allocations, code load)
    t0 = time.perf_counter()
    r = pb.anneal(XP, rf, opts)
    dt = time.perf_counter() - t0
    c = pb.counters()
    nfev = int(r["nfev"].sum()); nit = int(r["nit"].sum())
    print(json.dumps({"mode": "ladder", "workload": "D=%d N=%d B=%d nbeta=%d" % (D, N, B, nb),
                      "seconds": dt, "seed_evals": nfev, "seed_evals_per_s": nfev / dt,
                      "lbfgs_iterations": nit, "cycles": c["cycles"],
                      "us_per_cycle": dt * 1e6 / max(1, c["cycles"] - 0),
                      "A_final_median": float(np.median(r["A"][:, -1])),
                      "k_final_median": float(np.median(r["pest"][:, -1, 0])),
                      "status_counts": np.bincount(r["status"].ravel(), minlength=3).tolist()}), flush=True)
    if not args.no_cpu:
        # the same ladder for ONE of these seeds by the C oracle (vao_anneal: restated L-BFGS-B +
        # fused forward/adjoint) on one host core -- the reference's execution model
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import va_oracle
        opb = va_oracle.Problem(D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P[0], [0], disc="trapezoid")
        t0 = time.perf_counter()
        ro = opb.anneal(XP[0], 1.5, np.arange(nb), opts)
        dtc = time.perf_counter() - t0
        print(json.dumps({"mode": "ladder_cpu_baseline", "kind": "port", "cores": 1, "seeds": 1,
                          "seconds": dtc, "seed_evals": int(ro["nfev"].sum()),
                          "seed_evals_per_s": int(ro["nfev"].sum()) / dtc,
                          "A_final": float(ro["A"][-1]), "device_A_final_seed0": float(r["A"][0, -1]),
                          "device_seconds_per_seed": dt / B}), flush=True)
    pb.close()


def launch_ranks(n, argv, port=None):
    """Run this script as n ranks of ONE node under torch.distributed.run (the same command line the driver uses)
    and return the job's exit code.  Called before anything has touched the GPU: the ranks are child processes."""
    import socket
    import subprocess
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def bench_env():
    """every VA_* / VARANNEAL_AMD_* variable set in this process: they select kernels or code paths, so they are
    part of what was measured (config.env)"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith(("VA_", "VARANNEAL_AMD_"))}


def dry_run(args, rank, world, env_set):
    """The launcher / rank bookkeeping without a GPU (tests/test_bench_launcher.py): joins the process group, counts
    its ranks with an all-reduce, and rank 0 prints the fields of the contract line that depend on the launch."""
    ranks = 1
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
        one = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(one)
        ranks = int(one[0])
        assert ranks == dist.get_world_size() == args.gpus
        dist.destroy_process_group()
    if rank == 0:
        w = WORKLOADS.get(args.workload) or NNET_WORKLOADS.get(args.workload) or dict(name="mnist_shape", B=1)
        print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "scaling": "weak", "config": {"workload": w["name"], "seeds_per_gpu": w["B"],
                                                        "rccl_ranks": ranks, "env": env_set}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks of the job, one per GPU (default: WORLD_SIZE if the job was launched already, else 1)")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS) + sorted(NNET_WORKLOADS) + ["mnist_shape"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra sub-records (C4 shape, C3 ladder)")
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--mode", default="eval", choices=["eval", "ladder"],
                    help="eval: the contract line (batched A/gradA launches); ladder: a whole "
                         "RF ladder through va_anneal, extra information (stderr-style JSON)")
    ap.add_argument("--nbeta", type=int, default=30)
    ap.add_argument("--generated", action="store_true",
                    help="run the workload's Lorenz-96 as a user would supply it: a Python callable traced and "
                         "compiled by varanneal_amd.codegen (not the built-in right-hand side)")
    ap.add_argument("--eval-kernel", type=int, default=0, help="0 auto, 1 flat-mapped, 3 workgroup column runs, 4 wave-private column runs, 5 streaming column strips")
    ap.add_argument("--tune", default="", help="comma-separated va_problem_tune knobs for the measured handle, e.g. fold=0,graph=0 "
                                               "(recorded in config.tune; none changes a result)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal without a GPU: every rank joins the process group (--backend gloo), "
                         "rank 0 prints the contract line's bookkeeping fields (n_gpus, config.rccl_ranks) and nothing is timed")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of a multi-rank run (nccl = RCCL; gloo only for the CPU rehearsal "
                         "of the launcher in tests/)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the multi-rank timed path on a box with ONE GPU: every rank runs its shard on cuda:0 "
                         "(needs --backend gloo: RCCL refuses two ranks on one device); recorded in config.rehearsal")
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))      # (torchrun ... bench.py without --gpus: the job's size)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.share_gpu and args.backend != "gloo":
        ap.error("--share-gpu needs --backend gloo")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks (one process per GPU) as CHILD processes of a
        # launcher that never touches the GPU, and leave with their exit code.  The reference's counterpart of
        # this fan-out is its array job: examples/nnet_barimages/SGEcluster/submit_multiM.sh:14-30.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the job has WORLD_SIZE=%d ranks\n" % (args.gpus, world))
        sys.exit(2)
    env_set = bench_env()

    if args.dry_run:
        dry_run(args, rank, world, env_set)
        return

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            sys.stderr.write("bench.py: process group has %d ranks, --gpus %d\n" % (dist.get_world_size(), args.gpus))
            sys.exit(2)
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)

    if args.workload == "mnist_shape":
        if rank == 0:
            print(json.dumps(dict(mode="mnist_shape", **mnist_shape_run(local_rank))), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.workload in NNET_WORKLOADS:
        nnet_main(args, rank, local_rank, world, dist, torch)
        if dist is not None:
            dist.destroy_process_group()
        return
    from varanneal_amd import _capi, twin
    w = WORKLOADS[args.workload]
    D, N, B = w["D"], w["N"], w["B"]
    Y, Lidx, XP, P = make_inputs(D, N, B, rank)
    rhs = "lorenz96"
    if args.generated:
        from varanneal_amd import codegen

        def l96_user(t, x, p):
            return np.roll(x, 1, 1) * (np.roll(x, -1, 1) - np.roll(x, 2, 1)) - x + p[0]
        mod = codegen.module_for(l96_user, D, 1, col_variant=lambda ne, gh, reach: _capi.eval_plan(
            B, D, N, "trapezoid", ne, gh, tile_rows=args.tile_rows, eval_kernel=args.eval_kernel, reach=reach, Lidx=Lidx))
        rhs = _capi.load_rhs_module(mod["so"])
    pb = _capi.Problem(B, D, N, Y, Lidx, twin.DT, 4.0, 4e-6, P, [0], disc="trapezoid", rhs=rhs,
                       device=local_rank, tile_rows=args.tile_rows, eval_kernel=args.eval_kernel)
    tune = {k: int(v) for k, v in (kv.split("=") for kv in args.tune.split(",") if kv)}
    pb.tune(**tune)
    info = pb.info()
    if args.mode == "ladder":
        ladder_mode(args, pb, XP, P, D, N, B, Y, Lidx, local_rank)
        return
    A, me, fe, g = pb.action_grad(XP, RF_SCALE)          # paths now resident in HBM

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    wall, kernel_ms, gather_ms, ramp, launch_mode = timed_steps(pb, RF_SCALE, args.steps, args.warmup, A, dist, world, torch, barrier, tune.get("graph"))

    if rank == 0:
        balg = bytes_alg(B, N, D, 1, N, len(Lidx))
        kern_s = kernel_ms * 1e-3 / args.steps
        achieved = balg / kern_s / 1e9
        out = {
            "metric": "action+grad evals/sec, Lorenz-96 D=%d N=%d" % (D, N),
            "value": world * B * args.steps / wall, "unit": "evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (Lorenz-96 twin experiment, k=8.17, sigma=0.5, seeded)",
            "config": {"workload": w["name"], "seeds_per_gpu": B, "D": D, "N": N, "L": len(Lidx),
                       "disc": "trapezoid", "tile_rows": info["tile_rows"], "ntiles": info["ntiles"],
                       "parallelism": "seeds sharded, %d per GPU" % B, "final_gather_ms": gather_ms,
                       "rccl_ranks": dist.get_world_size() if dist is not None else 1, "ramp_launches": ramp, "launch_mode": launch_mode, "env": env_set, "tune": tune,
                       "rehearsal": ("%d ranks share one GPU, collectives over gloo" % world) if args.share_gpu else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "step": "complete S1 evaluation: A, me, fe formed in the same launch (va_epilogue.h)",
                         "frac": achieved / HBM_PEAK_GBS,
                         # the same fraction from the host's wall clock around the K steps (the clock `value` uses)
                         "frac_wall": balg * args.steps / wall / 1e9 / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(w["name"]),
                         "kernel": kernel_name(D, info, args.generated),
                         "kernel_us": kern_s * 1e6, "bytes_alg_per_launch": balg},
            "cpu_baseline": None,
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(D, N, Y, Lidx, XP, P)
            out["cpu_baseline_all"] = cpu_baseline_all(D, N, Y, Lidx, XP, P)
            out["cpu_baseline_numpy"] = cpu_baseline_numpy(D, N, Y, Lidx, XP, P)
        if world == 1 and args.workload == "c3" and not args.no_extra:
            pb.close()
            out["extra"] = {"ladder": extra_ladder(local_rank, D, N, B, Y, Lidx, XP, P),
                            "ladder_c1": extra_ladder_small(local_rank, 200), "ladder_c2": extra_ladder_small(local_rank, 1000),
                            "c4": extra_c4(local_rank),
                            "c3_sh": extra_variant(local_rank, disc="SimpsonHermite", N=1001),      # (what Lorenz96_anneal.py:85 runs)
                            "c4_sh": extra_variant(local_rank, steps=30, D=200, disc="SimpsonHermite", N=5001),
                            "c4_rf": extra_variant(local_rank, steps=30, D=200, N=5000, rf_vec=True),
                            "lin_d20": extra_linear(local_rank),
                            "c5": extra_nnet(local_rank, "c5", 2000), "c5x": extra_nnet(local_rank, "c5x", 40),
                            "mnist_shape": mnist_shape_run(local_rank)}
        print(json.dumps(out), flush=True)
    pb.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
