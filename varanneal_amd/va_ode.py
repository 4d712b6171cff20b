"""va_ode.Annealer -- drop-in for `varanneal.va_ode.Annealer` on MI355X.

Same surface as the reference (varanneal/va_ode.py:43-905):

    set_model(f, D) / set_data(data, stim, t, nstart, N) / set_data_fromfile(...)
    anneal(X0, P0, alpha, beta_array, RM, RF0, Lidx, Pidx, dt_model=None,
           init_to_data=True, action='A_gaussian', disc='trapezoid',
           method='L-BFGS-B', bounds=None, opt_args=None, adolcID=0,
           track_paths=None, track_params=None, track_action_errors=None)
    anneal_init(...) / anneal_step()
    save_paths / save_params / save_action_errors / save_as_minAone
    attributes N_data, minpaths, A_array, me_array, fe_array, P, exitflags, ...

What differs underneath: the action, its gradient (hand-coded adjoint, no ADOL-C
tape -- `adolcID` is accepted and ignored) and the L-BFGS minimisation over the
whole RF ladder run as HIP kernels behind include/varanneal_amd.h; there is no
CPU fallback.  Extensions are keyword-only or shape-triggered so reference scripts
run unchanged:

  * batched seeds: pass X0 with shape (B, N_model, D) and P0 with shape (B, NP);
    every result array gains a leading B axis.  Seeds climb the ladder
    independently on the device.
  * `device=`, `verbose=`, `fused=` keyword-only arguments of anneal()/anneal_init().

A model function that is not one of the hand-written built-ins (varanneal_amd.rhs) is
traced once, differentiated symbolically and compiled into a device module
(varanneal_amd.codegen) -- including the reference's `f(t, x, (p, stim))` stimulus
convention (va_ode.py:345-375).

Time-dependent parameters (upstream: P0 of shape (N_model, NP) with a 2-D X0,
va_ode.py:170-188, 565-570) are supported for the discretisations that work upstream
(trapezoid, SimpsonHermite); the model then receives the rows' own parameters, p[:, k].
Here any subset Pidx may be estimated and bounds work (upstream's anneal_step and bounds
branches for this case are broken, va_ode.py:597-601, 715-732).  A batch uses P0 (B, N_model, NP).

Full measurement precision matrices (`RM` of shape (L,L) or (N_data,L,L), va_ode.py:149-152) and full
model-error precision matrices (`RF0` of shape (D,D) or (N_model-1,D,D), va_ode.py:211-217; contracted per
row, which is what upstream's Simpson-Hermite branch does and its other branches mean) are supported
(flat tile kernel).  Not implemented (raise NotImplementedError): user-defined action callables, method='LM'.
"""
from __future__ import print_function

import time

import os

import numpy as np

from . import _capi, rhs as _rhs
from ._hipmin import HIPmin, alpha_pow as _alpha_pow

_DISCS = ("euler", "trapezoid", "SimpsonHermite", "forwardmap")


class Annealer(HIPmin):
    def __init__(self):
        self.taped = False                    # reference attribute (va_ode.py:53); unused here
        self.annealing_initialized = False
        self._pb = None
        self.stim = None

    # ------------------------------------------------------------------ model / data
    def set_model(self, f, D):
        """f(t, x, p) acting on time slices (va_ode.py:56-67), or a registry name."""
        self.f = f
        self.D = int(D)
        self._rhs_name = _rhs.recognise(f, self.D)

    def set_data_fromfile(self, data_file, stim_file=None, nstart=0, N=None):
        """va_ode.py:69-96 (which raises NameError upstream); here it simply works."""
        data = np.load(data_file) if data_file.endswith("npy") else np.loadtxt(data_file)
        stim = None
        if stim_file is not None:
            stim = np.load(stim_file) if stim_file.endswith("npy") else np.loadtxt(stim_file)
        self.set_data(data, stim=stim, nstart=nstart, N=N)

    def set_data(self, data, stim=None, t=None, nstart=0, N=None):
        """va_ode.py:98-124: with `t` given, `data` holds observations only; otherwise
        column 0 of `data` (and of `stim`) is time."""
        self.N_data = data.shape[0] if N is None else N
        sl = slice(nstart, nstart + self.N_data)
        if t is None:
            self.t_data = data[sl, 0]
            self.Y = data[sl, 1:]
            self.stim = stim[sl, 1:] if stim is not None else None
        else:
            self.t_data = t[sl]
            self.Y = data[sl]
            self.stim = stim[sl] if stim is not None else None
        self.dt_data = self.t_data[1] - self.t_data[0]

    # ------------------------------------------------------------------ annealing
    def anneal(self, X0, P0, alpha, beta_array, RM, RF0, Lidx, Pidx, dt_model=None,
               init_to_data=True, action='A_gaussian', disc='trapezoid',
               method='L-BFGS-B', bounds=None, opt_args=None, adolcID=0,
               track_paths=None, track_params=None, track_action_errors=None,
               *, device=None, verbose=True, fused=None, n_seeds=None, devices=None, gather=True,
               bounded_minimiser=None):
        """Full ladder (va_ode.py:459-528).  With no per-step tracking and the device
        minimiser, the whole ladder runs in one C-ABI call (`fused`).

        Keyword-only additions (reference scripts run unchanged without them):
          n_seeds  X0 / P0 carry `n_seeds` independent initial guesses on their leading axis.  Under an
                   initialised torch.distributed job (one process per GPU; backend "nccl" is RCCL) every
                   rank anneals the block [r*n/R, (r+1)*n/R) of them -- the reference's SGE array job
                   (examples/nnet_barimages/SGEcluster/submit_multiM.sh:14-30) without the cluster -- and
                   ONE all-gather leaves the per-seed tables of all seeds in `self.gathered` on every
                   rank.  A seed's results do not depend on how many ranks share the work.
          devices  device ordinals to use (rank r takes devices[r % len]); default LOCAL_RANK
          gather   False: skip the collective (`self.gathered` then holds this rank's seeds only)
          bounded_minimiser  with `bounds`: 'scipy' = SciPy's L-BFGS-B on the host around the device
                   evaluator, exactly the reference's call (_autodiffmin.py:85-86; one seed; the default
                   for a single seed); 'device' = L-BFGS-B itself on the device (generalised Cauchy point +
                   subspace minimisation, csrc/va_lbfgsb.hip): every seed at once, nothing leaves HBM, SciPy's
                   iterates step for step (the default for batched seeds)"""
        if n_seeds is not None:
            from . import parallel
            X0 = np.asarray(X0); P0 = np.asarray(P0, dtype=np.float64)
            if X0.ndim != 3 or X0.shape[0] != n_seeds or P0.shape[0] != n_seeds:
                raise ValueError("n_seeds=%d needs X0 of shape (n_seeds, N_model, D) and P0 with n_seeds rows" % n_seeds)
            rank, world = parallel.rank_world()
            lo, hi = parallel.seed_range(n_seeds, rank, world)
            self.n_seeds, self.seed_lo, self.seed_hi = n_seeds, lo, hi
            if device is None:
                device = parallel.local_device(devices, rank)
            local = None
            if hi > lo:
                # (views: init_to_data and the parameter write-back still reach the caller's arrays)
                self.anneal(X0[lo:hi], P0[lo:hi], alpha, beta_array, RM, RF0, Lidx, Pidx, dt_model,
                            init_to_data, action, disc, method, bounds, opt_args, adolcID,
                            track_paths, track_params, track_action_errors,
                            device=device, verbose=verbose, fused=fused, bounded_minimiser=bounded_minimiser)
                ND = self.N_model * self.D
                local = {"A": self._A, "me": self._me, "fe": self._fe, "params": self._mp[:, :, ND:],
                         "exitflags": self._flags, "nit": self._nit, "nfev": self._nfev}
            if world > 1 and gather:
                if local is None:                     # more ranks than seeds: an empty block still joins the collective
                    nb, npw = len(beta_array), P0.reshape(n_seeds, -1).shape[1]
                    local = {"A": np.zeros((0, nb)), "me": np.zeros((0, nb)), "fe": np.zeros((0, nb)),
                             "params": np.zeros((0, nb, npw)), "exitflags": np.zeros((0, nb), np.int8),
                             "nit": np.zeros((0, nb), np.int32), "nfev": np.zeros((0, nb), np.int64)}
                self.gathered = parallel.gather_tables(local, n_seeds, device=device)
            else:
                self.gathered = local
            return
        if device is None:
            device = 0
        if self.annealing_initialized is False:       # reference: flag is never set (va_ode.py:468,705)
            self.anneal_init(X0, P0, alpha, beta_array, RM, RF0, Lidx, Pidx, dt_model,
                             init_to_data, action, disc, method, bounds, opt_args, adolcID,
                             device=device, verbose=verbose, bounded_minimiser=bounded_minimiser)
        tracking = any(t is not None for t in (track_paths, track_params, track_action_errors))
        if fused is None:
            fused = (not tracking) and self._device_minimiser
        if fused:
            if not self._device_minimiser:
                raise ValueError("fused=True needs method='L-BFGS-B' with bounds=None")
            self._anneal_fused()
            return
        for _ in self.beta_array:
            if self.verbose:
                print('------------------------------')
                print('Step %d of %d' % (self.betaidx + 1, len(self.beta_array)))
                print('beta = %d, RF = %.8e' % (self.beta, self._rf_print()))
                print('')
            self.anneal_step()
            if track_paths is not None:
                self.save_paths(track_paths['filename'], track_paths.get('dtype', np.float64),
                                track_paths.get('fmt', "%.8e"))
            if track_params is not None:
                self.save_params(track_params['filename'], track_params.get('dtype', np.float64),
                                 track_params.get('fmt', "%.8e"))
            if track_action_errors is not None:
                self.save_action_errors(track_action_errors['filename'],
                                        track_action_errors.get('cmpt', 0),
                                        track_action_errors.get('dtype', np.float64),
                                        track_action_errors.get('fmt', "%.8e"))

    def anneal_init(self, X0, P0, alpha, beta_array, RM, RF0, Lidx, Pidx, dt_model=None,
                    init_to_data=True, action='A_gaussian', disc='trapezoid',
                    method='L-BFGS-B', bounds=None, opt_args=None, adolcID=0,
                    *, device=0, verbose=True, bounded_minimiser=None):
        """va_ode.py:531-705."""
        if method not in ('L-BFGS-B', 'NCG', 'LM', 'TNC'):
            print("ERROR: Optimization routine not recognized. Annealing not initialized.")
            return None
        if method == 'LM':
            raise NotImplementedError("method='LM' is dead code upstream (_autodiffmin.py:157)")
        self.method = method
        self.verbose = verbose
        if not hasattr(self, "f"):
            raise ValueError("set_model() has not been called")
        # separate dt_data and dt_model are not supported upstream with a stimulus (va_ode.py:544-547)
        if dt_model is not None and dt_model != self.dt_data and self.stim is not None:
            raise ValueError("Separate dt_data and dt_model currently not supported with an external stimulus.")
        if action != 'A_gaussian':
            raise NotImplementedError("only action='A_gaussian' is implemented")
        if disc not in _DISCS:
            raise ValueError("unknown discretisation %r (expected one of %s)" % (disc, _DISCS))
        self.disc_name = disc

        # time grid (va_ode.py:549-558)
        if dt_model is None:
            self.dt_model = self.dt_data
            self.N_model = self.N_data
            self.merr_nskip = 1
            self.t_model = np.copy(self.t_data)
        else:
            self.dt_model = dt_model
            self.merr_nskip = int(self.dt_data / self.dt_model)
            self.N_model = (self.N_data - 1) * self.merr_nskip + 1
            self.t_model = np.linspace(self.t_data[0], self.t_data[-1], self.N_model)
        self.opt_args = opt_args

        # seeds / parameters
        X0 = np.asarray(X0)
        P0 = np.asarray(P0, dtype=np.float64)
        self._batched = X0.ndim == 3
        if self._batched:
            if P0.ndim not in (2, 3) or P0.shape[0] != X0.shape[0]:
                raise ValueError("batched X0 (B,N,D) needs P0 of shape (B,NP), or (B,N_model,NP) for "
                                 "time-dependent parameters")
            self.B = X0.shape[0]
            self._tdp = P0.ndim == 3
        else:
            if P0.ndim not in (1, 2):
                raise ValueError("P0 must be 1-D (static) or (N_model, NP) (time-dependent, va_ode.py:565-570)")
            self.B = 1
            self._tdp = P0.ndim == 2
        if self._tdp:
            # upstream's euler / forwardmap branches slice p one row short (va_ode.py:345-349,
            # 443-447 against :174-175): only the discretisations that work there are offered
            if disc not in ("trapezoid", "SimpsonHermite"):
                raise NotImplementedError("time-dependent parameters with disc=%r (inconsistent upstream)" % disc)
            if P0.shape[-2] != self.N_model:
                raise ValueError("time-dependent P0 must have N_model = %d rows" % self.N_model)
        if X0.shape[-2:] != (self.N_model, self.D):
            raise ValueError("X0 must have shape (N_model, D) = (%d, %d)" % (self.N_model, self.D))
        self.P = P0                                   # reference keeps a reference to the caller's array
        self.NP = P0.shape[-1]
        self.Pidx = list(Pidx)
        self.NPest = len(self.Pidx)
        self.Lidx = list(Lidx)
        self.L = len(self.Lidx)
        if np.shape(self.Y)[1] != self.L:
            raise ValueError("data has %d observed columns but Lidx has %d entries" % (np.shape(self.Y)[1], self.L))
        stim = None
        if self.stim is not None:
            stim = np.asarray(self.stim, dtype=np.float64)
        if self._rhs_name is not None and stim is None:
            impl, NPr, _ = _rhs.REGISTRY[self._rhs_name]
            if self.NP != NPr:
                raise ValueError("RHS %r takes %d parameter(s), P0 has %d" % (self._rhs_name, NPr, self.NP))
            rhs_id = self._rhs_name
        else:
            rhs_id = None                             # any other callable: a generated module, built below
                                                      # once the weights and bounds (its kernel variant) are known

        # RM / RF0 broadcasting (va_ode.py:612-640)
        if isinstance(RM, list):
            RM = np.array(RM)
        if isinstance(RM, np.ndarray) and RM.ndim > 0:
            if RM.shape == (self.L,):
                self.RM = np.resize(RM, (self.N_data, self.L))
            elif RM.shape == (self.N_data, self.L):
                self.RM = RM
            elif RM.shape == (self.L, self.L):
                self.RM = np.resize(RM, (self.N_data, self.L, self.L))      # va_ode.py:617-618
            elif RM.shape == (self.N_data, self.L, self.L):
                self.RM = RM
            else:
                raise ValueError("ERROR: RM has an invalid shape.")
        else:
            self.RM = float(RM)
        if isinstance(RF0, list):
            RF0 = np.array(RF0)
        if isinstance(RF0, np.ndarray) and RF0.ndim > 0:
            if RF0.shape == (self.D,):
                self.RF0 = np.resize(RF0, (self.N_model - 1, self.D))
            elif RF0.shape == (self.N_model - 1, self.D):
                self.RF0 = RF0
            elif RF0.shape == (self.D, self.D):
                # full matrices (va_ode.py:631-632): diff_n . (RF_n . diff_n) per time step, as upstream's
                # Simpson-Hermite branch contracts them (:211-217; the other branch, :218-222, slips to the
                # whole diff array -- the per-row contraction is what is computed for every discretisation)
                self.RF0 = np.resize(RF0, (self.N_model - 1, self.D, self.D))
            elif RF0.shape == (self.N_model - 1, self.D, self.D):
                self.RF0 = RF0
            else:
                raise ValueError("ERROR: RF0 has an invalid shape.")
        else:
            self.RF0 = float(RF0)

        # ladder (va_ode.py:643-650; beta is truncated to uint16 upstream, kept)
        self.alpha = alpha
        self.beta_array = np.array(beta_array, dtype=np.uint16)
        self.Nbeta = len(self.beta_array)
        self._rf_scale = _alpha_pow(self.alpha, self.beta_array)
        self.betaidx = 0
        self.beta = self.beta_array[0]
        self.RF = self.RF0 * _alpha_pow(self.alpha, self.beta)

        # bounds (va_ode.py:582-605): expanded exactly as upstream, used by the SciPy route
        if bounds is not None:
            state_b, param_b = list(bounds[:self.D]), list(bounds[self.D:])
            self.bounds = [state_b[i] for _ in range(self.N_model) for i in range(self.D)]
            # (upstream's time-dependent branch, va_ode.py:597-601, names undefined variables;
            # this is what it means: the parameter bounds repeated for every time point)
            self.bounds += [param_b[i] for _ in range(self.N_model if self._tdp else 1) for i in range(self.NPest)]
        else:
            self.bounds = None
        if bounded_minimiser is None:
            # L-BFGS-B itself on the device (csrc/va_lbfgsb.hip follows SciPy's iterates step for step:
            # tests/test_gpu_codegen.py); 'scipy' keeps SciPy on the host around the device evaluator
            bounded_minimiser = 'device'
        if bounded_minimiser not in ('scipy', 'device'):
            raise ValueError("bounded_minimiser must be 'scipy' or 'device'")
        self._device_bounds = bounds is not None and method == 'L-BFGS-B' and bounded_minimiser == 'device'
        self._device_minimiser = (method == 'L-BFGS-B' and (bounds is None or self._device_bounds))
        if not self._device_minimiser and self.B != 1:
            raise ValueError("NCG / TNC / bounded_minimiser='scipy' run SciPy on the host around the device "
                             "evaluator: one seed only")

        # initial path (va_ode.py:666-693); init_to_data overwrites the caller's X0 in place, as upstream
        if init_to_data is True:
            X0[..., ::self.merr_nskip, self.Lidx] = self.Y[:]
        ND = self.N_model * self.D
        Xf = np.reshape(np.asarray(X0, dtype=np.float64), (self.B, ND))
        npw = self.N_model * self.NP if self._tdp else self.NP          # stored parameter block, time-major
        Pf = np.reshape(P0, (self.B, npw))
        # positions of the estimated entries inside that block, in path-vector order (va_ode.py:183-188)
        self._estpos = ([n * self.NP + k for n in range(self.N_model) for k in self.Pidx] if self._tdp
                        else list(self.Pidx))
        self._mp = np.zeros((self.B, self.Nbeta, ND + npw), dtype=np.float64)
        self._mp[:, 0, :ND] = Xf
        self._mp[:, 0, ND:] = Pf
        self._A = np.zeros((self.B, self.Nbeta)); self._me = np.zeros((self.B, self.Nbeta))
        self._fe = np.zeros((self.B, self.Nbeta))
        self._flags = np.zeros((self.B, self.Nbeta), dtype=np.int8)
        self._nit = np.zeros((self.B, self.Nbeta), dtype=np.int32)
        self._nfev = np.zeros((self.B, self.Nbeta), dtype=np.int64)
        self._Pfull = np.array(Pf, dtype=np.float64)
        self.adolcID = adolcID                        # accepted, unused: there is no tape

        if rhs_id is None:
            # trace the callable, differentiate it, emit HIP, compile a module.  A model with a column form
            # (codegen.column_form: stencils, small dense systems) or a ghosted form (codegen.ghost_form: wide
            # stencils) also gets the ONE instantiation of the column-run kernel this problem's geometry calls
            # for (va_eval_plan) -- the kernels the built-in Lorenz-96 runs on
            from . import codegen
            nstim = 0 if stim is None else (1 if stim.ndim == 1 else stim.shape[1])

            def variant(ne, ghost, reach=None):
                return _capi.eval_plan(self.B, self.D, self.N_model, disc, ne, ghost,
                                        rm_array=isinstance(self.RM, np.ndarray), rm_full=np.ndim(self.RM) == 3,
                                        rf_array=isinstance(self.RF0, np.ndarray), rf_full=np.ndim(self.RF0) == 3,
                                        merr_nskip=self.merr_nskip,
                                        bounded=self._device_bounds, p_time_dependent=self._tdp,
                                        reach=reach, Lidx=self.Lidx)
            mod = codegen.module_for(self.f, self.D, self.NP, nstim, 1 if stim is None else stim.ndim,
                                     p_rows=self._tdp, col_variant=variant)
            rhs_id = _capi.load_rhs_module(mod["so"])
            self._rhs_module = mod

        # device image
        if self._pb is not None:
            self._pb.close()
        self._pb = _capi.Problem(self.B, self.D, self.N_model, np.asarray(self.Y, dtype=np.float64),
                                 self.Lidx, float(self.dt_model), self.RM, self.RF0,
                                 self._Pfull.reshape(self.B, self.N_model, self.NP) if self._tdp else self._Pfull,
                                 self.Pidx, disc=disc, rhs=rhs_id, merr_nskip=self.merr_nskip,
                                 p_time_dependent=self._tdp,
                                 t_model=np.asarray(self.t_model, dtype=np.float64), stim=stim,
                                 lbfgs_m=int((opt_args or {}).get("maxcor", 10)),
                                 max_beta=self.Nbeta, keep_paths=1, device=device,
                                 bounds=self.bounds if self._device_bounds else None)
        self.initalized = True                        # sic (va_ode.py:705)

    # views with the reference's shapes
    def _view(self, a):
        return a if self._batched else a[0]

    minpaths = property(lambda self: self._view(self._mp))
    A_array = property(lambda self: self._view(self._A))
    me_array = property(lambda self: self._view(self._me))
    fe_array = property(lambda self: self._view(self._fe))
    exitflags = property(lambda self: self._view(self._flags))
    nit_array = property(lambda self: self._view(self._nit))
    nfev_array = property(lambda self: self._view(self._nfev))

    def _rf_print(self):
        return float(np.ravel(self.RF)[0])

    def _xp0(self, k):
        """start point of ladder step k: previous minimiser, estimated parameters only
        (va_ode.py:715-732)"""
        ND = self.N_model * self.D
        src = self._mp[:, k - 1 if k > 0 else 0]
        return np.concatenate([src[:, :ND], src[:, ND:][:, self._estpos]], axis=1)

    def _write_back_P(self):
        """estimated values into the caller's P array (va_ode.py:750-769)"""
        if self.NPest == 0:
            return
        if self._tdp:
            est = self._Pfull.reshape(self.B, self.N_model, self.NP)[:, :, self.Pidx]
            if self._batched:
                self.P[:, :, self.Pidx] = est
            else:
                self.P[:, self.Pidx] = est[0]
        elif self._batched:
            self.P[:, self.Pidx] = self._Pfull[:, self.Pidx]
        else:
            self.P[self.Pidx] = self._Pfull[0, self.Pidx]

    def _store(self, k, x, A, me, fe, flag, nit, nfev):
        ND = self.N_model * self.D
        self._Pfull[:, self._estpos] = x[:, ND:]
        self._write_back_P()
        self._A[:, k] = A; self._me[:, k] = me; self._fe[:, k] = fe      # :773-775
        self._mp[:, k, :ND] = x[:, :ND]; self._mp[:, k, ND:] = self._Pfull  # :776
        self._flags[:, k] = flag; self._nit[:, k] = nit; self._nfev[:, k] = nfev

    def anneal_step(self):
        """One ladder step for every seed (va_ode.py:707-789)."""
        k = self.betaidx
        XP0 = self._xp0(k)
        rf = float(self._rf_scale[k])
        t0 = time.time()
        if self._device_minimiser:
            r = self._pb.minimize_lbfgs(XP0, rf, self.opt_args)
            x, A, me, fe, flag, nit, nfev = r["x"], r["A"], r["me"], r["fe"], r["status"], r["nit"], r["nfev"]
            msg = None
        else:
            x, A, me, fe, flag, nit, nfev, msg = self._minimize_scipy(XP0, rf)
        self._store(k, x, A, me, fe, flag, nit, nfev)
        if self.verbose:
            print("Optimization complete!")
            print("Time = {0} s".format(time.time() - t0))
            print("Exit flag = {0}".format(flag[0] if self.B == 1 else flag))
            if msg is not None:
                print("Exit message: {0}".format(msg))
            print("Iterations = {0}".format(nit[0] if self.B == 1 else nit))
            print("Obj. function value = {0}\n".format(A[0] if self.B == 1 else A))
        if self.betaidx < len(self.beta_array) - 1:                   # va_ode.py:779-782
            self.betaidx += 1
            self.beta = self.beta_array[self.betaidx]
            self.RF = self.RF0 * _alpha_pow(self.alpha, self.beta)
        self.taped = False

    def _anneal_fused(self):
        """Remaining ladder steps in one va_anneal call; seeds advance independently."""
        k0 = self.betaidx
        XP0 = self._xp0(k0)
        t0 = time.time()
        # the minimising path of every step (va_ode.py:776) comes back in the same call
        r = self._pb.anneal(XP0, self._rf_scale[k0:], self.opt_args, want_paths=True)
        ND = self.N_model * self.D
        nb = self.Nbeta - k0
        self._A[:, k0:] = r["A"]; self._me[:, k0:] = r["me"]; self._fe[:, k0:] = r["fe"]
        self._flags[:, k0:] = r["status"]; self._nit[:, k0:] = r["nit"]; self._nfev[:, k0:] = r["nfev"]
        if self._tdp:                                 # rows come back as [X | p_est], time-major
            mp = r["minpaths"]
            self._mp[:, k0:, :ND] = mp[:, :, :ND]
            self._mp[:, k0:, ND:] = self._Pfull[:, None, :]
            self._mp[:, k0:, [ND + j for j in self._estpos]] = mp[:, :, ND:]
        else:
            self._mp[:, k0:] = r["minpaths"]
        self._Pfull[:] = self._mp[:, -1, ND:]
        self._write_back_P()
        self.betaidx = self.Nbeta - 1
        self.beta = self.beta_array[self.betaidx]
        self.RF = self.RF0 * _alpha_pow(self.alpha, self.beta)
        if self.verbose:
            dt = time.time() - t0
            for j in range(nb):
                k = k0 + j
                print('Step %d of %d  beta = %d  RF = %.8e  exit flag = %s  iterations = %s  A = %s'
                      % (k + 1, self.Nbeta, self.beta_array[k],
                         float(np.ravel(self.RF0)[0]) * self._rf_scale[k],
                         self._view(self._flags)[..., k], self._view(self._nit)[..., k],
                         self._view(self._A)[..., k]))
            print("\nLadder of %d steps x %d seed(s): %.3f s, %d action+gradient evaluations"
                  % (nb, self.B, dt, int(self._nfev[:, k0:].sum())))

    # ------------------------------------------------------------------ S1 evaluator
    def _eval(self, XP, want_grad):
        XP = np.asarray(XP, dtype=np.float64)
        single = XP.ndim == 1
        X2 = np.tile(XP, (self.B, 1)) if single else XP
        rf = float(self._rf_scale[self.betaidx])
        A, me, fe, g = self._pb.action_grad(X2, rf, want_grad=want_grad)
        if single:
            return A[0], me[0], fe[0], (g[0] if want_grad else None)
        return A, me, fe, g

    def A_gaussian(self, XP):
        return self._eval(XP, False)[0]

    A = A_gaussian

    def me_gaussian(self, X):
        """Measurement error of a path (va_ode.py:138-158); X may omit the parameters."""
        X = np.asarray(X, dtype=np.float64)
        ND = self.N_model * self.D
        if X.shape[-1] == ND:
            pad = self._Pfull[:, self._estpos] if X.ndim == 2 else self._Pfull[0, self._estpos]
            X = np.concatenate([X, pad], axis=-1)
        return self._eval(X, False)[1]

    def fe_gaussian(self, XP):
        return self._eval(XP, False)[2]

    def _minimize_scipy(self, XP0, rf):
        """bounds / NCG / TNC: SciPy on the host exactly as _autodiffmin.py:72-146 calls it,
        with the device kernel in the role of A_gradA_taped."""
        import scipy.optimize as opt
        meth = {'L-BFGS-B': 'L-BFGS-B', 'NCG': 'CG', 'TNC': 'TNC'}[self.method]

        def fg(z):
            A, me, fe, g = self._pb.action_grad(z[None, :], rf)
            return A[0], g[0]
        kw = dict(method=meth, jac=True, options=self.opt_args)
        if meth != 'CG':
            kw["bounds"] = self.bounds
        res = opt.minimize(fg, XP0[0], **kw)
        A, me, fe, _ = self._pb.action_grad(res.x[None, :], rf, want_grad=False)
        return (res.x[None, :], np.array([res.fun]), me, fe, np.array([res.status]),
                np.array([res.nit]), np.array([res.nfev]), res.message)

    # ------------------------------------------------------------------ savers (va_ode.py:794-889)
    def save_paths(self, filename, dtype=np.float64, fmt="%.8e"):
        ND = self.N_model * self.D
        sav = np.reshape(self._mp[:, :, :ND], (self.B, self.Nbeta, self.N_model, self.D))
        ts = np.resize(np.reshape(self.t_model, (self.N_model, 1)), (self.B, self.Nbeta, self.N_model, 1))
        sav = np.concatenate((ts, sav), axis=3)
        sav = sav if self._batched else sav[0]
        if filename.endswith('.npy'):
            np.save(filename, sav.astype(dtype))
        else:
            np.savetxt(filename, sav.reshape(-1, self.D + 1), fmt=fmt)

    def save_params(self, filename, dtype=np.float64, fmt="%.8e"):
        if self.NPest == 0:
            print("WARNING: You did not estimate any parameters.  Writing fixed "
                  "parameter values to file anyway.")
        ND = self.N_model * self.D
        sav = self._mp[:, :, ND:]
        if self._tdp:                                 # (Nbeta, N_model, NP), va_ode.py:824-840
            sav = sav.reshape(self.B, self.Nbeta, self.N_model, self.NP)
        sav = sav if self._batched else sav[0]
        if filename.endswith('.npy'):
            np.save(filename, sav.astype(dtype))
        else:
            np.savetxt(filename, sav.reshape(-1, self.NP), fmt=fmt)

    def save_action_errors(self, filename, cmpt=0, dtype=np.float64, fmt="%.8e"):
        sav = np.zeros((self.B, self.Nbeta, 5))
        sav[:, :, 0] = self.beta_array
        sav[:, :, 1] = self._A; sav[:, :, 2] = self._me; sav[:, :, 3] = self._fe
        rf0 = float(np.ravel(self.RF0)[0])            # RF0[0, 0] for array-valued RF0 (va_ode.py:861)
        sav[:, :, 4] = self._fe / (rf0 * _alpha_pow(self.alpha, self.beta_array))
        sav = sav if self._batched else sav[0]
        if filename.endswith('.npy'):
            np.save(filename, sav.astype(dtype))
        else:
            np.savetxt(filename, sav.reshape(-1, 5), fmt=fmt)

    def save_as_minAone(self, savedir='', savefile=None, seed=0):
        """minAone-style text rows [beta, exitflag, A, path..., params...] (va_ode.py:875-889).
        Upstream saves never-written exitflags; here they hold the minimiser's status."""
        if savedir.endswith('/') is False:
            savedir += '/'
        if savefile is None:
            savefile = savedir + 'D%d_M%d_PATH%d.dat' % (self.D, self.L, self.adolcID)
        else:
            savefile = savedir + savefile
        betaR = self.beta_array.reshape((self.Nbeta, 1))
        exitR = self._flags[seed].reshape((self.Nbeta, 1))
        AR = self._A[seed].reshape((self.Nbeta, 1))
        np.savetxt(savefile, np.hstack((betaR, exitR, AR, self._mp[seed])))

    def gen_xtrace(self):
        """kept for API compatibility (va_ode.py:894-905); nothing is taped here"""
        return np.random.rand(self._mp.shape[-1] - (self._Pfull.shape[1] - len(self._estpos)))

    def close(self):
        if self._pb is not None:
            self._pb.close()
            self._pb = None
