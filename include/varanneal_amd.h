/* include/varanneal_amd.h -- C-ABI of the MI355X-native variational-annealing
 * hot path (libvaranneal_amd.so, HIP for gfx950).
 *
 * The reference (paulrozdeba/varanneal) has no FFI: its seam is the ADmin
 * mixin that va_ode.Annealer inherits (varanneal/va_ode.py:41,43).  Each entry
 * point below names the reference interface it stands in for; the ctypes
 * binding a maintainer would add is shown in INTEGRATION.md and implemented in
 * varanneal_amd/_capi.py.
 *
 * Conventions
 *   - every function returns VA_OK (0) or a negative VA_E* code; never exits,
 *     never throws (reference: print + sys.exit(1), va_ode.py:622-623,637-638).
 *     va_last_error() returns a thread-local message for the last failure.
 *   - caller owns every pointer it passes; host inputs need only stay valid
 *     for the duration of the call.
 *   - a handle is bound to one device and one HIP stream and is NOT re-entrant;
 *     distinct handles may be used from distinct threads.
 *   - all arithmetic is float64.
 *   - path vectors use the reference's packing (va_ode.py:688-693):
 *         XP[b] = [ X[b] flattened (N_model*D, time-major) | p_est (NPest) ]
 *     `ld` is the stride in doubles between consecutive seeds (>= n_var).
 *     `mem` says where XP/grad/outputs live: host (copied in/out) or device
 *     (used in place, no PCIe traffic).
 */
#ifndef VARANNEAL_AMD_H
#define VARANNEAL_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VA_ABI_VERSION 12

enum { VA_OK = 0, VA_EINVAL = -1, VA_ENOMEM = -2, VA_EHIP = -3, VA_EUNSUPPORTED = -4,
       VA_ESTATE = -5 };

/* discretisations: va_ode.py:341-356 (euler), 358-380 (trapezoid),
 * 404-437 (SimpsonHermite, needs odd N_model), 439-454 (forwardmap) */
enum { VA_DISC_EULER = 0, VA_DISC_TRAPEZOID = 1, VA_DISC_SIMPSON_HERMITE = 2,
       VA_DISC_FORWARDMAP = 3 };

/* right-hand sides (the user `f(t,x,p)` of set_model, va_ode.py:56-67).
 * LORENZ96: built in; examples/Lorenz96_D20/Lorenz96_anneal.py:15-16, NP = 1 (forcing k).
 * ids >= VA_RHS_USER_BASE: generated-code modules registered with va_rhs_load_module
 * (varanneal_amd/codegen.py traces the user's Python callable, emits f, J^T v and
 * (df/dp)^T v as HIP and compiles them for gfx950). */
enum { VA_RHS_LORENZ96 = 0, VA_RHS_USER_BASE = 1000 };

enum { VA_MEM_HOST = 0, VA_MEM_DEVICE = 1 };

typedef struct va_problem_s *va_handle;

/* Everything anneal_init() freezes (va_ode.py:531-705). */
typedef struct va_problem_desc {
    int32_t struct_size;      /* = sizeof(va_problem_desc)                          */
    int32_t device;           /* HIP device ordinal                                  */
    int32_t batch;            /* B: independent initial paths ("seeds") resident     */
    int32_t D;                /* state dimension (set_model)                         */
    int32_t N_model;          /* time points of the path                             */
    int32_t N_data;           /* observation times                                   */
    int32_t merr_nskip;       /* int(dt_data/dt_model), va_ode.py:556                */
    int32_t L;                /* observed components                                 */
    const int32_t *Lidx;      /* [L]                                                 */
    const double *Y;          /* [N_data*L] observations, shared by all seeds        */
    double dt_model;
    int32_t rm_kind;          /* 0: scalar rm;  1: rm_array [N_data*L] (va_ode.py:147-148);
                               * 2: rm_array [N_data*L*L], full precision matrices (va_ode.py:149-152) */
    double rm;
    const double *rm_array;
    int32_t rf_kind;          /* 0: scalar rf0; 1: rf0_array [(N_model-1)*D] (va_ode.py:203-209);
                               * 2: rf0_array [(N_model-1)*D*D] full matrices, diff_n.(RF_n.diff_n) (va_ode.py:211-217) */
    double rf0;
    const double *rf0_array;
    int32_t NP;               /* parameters of the RHS                               */
    int32_t NPest;            /* how many are estimated                              */
    const int32_t *Pidx;      /* [NPest] indices into the parameter vector           */
    const double *P;          /* [B*NP] full parameter vector of every seed          */
    int32_t disc;             /* VA_DISC_*                                           */
    int32_t rhs;              /* VA_RHS_*                                            */
    int32_t lbfgs_m;          /* history pairs kept on device (SciPy maxcor, 10)     */
    int32_t max_beta;         /* longest ladder va_anneal will be asked for (>=1)    */
    int32_t keep_paths;       /* 1: keep every beta step's path on device (minpaths) */
    int32_t tile_rows;        /* 0 = auto; time rows per workgroup                   */
    int32_t eval_kernel;      /* 0 = auto; 1 = flat-mapped, 3 = workgroup column runs, 4 = wave-private column runs,
                               * 5 = streaming column strips (a kernel that does not apply falls back)          */
    const double *t_model;    /* NULL or [N_model]: times passed to a non-autonomous RHS (va_ode.py:553,558) */
    const double *stim;       /* NULL or [N_model*n_stim]: external stimulus rows, f(t,x,(p,stim)) (va_ode.py:345-375) */
    int32_t n_stim;
    int32_t p_time_dependent; /* 1: P is [B][N_model][NP] (P0.ndim == 2 upstream, va_ode.py:170-188); the path
                               * vector is then [X (N_model*D) | p_est (N_model*NPest), time-major],
                               * n_var = N_model*(D+NPest); trapezoid and SimpsonHermite only (upstream's
                               * euler/forwardmap branches slice p one row short, va_ode.py:345-349) */
    void *stream;             /* hipStream_t to run on; NULL = library-owned stream  */
    const double *lower;      /* NULL, or [n_var] box bounds of the path vector [X | p_est] shared by all seeds     */
    const double *upper;      /* (va_ode.py:582-605 expands `bounds` to this order); -/+HUGE_VAL = no bound.        */
                              /* With bounds va_minimize_lbfgs / va_anneal run L-BFGS-B itself on the device: the    */
                              /* direction step is csrc/va_lbfgsb.hip (generalised Cauchy point + subspace            */
                              /* minimisation, one workgroup per seed) in k_direction's place; same line search,     */
                              /* stopping rules on the projected gradient.  Iterates follow SciPy's step for step    */
                              /* (tests/test_gpu_codegen.py::test_bounded_device_minimiser_is_lbfgsb_step_for_step). */
} va_problem_desc;

/* SciPy option names (reference passes opt_args through, _autodiffmin.py:85-86) */
typedef struct va_lbfgs_opts {
    int32_t maxcor;           /* <= desc.lbfgs_m                                     */
    double ftol;              /* (f_k - f_{k+1})/max(|f_k|,|f_{k+1}|,1) <= ftol      */
    double gtol;              /* max|g_i| <= gtol                                    */
    int32_t maxiter;
    int64_t maxfun;
    int32_t maxls;
} va_lbfgs_opts;

int32_t va_abi_version(void);
const char *va_last_error(void);
int va_device_count(int32_t *count);

/* Register a compiled right-hand-side module (a shared object built from
 * varanneal_amd/csrc/va_user_rhs.hip + a generated header); returns its rhs id. */
int va_rhs_load_module(const char *path, int32_t *rhs_id);

/* For the code generator: which instantiation of a column-run evaluation kernel a problem of this shape
 * would run for a model that has a column form publishing `ne` products per element (wave-private kernel
 * k_eval4, narrow states; 0 = no such form) and / or a ghosted form with `ghost` ghost columns per side
 * (workgroup kernel k_eval3, wide states; 0 = none).  Reads only the sizes, kinds and flags of `desc`
 * (pointers other than lower/upper are not followed; no GPU call).
 * out[4] = (eval kernel: 0 flat / 3 / 4, disc, rows per lane run K,
 *           kernel 4: 1 if scalar weights else 0; kernel 3: threads per workgroup).
 * A module built for exactly that instantiation runs it; any other problem runs the module's flat kernel. */
int va_eval_plan(const va_problem_desc *desc, int32_t ne, int32_t ghost, int32_t *out);
/* The same for a column form whose reaches are known: reach[4] = {xl, xr, gl, gr} = how many columns to the left /
 * right f reads (xl, xr) and the adjoint gather receives from (gl, gr).  Wide even states (D > 64) with
 * scalar or per-row weights (any merr_nskip) then run the
 * streaming kernel k_eval5 (csrc/va_eval5.h): out = (5, disc, 0, 0).  Follows desc->Lidx (needs L and Lidx). */
int va_eval_plan_reach(const va_problem_desc *desc, int32_t ne, int32_t ghost, const int32_t *reach, int32_t *out);

int va_problem_create(const va_problem_desc *desc, va_handle *out);
void va_problem_destroy(va_handle h);

/* n_var = N_model*D + NPest; ld_internal = device stride the library uses. */
int va_problem_info(va_handle h, int64_t *n_var, int64_t *ld_internal, int32_t *tile_rows,
                    int32_t *ntiles);

/* Performance knobs of a handle.  None changes a result: the same sums are formed in the same order whichever
 * launch forms them (tests/test_gpu_parity.py::test_tuning_knobs_leave_every_bit_alone).  There are no environment
 * variables behind the library's choices; bench.py reports the knobs it was asked to set. */
#define VA_TUNE_FOLD 1       /* 1: the seed's last-arriving workgroup forms A / runs the line-search step inside the
                              * evaluation kernel (default while the grid is <= 8 workgroups per CU); 0: tail kernels */
#define VA_TUNE_GRAD_SC1 2   /* 1: gradient stores write through (default with FOLD)                                   */
#define VA_TUNE_PRIO 3       /* 1: later-dispatched workgroups of a CU issue at higher priority (default); 2: a seed's  */
                             /* first and last tile (the edge variant of the rows phase) issue ahead; 0: no priorities */
#define VA_TUNE_GRAPH 4      /* 1: ladder cycles and timed evaluations are replayed from a hipGraph (default)        */
#define VA_TUNE_PERSIST 5    /* 1: ladders of few seeds on short paths run as ONE launch of the persistent
                              * per-seed kernel (csrc/va_persist.h: every vector of the minimisation resident in LDS),
                              * when the problem is eligible (default); 0: always the three-launch cycle.  Same
                              * arithmetic, different order of the partial sums: results agree to rounding, not bit for bit */
#define VA_TUNE_PERSIST_ROWS 6   /* time rows per workgroup of the persistent kernel (0: the library's choice); VA_EINVAL when not admissible */
#define VA_TUNE_NNET_FUSED 7     /* network handles whose layers are at most 128 wide (at most 64 layers, scalar RM): 1 forward and
                                 * state-gradient products in one kernel (k_nnet_fb), 0 the separate k_nnet_fwd + k_nnet_bwd_x.
                                 * The library's own choice: 1 when blocks of 32 examples x seeds >= 2 x CUs and the activation is a
                                 * built-in other than softplus.  2 + t: as 1, the first workgroups' starts spread over t us (measurement) */
int va_problem_tune(va_handle h, int32_t what, int32_t value);
/* 1 if va_anneal / va_minimize_lbfgs on this handle run the persistent per-seed kernel, with its geometry
 * (workgroups per seed, time rows per workgroup); 0 otherwise. */
int va_problem_persistent(va_handle h, int32_t *workgroups_per_seed, int32_t *rows_per_workgroup);

/* Which evaluation kernel the handle runs (the values of va_problem_desc.eval_kernel: 1 flat,
 * 3 workgroup column runs, 4 wave-private column runs, 5 streaming column strips) and the rows per lane run
 * (0 for the flat kernel, 2 = rows per ring slot for the streaming kernel). */
int va_problem_eval_kernel(va_handle h, int32_t *eval_kernel, int32_t *run_rows);

/* S1 evaluator -- replaces ADmin.A_gradA_taped (_autodiffmin.py:57-58), batched:
 * (A, me, fe, grad A) for all B seeds at RF = RF0*rf_scale.  me/fe follow
 * me_gaussian/fe_gaussian (va_ode.py:138-234).  A/me/fe: [B]; grad: [B*ldg] or NULL.
 * Outputs live where `mem` says. */
int va_action_grad(va_handle h, const double *XP, int64_t ld, int32_t mem, double rf_scale,
                   double *A, double *me, double *fe, double *grad, int64_t ldg);

/* S2 minimiser -- replaces ADmin.min_lbfgs_scipy (_autodiffmin.py:72-95) for
 * bounds=None: device-resident batched L-BFGS with L-BFGS-B's stopping rules.
 * status[b]: 0 converged, 1 maxiter/maxfun reached, 2 abnormal (SciPy warnflag).
 * XP_inout (mem) receives the minimisers; Amin/status/nit/nfev are HOST arrays [B]. */
int va_minimize_lbfgs(va_handle h, double *XP_inout, int64_t ld, int32_t mem, double rf_scale,
                      const va_lbfgs_opts *opts, double *Amin, double *me, double *fe,
                      int32_t *status, int32_t *nit, int64_t *nfev);

/* S3 ladder -- replaces the beta loop of Annealer.anneal + anneal_step
 * (va_ode.py:474-490, 707-789) for all B seeds, each seed climbing the ladder
 * at its own pace, without returning to the host between steps.
 *   rf_scale[nbeta]  = alpha**beta_k (computed by the caller, va_ode.py:650,782)
 *   XP_inout         start paths in, final-step minimisers out
 *   ame  [B*nbeta*3] HOST: (A, me, fe) per seed per step (va_ode.py:773-775)
 *   pest [B*nbeta*NPest] HOST: estimated parameters per step (may be NULL)
 *   status/nit/nfev [B*nbeta] HOST (may be NULL)
 *   minpaths [B*nbeta*(N_model*D+NP)] HOST or NULL; needs desc.keep_paths=1 */
int va_anneal(va_handle h, double *XP_inout, int64_t ld, int32_t mem, const double *rf_scale,
              int32_t nbeta, const va_lbfgs_opts *opts, double *ame, double *pest,
              int32_t *status, int32_t *nit, int64_t *nfev, double *minpaths);

/* Copy the path stored for (seed, beta step) by the last va_anneal (keep_paths=1):
 * out[N_model*D + NP] HOST. */
int va_get_minpath(va_handle h, int32_t seed, int32_t beta_idx, double *out);

/* Measurement hook for bench.py: `iters` complete S1 evaluations (the same launch va_action_grad
 * makes: A, me, fe and the full gradient are formed every time) on the resident paths (those of
 * the last va_action_grad / va_anneal call), bracketed by HIP events on the handle's stream;
 * returns elapsed ms. */
int va_eval_timed(va_handle h, double rf_scale, int32_t iters, float *elapsed_ms);
/* Everything va_eval_timed(h, rf_scale, iters, ..) would otherwise do before its first launch -- arming the seeds,
 * capturing / instantiating / uploading the hipGraph of the chunk of launches it replays -- done now, so that a caller
 * who brackets va_eval_timed with a wall clock of its own (bench.py) times the launches only.  The graph is keyed on
 * the bytes of the device image it captured: any later change of the handle makes va_eval_timed rebuild it. */
int va_eval_timed_prepare(va_handle h, double rf_scale, int32_t iters);

/* Measurement hook for bench.py: `iters` launches each of the two L-BFGS vector kernels (k_update,
 * k_direction) with every seed's history full (lbfgs_m pairs), bracketed by HIP events; returns the
 * elapsed ms of each.  Overwrites the resident paths and the L-BFGS state: evaluate / anneal again
 * from host data afterwards.  VA_EUNSUPPORTED for a bounded handle (its third launch is k_lbfgsb_dir). */
int va_lbfgs_timed(va_handle h, int32_t iters, float *ms_update, float *ms_direction);
/* Measurement only: `iters` evaluation launches as a ladder cycle makes them (every seed at a line-search trial
 * point x + stp*d, the tail runs one line-search step), each after re-arming the seeds; *ms_eval = the launches'
 * time with the re-arming kernel's own time (measured in a second pass) subtracted.  Leaves the seeds idle. */
int va_eval_ls_timed(va_handle h, double rf_scale, int32_t iters, float *ms_eval);

/* The outputs the last S1 evaluation (va_action_grad or va_eval_timed) left on the device:
 * A/me/fe [B], grad [B*ldg] (any may be NULL), HOST arrays. */
int va_read_eval_outputs(va_handle h, double *A, double *me, double *fe, double *grad, int64_t ldg);

/* Profiling hook: copy the first n doubles of the L-BFGS inner-product partial table to the host. */
int va_debug_read_partials(va_handle h, double *out, int64_t n);

/* Profiling hook: the per-phase tick sums (100 MHz) a -DVA_PZ_STAMPS measurement build of the persistent kernel leaves
 * behind (n <= 14 doubles; zeros from the product library). */
int va_debug_read_persist(va_handle h, double *out, int64_t n);

/* Cumulative counters since create: batched eval launches, seed-evaluations,
 * L-BFGS cycles. */
int va_get_counters(va_handle h, int64_t *eval_launches, int64_t *seed_evals, int64_t *cycles);

/* ---- the single collective of a multi-GPU job -------------------------------
 * Independent annealing runs are sharded over the GPUs of a node, one process per GPU; the reference's
 * counterpart is an SGE array job whose tasks leave their results in files
 * (examples/nnet_barimages/SGEcluster/submit_multiM.sh:14-30, SGE_bardata_anneal.py:129-132).  Nothing is
 * exchanged while the ladders run; at the end ONE RCCL all-gather over xGMI collects the per-seed result
 * tables (KBs: latency-bound).  librccl.so is loaded on first use (dlopen); a job that never gathers
 * does not need it.  The Python host normally issues the same collective through torch.distributed
 * (varanneal_amd/parallel.py: backend "nccl" is RCCL); these entry points serve callers without torch. */
typedef struct va_comm_s *va_comm;
#define VA_COMM_ID_BYTES 128
/* rank 0 calls this once and hands the 128 bytes to the other ranks (file, socket, MPI, ...) */
int va_comm_unique_id(char id[VA_COMM_ID_BYTES]);
/* every rank: join the communicator of `world` ranks on `device` (blocks until all have joined) */
int va_comm_create(const char id[VA_COMM_ID_BYTES], int32_t world, int32_t rank, int32_t device, va_comm *out);
void va_comm_destroy(va_comm c);
/* After va_anneal(h, ..., nbeta, ...): all-gather the result tables of every rank's B seeds (the same B
 * on all ranks) on h's stream.  HOST outputs, rank-major then seed-major:
 *   table  [world*B][nbeta][3 + NPest] = (A, me, fe, estimated parameters) per seed and ladder step
 *   status [world*B][nbeta]            (may be NULL) */
int va_gather_results(va_handle h, va_comm c, int32_t nbeta, double *table, int32_t *status);

/* ---- feed-forward-network action (reference: varanneal/va_nnet.py) -------------------
 * va_nnet.Annealer estimates the neuron states of M training examples and (a subset of)
 * the weights/biases of an all-to-all layered network; the "model error" is the mismatch
 * x_{n+1} - f(W_n x_n + b_n) of every layer transition (va_nnet.py:175-255), the
 * measurement error acts on observed input/output neurons (va_nnet.py:117-173).
 * A handle made here is used with the SAME S1/S2/S3 entry points above; its path vector is
 *     XP[b] = [ X[b]: M examples x NDnet states, example-major | p_est (NPest) ]   (va_nnet.py:440)
 * with NDnet = sum(structure); the flat parameter vector is W_0 (structure[1] x structure[0],
 * row-major), b_0, W_1, b_1, ... (va_nnet.py:194-207).  minpaths rows returned by va_anneal
 * for such a handle are [X | p_est] (n_var wide); the caller scatters p_est into P. */
enum { VA_ACT_SIGMOID = 0,   /* 1/(1+exp(-(W x + b))): examples/nnet_twin/nnet_twin_anneal.py:20-22 */
       VA_ACT_TANH = 1, VA_ACT_LINEAR = 2,
       VA_ACT_RELU = 3,      /* max(W x + b, 0) */
       VA_ACT_SOFTPLUS = 4,  /* log(1 + exp(W x + b)) */
       VA_ACT_USER_BASE = 1000 /* ids >= this: generated activation modules (va_act_load_module) */ };

typedef struct va_nnet_desc {
    int32_t struct_size;      /* = sizeof(va_nnet_desc)                                  */
    int32_t device;
    int32_t batch;            /* B independent initial guesses ("seeds")                  */
    int32_t n_layers;         /* len(structure), >= 2 (set_structure, va_nnet.py:59-69)   */
    const int32_t *structure; /* [n_layers] neurons per layer                             */
    int32_t M;                /* training examples (set_input_data, va_nnet.py:78-91)     */
    int32_t L_in, L_out;      /* observed input / output neurons (va_nnet.py:324-332)     */
    const int32_t *Lidx_in;   /* [L_in]  indices into the input layer                     */
    const int32_t *Lidx_out;  /* [L_out] indices into the output layer                    */
    const double *data_in;    /* [M][L_in]                                                */
    const double *data_out;   /* [M][L_out]                                               */
    double rm_in, rm_out;     /* RM scalar -> both equal; RM = [a, b] -> (a, b) (va_nnet.py:132-147) */
    double rf0;               /* RF = rf0 * rf_scale                                      */
    int32_t NP, NPest;        /* all parameters / estimated ones                          */
    const int32_t *Pidx;      /* [NPest] indices into the flat parameter vector           */
    const double *P;          /* [batch][NP] initial/fixed values                         */
    int32_t activation;       /* VA_ACT_* or an id from va_act_load_module                */
    int32_t lbfgs_m;          /* history pairs kept on the device (0 -> 10)               */
    int32_t max_beta;         /* longest ladder va_anneal will be given                   */
    int32_t keep_paths;       /* store every step's minimiser                             */
    void *stream;             /* hipStream_t to use, or NULL for a private one            */
    const double *rm_in_matrix;  /* NULL, or [L_in][L_in]:  RM = [RMin, RMout] with full matrices,       */
    const double *rm_out_matrix; /* NULL, or [L_out][L_out]: diff.(RM.diff) per example (va_nnet.py:136-139); both or neither */
} va_nnet_desc;

/* Register a compiled activation module (a shared object built from varanneal_amd/csrc/va_user_act.hip + a
 * generated header: the elementwise g of a user layer map f(x, W, b) = g(W x + b), va_nnet.py:71, 260-264);
 * returns the id to put in va_nnet_desc.activation. */
int va_act_load_module(const char *path, int32_t *act_id);

/* Everything va_nnet.Annealer.anneal_init() freezes (va_nnet.py:288-450). */
int va_nnet_problem_create(const va_nnet_desc *desc, va_handle *out);

#ifdef __cplusplus
}
#endif
#endif /* VARANNEAL_AMD_H */
