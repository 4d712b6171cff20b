/* oracle/va_oracle.h -- CPU restatement of the variational-annealing hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; varanneal_amd/ never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against the .npz fixtures in tests/golden, which oracle/gen_golden.py produced by
 * running the reference's own action code (varanneal/va_ode.py:130-234,
 * 341-454 under the py3 loader oracle/_refload.py) -- values directly,
 * gradients by complex-step through the reference's A (ADOL-C, the
 * un-vendored third-party AD engine the reference calls at
 * _autodiffmin.py:57-58, is not installed anywhere in this image).
 */
#ifndef VA_ORACLE_H
#define VA_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VAO_DISC_EULER = 0, VAO_DISC_TRAPEZOID = 1, VAO_DISC_SIMPSON_HERMITE = 2,
       VAO_DISC_FORWARDMAP = 3 };
enum { VAO_RHS_LORENZ96 = 0, VAO_RHS_NAKL = 1 /* reserved */ };

typedef struct {
    int32_t D, N_model, N_data, merr_nskip, L;
    const int32_t *Lidx;      /* [L] observed state indices              */
    const double  *Y;         /* [N_data*L] observations                 */
    double         dt_model;
    const double  *rm_array;  /* NULL -> scalar rm; else [N_data*L] diag */
    double         rm;
    const double  *rf0_array; /* NULL -> scalar; else [(N_model-1)*D]    */
    double         rf0;       /* RF = rf0 * rf_scale (rf_scale=alpha^beta) */
    int32_t NP, NPest;
    const int32_t *Pidx;      /* [NPest]                                 */
    const double  *P;         /* [NP] full parameter vector (fixed part) */
    int32_t disc, rhs;
} vao_problem;

/* One (A, me, fe, grad A) evaluation.  XP = [X (N*D row-major) | p_est].
 * grad may be NULL (value only).  Returns 0, or <0 on bad arguments. */
int vao_action_grad(const vao_problem *pb, const double *XP, double rf_scale,
                    double *A, double *me, double *fe, double *grad);

typedef struct {
    int32_t m;          /* maxcor (SciPy default 10)  */
    double  ftol, gtol; /* SciPy names: factr*epsmch, pgtol */
    int32_t maxiter;
    int64_t maxfun;
    int32_t maxls;      /* SciPy default 20 */
} vao_lbfgs_opts;

/* Unbounded L-BFGS following the published L-BFGS-B 3.0 algorithm on its
 * unconstrained path (what scipy.optimize.minimize(method='L-BFGS-B',
 * bounds=None) executes; reference call site _autodiffmin.py:85-86).
 * status: 0 converged, 1 maxiter/maxfun, 2 abnormal (SciPy warnflag).   */
/* The same evaluation for `nseeds` independent seeds that share Y, Lidx, RM, RF0 (pb[s] differ in P
 * only), one seed per OpenMP thread: the all-host-cores CPU baseline of bench.py.
 * XP: [nseeds][n_var], A/me/fe: [nseeds], grad: [nseeds][n_var] or NULL.  Returns the first error. */
int vao_action_grad_batch(const vao_problem *const *pb, int nseeds, const double *XP, double rf_scale,
                          double *A, double *me, double *fe, double *grad);
/* threads an OpenMP region of this library runs on */
int vao_num_threads(void);
void vao_set_num_threads(int n);

/* objective callback of vao_lbfgs_generic: value and gradient at x; non-zero return = failure */
typedef int (*vao_fg_t)(void *ctx, const double *x, double *f, double *g);
int vao_lbfgs_generic(int32_t n, double *x, vao_fg_t fg, void *ctx,
                      const vao_lbfgs_opts *o, double *Amin, int32_t *status,
                      int32_t *nit_out, int64_t *nfev_out);
/* with box bounds lo / hi (n entries each, +-HUGE_VAL = none), handled as the device minimiser handles
 * them: active-set truncation of the same L-BFGS (see va_oracle.c) */
int vao_lbfgs_bounded(int32_t n, double *x, vao_fg_t fg, void *ctx, const double *lo, const double *hi,
                      const vao_lbfgs_opts *o, double *Amin, int32_t *status,
                      int32_t *nit_out, int64_t *nfev_out);
/* L-BFGS-B ITSELF for a problem with bounds: generalised Cauchy point + subspace minimisation, the published
 * algorithm restated routine for routine (va_lbfgsb.inc.c) -- what scipy.optimize.minimize(method='L-BFGS-B',
 * bounds=...) runs at the reference's call site _autodiffmin.py:85-86; pinned against it step for step. */
int vao_lbfgsb(int32_t n, double *x, vao_fg_t fg, void *ctx, const double *lo, const double *hi,
               const vao_lbfgs_opts *o, double *Amin, int32_t *status, int32_t *nit_out, int64_t *nfev_out);
int vao_minimize_lbfgs(const vao_problem *pb, double *XP_inout, double rf_scale,
                       const vao_lbfgs_opts *o, double *Amin, int32_t *status,
                       int32_t *nit, int64_t *nfev);

/* Full ladder for one seed (va_ode.py:459-490, 707-789).  out_minpaths is
 * [nbeta*(N*D+NP)], out_ame is [nbeta*3] = (A, me, fe) per beta step.   */
int vao_anneal(const vao_problem *pb, const double *XP0, double alpha,
               const uint16_t *beta, int32_t nbeta, const vao_lbfgs_opts *o,
               double *out_minpaths, double *out_ame, int32_t *out_status,
               int32_t *out_nit, int64_t *out_nfev);

#ifdef __cplusplus
}
#endif
#endif
