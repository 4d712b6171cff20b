/* oracle/va_oracle.c -- CPU restatement of the variational-annealing hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT (see va_oracle.h for the rules and the
 * parity-pinning statement).  Plain C99, single thread, double precision,
 * written for clarity and to be structurally DIFFERENT from the HIP path:
 * the gradient here is a scatter-style reverse sweep (each residual pushes
 * its adjoint to the rows it read), the HIP kernel is gather-style.
 *
 * Reference lines followed (paths relative to /root/reference):
 *   A_gaussian            varanneal/va_ode.py:130-136
 *   me_gaussian           varanneal/va_ode.py:138-158
 *   fe_gaussian           varanneal/va_ode.py:160-234
 *   disc_euler            varanneal/va_ode.py:341-356
 *   disc_trapezoid        varanneal/va_ode.py:358-380
 *   disc_SimpsonHermite   varanneal/va_ode.py:404-437
 *   disc_forwardmap       varanneal/va_ode.py:439-454
 *   l96 RHS               examples/Lorenz96_D20/Lorenz96_anneal.py:15-16
 *   min_lbfgs_scipy       varanneal/_autodiffmin.py:72-95  (-> SciPy L-BFGS-B)
 *   anneal / anneal_step  varanneal/va_ode.py:459-490, 707-789
 *
 * Third-party algorithm restated (absent from /root/reference, un-pinned by
 * it, README.md:27 "scipy >= 0.18.1"): L-BFGS-B 3.0 (Byrd, Lu, Nocedal, Zhu;
 * Morales & Nocedal 2011) on its unconstrained path, with the MINPACK-2
 * dcsrch/dcstep line search (More' & Thuente 1994).  Behaviourally pinned
 * against scipy 1.15.3 in tests/test_oracle_lbfgs.py.
 */
#include "va_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ RHS */
/* f_i = x_{i-1} (x_{i+1} - x_{i-2}) - x_i + k, cyclic in i.
 * np.roll(x,1,1)[i] = x[i-1]; np.roll(x,-1,1)[i] = x[i+1]; np.roll(x,2,1)[i] = x[i-2]. */
static void l96_f(int D, const double *x, const double *p, double *f)
{
    for (int i = 0; i < D; ++i) {
        int im1 = (i + D - 1) % D, im2 = (i + D - 2) % D, ip1 = (i + 1) % D;
        f[i] = x[im1] * (x[ip1] - x[im2]) - x[i] + p[0];
    }
}
/* gx += J^T s ; gp += (df/dp)^T s  (scatter form: loop over outputs i). */
static void l96_vjp(int D, const double *x, const double *p, const double *s,
                    double *gx, double *gp)
{
    (void)p;
    for (int i = 0; i < D; ++i) {
        int im1 = (i + D - 1) % D, im2 = (i + D - 2) % D, ip1 = (i + 1) % D;
        gx[im1] += s[i] * (x[ip1] - x[im2]);
        gx[ip1] += s[i] * x[im1];
        gx[im2] -= s[i] * x[im1];
        gx[i]   -= s[i];
        gp[0]   += s[i];
    }
}

typedef void (*rhs_f_t)(int, const double *, const double *, double *);
typedef void (*rhs_vjp_t)(int, const double *, const double *, const double *,
                          double *, double *);

static int rhs_lookup(int id, rhs_f_t *f, rhs_vjp_t *v)
{
    switch (id) {
    case VAO_RHS_LORENZ96: *f = l96_f; *v = l96_vjp; return 0;
    default: return -1;
    }
}

/* --------------------------------------------------------------- action */
int vao_action_grad(const vao_problem *pb, const double *XP, double rf_scale,
                    double *A_out, double *me_out, double *fe_out, double *grad)
{
    const int D = pb->D, N = pb->N_model, L = pb->L, ND = N * D;
    rhs_f_t rf; rhs_vjp_t rv;
    if (rhs_lookup(pb->rhs, &rf, &rv)) return -1;
    if (pb->disc == VAO_DISC_SIMPSON_HERMITE && (N % 2) == 0) return -2;
    if (N < 2 || D < 1) return -3;

    /* parameter vector: copy of P with the estimated entries taken from XP
     * (va_ode.py:165-181). */
    /* scratch: one block per thread, kept between calls (buffers of this size would otherwise be
     * mmap'ed and unmapped on every call, which serialises the threads of vao_action_grad_batch
     * in the kernel) */
    static _Thread_local double *scratch = NULL;
    static _Thread_local size_t scratch_cap = 0;
    const size_t npp = (size_t)(pb->NP > 0 ? pb->NP : 1);
    const size_t need = 2 * (size_t)ND + 2 * npp;
    if (need > scratch_cap) {
        free(scratch);
        scratch = (double *)malloc(sizeof(double) * need);
        scratch_cap = scratch ? need : 0;
        if (!scratch) return -4;
    }
    double *f = scratch;                                    /* f at every row   */
    double *sadj = f + ND;                                  /* adjoint of f rows */
    double *p = sadj + ND;
    double *gpfull = p + npp;
    memset(sadj, 0, sizeof(double) * ND);
    memset(gpfull, 0, sizeof(double) * npp);
    for (int k = 0; k < pb->NP; ++k) p[k] = pb->P[k];
    for (int k = 0; k < pb->NPest; ++k) p[pb->Pidx[k]] = XP[ND + k];

    for (int n = 0; n < N; ++n) rf(D, XP + n * D, p, f + n * D);
    if (grad) memset(grad, 0, sizeof(double) * (ND + pb->NPest));

    /* measurement error, va_ode.py:138-158 */
    double me = 0.0;
    const double cme = 1.0 / ((double)L * pb->N_data);
    for (int n = 0; n < pb->N_data; ++n) {
        const double *xr = XP + (size_t)n * pb->merr_nskip * D;
        for (int l = 0; l < L; ++l) {
            double diff = xr[pb->Lidx[l]] - pb->Y[n * L + l];
            double w = pb->rm_array ? pb->rm_array[n * L + l] : pb->rm;
            me += w * diff * diff;
            if (grad) grad[(size_t)n * pb->merr_nskip * D + pb->Lidx[l]] += 2.0 * w * diff * cme;
        }
    }
    me *= cme;

    /* model error, va_ode.py:160-234 */
    const double dt = pb->dt_model;
    const double cfe = 1.0 / ((double)D * (N - 1));
    double fe = 0.0;
#define RFW(n, i) ((pb->rf0_array ? pb->rf0_array[(n) * D + (i)] : pb->rf0) * rf_scale)
    if (pb->disc == VAO_DISC_SIMPSON_HERMITE) {
        /* va_ode.py:192-195, 204-207, 229-230, 430-435 */
        for (int n = 0; n + 2 < N; n += 2) {
            const double *x0 = XP + n * D, *x1 = x0 + D, *x2 = x1 + D;
            const double *f0 = f + n * D, *f1 = f0 + D, *f2 = f1 + D;
            for (int i = 0; i < D; ++i) {
                double v1 = (f0[i] + 4.0 * f1[i] + f2[i]) * (2.0 * dt) / 6.0;
                double v2 = (x0[i] + x2[i]) / 2.0 + (f0[i] - f2[i]) * (2.0 * dt) / 8.0;
                double d1 = x2[i] - x0[i] - v1;
                double d2 = x1[i] - v2;
                double w1 = RFW(n, i), w2 = RFW(n + 1, i);
                fe += w1 * d1 * d1 + w2 * d2 * d2;
                if (grad) {
                    double q1 = 2.0 * cfe * w1 * d1, q2 = 2.0 * cfe * w2 * d2;
                    grad[(n + 2) * D + i] += q1 - 0.5 * q2;
                    grad[n * D + i]       += -q1 - 0.5 * q2;
                    grad[(n + 1) * D + i] += q2;
                    sadj[n * D + i]       += -q1 * (2.0 * dt) / 6.0 - q2 * (2.0 * dt) / 8.0;
                    sadj[(n + 1) * D + i] += -q1 * 4.0 * (2.0 * dt) / 6.0;
                    sadj[(n + 2) * D + i] += -q1 * (2.0 * dt) / 6.0 + q2 * (2.0 * dt) / 8.0;
                }
            }
        }
    } else {
        for (int n = 0; n + 1 < N; ++n) {
            const double *x0 = XP + n * D, *x1 = x0 + D;
            const double *f0 = f + n * D, *f1 = f0 + D;
            for (int i = 0; i < D; ++i) {
                double diff;
                if (pb->disc == VAO_DISC_TRAPEZOID)
                    diff = x1[i] - x0[i] - dt * (f0[i] + f1[i]) / 2.0;
                else if (pb->disc == VAO_DISC_EULER)
                    diff = x1[i] - x0[i] - dt * f0[i];
                else /* forwardmap, va_ode.py:197 */
                    diff = x1[i] - f0[i];
                double w = RFW(n, i);
                fe += w * diff * diff;
                if (grad) {
                    double q = 2.0 * cfe * w * diff;
                    grad[(n + 1) * D + i] += q;
                    if (pb->disc == VAO_DISC_TRAPEZOID) {
                        grad[n * D + i] -= q;
                        sadj[n * D + i] -= 0.5 * dt * q;
                        sadj[(n + 1) * D + i] -= 0.5 * dt * q;
                    } else if (pb->disc == VAO_DISC_EULER) {
                        grad[n * D + i] -= q;
                        sadj[n * D + i] -= dt * q;
                    } else {
                        sadj[n * D + i] -= q;
                    }
                }
            }
        }
    }
#undef RFW
    fe *= cfe;

    if (grad) {
        for (int n = 0; n < N; ++n)
            rv(D, XP + n * D, p, sadj + n * D, grad + n * D, gpfull);
        for (int k = 0; k < pb->NPest; ++k) grad[ND + k] = gpfull[pb->Pidx[k]];
    }
    *A_out = me + fe; *me_out = me; *fe_out = fe;
    return 0;
}

/* ------------------------------------------------- More'-Thuente dcsrch */
enum { LS_START = 0, LS_FG = 1, LS_CONV = 2, LS_WARN = 3, LS_ERROR = 4 };

typedef struct {
    int brackt, stage;
    double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
} dcsrch_state;

static void dcstep(double *stx, double *fx, double *dx, double *sty, double *fy,
                   double *dy, double *stp, double fp, double dp, int *brackt,
                   double stpmin, double stpmax)
{
    double sgnd = dp * (*dx / fabs(*dx));
    double theta, s, gamma, p, q, r, stpc, stpq, stpf;
    if (fp > *fx) {                                   /* case 1 */
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
        gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp < *stx) gamma = -gamma;
        p = (gamma - *dx) + theta; q = ((gamma - *dx) + gamma) + dp; r = p / q;
        stpc = *stx + r * (*stp - *stx);
        stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2.0) * (*stp - *stx);
        if (fabs(stpc - *stx) < fabs(stpq - *stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2.0;
        *brackt = 1;
    } else if (sgnd < 0.0) {                          /* case 2 */
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
        gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta; q = ((gamma - dp) + gamma) + *dx; r = p / q;
        stpc = *stp + r * (*stx - *stp);
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc; else stpf = stpq;
        *brackt = 1;
    } else if (fabs(dp) < fabs(*dx)) {                /* case 3 */
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = fmax(fabs(theta), fmax(fabs(*dx), fabs(dp)));
        gamma = s * sqrt(fmax(0.0, (theta / s) * (theta / s) - (*dx / s) * (dp / s)));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta; q = (gamma + (*dx - dp)) + gamma; r = p / q;
        if (r < 0.0 && gamma != 0.0) stpc = *stp + r * (*stx - *stp);
        else if (*stp > *stx) stpc = stpmax;
        else stpc = stpmin;
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (*brackt) {
            if (fabs(stpc - *stp) < fabs(stpq - *stp)) stpf = stpc; else stpf = stpq;
            if (*stp > *stx) stpf = fmin(*stp + 0.66 * (*sty - *stp), stpf);
            else stpf = fmax(*stp + 0.66 * (*sty - *stp), stpf);
        } else {
            if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc; else stpf = stpq;
            stpf = fmin(stpmax, stpf); stpf = fmax(stpmin, stpf);
        }
    } else {                                          /* case 4 */
        if (*brackt) {
            theta = 3.0 * (fp - *fy) / (*sty - *stp) + *dy + dp;
            s = fmax(fabs(theta), fmax(fabs(*dy), fabs(dp)));
            gamma = s * sqrt((theta / s) * (theta / s) - (*dy / s) * (dp / s));
            if (*stp > *sty) gamma = -gamma;
            p = (gamma - dp) + theta; q = ((gamma - dp) + gamma) + *dy; r = p / q;
            stpc = *stp + r * (*sty - *stp);
            stpf = stpc;
        } else if (*stp > *stx) stpf = stpmax;
        else stpf = stpmin;
    }
    if (fp > *fx) { *sty = *stp; *fy = fp; *dy = dp; }
    else {
        if (sgnd < 0.0) { *sty = *stx; *fy = *fx; *dy = *dx; }
        *stx = *stp; *fx = fp; *dx = dp;
    }
    *stp = stpf;
}

/* returns new task; *stp updated in place. */
static int dcsrch(double f, double g, double *stp, double ftol, double gtol,
                  double xtol, double stpmin, double stpmax, int task,
                  dcsrch_state *st)
{
    const double xtrapl = 1.1, xtrapu = 4.0;
    if (task == LS_START) {
        if (*stp < stpmin || *stp > stpmax || g >= 0.0) return LS_ERROR;
        st->brackt = 0; st->stage = 1; st->finit = f; st->ginit = g;
        st->gtest = ftol * g; st->width = stpmax - stpmin; st->width1 = st->width / 0.5;
        st->stx = 0.0; st->fx = f; st->gx = g; st->sty = 0.0; st->fy = f; st->gy = g;
        st->stmin = 0.0; st->stmax = *stp + xtrapu * *stp;
        return LS_FG;
    }
    double ftest = st->finit + *stp * st->gtest;
    if (st->stage == 1 && f <= ftest && g >= 0.0) st->stage = 2;
    int out = LS_FG;
    if (st->brackt && (*stp <= st->stmin || *stp >= st->stmax)) out = LS_WARN;
    if (st->brackt && st->stmax - st->stmin <= xtol * st->stmax) out = LS_WARN;
    if (*stp == stpmax && f <= ftest && g <= st->gtest) out = LS_WARN;
    if (*stp == stpmin && (f > ftest || g >= st->gtest)) out = LS_WARN;
    if (f <= ftest && fabs(g) <= gtol * (-st->ginit)) out = LS_CONV;
    if (out != LS_FG) return out;

    if (st->stage == 1 && f <= st->fx && f > ftest) {
        double fm = f - *stp * st->gtest, fxm = st->fx - st->stx * st->gtest,
               fym = st->fy - st->sty * st->gtest, gm = g - st->gtest,
               gxm = st->gx - st->gtest, gym = st->gy - st->gtest;
        dcstep(&st->stx, &fxm, &gxm, &st->sty, &fym, &gym, stp, fm, gm, &st->brackt,
               st->stmin, st->stmax);
        st->fx = fxm + st->stx * st->gtest; st->fy = fym + st->sty * st->gtest;
        st->gx = gxm + st->gtest; st->gy = gym + st->gtest;
    } else {
        dcstep(&st->stx, &st->fx, &st->gx, &st->sty, &st->fy, &st->gy, stp, f, g,
               &st->brackt, st->stmin, st->stmax);
    }
    if (st->brackt) {
        if (fabs(st->sty - st->stx) >= 0.66 * st->width1)
            *stp = st->stx + 0.5 * (st->sty - st->stx);
        st->width1 = st->width; st->width = fabs(st->sty - st->stx);
    }
    if (st->brackt) {
        st->stmin = fmin(st->stx, st->sty); st->stmax = fmax(st->stx, st->sty);
    } else {
        st->stmin = *stp + xtrapl * (*stp - st->stx);
        st->stmax = *stp + xtrapu * (*stp - st->stx);
    }
    *stp = fmax(*stp, stpmin); *stp = fmin(*stp, stpmax);
    if ((st->brackt && (*stp <= st->stmin || *stp >= st->stmax)) ||
        (st->brackt && st->stmax - st->stmin <= xtol * st->stmax))
        *stp = st->stx;
    return LS_FG;
}

/* --------------------------------------------------------------- L-BFGS */
static double ddot(int n, const double *a, const double *b)
{
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* The minimiser on any objective: fg(ctx, x, &f, g) returns non-zero on failure.  (The arbiter for
 * actions other than the ODE one -- oracle/va_nnet_oracle.py drives it through a ctypes callback.)
 *
 * lo / hi (NULL, or n entries each, +-HUGE_VAL for "none"): box bounds, handled the way the DEVICE
 * minimiser handles them (csrc/va_core.h, ls_step / k_direction) -- an active-set truncation of the
 * same L-BFGS, NOT L-BFGS-B's generalised Cauchy point + subspace minimisation:
 *   - x0 is projected onto the box; convergence is tested on the projected gradient (L-BFGS-B's projgr);
 *   - components of d = -H g that would leave the box from a bound they sit on are set to zero;
 *   - the line search runs up to stpmx, the largest step that keeps x + stp d inside the box
 *     (trial points are clamped, so every iterate is feasible to the last bit);
 *   - first trial step min(1/|d|, stpmx) at iteration 0, min(1, stpmx) afterwards;
 *   - the ftol test is skipped after a step that ended on a bound (it was cut short).
 * With no bound active it takes the steps of the unbounded code. */
static double proj_grad(double x, double g, double l, double u)
{
    if (g < 0.0) return fmax(x - u, g);      /* (x - u = -inf without an upper bound) */
    return fmin(x - l, g);
}

int vao_lbfgs_bounded(int32_t n, double *x, vao_fg_t fg, void *ctx, const double *lo, const double *hi,
                      const vao_lbfgs_opts *o, double *Amin, int32_t *status,
                      int32_t *nit_out, int64_t *nfev_out)
{
    const int m = o->m, bnd = lo && hi;
    const double epsmch = DBL_EPSILON, big = 1e10;
    double *g = (double *)malloc(sizeof(double) * n), *d = (double *)malloc(sizeof(double) * n);
    double *t = (double *)malloc(sizeof(double) * n), *r = (double *)malloc(sizeof(double) * n);
    double *S = (double *)malloc(sizeof(double) * (size_t)m * n);
    double *Y = (double *)malloc(sizeof(double) * (size_t)m * n);
    double *rho = (double *)malloc(sizeof(double) * m), *al = (double *)malloc(sizeof(double) * m);
    double f, fold = 0.0, theta = 1.0;
    int col = 0, head = 0;   /* circular history: slot (head+j)%m, j=0 oldest */
    int iter = 0, rc = 0, warn = 2;
    int64_t nfev = 0;

    if (bnd) for (int i = 0; i < n; ++i) x[i] = fmin(fmax(x[i], lo[i]), hi[i]);
    if (fg(ctx, x, &f, g)) { rc = -1; goto done; }
    nfev = 1;
    {
        double sb = 0.0;
        for (int i = 0; i < n; ++i) sb = fmax(sb, fabs(bnd ? proj_grad(x[i], g[i], lo[i], hi[i]) : g[i]));
        if (sb <= o->gtol) { warn = 0; goto done; }
    }
    for (;;) {
        /* direction d = -H g (two-loop; H0 = 1/theta) */
        for (int i = 0; i < n; ++i) d[i] = -g[i];
        for (int j = col - 1; j >= 0; --j) {
            int s = (head + j) % m;
            al[j] = rho[s] * ddot(n, S + (size_t)s * n, d);
            for (int i = 0; i < n; ++i) d[i] -= al[j] * Y[(size_t)s * n + i];
        }
        for (int i = 0; i < n; ++i) d[i] /= theta;
        for (int j = 0; j < col; ++j) {
            int s = (head + j) % m;
            double be = rho[s] * ddot(n, Y + (size_t)s * n, d);
            for (int i = 0; i < n; ++i) d[i] += (al[j] - be) * S[(size_t)s * n + i];
        }
        double stpmx = big;
        if (bnd) {
            for (int i = 0; i < n; ++i) {
                if ((x[i] <= lo[i] && d[i] < 0.0) || (x[i] >= hi[i] && d[i] > 0.0)) d[i] = 0.0;
                if (d[i] > 0.0) stpmx = fmin(stpmx, (hi[i] - x[i]) / d[i]);
                else if (d[i] < 0.0) stpmx = fmin(stpmx, (lo[i] - x[i]) / d[i]);
            }
        }
        /* line search (lnsrlb) */
        double dtd = ddot(n, d, d), dnorm = sqrt(dtd), stp;
        stp = (iter == 0) ? fmin(1.0 / dnorm, stpmx) : fmin(1.0, stpmx);
        memcpy(t, x, sizeof(double) * n); memcpy(r, g, sizeof(double) * n);
        fold = f;
        int ifun = 0, iback = 0, info = 0, task = LS_START;
        double gd = 0.0, gdold = 0.0;
        dcsrch_state ls;
        for (;;) {
            gd = ddot(n, d, g);
            if (ifun == 0) { gdold = gd; if (gd >= 0.0) { info = -4; break; } }
            task = dcsrch(f, gd, &stp, 1e-3, 0.9, 0.1, 0.0, stpmx, task, &ls);
            if (task == LS_ERROR) { info = -4; break; }
            if (task == LS_CONV || task == LS_WARN) break;
            ++ifun; iback = ifun - 1;
            if (iback >= o->maxls) break;
            ++nfev;
            for (int i = 0; i < n; ++i) x[i] = stp * d[i] + t[i];
            if (bnd) for (int i = 0; i < n; ++i) x[i] = fmin(fmax(x[i], lo[i]), hi[i]);
            if (fg(ctx, x, &f, g)) { rc = -1; goto done; }
        }
        if (info != 0 || iback >= o->maxls) {
            memcpy(x, t, sizeof(double) * n); memcpy(g, r, sizeof(double) * n); f = fold;
            if (col == 0) { warn = 2; ++iter; goto done; }   /* ABNORMAL_TERMINATION_IN_LNSRCH */
            col = 0; head = 0; theta = 1.0;                   /* RESTART_FROM_LNSRCH */
            continue;
        }
        /* NEW_X */
        ++iter;
        if (iter >= o->maxiter) { warn = 1; goto done; }      /* SciPy wrapper order */
        if (nfev > o->maxfun) { warn = 1; goto done; }
        double sb = 0.0;
        for (int i = 0; i < n; ++i) sb = fmax(sb, fabs(bnd ? proj_grad(x[i], g[i], lo[i], hi[i]) : g[i]));
        if (sb <= o->gtol) { warn = 0; goto done; }
        if (!(bnd && stp >= stpmx)) {       /* (a step that ended on a bound was cut short: no verdict on progress) */
            double dd = fmax(fmax(fabs(fold), fabs(f)), 1.0);
            if (fold - f <= o->ftol * dd) { warn = 0; goto done; }
        }
        /* BFGS update (matupd) with the L-BFGS-B skip rule */
        for (int i = 0; i < n; ++i) r[i] = g[i] - r[i];
        double rr = ddot(n, r, r), dr, ddum;
        if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
        else { dr = (gd - gdold) * stp; for (int i = 0; i < n; ++i) d[i] *= stp; ddum = -gdold * stp; }
        if (dr <= epsmch * ddum) continue;                    /* skip update */
        int slot;
        if (col < m) { slot = (head + col) % m; ++col; }
        else { slot = head; head = (head + 1) % m; }
        memcpy(S + (size_t)slot * n, d, sizeof(double) * n);
        memcpy(Y + (size_t)slot * n, r, sizeof(double) * n);
        rho[slot] = 1.0 / dr;
        theta = rr / dr;
    }
done:
    *Amin = f; *status = warn; *nit_out = iter; *nfev_out = nfev;
    free(g); free(d); free(t); free(r); free(S); free(Y); free(rho); free(al);
    return rc;
}

int vao_lbfgs_generic(int32_t n, double *x, vao_fg_t fg, void *ctx,
                      const vao_lbfgs_opts *o, double *Amin, int32_t *status,
                      int32_t *nit_out, int64_t *nfev_out)
{
    return vao_lbfgs_bounded(n, x, fg, ctx, NULL, NULL, o, Amin, status, nit_out, nfev_out);
}

typedef struct { const vao_problem *pb; double rf_scale; } ode_ctx;
static int ode_fg(void *c, const double *x, double *f, double *g)
{
    const ode_ctx *k = (const ode_ctx *)c;
    double me, fe;
    return vao_action_grad(k->pb, x, k->rf_scale, f, &me, &fe, g);
}

int vao_minimize_lbfgs(const vao_problem *pb, double *x, double rf_scale,
                       const vao_lbfgs_opts *o, double *Amin, int32_t *status,
                       int32_t *nit_out, int64_t *nfev_out)
{
    ode_ctx c = {pb, rf_scale};
    return vao_lbfgs_generic(pb->N_model * pb->D + pb->NPest, x, ode_fg, &c, o, Amin, status, nit_out, nfev_out);
}

/* --------------------------------------------------------------- ladder */
int vao_anneal(const vao_problem *pb0, const double *XP0, double alpha,
               const uint16_t *beta, int32_t nbeta, const vao_lbfgs_opts *o,
               double *out_minpaths, double *out_ame, int32_t *out_status,
               int32_t *out_nit, int64_t *out_nfev)
{
    vao_problem pb = *pb0;
    const int ND = pb.N_model * pb.D, nv = ND + pb.NPest, wide = ND + pb.NP;
    double *P = (double *)malloc(sizeof(double) * (pb.NP > 0 ? pb.NP : 1));
    double *xp = (double *)malloc(sizeof(double) * nv);
    for (int k = 0; k < pb.NP; ++k) P[k] = pb0->P[k];
    pb.P = P;
    memcpy(xp, XP0, sizeof(double) * nv);
    int rc = 0;
    for (int b = 0; b < nbeta && rc == 0; ++b) {
        double rf_scale = pow(alpha, (double)beta[b]);        /* va_ode.py:650,782 */
        double A, me, fe; int32_t st, nit; int64_t nfev;
        rc = vao_minimize_lbfgs(&pb, xp, rf_scale, o, &A, &st, &nit, &nfev);
        for (int k = 0; k < pb.NPest; ++k) P[pb.Pidx[k]] = xp[ND + k];   /* :750-756 */
        double A2;
        vao_action_grad(&pb, xp, rf_scale, &A2, &me, &fe, NULL);        /* :774-775 */
        out_ame[3 * b] = A; out_ame[3 * b + 1] = me; out_ame[3 * b + 2] = fe;
        memcpy(out_minpaths + (size_t)b * wide, xp, sizeof(double) * ND);
        memcpy(out_minpaths + (size_t)b * wide + ND, P, sizeof(double) * pb.NP);
        out_status[b] = st; out_nit[b] = nit; out_nfev[b] = nfev;
    }
    free(P); free(xp);
    return rc;
}


/* ------------------------------------------------------- batch of seeds */
int vao_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* threads the next batch calls use (a container's CPU quota can be far below the cores it can see) */
void vao_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int vao_action_grad_batch(const vao_problem *const *pb, int nseeds, const double *XP, double rf_scale,
                          double *A, double *me, double *fe, double *grad)
{
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int s = 0; s < nseeds; ++s) {
        const size_t nv = (size_t)pb[s]->N_model * pb[s]->D + pb[s]->NPest;
        int r = vao_action_grad(pb[s], XP + s * nv, rf_scale, A + s, me + s, fe + s, grad ? grad + s * nv : NULL);
        if (r) {
#pragma omp critical
            if (!rc) rc = r;
        }
    }
    return rc;
}

/* L-BFGS-B with bounds (generalised Cauchy point + subspace minimisation): its own file, same translation unit */
#include "va_lbfgsb.inc.c"
