// va_core.h -- per-tile action/adjoint arithmetic and the per-seed L-BFGS state
// machine, written once as __host__ __device__ code.
//
// The HIP kernels in va_kernels.hip give these functions their parallel
// decomposition (workgroup = one tile of time rows of one seed, LDS staging,
// wave64 reductions).  tests/cpu_emul/ compiles the same header with g++ and
// drives the phases serially, so the index/halo/state-machine logic is covered
// by the CPU test-suite; that emulator is test infrastructure and is not linked
// into the product library.
//
// Reference arithmetic this restates for the device (paths in /root/reference):
//   me_gaussian  varanneal/va_ode.py:138-158     fe_gaussian  :160-234
//   disc_euler :341-356  disc_trapezoid :358-380  disc_SimpsonHermite :404-437
//   disc_forwardmap :439-454   l96  examples/Lorenz96_D20/Lorenz96_anneal.py:15-16
//   min_lbfgs_scipy varanneal/_autodiffmin.py:72-95 -> L-BFGS-B 3.0 (unbounded path)
//   anneal_step  varanneal/va_ode.py:707-789
//
// Gradient: no tape.  With q = (2 RF/(D(N-1))) * w * residual, the adjoint of the
// discretisation stencil at time row m is
//     dA/dx_m = direct_m(q) + J_m^T s_m(q)   (+ measurement term)
//     dA/dp   = sum_m (df_m/dp)^T s_m(q)
// where direct_m and s_m are fixed linear combinations of q at rows m-2..m+1
// (see disc_direct_s below); J_m^T s is hand-coded per RHS.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VA_HD __host__ __device__ inline
// the line-search state machine runs as the tail of the evaluation kernel: a real call would give
// every wave of that kernel a scratch frame
#define VA_HD_FLAT __host__ __device__ __attribute__((always_inline)) inline
#else
#define VA_HD inline
#define VA_HD_FLAT inline
#endif

namespace va {

enum { DISC_EULER = 0, DISC_TRAPEZOID = 1, DISC_SH = 2, DISC_FWDMAP = 3 };
enum { PH_IDLE = 0, PH_START = 1, PH_LS = 2, PH_FINISHED = 3 };
enum { UPD_X = 1, UPD_G = 2, UPD_HIST = 4, UPD_STORE = 8 };
enum { LS_START = 0, LS_FG = 1, LS_CONV = 2, LS_WARN = 3, LS_ERROR = 4 };

constexpr int MAX_M = 32;        // history pairs
constexpr int RHS_MAX_NP = 24;   // parameters the tuned kernels and the rows of partial sums carry (NaKL: 18)
constexpr int RHS_BIG_NP = 128;  // parameters a right-hand side may have on the flat kernel: the gradient partials of those
                                 // beyond RHS_MAX_NP travel in a table of their own (Dev::evp_big); the reference has no cap
                                 // (varanneal/va_ode.py:564-578)
// eval partial columns
enum { EP_ME = 0, EP_FE = 1, EP_GTD = 2, EP_GN2 = 3, EP_GMAX = 4, EP_GP = 5 };
constexpr int EP_N = EP_GP + RHS_MAX_NP;
constexpr int EP_BIG = EP_GP + RHS_BIG_NP;
// update partial columns: 5 fixed + 4 per old history slot
enum { UP_YGT = 0, UP_SGT = 1, UP_YY = 2, UP_SY = 3, UP_GTGT = 4, UP_OLD = 5 };
constexpr int UP_N = UP_OLD + 4 * MAX_M;
enum { DP_GD = 0, DP_DD = 1, DP_STPMX = 2, DP_N = 3 };   // g.d, d.d, largest step inside the box (min)

struct Dims {
    int D, N, ND, ld, L, N_data, nskip, NP, NPest, T, ntiles, B, m, disc, nchunks, chunk;
    int ghost;                 // column-run kernel (emode 3): ghost columns per side = RHS::GHOST (2 for Lorenz-96)
    int emode, RY, NT, maxr;   // eval kernel: 1 = flat-mapped, 3 = column-run (va_tile3.h), 4 = wave-private column runs (va_tile4.h)
    unsigned long long obsmask; // bit i set <=> state column i is observed (D <= 64; Lidx ascending on the device)
    int nprow;                 // eval partial rows per seed (= ntiles: one per workgroup; the network kernels have their own count)
    // time-dependent parameters (va_ode.py:170-188): P is (N, NPt) per seed and the vector is
    // [X (N*D) | p_est (N*NPe), time-major].  Then ND = N*D + N*NPe and NP = NPest = 0 for the
    // L-BFGS kernels (one flat run), and the flat tile kernel uses NPt / NPe.  Static: tdp = 0,
    // NPt = NP, NPe = NPest.
    int tdp, NPt, NPe;
    int lin;                   // 1: the right-hand side carries a dense constant linear part (RHS::LINEAR; va_eval_lin.h): one more staged array, J^T s of that part
    int bounded;               // bit 0: box bounds on the path vector (ProblemPtrs::lo / hi); bit 1: every variable has both bounds (L-BFGS-B's `boxed`)
    double dt, cme, cfe, rm, rf0;
};

// Read-only problem arrays (device pointers on the GPU, heap in the emulator).
struct ProblemPtrs {
    const int *lmap;       // [D] -> l or -1
    const double *Y;       // [N_data*L]
    const double *rm_arr;  // NULL or [N_data*L]
    const double *rf0_arr; // NULL or [(N-1)*D]
    const double *rf0_full; // NULL or [(N-1)*D*D]: full model-error precision matrices (va_ode.py:211-217), flat kernel only
    const int *Pidx;       // [NPest]
    const double *Pfull;   // [B*NP]
    const double *tmodel;  // NULL or [N]: model times, for non-autonomous right-hand sides
    const double *stim;    // NULL or [N*nstim]: external stimulus rows (va_ode.py:345-354)
    int nstim;
    const int *Lidx;       // [L] observed state columns (full-RM problems only)
    const double *rm_full; // NULL or [N_data*L*L]: full measurement precision matrices (va_ode.py:149-152)
    const double *lo, *hi; // NULL or [ld]: box bounds of the path vector (va_ode.py:582-605), +-HUGE_VAL = none
};

struct LsState {
    int brackt, stage;
    double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
};

struct Opts {
    int m, maxiter, maxls;
    long long maxfun;
    double ftol, gtol;
};

// Everything the line-search / ladder step touches: <= 512 bytes so that one wave
// can stage it in LDS with a single coalesced 8-byte load per lane (k_ls).
// (the scalar part: what ls_step reads and writes besides the slot order -- the persistent ladder kernel runs the step on a
// REGISTER copy of this part, va_persist.h; an array member would put the copy in scratch memory)
struct SeedScal {
    int phase, beta_idx, iter, col, head, ifun, iback, ls_task;
    int upd, slot, dir, store_idx, nold, pad0;      // pad0: bounded problems -- bit 0 a pair waits to enter S'Y / S'S (k_lbfgsb_dir), bit 1 the oldest pair was evicted
    long long nfev;
    double f, fold, me, fe, theta, stp, gd, gdold, rf_scale, gn2, dr;
    double stp_upd;         // accepted step the update kernel applies (stp is the NEXT trial step)
    LsState ls;
    double cg;
    double gd_dir;          // g.d of the direction in use (left by k_direction's last arriver)
    double stpmx;           // bounded problems: the largest step along d that stays inside the box
};
struct SeedHot : SeedScal {
    int order[MAX_M];       // history slots, oldest -> newest (after the pending update)
};
static_assert(sizeof(SeedHot) <= 512 && sizeof(SeedHot) % 8 == 0, "SeedHot must fit one wave-wide 8-byte load");

struct SeedState : SeedHot {
    double cY[MAX_M], cS[MAX_M];
    double a[MAX_M], b[MAX_M];
    double SY[MAX_M * MAX_M], YY[MAX_M * MAX_M];
};

// ---------------------------------------------------------------- RHS: Lorenz-96
// f_i = x_{i-1}(x_{i+1} - x_{i-2}) - x_i + k   (cyclic)
struct RhsL96 {
    static constexpr int NP = 1;
    static VA_HD double f(const double *x, int i, int D, const double *p, double, const double *)
    {
        int im1 = i == 0 ? D - 1 : i - 1;
        int im2 = im1 == 0 ? D - 1 : im1 - 1;
        int ip1 = i == D - 1 ? 0 : i + 1;
        return x[im1] * (x[ip1] - x[im2]) - x[i] + p[0];
    }
    // (J^T s)_j = s_{j+1}(x_{j+2} - x_{j-1}) + s_{j-1} x_{j-2} - s_{j+2} x_{j+1} - s_j
    static VA_HD double vjp(const double *x, const double *s, int j, int D, const double *, double, const double *)
    {
        int jm1 = j == 0 ? D - 1 : j - 1;
        int jm2 = jm1 == 0 ? D - 1 : jm1 - 1;
        int jp1 = j == D - 1 ? 0 : j + 1;
        int jp2 = jp1 == D - 1 ? 0 : jp1 + 1;
        return s[jp1] * (x[jp2] - x[jm1]) + s[jm1] * x[jm2] - s[jp2] * x[jp1] - s[j];
    }
    // acc[k] += s_i * df_i/dp_k
    static VA_HD void pgrad(const double *, const double *s, int i, int, const double *, double, const double *, double *acc)
    {
        acc[0] += s[i];
    }
};

// ---------------------------------------------------------------- tile context
template <int DISC> struct Halo {
    static constexpr int HL = (DISC == DISC_SH) ? 2 : 1;   // rows needed before the tile
    static constexpr int HR = 1;                           // rows needed after it
};

// A right-hand side f = A0 x + rest(x, p, t) whose dense CONSTANT linear part A0 is split off by the generator
// (codegen.linear_split): RHS::LINEAR is defined, RHS::f / vjp / pgrad describe the rest only, and the tables
// RHS::lin_A0() = A0 and RHS::lin_A0T() = A0^T (row-major, LIN_DP x LIN_DP, zero-padded to a multiple of 16) carry
// the linear part -- on the device through the matrix cores (va_eval_lin.h), here with plain loops for the emulator.
template <class RHS, class = void> struct rhs_linear { static constexpr bool value = false; };
template <class RHS> struct rhs_linear<RHS, decltype((void)RHS::LINEAR)> { static constexpr bool value = true; };

struct TileCtx {
    double *js = nullptr;        // [T*D] (LDS): J^T s of the dense linear part, device only
    int n0, R, use_d;            // first owned row, rows staged, trial point x + stp*d ?
    double stp, c;               // c = 2 * rf0_scale * cfe   (rf0 weight applied per element)
    double *xs, *fs, *qs;        // staged rows [R*D] (LDS); fs is re-used for s
    const double *xg, *dg;       // this seed's x (and d) in global memory
    const double *zg = nullptr;  // bounded problems: this seed's z (d = z - x), the trial point at step 1
    double *gtg;                 // this seed's gradient output
    long goff = 0;               // gtg / dg are indexed by (path index - goff): 0 for the seed's global vectors, the slice's
                                 // first element when they are a workgroup's LDS-resident slice (va_persist.h)
    const double *tmodel, *stim; // per-row time / stimulus (NULL / unused for autonomous RHS)
    int nstim;
    double *ps;                  // time-dependent parameters: staged rows [R*NPt] (LDS)
    double p[RHS_BIG_NP];        // (statically indexed: the entries a model does not have cost nothing)
};

template <int NV> struct ThreadAccT {
    double v[NV];
    VA_HD void clear() { for (int k = 0; k < NV; ++k) v[k] = 0.0; }
};
typedef ThreadAccT<EP_N> ThreadAcc;

// trial point; the SAME expression is used by the update kernel so that the
// accepted iterate is bit-identical to the point that was evaluated.
VA_HD double trial(double x, double stp, double d) { return fma(stp, d, x); }
// ... and inside the box to the last bit when there is one (stp <= stpmx keeps it there up to rounding)
VA_HD double clampb(double v, const ProblemPtrs &pp, long i) { return pp.lo ? fmin(fmax(v, pp.lo[i]), pp.hi[i]) : v; }
// The trial point of a bounded problem.  At step 1 it is z ITSELF -- L-BFGS-B's `if (stp == one) x = z` (lnsrlb) -- not
// x + (z - x): with |u - x| much larger than |u| the sum lands a few ulp inside the bound, the variable then no longer
// counts as sitting on it, and the next Cauchy step treats it as free.  z: the seed's Cauchy / subspace point (Dev::lb_z), or
// NULL where there are no bounds.
VA_HD double trial_b(double x, double stp, double d, const double *z, const ProblemPtrs &pp, long i)
{
    return (z && stp == 1.0) ? z[i] : clampb(trial(x, stp, d), pp, i);
}
// L-BFGS-B's projected gradient (projgr): what the convergence test looks at when there are bounds
VA_HD double proj_grad(double x, double g, double l, double u) { return g < 0.0 ? fmax(x - u, g) : fmin(x - l, g); }

template <class RHS>
VA_HD void tile_params(const Dims &dm, const ProblemPtrs &pp, int b, TileCtx &c)
{
#pragma unroll
    for (int k = 0; k < RHS::NP; ++k) c.p[k] = pp.Pfull[(size_t)b * dm.NP + k];
    for (int k = 0; k < dm.NPest; ++k) {
        double v = c.xg[dm.ND + k];
        if (c.use_d) v = trial_b(v, c.stp, c.dg[dm.ND + k], c.zg, pp, dm.ND + k);
        const int dst = pp.Pidx[k];
        // select chain instead of c.p[dst]: a runtime-indexed array would live in scratch
#pragma unroll
        for (int j = 0; j < RHS::NP; ++j) c.p[j] = (dst == j) ? v : c.p[j];
    }
}

// phase 1: stage rows [n0-HL, n0-HL+R) of x (or x + stp*d); rows outside [0,N) read as 0.
template <int DISC>
VA_HD void tile_load(const Dims &dm, const ProblemPtrs &pp, TileCtx &c, int tid, int nt)
{
    const long base = (long)(c.n0 - Halo<DISC>::HL) * dm.D;
    const int tot = c.R * dm.D;
    const long NDx = (long)dm.N * dm.D;          // (dm.ND also counts the parameter block when tdp)
    for (int e = tid; e < tot; e += nt) {
        long gi = base + e;
        double v = 0.0;
        if (gi >= 0 && gi < NDx) {
            v = c.xg[gi];
            if (c.use_d) v = trial_b(v, c.stp, c.dg[gi], c.zg, pp, gi);
        }
        c.xs[e] = v;
    }
}

// phase 1b (time-dependent parameters): stage the rows' own parameter vectors -- estimated
// entries from the trial point, the others from the fixed table (va_ode.py:177-188).
template <int DISC>
VA_HD void tile_load_p(const Dims &dm, const ProblemPtrs &pp, int b, TileCtx &c, int tid, int nt)
{
    const int NPt = dm.NPt, tot = c.R * NPt;
    const long NDx = (long)dm.N * dm.D;
    for (int e = tid; e < tot; e += nt) {
        const int lr = e / NPt, k = e - lr * NPt;
        const int row = c.n0 - Halo<DISC>::HL + lr;
        double v = 0.0;
        if (row >= 0 && row < dm.N) {
            int est = -1;
            for (int j = 0; j < dm.NPe; ++j) est = (pp.Pidx[j] == k) ? j : est;
            if (est >= 0) {
                const long gi = NDx + (long)row * dm.NPe + est;
                v = c.xg[gi];
                if (c.use_d) v = trial_b(v, c.stp, c.dg[gi], c.zg, pp, gi);
            } else v = pp.Pfull[((size_t)b * dm.N + row) * NPt + k];
        }
        c.ps[e] = v;
    }
}

// phase 2: f at every staged row that exists.
template <class RHS, int DISC>
VA_HD void tile_f(const Dims &dm, TileCtx &c, int tid, int nt)
{
    const int D = dm.D, tot = c.R * D;
    int lr = tid / D, i = tid - lr * D;
    const int dlr = nt / D, di = nt - dlr * D;
    for (int e = tid; e < tot; e += nt) {
        int row = c.n0 - Halo<DISC>::HL + lr;
        double fv = (row >= 0 && row < dm.N)
                        ? RHS::f(c.xs + lr * D, i, D, dm.tdp ? c.ps + lr * dm.NPt : c.p,
                                 c.tmodel ? c.tmodel[row] : 0.0, c.stim + (size_t)row * c.nstim)
                        : 0.0;
        if constexpr (rhs_linear<RHS>::value) {
#ifdef __HIP_DEVICE_COMPILE__
            fv += c.fs[e];                               // X A0^T, left there by lin_gemm (rows that do not exist are rows of zeros)
#else
            const double *A0 = RHS::lin_A0();
            for (int j = 0; j < D; ++j) fv += A0[i * RHS::LIN_DP + j] * c.xs[lr * D + j];
#endif
        }
        c.fs[e] = fv;
        lr += dlr; i += di;
        if (i >= D) { i -= D; ++lr; }
    }
}

// phase 3: weighted residual adjoints q for rows [n0-HL, n0+T); model-error sum over owned rows.
template <int DISC, class ACC>
VA_HD void tile_q(const Dims &dm, const ProblemPtrs &pp, TileCtx &c, ACC &acc, int tid, int nt)
{
    constexpr int HL = Halo<DISC>::HL;
    const int D = dm.D, N = dm.N, tot = (dm.T + HL) * D;
    const double dt = dm.dt;
    int lr = tid / D, i = tid - lr * D;
    const int dlr = nt / D, di = nt - dlr * D;
    for (int e = tid; e < tot; e += nt) {
        const int row = c.n0 - HL + lr;
        double q = 0.0;
        if (row >= 0) {
            const double *x0 = c.xs + lr * D, *f0 = c.fs + lr * D;
            double r = 0.0; bool have = false;
            if (DISC == DISC_SH) {
                if ((row & 1) == 0) {
                    if (row + 2 <= N - 1) {          // d1 of the interval starting at `row`
                        r = x0[2 * D + i] - x0[i]
                            - (f0[i] + 4.0 * f0[D + i] + f0[2 * D + i]) * (2.0 * dt) / 6.0;
                        have = true;
                    }
                } else if (row + 1 <= N - 1) {       // d2 of the interval starting at row-1
                    r = x0[i] - ((x0[i - D] + x0[i + D]) / 2.0
                                 + (f0[i - D] - f0[i + D]) * (2.0 * dt) / 8.0);
                    have = true;
                }
            } else if (row <= N - 2) {
                if (DISC == DISC_TRAPEZOID) r = x0[D + i] - x0[i] - dt * (f0[i] + f0[D + i]) / 2.0;
                else if (DISC == DISC_EULER) r = x0[D + i] - x0[i] - dt * f0[i];
                else r = x0[D + i] - f0[i];
                have = true;
            }
            if (have) {
                if (pp.rf0_full) q = r;       // the bare residual: tile_qfull contracts the row with its matrix
                else {
                    double w = pp.rf0_arr ? pp.rf0_arr[(size_t)row * D + i] : dm.rf0;
                    q = c.c * w * r;
                    if (lr >= HL && row < N) acc.v[EP_FE] += w * r * r;
                }
            }
        }
        c.qs[e] = q;
        lr += dlr; i += di;
        if (i >= D) { i -= D; ++lr; }
    }
}

// phase 3b (full RF matrices, va_ode.py:211-217): fe = sum_n r_n . (R_n r_n) with R_n = RF0[n] (row n's
// residual; Simpson-Hermite: d1 of the interval at even n, d2 at odd n -- RF[2i], RF[2i+1] upstream).  Reads the
// residuals tile_q left in c.qs and writes the adjoints q_n = (c/2) (R_n + R_n^T) r_n over c.fs (f is no
// longer needed); the caller then swaps the two arrays.  R is not assumed symmetric.
template <int DISC, class ACC>
VA_HD void tile_qfull(const Dims &dm, const ProblemPtrs &pp, TileCtx &c, ACC &acc, int tid, int nt)
{
    constexpr int HL = Halo<DISC>::HL;
    const int D = dm.D, N = dm.N, tot = (dm.T + HL) * D;
    int lr = tid / D, i = tid - lr * D;
    const int dlr = nt / D, di = nt - dlr * D;
    for (int e = tid; e < tot; e += nt) {
        const int row = c.n0 - HL + lr;
        double q = 0.0;
        if (row >= 0 && row <= N - 2) {      // (rows without a residual hold zeros in c.qs)
            const double *R = pp.rf0_full + (size_t)row * D * D;
            const double *r = c.qs + lr * D;
            double rowsum = 0.0, sym = 0.0;
            for (int k = 0; k < D; ++k) {
                rowsum += R[i * D + k] * r[k];
                sym += (R[i * D + k] + R[k * D + i]) * r[k];
            }
            q = 0.5 * c.c * sym;
            if (lr >= HL) acc.v[EP_FE] += r[i] * rowsum;
        }
        c.fs[e] = q;
        lr += dlr; i += di;
        if (i >= D) { i -= D; ++lr; }
    }
}

// direct_m and s_m from q (qm points at q[row m][i]; rows m-2..m+1 are addressable,
// out-of-range rows hold 0 by construction of tile_q).
template <int DISC>
VA_HD void disc_direct_s(const double *qm, int D, int m, double dt, double &direct, double &s)
{
    if (DISC == DISC_TRAPEZOID) {
        direct = qm[-D] - qm[0];
        s = -0.5 * dt * (qm[-D] + qm[0]);
    } else if (DISC == DISC_EULER) {
        direct = qm[-D] - qm[0];
        s = -dt * qm[0];
    } else if (DISC == DISC_FWDMAP) {
        direct = qm[-D];
        s = -qm[0];
    } else {                                          // Simpson-Hermite
        if ((m & 1) == 0) {
            // q1_m = q[m], q2_m = q[m+1], q1_{m-2} = q[m-2], q2_{m-2} = q[m-1]
            direct = -qm[0] - 0.5 * qm[D] + qm[-2 * D] - 0.5 * qm[-D];
            s = -(dt / 3.0) * (qm[0] + qm[-2 * D]) - (dt / 4.0) * (qm[D] - qm[-D]);
        } else {
            direct = qm[0];
            s = -(4.0 * dt / 3.0) * qm[-D];
        }
    }
}

// phase 4: s rows for the owned rows (written over fs).
template <int DISC>
VA_HD void tile_s(const Dims &dm, TileCtx &c, int tid, int nt)
{
    constexpr int HL = Halo<DISC>::HL;
    const int D = dm.D, tot = dm.T * D;
    int lt = tid / D, i = tid - lt * D;
    const int dlr = nt / D, di = nt - dlr * D;
    for (int e = tid; e < tot; e += nt) {
        const int lr = lt + HL, m = c.n0 + lt;
        double direct, s = 0.0;
        if (m < dm.N) disc_direct_s<DISC>(c.qs + lr * D + i, D, m, dm.dt, direct, s);
        c.fs[lr * D + i] = s;
        lt += dlr; i += di;
        if (i >= D) { i -= D; ++lt; }
    }
}

// phase 5: gradient rows of the tile + measurement term + parameter-gradient and
// line-search partial sums.
template <class RHS, int DISC, class ACC>
VA_HD void tile_g(const Dims &dm, const ProblemPtrs &pp, TileCtx &c, ACC &acc, int tid, int nt)
{
    constexpr int HL = Halo<DISC>::HL;
    const int D = dm.D, tot = dm.T * D;
    int lt = tid / D, j = tid - lt * D;
    const int dlr = nt / D, dj = nt - dlr * D;
    for (int e = tid; e < tot; e += nt) {
        const int lr = lt + HL, m = c.n0 + lt;
        if (m < dm.N) {
            double direct, sdummy;
            disc_direct_s<DISC>(c.qs + lr * D + j, D, m, dm.dt, direct, sdummy);
            const double *xr = c.xs + lr * D, *sr = c.fs + lr * D;
            const double tm = c.tmodel ? c.tmodel[m] : 0.0;
            const double *st = c.stim + (size_t)m * c.nstim;
            const double *pr = dm.tdp ? c.ps + lr * dm.NPt : c.p;
            double g = direct + RHS::vjp(xr, sr, j, D, pr, tm, st);
            if constexpr (rhs_linear<RHS>::value) {
#ifdef __HIP_DEVICE_COMPILE__
                g += c.js[lt * D + j];                   // S A0 (lin_gemm)
#else
                const double *A0 = RHS::lin_A0();
                for (int i = 0; i < D; ++i) g += sr[i] * A0[i * RHS::LIN_DP + j];
#endif
            }
            if (!dm.tdp) RHS::pgrad(xr, sr, j, D, pr, tm, st, acc.v + EP_GP);
            const int l = pp.lmap[j];
            if (l >= 0 && (m % dm.nskip) == 0) {
                const int nd = m / dm.nskip;
                const double diff = xr[j] - pp.Y[(size_t)nd * dm.L + l];
                if (pp.rm_full) {
                    // diff^T RM_n diff (va_ode.py:149-152): this element's share is its own row of the
                    // quadratic form; its derivative picks up row and column l of RM_n (not assumed symmetric)
                    const double *R = pp.rm_full + (size_t)nd * dm.L * dm.L;
                    double row = 0.0, sym = 0.0;
                    for (int k = 0; k < dm.L; ++k) {
                        const double dk = xr[pp.Lidx[k]] - pp.Y[(size_t)nd * dm.L + k];
                        row += R[l * dm.L + k] * dk;
                        sym += (R[l * dm.L + k] + R[k * dm.L + l]) * dk;
                    }
                    acc.v[EP_ME] += diff * row;
                    g += dm.cme * sym;
                } else {
                    const double w = pp.rm_arr ? pp.rm_arr[(size_t)nd * dm.L + l] : dm.rm;
                    acc.v[EP_ME] += w * diff * diff;
                    g += 2.0 * dm.cme * w * diff;
                }
            }
            const long gi = (long)m * D + j;
            c.gtg[gi - c.goff] = g;
            if (c.use_d) acc.v[EP_GTD] += g * c.dg[gi - c.goff];
            acc.v[EP_GN2] += g * g;
            acc.v[EP_GMAX] = fmax(acc.v[EP_GMAX], fabs(pp.lo ? proj_grad(xr[j], g, pp.lo[gi], pp.hi[gi]) : g));
        }
        lt += dlr; j += dj;
        if (j >= D) { j -= D; ++lt; }
    }
}

// phase 5b (time-dependent parameters): dA/dp_m = (df/dp)^T s_m row by row -- one thread per
// owned row walks the D state components (generic path, not a tuned one).
template <class RHS, int DISC, class ACC>
VA_HD void tile_gp(const Dims &dm, const ProblemPtrs &pp, TileCtx &c, ACC &acc, int tid, int nt)
{
    constexpr int HL = Halo<DISC>::HL;
    const int D = dm.D;
    const long NDx = (long)dm.N * D;
    for (int lt = tid; lt < dm.T; lt += nt) {
        const int lr = lt + HL, m = c.n0 + lt;
        if (m >= dm.N) continue;
        const double *xr = c.xs + lr * D, *sr = c.fs + lr * D, *pr = c.ps + lr * dm.NPt;
        const double tm = c.tmodel ? c.tmodel[m] : 0.0;
        const double *st = c.stim + (size_t)m * c.nstim;
        double gp[RHS_MAX_NP];
        for (int k = 0; k < RHS_MAX_NP; ++k) gp[k] = 0.0;
        for (int j = 0; j < D; ++j) RHS::pgrad(xr, sr, j, D, pr, tm, st, gp);
        for (int e = 0; e < dm.NPe; ++e) {
            const int dst = pp.Pidx[e];
            double g = 0.0;
            for (int k = 0; k < RHS_MAX_NP; ++k) g = (k == dst) ? gp[k] : g;      // no runtime-indexed array
            const long gi = NDx + (long)m * dm.NPe + e;
            c.gtg[gi] = g;
            if (c.use_d) acc.v[EP_GTD] += g * c.dg[gi];
            acc.v[EP_GN2] += g * g;
            double pv = 0.0;
            for (int k = 0; k < RHS_MAX_NP; ++k) pv = (k == dst) ? pr[k < dm.NPt ? k : 0] : pv;
            acc.v[EP_GMAX] = fmax(acc.v[EP_GMAX], fabs(pp.lo ? proj_grad(pv, g, pp.lo[gi], pp.hi[gi]) : g));
        }
    }
}

// ---------------------------------------------------------------- dcsrch / dcstep
// MINPACK-2 line search (More' & Thuente 1994) as used by L-BFGS-B's lnsrlb.
VA_HD_FLAT void dcstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy,
                  double &stp, double fp, double dp, int &brackt, double stpmin, double stpmax)
{
    const double sgnd = dp * (dx / fabs(dx));
    double theta, s, gamma, p, q, r, stpc, stpq, stpf;
    if (fp > fx) {
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
        gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp < stx) gamma = -gamma;
        p = (gamma - dx) + theta; q = ((gamma - dx) + gamma) + dp; r = p / q;
        stpc = stx + r * (stp - stx);
        stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
        stpf = (fabs(stpc - stx) < fabs(stpq - stx)) ? stpc : stpc + (stpq - stpc) / 2.0;
        brackt = 1;
    } else if (sgnd < 0.0) {
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
        gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
        if (stp > stx) gamma = -gamma;
        p = (gamma - dp) + theta; q = ((gamma - dp) + gamma) + dx; r = p / q;
        stpc = stp + r * (stx - stp);
        stpq = stp + (dp / (dp - dx)) * (stx - stp);
        stpf = (fabs(stpc - stp) > fabs(stpq - stp)) ? stpc : stpq;
        brackt = 1;
    } else if (fabs(dp) < fabs(dx)) {
        theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
        s = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
        gamma = s * sqrt(fmax(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
        if (stp > stx) gamma = -gamma;
        p = (gamma - dp) + theta; q = (gamma + (dx - dp)) + gamma; r = p / q;
        if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
        else if (stp > stx) stpc = stpmax;
        else stpc = stpmin;
        stpq = stp + (dp / (dp - dx)) * (stx - stp);
        if (brackt) {
            stpf = (fabs(stpc - stp) < fabs(stpq - stp)) ? stpc : stpq;
            if (stp > stx) stpf = fmin(stp + 0.66 * (sty - stp), stpf);
            else stpf = fmax(stp + 0.66 * (sty - stp), stpf);
        } else {
            stpf = (fabs(stpc - stp) > fabs(stpq - stp)) ? stpc : stpq;
            stpf = fmin(stpmax, stpf); stpf = fmax(stpmin, stpf);
        }
    } else {
        if (brackt) {
            theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
            s = fmax(fabs(theta), fmax(fabs(dy), fabs(dp)));
            gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
            if (stp > sty) gamma = -gamma;
            p = (gamma - dp) + theta; q = ((gamma - dp) + gamma) + dy; r = p / q;
            stpf = stp + r * (sty - stp);
        } else if (stp > stx) stpf = stpmax;
        else stpf = stpmin;
    }
    {
        // interval update, written as selects on values (branches that assign through the reference
        // parameters make the device compiler keep the six of them in scratch memory)
        const bool hi = fp > fx, flip = !hi && sgnd < 0.0;
        const double nsty = hi ? stp : (flip ? stx : sty), nfy = hi ? fp : (flip ? fx : fy), ndy = hi ? dp : (flip ? dx : dy);
        const double nstx = hi ? stx : stp, nfx = hi ? fx : fp, ndx = hi ? dx : dp;
        sty = nsty; fy = nfy; dy = ndy; stx = nstx; fx = nfx; dx = ndx;
    }
    stp = stpf;
}

VA_HD_FLAT int dcsrch(double f, double g, double &stp, double ftol, double gtol, double xtol,
                 double stpmin, double stpmax, int task, LsState &st)
{
    const double xtrapl = 1.1, xtrapu = 4.0;
    if (task == LS_START) {
        if (stp < stpmin || stp > stpmax || g >= 0.0) return LS_ERROR;
        st.brackt = 0; st.stage = 1; st.finit = f; st.ginit = g; st.gtest = ftol * g;
        st.width = stpmax - stpmin; st.width1 = st.width / 0.5;
        st.stx = 0.0; st.fx = f; st.gx = g; st.sty = 0.0; st.fy = f; st.gy = g;
        st.stmin = 0.0; st.stmax = stp + xtrapu * stp;
        return LS_FG;
    }
    const double ftest = st.finit + stp * st.gtest;
    if (st.stage == 1 && f <= ftest && g >= 0.0) st.stage = 2;
    int out = LS_FG;
    if (st.brackt && (stp <= st.stmin || stp >= st.stmax)) out = LS_WARN;
    if (st.brackt && st.stmax - st.stmin <= xtol * st.stmax) out = LS_WARN;
    if (stp == stpmax && f <= ftest && g <= st.gtest) out = LS_WARN;
    if (stp == stpmin && (f > ftest || g >= st.gtest)) out = LS_WARN;
    if (f <= ftest && fabs(g) <= gtol * (-st.ginit)) out = LS_CONV;
    if (out != LS_FG) return out;
    {
        // (one call of dcstep on local copies, for the modified function or the function itself:
        // two calls on different storage end up behind pointers, i.e. in scratch memory on the device)
        const bool mod = st.stage == 1 && f <= st.fx && f > ftest;
        const double sh = mod ? st.gtest : 0.0;
        double stx = st.stx, sty = st.sty;
        double fxm = st.fx - stx * sh, fym = st.fy - sty * sh, gxm = st.gx - sh, gym = st.gy - sh;
        const double fm = f - stp * sh, gm = g - sh;
        int brackt = st.brackt;
        dcstep(stx, fxm, gxm, sty, fym, gym, stp, fm, gm, brackt, st.stmin, st.stmax);
        st.stx = stx; st.sty = sty; st.brackt = brackt;
        st.fx = fxm + stx * sh; st.fy = fym + sty * sh;
        st.gx = gxm + sh; st.gy = gym + sh;
    }
    if (st.brackt) {
        if (fabs(st.sty - st.stx) >= 0.66 * st.width1) stp = st.stx + 0.5 * (st.sty - st.stx);
        st.width1 = st.width; st.width = fabs(st.sty - st.stx);
        st.stmin = fmin(st.stx, st.sty); st.stmax = fmax(st.stx, st.sty);
    } else {
        st.stmin = stp + xtrapl * (stp - st.stx);
        st.stmax = stp + xtrapu * (stp - st.stx);
    }
    stp = fmax(stp, stpmin); stp = fmin(stp, stpmax);
    if ((st.brackt && (stp <= st.stmin || stp >= st.stmax)) ||
        (st.brackt && st.stmax - st.stmin <= xtol * st.stmax))
        stp = st.stx;
    return LS_FG;
}

// ---------------------------------------------------------------- per-seed results
struct SeedResults {
    double *ame;          // [nbeta*3]
    double *pest;         // [nbeta*NPest] or NULL
    int *status, *nit;    // [nbeta]
    long long *nfev;      // [nbeta]
};

// close the current beta step (va_ode.py:773-782): record, then move to the next RF
// or finish.  `accepted`: the trial point becomes the stored minimiser.
VA_HD_FLAT void finish_step(SeedScal &s, int status, bool accepted, const double *rf_ladder, int nbeta,
                       const SeedResults &r, int *n_active_dec)
{
    const int k = s.beta_idx;
    r.ame[3 * k] = s.f; r.ame[3 * k + 1] = s.me; r.ame[3 * k + 2] = s.fe;
    r.status[k] = status; r.nit[k] = s.iter; r.nfev[k] = s.nfev;
    s.upd = (accepted ? UPD_X : 0) | UPD_STORE;
    s.store_idx = k; s.dir = 0; s.nold = 0;
    if (k + 1 < nbeta) {
        s.beta_idx = k + 1; s.rf_scale = rf_ladder[k + 1]; s.phase = PH_START;
    } else {
        s.phase = PH_FINISHED; *n_active_dec = 1;
    }
}

VA_HD_FLAT void begin_linesearch(SeedScal &s)
{
    const double big = 1e10;
    s.ifun = 0; s.iback = 0; s.ls_task = LS_START;
    s.stp = (s.iter == 0) ? fmin(1.0 / sqrt(s.gn2), big) : 1.0;   // lnsrlb; d = -g at iter 0
    s.dir = 1;
}

VA_HD_FLAT void reset_memory(SeedScal &s) { s.col = 0; s.head = 0; s.theta = 1.0; s.nold = 0; s.pad0 = 0; }

// K2: consume one evaluation.  ev[] = eval partial sums INCLUDING the parameter tail
// contributions; dirp[] = (g.d, d.d) of the direction in use.
// (`ls`: the More'-Thuente state of the step -- s.ls itself, or a register copy of it that the caller writes back: the
// persistent ladder kernel keeps the seed's state in LDS, where every field access of dcsrch / dcstep would be a dependent
// round trip)
VA_HD_FLAT void ls_step(SeedScal &s, LsState &ls, int *order, const double *ev, const double *dirp, const Opts &o,
                   const double *rf_ladder, int nbeta, const SeedResults &r, int *n_active_dec,
                   double cme, double cfe, bool bounded = false)
{
    const double epsmch = 2.220446049250313e-16, big = 1e10;
    const double stpmax = bounded ? s.stpmx : big;       // bounded: the line search stops at the box
    s.upd = 0; s.dir = 0; s.nold = 0; s.store_idx = -1;
    const double me = ev[EP_ME] * cme, fe = ev[EP_FE] * cfe * s.rf_scale;
    const double ft = me + fe;
    if (s.phase == PH_START) {
        // first evaluation of a minimisation (setulb FG_START)
        s.f = ft; s.me = me; s.fe = fe; s.gn2 = ev[EP_GN2];
        s.iter = 0; s.nfev = 1; reset_memory(s);
        if (ev[EP_GMAX] <= o.gtol) { finish_step(s, 0, false, rf_ladder, nbeta, r, n_active_dec); return; }
        s.upd = UPD_G; s.phase = PH_LS;
        begin_linesearch(s);
        return;
    }
    if (s.phase != PH_LS) return;
    bool fail = false;
    if (s.ifun == 0) {
        // lnsrlb entry: directional derivative at the start point
        s.gdold = dirp[DP_GD]; s.fold = s.f;
        if (s.gdold >= 0.0) fail = true;
        else {
            s.ls_task = dcsrch(s.f, s.gdold, s.stp, 1e-3, 0.9, 0.1, 0.0, stpmax, LS_START, ls);
            if (s.ls_task == LS_ERROR) fail = true;
            else { s.ifun = 1; s.iback = 0; s.nfev += 1; }    // the evaluation we are consuming
        }
    }
    double stp_eval = s.stp;
    if (!fail) {
        s.gd = ev[EP_GTD];
        s.ls_task = dcsrch(ft, s.gd, s.stp, 1e-3, 0.9, 0.1, 0.0, stpmax, LS_FG, ls);
        if (s.ls_task == LS_FG) {
            s.ifun += 1; s.iback = s.ifun - 1;
            if (s.iback >= o.maxls) fail = true;
            else { s.nfev += 1; return; }                     // next trial at the new s.stp
        }
    }
    if (fail) {
        // restore the previous iterate (x, g, f untouched during the search)
        if (s.col == 0) { finish_step(s, 2, false, rf_ladder, nbeta, r, n_active_dec); return; }
        reset_memory(s);                                      // RESTART_FROM_LNSRCH
        s.ifun = 0; s.iback = 0; s.ls_task = LS_START; s.stp = 1.0; s.dir = 1;
        return;
    }
    // NEW_X: the point evaluated at stp_eval is accepted (dcsrch leaves stp unchanged on CONV/WARN)
    s.stp = stp_eval; s.stp_upd = stp_eval;
    const double fold = s.f;
    s.f = ft; s.me = me; s.fe = fe; s.gn2 = ev[EP_GN2];
    s.iter += 1;
    if (s.iter >= o.maxiter) { finish_step(s, 1, true, rf_ladder, nbeta, r, n_active_dec); return; }
    if (s.nfev > o.maxfun) { finish_step(s, 1, true, rf_ladder, nbeta, r, n_active_dec); return; }
    if (ev[EP_GMAX] <= o.gtol) { finish_step(s, 0, true, rf_ladder, nbeta, r, n_active_dec); return; }
    {
        const double dd = fmax(fmax(fabs(fold), fabs(ft)), 1.0);
        if (fold - ft <= o.ftol * dd) { finish_step(s, 0, true, rf_ladder, nbeta, r, n_active_dec); return; }
    }
    // BFGS pair (matupd) unless L-BFGS-B's curvature rule skips it
    const double dr = (s.gd - s.gdold) * s.stp, ddum = -s.gdold * s.stp;
    s.upd = UPD_X | UPD_G;
    s.nold = s.col;
    if (dr > epsmch * ddum) {
        int slot;
        bool evicted = false;
        if (s.col < o.m) { slot = (s.head + s.col) % o.m; s.nold = s.col; s.col += 1; }
        else {
            // the oldest pair is evicted: order[] shifts left
            evicted = true;
            slot = s.head; s.head = (s.head + 1) % o.m;
            for (int j = 0; j + 1 < s.col; ++j) order[j] = order[j + 1];
            s.nold = s.col - 1;
        }
        order[s.col - 1] = slot;
        s.slot = slot; s.dr = dr; s.upd |= UPD_HIST;
        // (bounded problems, k_lbfgsb_dir: a pair to enter S'Y / S'S; bit 1: the oldest pair was evicted)
        s.pad0 = 1 | (evicted ? 2 : 0);
    }
    begin_linesearch(s);
}

VA_HD_FLAT void ls_step(SeedHot &s, const double *ev, const double *dirp, const Opts &o,
                   const double *rf_ladder, int nbeta, const SeedResults &r, int *n_active_dec,
                   double cme, double cfe, bool bounded = false)
{
    ls_step(static_cast<SeedScal &>(s), s.ls, s.order, ev, dirp, o, rf_ladder, nbeta, r, n_active_dec, cme, cfe, bounded);
}

// K4: Gram update + L-BFGS direction in coefficient space (compact two-loop).
// up[] = update-kernel dot products (UP_* layout, old slots in order[0..nold)).
// Produces d = cg*g + sum_j cY[slot]*Y[slot] + cS[slot]*S[slot].
// Works on a view so that the device can run it on LDS copies (k_coeffs) and the
// host/emulator on the SeedState itself.
struct CoefView {
    int upd, slot, nold, col;
    const int *order;
    double dr;
    double *theta, *cg;
    double *SY, *YY;            // [MAX_M*MAX_M], physical-slot indexed
    double *a, *b, *cY, *cS;    // [MAX_M]
    double *c, *e, *al;         // [MAX_M] work arrays
};

VA_HD void direction_coeffs_view(const CoefView &s, const double *up)
{
    const int M = MAX_M;
    const bool hist = (s.upd & UPD_HIST) != 0;
    const int nold = s.nold, col = s.col;
    if (hist) {
        const int sn = s.slot;
        for (int j = 0; j < nold; ++j) {
            const int sj = s.order[j];
            s.SY[sj * M + sn] = up[UP_OLD + 4 * j + 2];          // S_j . y
            s.YY[sj * M + sn] = s.YY[sn * M + sj] = up[UP_OLD + 4 * j + 3];
        }
        s.SY[sn * M + sn] = s.dr;                                // s.y as the line search saw it
        s.YY[sn * M + sn] = up[UP_YY];
        *s.theta = up[UP_YY] / s.dr;
    }
    for (int j = 0; j < nold; ++j) {
        s.a[j] = up[UP_OLD + 4 * j + 0];                         // S_j . g
        s.b[j] = up[UP_OLD + 4 * j + 1];                         // Y_j . g
    }
    if (hist) { s.a[col - 1] = up[UP_SGT]; s.b[col - 1] = up[UP_YGT]; }
    // The two-loop recursion in its compact form (Byrd, Nocedal & Schnabel 1994): with
    // R_ij = S_i.Y_j (i <= j, oldest first), D = diag(R), a = S^T g, b = Y^T g,
    //     p = R^{-1} a,   q = (D + gamma Y^T Y) p - gamma b,   u = R^{-T} q,
    //     d = -H g = -gamma g - S u + gamma Y p.
    // Both triangular solves are written column-oriented (solve one unknown, then update the
    // remaining right-hand sides): that is the form k_coeffs runs with one lane per pair.
    const double gamma = 1.0 / *s.theta;
    double *p = s.c, *q = s.e, *rhs = s.al;
    for (int j = 0; j < col; ++j) rhs[j] = s.a[j];
    for (int i = col - 1; i >= 0; --i) {                         // back substitution, R upper
        const int si = s.order[i];
        p[i] = rhs[i] * (1.0 / s.SY[si * M + si]);
        for (int j = 0; j < i; ++j) rhs[j] -= s.SY[s.order[j] * M + si] * p[i];
    }
    for (int i = 0; i < col; ++i) {
        const int si = s.order[i];
        double acc = 0.0;
        for (int k = 0; k < col; ++k) acc += s.YY[si * M + s.order[k]] * p[k];
        q[i] = s.SY[si * M + si] * p[i] + gamma * acc - gamma * s.b[i];
    }
    for (int i = 0; i < col; ++i) {                              // forward substitution, R^T lower
        const int si = s.order[i];
        q[i] = q[i] * (1.0 / s.SY[si * M + si]);                 // u_i
        for (int j = i + 1; j < col; ++j) q[j] -= s.SY[si * M + s.order[j]] * q[i];
    }
    *s.cg = -gamma;
    for (int j = 0; j < MAX_M; ++j) { s.cY[j] = 0.0; s.cS[j] = 0.0; }
    for (int j = 0; j < col; ++j) { s.cY[s.order[j]] = gamma * p[j]; s.cS[s.order[j]] = -q[j]; }
}

VA_HD void direction_coeffs(SeedState &s, const double *up, const Opts &o)
{
    (void)o;
    double c[MAX_M], e[MAX_M], al[MAX_M];
    CoefView v;
    v.upd = s.upd; v.slot = s.slot; v.nold = s.nold; v.col = s.col; v.order = s.order; v.dr = s.dr;
    v.theta = &s.theta; v.cg = &s.cg; v.SY = s.SY; v.YY = s.YY; v.a = s.a; v.b = s.b;
    v.cY = s.cY; v.cS = s.cS; v.c = c; v.e = e; v.al = al;
    direction_coeffs_view(v, up);
}

}  // namespace va
