// va_lbfgsb.hip -- the direction step of L-BFGS-B for problems WITH box bounds, on the device: one workgroup per seed.
//
// Reference: va_ode.py:582-605 builds `bounds`, _autodiffmin.py:85-86 hands them to SciPy's L-BFGS-B.  The batched
// state machine (va_core.h: ls_step in the tail of the evaluation kernel; k_update) keeps its line search, stopping
// rules and history ring; what changes with bounds is how the next trial direction is found -- and that is this
// kernel, launched in k_direction's place: the published algorithm's
//     matupd / formt   the new pair enters S'Y (lower triangle), S'S (upper), theta = y'y / s'y, T = theta S'S + L D^-1 L'
//     cauchy           generalised Cauchy point along the projected steepest-descent path (breakpoints taken in
//                      increasing order; each crossing a rank-one update of the 2m-vectors p, c through the middle matrix)
//     formk / cmprlb   LEL' factorisation of the reduced middle matrix over the free variables at the Cauchy point; reduced gradient
//     subsm            subspace minimisation + the projection step of L-BFGS-B 3.0 (Morales & Nocedal 2011)
//     lnsrlb (head)    d = z - x, the largest feasible step, the first trial step
// (Byrd, Lu, Nocedal & Zhu 1995; Zhu et al., Algorithm 778).  Loops over the n variables are strided over the
// workgroup with fixed-order reductions (bitwise reproducible); the 2m x 2m dense algebra is done by one lane in LDS.
// The breakpoint heap of the serial code becomes a workgroup-wide arg-min per crossing: same order but for exact ties.
// The CPU oracle (oracle/va_lbfgsb.inc.c, pinned to SciPy step for step) is the checker of this port.
#include "va_device.h"
#include "va_eval_flat.h"

namespace va {

constexpr int LB_THREADS = 256;

struct LbCtx {
    int n, m, col, tid, lane, wave;
    const int *order;              // history slots, oldest -> newest
    const double *S, *Y;           // the seed's rings [m][ld]
    size_t ld;
    const double *lo, *hi;         // [n] (+-HUGE_VAL = none)
    double *sy, *ss, *wt, *wn, *wn1, *wa, *red;      // LDS
    int *ired;
};
#define LWS(i, j) c.S[(size_t)c.order[(j) - 1] * c.ld + ((i) - 1)]
#define LWY(i, j) c.Y[(size_t)c.order[(j) - 1] * c.ld + ((i) - 1)]
#define LSY(i, j) c.sy[((j) - 1) * c.m + ((i) - 1)]
#define LSS(i, j) c.ss[((j) - 1) * c.m + ((i) - 1)]
#define LWT(i, j) c.wt[((j) - 1) * c.m + ((i) - 1)]
#define LWN(i, j) c.wn[((j) - 1) * (2 * c.m) + ((i) - 1)]
#define LWN1(i, j) c.wn1[((j) - 1) * (2 * c.m) + ((i) - 1)]

__device__ __forceinline__ int lb_nbd(double l, double u) { return l > -HUGE_VAL ? (u < HUGE_VAL ? 2 : 1) : (u < HUGE_VAL ? 3 : 0); }

// workgroup sums / min / max with a fixed order; every thread gets the result
__device__ __forceinline__ double lb_bsum(const LbCtx &c, double v)
{
    v = wave_sum(v);
    __syncthreads();
    if (c.lane == 0) c.red[c.wave] = v;
    __syncthreads();
    return ((c.red[0] + c.red[1]) + c.red[2]) + c.red[3];
}
__device__ __forceinline__ double lb_bmax(const LbCtx &c, double v)
{
    v = wave_max(v);
    __syncthreads();
    if (c.lane == 0) c.red[c.wave] = v;
    __syncthreads();
    return fmax(fmax(c.red[0], c.red[1]), fmax(c.red[2], c.red[3]));
}
__device__ __forceinline__ double lb_bmin(const LbCtx &c, double v) { return -lb_bmax(c, -v); }
// smallest value and, among equals, the smallest index
__device__ __forceinline__ void lb_bargmin(const LbCtx &c, double &v, int &idx)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    __syncthreads();
    if (c.lane == 0) { c.red[c.wave] = v; c.ired[c.wave] = idx; }
    __syncthreads();
    v = c.red[0]; idx = c.ired[0];
    for (int w = 1; w < 4; ++w)
        if (c.red[w] < v || (c.red[w] == v && c.ired[w] < idx)) { v = c.red[w]; idx = c.ired[w]; }
}

// LINPACK dpofa / dtrsl on small matrices in LDS (one lane)
__device__ int lb_dpofa(double *a, int lda, int n)
{
#define A_(i, j) a[((j) - 1) * lda + ((i) - 1)]
    for (int j = 1; j <= n; ++j) {
        double s = 0.0;
        for (int k = 1; k <= j - 1; ++k) {
            double t = A_(k, j);
            for (int q = 1; q <= k - 1; ++q) t -= A_(q, k) * A_(q, j);
            t = t / A_(k, k);
            A_(k, j) = t;
            s += t * t;
        }
        s = A_(j, j) - s;
        if (s <= 0.0) return j;
        A_(j, j) = sqrt(s);
    }
    return 0;
#undef A_
}
__device__ int lb_dtrsl(const double *t, int ldt, int n, double *b, int job)
{
#define T_(i, j) t[((j) - 1) * ldt + ((i) - 1)]
    for (int j = 1; j <= n; ++j)
        if (T_(j, j) == 0.0) return j;
    if (job == 1) {
        b[n - 1] = b[n - 1] / T_(n, n);
        for (int jj = 2; jj <= n; ++jj) {
            const int j = n - jj + 1;
            const double temp = -b[j];
            for (int q = 1; q <= j; ++q) b[q - 1] += temp * T_(q, j + 1);
            b[j - 1] = b[j - 1] / T_(j, j);
        }
    } else {
        b[0] = b[0] / T_(1, 1);
        for (int j = 2; j <= n; ++j) {
            double s = 0.0;
            for (int q = 1; q <= j - 1; ++q) s += T_(q, j) * b[q - 1];
            b[j - 1] = (b[j - 1] - s) / T_(j, j);
        }
    }
    return 0;
#undef T_
}
// middle matrix times v (one lane)
__device__ int lb_bmv(const LbCtx &c, const double *v, double *p)
{
    const int col = c.col;
    if (col == 0) return 0;
    p[col] = v[col];
    for (int i = 2; i <= col; ++i) {
        double sum = 0.0;
        for (int k = 1; k <= i - 1; ++k) sum += LSY(i, k) * v[k - 1] / LSY(k, k);
        p[col + i - 1] = v[col + i - 1] + sum;
    }
    if (lb_dtrsl(c.wt, c.m, col, p + col, 11)) return 1;
    for (int i = 1; i <= col; ++i) p[i - 1] = v[i - 1] / sqrt(LSY(i, i));
    if (lb_dtrsl(c.wt, c.m, col, p + col, 1)) return 1;
    for (int i = 1; i <= col; ++i) p[i - 1] = -p[i - 1] / sqrt(LSY(i, i));
    for (int i = 1; i <= col; ++i) {
        double sum = 0.0;
        for (int k = i + 1; k <= col; ++k) sum += LSY(k, i) * p[col + k - 1] / LSY(i, i);
        p[i - 1] += sum;
    }
    return 0;
}
__device__ int lb_formt(const LbCtx &c, double theta)
{
    const int col = c.col;
    for (int j = 1; j <= col; ++j) LWT(1, j) = theta * LSS(1, j);
    for (int i = 2; i <= col; ++i)
        for (int j = i; j <= col; ++j) {
            const int k1 = (i < j ? i : j) - 1;
            double ddum = 0.0;
            for (int k = 1; k <= k1; ++k) ddum += LSY(i, k) * LSY(j, k) / LSY(k, k);
            LWT(i, j) = ddum + theta * LSS(i, j);
        }
    return lb_dpofa(c.wt, c.m, col) ? -3 : 0;
}

// broadcast one lane-0 int through LDS
__device__ __forceinline__ int lb_bcast(const LbCtx &c, int v)
{
    __syncthreads();
    if (c.tid == 0) c.ired[4] = v;
    __syncthreads();
    return c.ired[4];
}
__device__ __forceinline__ double lb_bcastd(const LbCtx &c, double v)
{
    __syncthreads();
    if (c.tid == 0) c.red[4] = v;
    __syncthreads();
    return c.red[4];
}

__global__ __launch_bounds__(LB_THREADS) void k_lbfgsb_dir(const Dev dv)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const Dims &dm = dv.dm;
    const int b = blockIdx.x;
    SeedState &s = dv.st[b];
    if (!s.dir) return;
    const int n = dm.ND + dm.NPest, m = dm.m;
    const double epsmch = 2.220446049250313e-16, big = 1e10;
    LbCtx c;
    c.n = n; c.m = m; c.tid = threadIdx.x; c.lane = threadIdx.x & 63; c.wave = threadIdx.x >> 6;
    c.order = s.order; c.ld = dm.ld;
    c.S = dv.S + (size_t)b * m * dm.ld; c.Y = dv.Y + (size_t)b * m * dm.ld;
    c.lo = dv.pp.lo; c.hi = dv.pp.hi;
    c.sy = lds; c.ss = c.sy + m * m; c.wt = c.ss + m * m; c.wn = c.wt + m * m; c.wn1 = c.wn + 4 * m * m;
    c.wa = c.wn1 + 4 * m * m; c.red = c.wa + 8 * m; c.ired = reinterpret_cast<int *>(c.red + 8);
    double *p = c.wa, *cc = c.wa + 2 * m, *wbp = c.wa + 4 * m, *v = c.wa + 6 * m;
    const int tid = c.tid;
    const size_t vo = (size_t)b * dm.ld;
    const double *x = dv.x + vo, *g = dv.g + vo;
    double *d = dv.d + vo, *z = dv.lb_z + vo, *r = dv.lb_r + vo, *xp = dv.lb_xp + vo, *t = dv.lb_t + vo;
    int *iwhere = dv.lb_iwhere + vo;
    double *mat = dv.lb_mat + (size_t)b * 3 * m * m;
    for (int e = tid; e < 3 * m * m; e += LB_THREADS) c.sy[e] = mat[e];        // (sy, ss, wt are contiguous)
    __syncthreads();
    int col = s.col;
    double theta = s.theta;
    const int cnstnd = dm.bounded & 1, boxed = (dm.bounded >> 1) & 1;
    const int flags = s.pad0;                                                  // bit 0: a pair was stored, bit 1: the oldest was evicted

    // ---- matupd / formt: the pair k_update stored (S_new = stp d, Y_new = g_new - g_old) enters the small matrices
    if (flags & 1) {
        c.col = col;
        if (flags & 2) {                                                       // move old information
            if (tid == 0)
                for (int j = 1; j <= col - 1; ++j) {
                    for (int q = 1; q <= j; ++q) LSS(q, j) = LSS(q + 1, j + 1);
                    for (int q = 0; q < col - j; ++q) LSY(j + q, j) = LSY(j + 1 + q, j + 1);
                }
            __syncthreads();
        }
        double rr = 0.0;
        for (int i = 1 + tid; i <= n; i += LB_THREADS) { const double yv = LWY(i, col); rr += yv * yv; }
        rr = lb_bsum(c, rr);
        for (int j = 1; j <= col - 1; ++j) {
            double a = 0.0, bq = 0.0;
            for (int i = 1 + tid; i <= n; i += LB_THREADS) { const double sn = LWS(i, col); a += sn * LWY(i, j); bq += LWS(i, j) * sn; }
            a = lb_bsum(c, a); bq = lb_bsum(c, bq);
            if (tid == 0) { LSY(col, j) = a; LSS(j, col) = bq; }
        }
        theta = rr / s.dr;
        int info = 0;
        if (tid == 0) {
            const double stp = s.stp_upd, dtd = dv.lb_dtd[b];
            LSS(col, col) = stp == 1.0 ? dtd : stp * stp * dtd;
            LSY(col, col) = s.dr;
            info = lb_formt(c, theta);
        }
        info = lb_bcast(c, info);
        if (info != 0) { col = 0; theta = 1.0; }                               // refresh the memory
    }
    // projected gradient norm (cauchy returns x itself when it is zero)
    double sbg = 0.0;
    for (int i = tid; i < n; i += LB_THREADS) {
        double gi = g[i];
        const int nb = lb_nbd(c.lo[i], c.hi[i]);
        if (nb != 0) { if (gi < 0.0) { if (nb >= 2) gi = fmax(x[i] - c.hi[i], gi); } else { if (nb <= 2) gi = fmin(x[i] - c.lo[i], gi); } }
        sbg = fmax(sbg, fabs(gi));
    }
    const double sbgnrm = lb_bmax(c, sbg);

    for (int attempt = 0; attempt < 2; ++attempt) {
        c.col = col;
        const int col2 = 2 * col;
        int info = 0;
        // ================================================================ cauchy
        if (!cnstnd && col > 0) {
            for (int i = tid; i < n; i += LB_THREADS) { z[i] = x[i]; iwhere[i] = -1; }
        } else if (sbgnrm <= 0.0) {
            for (int i = tid; i < n; i += LB_THREADS) z[i] = x[i];
        } else {
            double f1 = 0.0;
            int nbreak = 0, nmove = 0, unb = 0;
            for (int i = tid; i < n; i += LB_THREADS) {
                const double neggi = -g[i], l = c.lo[i], u = c.hi[i];
                const int nb = lb_nbd(l, u);
                int iw = nb == 0 ? -1 : ((nb == 2 && u - l <= 0.0) ? 3 : 0);
                double tl = 0.0, tu = 0.0, tb = HUGE_VAL, di = 0.0;
                if (iw != 3 && iw != -1) {
                    if (nb <= 2) tl = x[i] - l;
                    if (nb >= 2) tu = u - x[i];
                    const bool xlower = nb <= 2 && tl <= 0.0, xupper = nb >= 2 && tu <= 0.0;
                    iw = 0;
                    if (xlower) { if (neggi <= 0.0) iw = 1; }
                    else if (xupper) { if (neggi >= 0.0) iw = 2; }
                    else { if (fabs(neggi) <= 0.0) iw = -3; }
                }
                if (iw == 0 || iw == -1) {
                    di = neggi;
                    f1 -= neggi * neggi;
                    if (nb <= 2 && nb != 0 && neggi < 0.0) { tb = tl / (-neggi); nbreak++; }
                    else if (nb >= 2 && neggi > 0.0) { tb = tu / neggi; nbreak++; }
                    else { nmove++; if (fabs(neggi) > 0.0) unb = 1; }
                }
                iwhere[i] = iw; d[i] = di; t[i] = tb; z[i] = x[i];
            }
            f1 = lb_bsum(c, f1);
            nbreak = (int)lb_bsum(c, (double)nbreak);
            nmove = (int)lb_bsum(c, (double)nmove);
            const int bnded = lb_bsum(c, (double)unb) == 0.0;
            for (int j = 1; j <= col; ++j) {
                double a = 0.0, bq = 0.0;
                for (int i = 1 + tid; i <= n; i += LB_THREADS) { const double di = d[i - 1]; a += LWY(i, j) * di; bq += LWS(i, j) * di; }
                a = lb_bsum(c, a); bq = lb_bsum(c, bq);
                if (tid == 0) { p[j - 1] = a; p[col + j - 1] = theta * bq; }
            }
            if (!(nbreak == 0 && nmove == 0)) {
                if (tid == 0) for (int j = 0; j < col2; ++j) cc[j] = 0.0;
                double f2 = -theta * f1;
                const double f2_org = f2;
                if (col > 0) {
                    if (tid == 0) { info = lb_bmv(c, p, v); double dd = 0.0; for (int j = 0; j < col2; ++j) dd += v[j] * p[j]; c.red[5] = dd; }
                    info = lb_bcast(c, info);
                    f2 -= c.red[5];
                }
                double dtm = -f1 / f2, tsum = 0.0, tj = 0.0;
                bool done_all = false;
                if (info == 0 && nbreak > 0) {
                    int nleft = nbreak;
                    for (;;) {
                        // the next breakpoint
                        double tv = HUGE_VAL; int ti = 0x7fffffff;
                        for (int i = tid; i < n; i += LB_THREADS) { const double q = t[i]; if (q < tv) { tv = q; ti = i; } }
                        lb_bargmin(c, tv, ti);
                        const double tj0 = tj;
                        tj = tv;
                        const int ibp = ti;
                        const double dt = tj - tj0;
                        if (dtm < dt) break;
                        tsum += dt; nleft--;
                        const double dibp = d[ibp];
                        double zibp;
                        __syncthreads();
                        if (dibp > 0.0) zibp = c.hi[ibp] - x[ibp]; else zibp = c.lo[ibp] - x[ibp];
                        if (tid == 0) {
                            d[ibp] = 0.0; t[ibp] = HUGE_VAL;
                            if (dibp > 0.0) { z[ibp] = c.hi[ibp]; iwhere[ibp] = 2; } else { z[ibp] = c.lo[ibp]; iwhere[ibp] = 1; }
                        }
                        __syncthreads();
                        if (nleft == 0 && nbreak == n) { dtm = dt; done_all = true; break; }
                        const double dibp2 = dibp * dibp;
                        f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;
                        f2 = f2 - theta * dibp2;
                        if (col > 0) {
                            if (tid == 0) {
                                for (int j = 0; j < col2; ++j) cc[j] += dt * p[j];
                                for (int j = 1; j <= col; ++j) { wbp[j - 1] = LWY(ibp + 1, j); wbp[col + j - 1] = theta * LWS(ibp + 1, j); }
                                info = lb_bmv(c, wbp, v);
                                double wmc = 0.0, wmp = 0.0, wmw = 0.0;
                                for (int j = 0; j < col2; ++j) { wmc += cc[j] * v[j]; wmp += p[j] * v[j]; wmw += wbp[j] * v[j]; }
                                for (int j = 0; j < col2; ++j) p[j] -= dibp * wbp[j];
                                c.red[5] = dibp * wmc; c.red[6] = 2.0 * dibp * wmp - dibp2 * wmw;
                            }
                            info = lb_bcast(c, info);
                            if (info != 0) break;
                            f1 += c.red[5]; f2 += c.red[6];
                        }
                        f2 = fmax(epsmch * f2_org, f2);
                        if (nleft > 0) { dtm = -f1 / f2; continue; }
                        else if (bnded) { f1 = 0.0; f2 = 0.0; dtm = 0.0; }
                        else dtm = -f1 / f2;
                        break;
                    }
                }
                if (info == 0) {
                    if (!done_all) {
                        if (dtm <= 0.0) dtm = 0.0;
                        tsum += dtm;
                        __syncthreads();
                        for (int i = tid; i < n; i += LB_THREADS) z[i] += tsum * d[i];
                    }
                    if (col > 0 && tid == 0) for (int j = 0; j < col2; ++j) cc[j] += dtm * p[j];
                }
            }
        }
        __syncthreads();
        if (info != 0) { col = 0; theta = 1.0; continue; }                     // singular triangular system: refresh and redo
        // ================================================================ free variables at the Cauchy point
        int nfree = 0;
        for (int i = tid; i < n; i += LB_THREADS) nfree += iwhere[i] <= 0 ? 1 : 0;
        nfree = (int)lb_bsum(c, (double)nfree);
        if (nfree != 0 && col != 0) {
            // ============================================================ formk
            for (int iy = 1; iy <= col; ++iy)
                for (int jy = 1; jy <= iy; ++jy) {
                    double t1 = 0.0, t2 = 0.0;
                    for (int i = 1 + tid; i <= n; i += LB_THREADS) {
                        if (iwhere[i - 1] <= 0) t1 += LWY(i, iy) * LWY(i, jy); else t2 += LWS(i, iy) * LWS(i, jy);
                    }
                    t1 = lb_bsum(c, t1); t2 = lb_bsum(c, t2);
                    if (tid == 0) { LWN1(iy, jy) = t1; LWN1(m + iy, m + jy) = t2; }
                }
            for (int is0 = 1; is0 <= col; ++is0)
                for (int jy = 1; jy <= col; ++jy) {
                    double tt = 0.0;
                    const bool fr = is0 <= jy;
                    for (int i = 1 + tid; i <= n; i += LB_THREADS)
                        if ((iwhere[i - 1] <= 0) == fr) tt += LWS(i, is0) * LWY(i, jy);
                    tt = lb_bsum(c, tt);
                    if (tid == 0) LWN1(m + is0, jy) = tt;
                }
            __syncthreads();
            if (tid == 0) {
                const int m2 = 2 * m;
                for (int iy = 1; iy <= col; ++iy) {
                    const int is = col + iy, is1 = m + iy;
                    for (int jy = 1; jy <= iy; ++jy) { const int js = col + jy, js1 = m + jy; LWN(jy, iy) = LWN1(iy, jy) / theta; LWN(js, is) = LWN1(is1, js1) * theta; }
                    for (int jy = 1; jy <= iy - 1; ++jy) LWN(jy, is) = -LWN1(is1, jy);
                    for (int jy = iy; jy <= col; ++jy) LWN(jy, is) = LWN1(is1, jy);
                    LWN(iy, iy) += LSY(iy, iy);
                }
                info = lb_dpofa(c.wn, m2, col) ? -1 : 0;
                for (int js = col + 1; js <= col2 && info == 0; ++js)
                    if (lb_dtrsl(c.wn, m2, col, &LWN(1, js), 11)) info = -1;
                if (info == 0) {
                    for (int is = col + 1; is <= col2; ++is)
                        for (int js = is; js <= col2; ++js) {
                            double sm = 0.0;
                            for (int q = 1; q <= col; ++q) sm += LWN(q, is) * LWN(q, js);
                            LWN(is, js) += sm;
                        }
                    if (lb_dpofa(&LWN(col + 1, col + 1), m2, col)) info = -2;
                }
                // cmprlb: p = M c
                if (info == 0 && cnstnd) info = lb_bmv(c, cc, p) ? -8 : 0;
            }
            info = lb_bcast(c, info);
            if (info != 0) { col = 0; theta = 1.0; continue; }
            // ============================================================ cmprlb: r = -Z'B(xcp - x) - Z'g  (full-length, free entries used)
            for (int i = tid; i < n; i += LB_THREADS) {
                double ri = 0.0;
                if (!cnstnd) ri = -g[i];
                else if (iwhere[i] <= 0) {
                    ri = -theta * (z[i] - x[i]) - g[i];
                    for (int j = 1; j <= col; ++j) ri += LWY(i + 1, j) * p[j - 1] + LWS(i + 1, j) * (theta * p[col + j - 1]);
                }
                r[i] = ri;
            }
            __syncthreads();
            // ============================================================ subsm
            for (int j = 1; j <= col; ++j) {
                double a = 0.0, bq = 0.0;
                for (int i = 1 + tid; i <= n; i += LB_THREADS)
                    if (iwhere[i - 1] <= 0) { const double ri = r[i - 1]; a += LWY(i, j) * ri; bq += LWS(i, j) * ri; }
                a = lb_bsum(c, a); bq = lb_bsum(c, bq);
                if (tid == 0) { v[j - 1] = a; v[col + j - 1] = theta * bq; }
            }
            if (tid == 0) {
                const int m2 = 2 * m;
                info = lb_dtrsl(c.wn, m2, col2, v, 11) ? 1 : 0;
                if (info == 0) { for (int j = 0; j < col; ++j) v[j] = -v[j]; info = lb_dtrsl(c.wn, m2, col2, v, 1) ? 1 : 0; }
            }
            info = lb_bcast(c, info);
            if (info != 0) { col = 0; theta = 1.0; continue; }
            int iword = 0;
            for (int i = tid; i < n; i += LB_THREADS) {
                xp[i] = z[i];
                if (iwhere[i] <= 0) {
                    double dk = r[i];
                    for (int j = 1; j <= col; ++j) dk += LWY(i + 1, j) * v[j - 1] / theta + LWS(i + 1, j) * v[col + j - 1];
                    dk *= 1.0 / theta;
                    r[i] = dk;
                    const double l = c.lo[i], u = c.hi[i], xk = z[i];
                    const int nb = lb_nbd(l, u);
                    double xn;
                    if (nb == 0) xn = xk + dk;
                    else if (nb == 1) { xn = fmax(l, xk + dk); if (xn == l) iword = 1; }
                    else if (nb == 2) { xn = fmin(u, fmax(l, xk + dk)); if (xn == l || xn == u) iword = 1; }
                    else { xn = fmin(u, xk + dk); if (xn == u) iword = 1; }
                    z[i] = xn;
                }
            }
            iword = lb_bsum(c, (double)iword) != 0.0;
            if (iword) {
                double ddp = 0.0;
                for (int i = tid; i < n; i += LB_THREADS) ddp += (z[i] - x[i]) * g[i];
                ddp = lb_bsum(c, ddp);
                if (ddp > 0.0) {
                    // the projected point is not a descent step: back to the Cauchy point and a truncated step along d
                    double al = 1.0; int ibd = 0x7fffffff;
                    for (int i = tid; i < n; i += LB_THREADS) {
                        z[i] = xp[i];
                        if (iwhere[i] <= 0) {
                            const double dk = r[i], l = c.lo[i], u = c.hi[i];
                            const int nb = lb_nbd(l, u);
                            double t1 = 1.0;
                            if (nb != 0) {
                                if (dk < 0.0 && nb <= 2) { const double t2 = l - xp[i]; if (t2 >= 0.0) t1 = 0.0; else if (dk < t2) t1 = t2 / dk; }
                                else if (dk > 0.0 && nb >= 2) { const double t2 = u - xp[i]; if (t2 <= 0.0) t1 = 0.0; else if (dk > t2) t1 = t2 / dk; }
                                if (t1 < al) { al = t1; ibd = i; }
                            }
                        }
                    }
                    lb_bargmin(c, al, ibd);
                    __syncthreads();
                    if (al < 1.0 && tid == 0) {
                        const double dk = r[ibd];
                        if (dk > 0.0) { z[ibd] = c.hi[ibd]; r[ibd] = 0.0; } else if (dk < 0.0) { z[ibd] = c.lo[ibd]; r[ibd] = 0.0; }
                    }
                    __syncthreads();
                    for (int i = tid; i < n; i += LB_THREADS) if (iwhere[i] <= 0) z[i] += al * r[i];
                }
            }
        }
        break;
    }
    __syncthreads();
    // ==================================================================== lnsrlb, first part: d = z - x, feasible step, first trial
    double dtd = 0.0, gd = 0.0, smx = big;
    const int iter = s.iter;
    for (int i = tid; i < n; i += LB_THREADS) {
        const double di = z[i] - x[i];
        d[i] = di;
        dtd += di * di; gd += g[i] * di;
        if (cnstnd && iter != 0) {
            const double l = c.lo[i], u = c.hi[i];
            const int nb = lb_nbd(l, u);
            if (nb != 0) {
                if (di < 0.0 && nb <= 2) { const double a2 = l - x[i]; if (a2 >= 0.0) smx = 0.0; else if (di * smx < a2) smx = a2 / di; }
                else if (di > 0.0 && nb >= 2) { const double a2 = u - x[i]; if (a2 <= 0.0) smx = 0.0; else if (di * smx > a2) smx = a2 / di; }
            }
        }
    }
    for (int i = n + tid; i < (int)dm.ld; i += LB_THREADS) d[i] = 0.0;
    dtd = lb_bsum(c, dtd); gd = lb_bsum(c, gd);
    smx = lb_bmin(c, smx);
    if (cnstnd && iter == 0) smx = 1.0;
    for (int e = tid; e < 3 * m * m; e += LB_THREADS) mat[e] = c.sy[e];
    if (tid == 0) {
        s.col = col; s.theta = theta;
        if (col == 0) { s.head = 0; s.nold = 0; }
        s.gd_dir = gd; s.stpmx = smx;
        s.stp = (iter == 0 && !boxed) ? fmin(1.0 / sqrt(dtd), smx) : 1.0;
        dv.lb_dtd[b] = dtd;
        s.pad0 = 0;
        if (!dv.sticky) s.dir = 0;
    }
}

size_t lbfgsb_lds_bytes(const Dims &dm) { return sizeof(double) * (size_t)(11 * dm.m * dm.m + 8 * dm.m + 16); }

void launch_lbfgsb_dir(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_lbfgsb_dir, dim3(dv.dm.B), dim3(LB_THREADS), lbfgsb_lds_bytes(dv.dm), s, dv);
}
hipError_t prepare_lbfgsb(const Dev &dv)
{
    if (lbfgsb_lds_bytes(dv.dm) <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute((const void *)k_lbfgsb_dir, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace va
