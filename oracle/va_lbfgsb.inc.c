/* oracle/va_lbfgsb.inc.c -- L-BFGS-B itself, for problems WITH bounds.  TEST INFRASTRUCTURE (see va_oracle.h).
 *
 * The reference hands box bounds to scipy.optimize.minimize(method='L-BFGS-B', bounds=...)
 * (varanneal/_autodiffmin.py:85-86, bounds built at va_ode.py:582-605).  That minimiser is a THIRD-PARTY
 * dependency of the reference (SciPy, un-pinned ">= 0.18.1" in its README; 1.15.3 is what this image holds), so
 * this is a restatement of its PUBLISHED algorithm -- R. H. Byrd, P. Lu, J. Nocedal, C. Zhu, "A limited memory
 * algorithm for bound constrained optimization", SIAM J. Sci. Comput. 16 (1995); C. Zhu et al., Algorithm 778,
 * ACM TOMS 23 (1997); J. L. Morales, J. Nocedal, "Remark on Algorithm 778" (2011: L-BFGS-B 3.0, the projection
 * step after the subspace minimisation) -- routine for routine: active / projgr / cauchy (generalised Cauchy
 * point along the projected gradient path, breakpoints from a heap) / freev / formk (LEL^T factorisation of the
 * reduced middle matrix) / cmprlb / subsm / lnsrlb (More'-Thuente dcsrch) / matupd / formt / bmv.
 * Pinned step for step (nit, nfev, iterates) against scipy.optimize.minimize in tests/test_oracle_lbfgs.py.
 *
 * Differences from the Fortran that cannot change an iterate beyond rounding: formk builds the inner-product
 * blocks from scratch every time instead of adding entering / subtracting leaving variables.
 * Included at the end of va_oracle.c (uses its dcsrch / ddot). */

typedef struct {
    int n, m;
    double *ws, *wy;          /* n x m, column-major: S and Y pairs in a ring */
    double *sy, *ss, *wt;     /* m x m */
    double *wn, *wn1;         /* 2m x 2m */
    double *z, *r, *d, *t, *xp, *wa;   /* n, n, n, n, n, 8m */
    int *index, *iwhere, *indx2, *nbd;
} lbw;

#define WS(i, j) w->ws[((j) - 1) * (size_t)w->n + ((i) - 1)]
#define WY(i, j) w->wy[((j) - 1) * (size_t)w->n + ((i) - 1)]
#define SY(i, j) w->sy[((j) - 1) * w->m + ((i) - 1)]
#define SS(i, j) w->ss[((j) - 1) * w->m + ((i) - 1)]
#define WT(i, j) w->wt[((j) - 1) * w->m + ((i) - 1)]
#define WN(i, j) w->wn[((j) - 1) * (2 * w->m) + ((i) - 1)]
#define WN1(i, j) w->wn1[((j) - 1) * (2 * w->m) + ((i) - 1)]

/* LINPACK dpofa: Cholesky factor R (upper, R'R = A) of the leading n x n block of a (leading dimension lda) */
static int lb_dpofa(double *a, int lda, int n)
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
}
/* LINPACK dtrsl, upper triangular t: job 1 = solve T x = b, job 11 = solve T' x = b */
static int lb_dtrsl(const double *t, int ldt, int n, double *b, int job)
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
#undef A_
}

/* product of the 2m x 2m middle matrix of the compact L-BFGS formula with v (length 2 col) */
static int lb_bmv(const lbw *w, int col, const double *v, double *p)
{
    const int m = w->m;
    if (col == 0) return 0;
    p[col] = v[col];
    for (int i = 2; i <= col; ++i) {
        double sum = 0.0;
        for (int k = 1; k <= i - 1; ++k) sum += SY(i, k) * v[k - 1] / SY(k, k);
        p[col + i - 1] = v[col + i - 1] + sum;
    }
    if (lb_dtrsl(w->wt, m, col, p + col, 11)) return 1;
    for (int i = 1; i <= col; ++i) p[i - 1] = v[i - 1] / sqrt(SY(i, i));
    if (lb_dtrsl(w->wt, m, col, p + col, 1)) return 1;
    for (int i = 1; i <= col; ++i) p[i - 1] = -p[i - 1] / sqrt(SY(i, i));
    for (int i = 1; i <= col; ++i) {
        double sum = 0.0;
        for (int k = i + 1; k <= col; ++k) sum += SY(k, i) * p[col + k - 1] / SY(i, i);
        p[i - 1] += sum;
    }
    return 0;
}

/* T = theta*SS + L*D^(-1)*L' (upper triangle), then its Cholesky factor in wt */
static int lb_formt(lbw *w, int col, double theta)
{
    for (int j = 1; j <= col; ++j) WT(1, j) = theta * SS(1, j);
    for (int i = 2; i <= col; ++i)
        for (int j = i; j <= col; ++j) {
            const int k1 = (i < j ? i : j) - 1;
            double ddum = 0.0;
            for (int k = 1; k <= k1; ++k) ddum += SY(i, k) * SY(j, k) / SY(k, k);
            WT(i, j) = ddum + theta * SS(i, j);
        }
    return lb_dpofa(w->wt, w->m, col) ? -3 : 0;
}

static double lb_projgr(int n, const double *l, const double *u, const int *nbd, const double *x, const double *g)
{
    double sbgnrm = 0.0;
    for (int i = 0; i < n; ++i) {
        double gi = g[i];
        if (nbd[i] != 0) {
            if (gi < 0.0) { if (nbd[i] >= 2) gi = fmax(x[i] - u[i], gi); }
            else { if (nbd[i] <= 2) gi = fmin(x[i] - l[i], gi); }
        }
        sbgnrm = fmax(sbgnrm, fabs(gi));
    }
    return sbgnrm;
}

/* heap of breakpoints: least member to t[n-1], the rest re-heaped in t[0..n-2] */
static void lb_hpsolb(int n, double *t, int *iorder, int iheap)
{
    if (iheap == 0) {
        for (int k = 2; k <= n; ++k) {
            const double ddum = t[k - 1];
            const int indxin = iorder[k - 1];
            int i = k;
            while (i > 1) {
                const int j = i / 2;
                if (ddum < t[j - 1]) { t[i - 1] = t[j - 1]; iorder[i - 1] = iorder[j - 1]; i = j; }
                else break;
            }
            t[i - 1] = ddum; iorder[i - 1] = indxin;
        }
    }
    if (n > 1) {
        int i = 1;
        const double out = t[0], ddum = t[n - 1];
        const int indxou = iorder[0], indxin = iorder[n - 1];
        for (;;) {
            int j = i + i;
            if (j <= n - 1) {
                if (t[j] < t[j - 1]) j = j + 1;
                if (t[j - 1] < ddum) { t[i - 1] = t[j - 1]; iorder[i - 1] = iorder[j - 1]; i = j; }
                else break;
            } else break;
        }
        t[i - 1] = ddum; iorder[i - 1] = indxin;
        t[n - 1] = out; iorder[n - 1] = indxou;
    }
}

/* generalised Cauchy point.  p, c, wbp, v: 2m each.  iorder: n.  t: n.  Returns info. */
static int lb_cauchy(lbw *w, const double *x, const double *l, const double *u, const double *g, int *iorder,
                     double *t, double *d, double *xcp, double theta, int col, int head, double *p, double *c,
                     double *wbp, double *v, int *nseg, double sbgnrm, double epsmch)
{
    const int n = w->n, m = w->m;
    const int *nbd = w->nbd;
    int *iwhere = w->iwhere;
    if (sbgnrm <= 0.0) { memcpy(xcp, x, sizeof(double) * n); return 0; }
    int bnded = 1, nfree = n + 1, nbreak = 0, ibkmin = 0;
    double bkmin = 0.0, f1 = 0.0;
    const int col2 = 2 * col;
    for (int i = 0; i < col2; ++i) p[i] = 0.0;
    for (int i = 1; i <= n; ++i) {
        const double neggi = -g[i - 1];
        double tl = 0.0, tu = 0.0;
        if (iwhere[i - 1] != 3 && iwhere[i - 1] != -1) {
            if (nbd[i - 1] <= 2) tl = x[i - 1] - l[i - 1];
            if (nbd[i - 1] >= 2) tu = u[i - 1] - x[i - 1];
            const int xlower = nbd[i - 1] <= 2 && tl <= 0.0;
            const int xupper = nbd[i - 1] >= 2 && tu <= 0.0;
            iwhere[i - 1] = 0;
            if (xlower) { if (neggi <= 0.0) iwhere[i - 1] = 1; }
            else if (xupper) { if (neggi >= 0.0) iwhere[i - 1] = 2; }
            else { if (fabs(neggi) <= 0.0) iwhere[i - 1] = -3; }
        }
        int pointr = head;
        if (iwhere[i - 1] != 0 && iwhere[i - 1] != -1) d[i - 1] = 0.0;
        else {
            d[i - 1] = neggi;
            f1 -= neggi * neggi;
            for (int j = 1; j <= col; ++j) {
                p[j - 1] += WY(i, pointr) * neggi;
                p[col + j - 1] += WS(i, pointr) * neggi;
                pointr = pointr % m + 1;
            }
            if (nbd[i - 1] <= 2 && nbd[i - 1] != 0 && neggi < 0.0) {
                nbreak++; iorder[nbreak - 1] = i; t[nbreak - 1] = tl / (-neggi);
                if (nbreak == 1 || t[nbreak - 1] < bkmin) { bkmin = t[nbreak - 1]; ibkmin = nbreak; }
            } else if (nbd[i - 1] >= 2 && neggi > 0.0) {
                nbreak++; iorder[nbreak - 1] = i; t[nbreak - 1] = tu / neggi;
                if (nbreak == 1 || t[nbreak - 1] < bkmin) { bkmin = t[nbreak - 1]; ibkmin = nbreak; }
            } else {
                nfree--; iorder[nfree - 1] = i;
                if (fabs(neggi) > 0.0) bnded = 0;
            }
        }
    }
    if (theta != 1.0) for (int j = 0; j < col; ++j) p[col + j] *= theta;
    memcpy(xcp, x, sizeof(double) * n);
    if (nbreak == 0 && nfree == n + 1) return 0;
    for (int j = 0; j < col2; ++j) c[j] = 0.0;
    double f2 = -theta * f1;
    const double f2_org = f2;
    if (col > 0) {
        if (lb_bmv(w, col, p, v)) return 1;
        f2 -= ddot(col2, v, p);
    }
    double dtm = -f1 / f2, tsum = 0.0;
    *nseg = 1;
    if (nbreak > 0) {
        int nleft = nbreak, iter = 1, ibp;
        double tj = 0.0;
        for (;;) {
            const double tj0 = tj;
            if (iter == 1) { tj = bkmin; ibp = iorder[ibkmin - 1]; }
            else {
                if (iter == 2) {
                    if (ibkmin != nbreak) { t[ibkmin - 1] = t[nbreak - 1]; iorder[ibkmin - 1] = iorder[nbreak - 1]; }
                }
                lb_hpsolb(nleft, t, iorder, iter - 2);
                tj = t[nleft - 1]; ibp = iorder[nleft - 1];
            }
            const double dt = tj - tj0;
            if (dtm < dt) break;                               /* the minimiser is inside this interval */
            tsum += dt; nleft--; iter++;
            const double dibp = d[ibp - 1];
            double zibp;
            d[ibp - 1] = 0.0;
            if (dibp > 0.0) { zibp = u[ibp - 1] - x[ibp - 1]; xcp[ibp - 1] = u[ibp - 1]; iwhere[ibp - 1] = 2; }
            else { zibp = l[ibp - 1] - x[ibp - 1]; xcp[ibp - 1] = l[ibp - 1]; iwhere[ibp - 1] = 1; }
            if (nleft == 0 && nbreak == n) { dtm = dt; goto L999; }
            (*nseg)++;
            const double dibp2 = dibp * dibp;
            f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;
            f2 = f2 - theta * dibp2;
            if (col > 0) {
                for (int j = 0; j < col2; ++j) c[j] += dt * p[j];
                int pointr = head;
                for (int j = 1; j <= col; ++j) {
                    wbp[j - 1] = WY(ibp, pointr);
                    wbp[col + j - 1] = theta * WS(ibp, pointr);
                    pointr = pointr % m + 1;
                }
                if (lb_bmv(w, col, wbp, v)) return 1;
                const double wmc = ddot(col2, c, v), wmp = ddot(col2, p, v), wmw = ddot(col2, wbp, v);
                for (int j = 0; j < col2; ++j) p[j] -= dibp * wbp[j];
                f1 += dibp * wmc;
                f2 += 2.0 * dibp * wmp - dibp2 * wmw;
            }
            f2 = fmax(epsmch * f2_org, f2);
            if (nleft > 0) { dtm = -f1 / f2; continue; }
            else if (bnded) { f1 = 0.0; f2 = 0.0; dtm = 0.0; }
            else dtm = -f1 / f2;
            break;
        }
    }
    if (dtm <= 0.0) dtm = 0.0;
    tsum += dtm;
    for (int i = 0; i < n; ++i) xcp[i] += tsum * d[i];
L999:
    if (col > 0) for (int j = 0; j < col2; ++j) c[j] += dtm * p[j];
    return 0;
}

/* free / active index sets at the Cauchy point */
static void lb_freev(lbw *w, int *nfree, int *nenter, int *ileave, int *wrk, int updatd, int cnstnd, int iter)
{
    const int n = w->n;
    int *index = w->index, *indx2 = w->indx2, *iwhere = w->iwhere;
    *nenter = 0; *ileave = n + 1;
    if (iter > 0 && cnstnd) {
        for (int i = 1; i <= *nfree; ++i) { const int k = index[i - 1]; if (iwhere[k - 1] > 0) { (*ileave)--; indx2[*ileave - 1] = k; } }
        for (int i = 1 + *nfree; i <= n; ++i) { const int k = index[i - 1]; if (iwhere[k - 1] <= 0) { (*nenter)++; indx2[*nenter - 1] = k; } }
    }
    *wrk = (*ileave < n + 1) || (*nenter > 0) || updatd;
    *nfree = 0;
    int iact = n + 1;
    for (int i = 1; i <= n; ++i) {
        if (iwhere[i - 1] <= 0) { (*nfree)++; index[*nfree - 1] = i; }
        else { iact--; index[iact - 1] = i; }
    }
}

/* LEL^T factorisation of the indefinite reduced matrix K (inner-product blocks rebuilt from the index sets) */
static int lb_formk(lbw *w, int nsub, double theta, int col, int head)
{
    const int n = w->n, m = w->m, m2 = 2 * m;
    const int *ind = w->index;
    /* lower triangle of N = [Y'ZZ'Y  L_a'+R_z'; L_a+R_z  S'AA'S] */
    int ipntr = head;
    for (int iy = 1; iy <= col; ++iy) {
        const int is = m + iy;
        int jpntr = head;
        for (int jy = 1; jy <= iy; ++jy) {
            const int js = m + jy;
            double temp1 = 0.0, temp2 = 0.0;
            for (int k = 1; k <= nsub; ++k) { const int k1 = ind[k - 1]; temp1 += WY(k1, ipntr) * WY(k1, jpntr); }
            for (int k = nsub + 1; k <= n; ++k) { const int k1 = ind[k - 1]; temp2 += WS(k1, ipntr) * WS(k1, jpntr); }
            WN1(iy, jy) = temp1; WN1(is, js) = temp2;
            jpntr = jpntr % m + 1;
        }
        ipntr = ipntr % m + 1;
    }
    ipntr = head;
    for (int is0 = 1; is0 <= col; ++is0) {
        const int is = m + is0;
        int jpntr = head;
        for (int jy = 1; jy <= col; ++jy) {
            double temp = 0.0;
            if (is0 <= jy) {     /* R_z: over the free variables */
                for (int k = 1; k <= nsub; ++k) { const int k1 = ind[k - 1]; temp += WS(k1, ipntr) * WY(k1, jpntr); }
            } else {             /* L_a: over the active variables */
                for (int k = nsub + 1; k <= n; ++k) { const int k1 = ind[k - 1]; temp += WS(k1, ipntr) * WY(k1, jpntr); }
            }
            WN1(is, jy) = temp;
            jpntr = jpntr % m + 1;
        }
        ipntr = ipntr % m + 1;
    }
    /* upper triangle of WN = [D+Y'ZZ'Y/theta  -L_a'+R_z'; -L_a+R_z  S'AA'S*theta] */
    for (int iy = 1; iy <= col; ++iy) {
        const int is = col + iy, is1 = m + iy;
        for (int jy = 1; jy <= iy; ++jy) {
            const int js = col + jy, js1 = m + jy;
            WN(jy, iy) = WN1(iy, jy) / theta;
            WN(js, is) = WN1(is1, js1) * theta;
        }
        for (int jy = 1; jy <= iy - 1; ++jy) WN(jy, is) = -WN1(is1, jy);
        for (int jy = iy; jy <= col; ++jy) WN(jy, is) = WN1(is1, jy);
        WN(iy, iy) += SY(iy, iy);
    }
    if (lb_dpofa(w->wn, m2, col)) return -1;
    const int col2 = 2 * col;
    for (int js = col + 1; js <= col2; ++js)
        if (lb_dtrsl(w->wn, m2, col, &WN(1, js), 11)) return -1;
    for (int is = col + 1; is <= col2; ++is)
        for (int js = is; js <= col2; ++js) {
            double s = 0.0;
            for (int q = 1; q <= col; ++q) s += WN(q, is) * WN(q, js);
            WN(is, js) += s;
        }
    if (lb_dpofa(&WN(col + 1, col + 1), m2, col)) return -2;
    return 0;
}

/* r = -Z'B(xcp - xk) - Z'g */
static int lb_cmprlb(lbw *w, const double *x, const double *g, double theta, int col, int head, int nfree, int cnstnd)
{
    const int n = w->n, m = w->m;
    double *r = w->r, *z = w->z, *wa = w->wa;
    if (!cnstnd && col > 0) { for (int i = 0; i < n; ++i) r[i] = -g[i]; return 0; }
    for (int i = 1; i <= nfree; ++i) { const int k = w->index[i - 1]; r[i - 1] = -theta * (z[k - 1] - x[k - 1]) - g[k - 1]; }
    if (lb_bmv(w, col, wa + 2 * m, wa)) return -8;
    int pointr = head;
    for (int j = 1; j <= col; ++j) {
        const double a1 = wa[j - 1], a2 = theta * wa[col + j - 1];
        for (int i = 1; i <= nfree; ++i) { const int k = w->index[i - 1]; r[i - 1] += WY(k, pointr) * a1 + WS(k, pointr) * a2; }
        pointr = pointr % m + 1;
    }
    return 0;
}

/* subspace minimisation over the free variables, then the projection step of L-BFGS-B 3.0 */
static int lb_subsm(lbw *w, int nsub, const double *l, const double *u, double *x /* xcp in, subspace min out */, double *d,
                    double theta, const double *xx, const double *gg, int col, int head, int *iword, double *wv)
{
    const int n = w->n, m = w->m, m2 = 2 * m, col2 = 2 * col;
    const int *ind = w->index, *nbd = w->nbd;
    double *xp = w->xp;
    if (nsub <= 0) return 0;
    int pointr = head;
    for (int i = 1; i <= col; ++i) {
        double temp1 = 0.0, temp2 = 0.0;
        for (int j = 1; j <= nsub; ++j) { const int k = ind[j - 1]; temp1 += WY(k, pointr) * d[j - 1]; temp2 += WS(k, pointr) * d[j - 1]; }
        wv[i - 1] = temp1; wv[col + i - 1] = theta * temp2;
        pointr = pointr % m + 1;
    }
    if (lb_dtrsl(w->wn, m2, col2, wv, 11)) return 1;
    for (int i = 0; i < col; ++i) wv[i] = -wv[i];
    if (lb_dtrsl(w->wn, m2, col2, wv, 1)) return 1;
    pointr = head;
    for (int jy = 1; jy <= col; ++jy) {
        const int js = col + jy;
        for (int i = 1; i <= nsub; ++i) { const int k = ind[i - 1]; d[i - 1] += WY(k, pointr) * wv[jy - 1] / theta + WS(k, pointr) * wv[js - 1]; }
        pointr = pointr % m + 1;
    }
    for (int i = 0; i < nsub; ++i) d[i] *= 1.0 / theta;
    /* the projection: d is the Newton direction of the subspace problem */
    *iword = 0;
    memcpy(xp, x, sizeof(double) * n);
    for (int i = 1; i <= nsub; ++i) {
        const int k = ind[i - 1];
        const double dk = d[i - 1];
        double xk = x[k - 1];
        if (nbd[k - 1] != 0) {
            if (nbd[k - 1] == 1) { x[k - 1] = fmax(l[k - 1], xk + dk); if (x[k - 1] == l[k - 1]) *iword = 1; }
            else if (nbd[k - 1] == 2) {
                xk = fmax(l[k - 1], xk + dk); x[k - 1] = fmin(u[k - 1], xk);
                if (x[k - 1] == l[k - 1] || x[k - 1] == u[k - 1]) *iword = 1;
            } else if (nbd[k - 1] == 3) { x[k - 1] = fmin(u[k - 1], xk + dk); if (x[k - 1] == u[k - 1]) *iword = 1; }
        } else x[k - 1] = xk + dk;
    }
    if (*iword == 0) return 0;
    double dd_p = 0.0;
    for (int i = 0; i < n; ++i) dd_p += (x[i] - xx[i]) * gg[i];
    if (dd_p > 0.0) {
        memcpy(x, xp, sizeof(double) * n);
        double alpha = 1.0, temp1 = alpha;
        int ibd = 0;
        for (int i = 1; i <= nsub; ++i) {
            const int k = ind[i - 1];
            const double dk = d[i - 1];
            if (nbd[k - 1] != 0) {
                if (dk < 0.0 && nbd[k - 1] <= 2) {
                    const double temp2 = l[k - 1] - x[k - 1];
                    if (temp2 >= 0.0) temp1 = 0.0; else if (dk * alpha < temp2) temp1 = temp2 / dk;
                } else if (dk > 0.0 && nbd[k - 1] >= 2) {
                    const double temp2 = u[k - 1] - x[k - 1];
                    if (temp2 <= 0.0) temp1 = 0.0; else if (dk * alpha > temp2) temp1 = temp2 / dk;
                }
                if (temp1 < alpha) { alpha = temp1; ibd = i; }
            }
        }
        if (alpha < 1.0) {
            const double dk = d[ibd - 1];
            const int k = ind[ibd - 1];
            if (dk > 0.0) { x[k - 1] = u[k - 1]; d[ibd - 1] = 0.0; }
            else if (dk < 0.0) { x[k - 1] = l[k - 1]; d[ibd - 1] = 0.0; }
        }
        for (int i = 1; i <= nsub; ++i) { const int k = ind[i - 1]; x[k - 1] += alpha * d[i - 1]; }
    }
    return 0;
}

static void lb_matupd(lbw *w, const double *d, const double *r, int *itail, int iupdat, int *col, int *head,
                      double *theta, double rr, double dr, double stp, double dtd)
{
    const int n = w->n, m = w->m;
    if (iupdat <= m) { *col = iupdat; *itail = (*head + iupdat - 2) % m + 1; }
    else { *itail = *itail % m + 1; *head = *head % m + 1; }
    memcpy(&WS(1, *itail), d, sizeof(double) * n);
    memcpy(&WY(1, *itail), r, sizeof(double) * n);
    *theta = rr / dr;
    if (iupdat > m) {
        for (int j = 1; j <= *col - 1; ++j) {
            for (int q = 1; q <= j; ++q) SS(q, j) = SS(q + 1, j + 1);
            for (int q = 0; q < *col - j; ++q) SY(j + q, j) = SY(j + 1 + q, j + 1);
        }
    }
    int pointr = *head;
    for (int j = 1; j <= *col - 1; ++j) {
        SY(*col, j) = ddot(n, d, &WY(1, pointr));
        SS(j, *col) = ddot(n, &WS(1, pointr), d);
        pointr = pointr % m + 1;
    }
    SS(*col, *col) = stp == 1.0 ? dtd : stp * stp * dtd;
    SY(*col, *col) = dr;
}

int vao_lbfgsb(int32_t n, double *x, vao_fg_t fg, void *ctx, const double *lo, const double *hi,
               const vao_lbfgs_opts *o, double *Amin, int32_t *status, int32_t *nit_out, int64_t *nfev_out)
{
    if (n < 1 || !x || !fg || !o || !lo || !hi || o->m < 1) return -1;
    const int m = o->m;
    const double epsmch = 2.220446049250313e-16, big = 1e10;
    const double tol = o->ftol, pgtol = o->gtol;             /* SciPy: factr * epsmch = ftol */
    lbw W, *w = &W;
    memset(w, 0, sizeof W);
    w->n = n; w->m = m;
    w->ws = calloc((size_t)n * m, sizeof(double)); w->wy = calloc((size_t)n * m, sizeof(double));
    w->sy = calloc((size_t)m * m, sizeof(double)); w->ss = calloc((size_t)m * m, sizeof(double));
    w->wt = calloc((size_t)m * m, sizeof(double));
    w->wn = calloc((size_t)4 * m * m, sizeof(double)); w->wn1 = calloc((size_t)4 * m * m, sizeof(double));
    w->z = calloc(n, sizeof(double)); w->r = calloc(n, sizeof(double)); w->d = calloc(n, sizeof(double));
    w->t = calloc(n, sizeof(double)); w->xp = calloc(n, sizeof(double)); w->wa = calloc((size_t)8 * m, sizeof(double));
    w->index = calloc(n, sizeof(int)); w->iwhere = calloc(n, sizeof(int)); w->indx2 = calloc(n, sizeof(int));
    w->nbd = calloc(n, sizeof(int));
    double *g = calloc(n, sizeof(double)), *l = calloc(n, sizeof(double)), *u = calloc(n, sizeof(double));
    double *z = w->z, *r = w->r, *d = w->d, *t = w->t, *wa = w->wa;
    int rc = 0, warnflag = 2;
    int nit = 0;
    int64_t nfev = 0;
    double f = 0.0;
    /* bounds -> nbd (SciPy: 0 none, 1 lower, 2 both, 3 upper), x0 clipped into the box (scipy _minimize_lbfgsb) */
    for (int i = 0; i < n; ++i) {
        const int hl = lo[i] > -HUGE_VAL, hu = hi[i] < HUGE_VAL;
        w->nbd[i] = hl ? (hu ? 2 : 1) : (hu ? 3 : 0);
        l[i] = hl ? lo[i] : 0.0; u[i] = hu ? hi[i] : 0.0;
        if (hl && hu && l[i] > u[i]) { rc = -1; goto done; }
        if (hl && x[i] < l[i]) x[i] = l[i];
        if (hu && x[i] > u[i]) x[i] = u[i];
    }
    int col = 0, head = 1, itail = 0, iupdat = 0, updatd = 0, iback = 0, ifun = 0, nfree = n, nseg = 0;
    int nenter = 0, ileave = 0, iword = 0, wrk = 0, info = 0, iter = 0;
    double theta = 1.0, fold = 0.0, dnorm = 0.0, gd = 0.0, gdold = 0.0, stp = 0.0, stpmx = 0.0, dtd = 0.0, sbgnrm;
    /* active: project x, classify the variables */
    int prjctd = 0, cnstnd = 0, boxed = 1;
    for (int i = 0; i < n; ++i) {
        if (w->nbd[i] > 0) {
            if (w->nbd[i] <= 2 && x[i] <= l[i]) { if (x[i] < l[i]) { prjctd = 1; x[i] = l[i]; } }
            else if (w->nbd[i] >= 2 && x[i] >= u[i]) { if (x[i] > u[i]) { prjctd = 1; x[i] = u[i]; } }
        }
    }
    (void)prjctd;
    for (int i = 0; i < n; ++i) {
        if (w->nbd[i] != 2) boxed = 0;
        if (w->nbd[i] == 0) w->iwhere[i] = -1;
        else { cnstnd = 1; w->iwhere[i] = (w->nbd[i] == 2 && u[i] - l[i] <= 0.0) ? 3 : 0; }
    }
    if (fg(ctx, x, &f, g)) { rc = -2; goto done; }
    nfev = 1;
    sbgnrm = lb_projgr(n, l, u, w->nbd, x, g);
    if (sbgnrm <= pgtol) { warnflag = 0; goto done; }

    for (;;) {                                                   /* ---- one iteration (label 222) */
        int skip_to_subsm = 0;
        iword = -1;
        if (!cnstnd && col > 0) { memcpy(z, x, sizeof(double) * n); wrk = updatd; nseg = 0; skip_to_subsm = 1; }
        if (!skip_to_subsm) {
            info = lb_cauchy(w, x, l, u, g, w->indx2, t, d, z, theta, col, head, wa, wa + 2 * m, wa + 4 * m, wa + 6 * m,
                             &nseg, sbgnrm, epsmch);
            if (info != 0) { info = 0; col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = 0; continue; }
            lb_freev(w, &nfree, &nenter, &ileave, &wrk, updatd, cnstnd, iter);
        }
        if (nfree != 0 && col != 0) {
            if (wrk) info = lb_formk(w, nfree, theta, col, head);
            if (info != 0) { info = 0; col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = 0; continue; }
            info = lb_cmprlb(w, x, g, theta, col, head, nfree, cnstnd);
            if (info == 0) info = lb_subsm(w, nfree, l, u, z, r, theta, x, g, col, head, &iword, wa);
            if (info != 0) { info = 0; col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = 0; continue; }
        }
        /* ---- line search (label 555 / 666) */
        for (int i = 0; i < n; ++i) d[i] = z[i] - x[i];
        dcsrch_state ls;
        int task = LS_START, restart = 0, abnormal = 0;
        /* first pass of lnsrlb */
        dtd = ddot(n, d, d); dnorm = sqrt(dtd);
        stpmx = big;
        if (cnstnd) {
            if (iter == 0) stpmx = 1.0;
            else {
                for (int i = 0; i < n; ++i) {
                    const double a1 = d[i];
                    if (w->nbd[i] != 0) {
                        if (a1 < 0.0 && w->nbd[i] <= 2) {
                            const double a2 = l[i] - x[i];
                            if (a2 >= 0.0) stpmx = 0.0; else if (a1 * stpmx < a2) stpmx = a2 / a1;
                        } else if (a1 > 0.0 && w->nbd[i] >= 2) {
                            const double a2 = u[i] - x[i];
                            if (a2 <= 0.0) stpmx = 0.0; else if (a1 * stpmx > a2) stpmx = a2 / a1;
                        }
                    }
                }
            }
        }
        stp = (iter == 0 && !boxed) ? fmin(1.0 / dnorm, stpmx) : 1.0;
        memcpy(t, x, sizeof(double) * n); memcpy(r, g, sizeof(double) * n);
        fold = f; ifun = 0; iback = 0;
        for (;;) {
            gd = ddot(n, g, d);
            if (ifun == 0) { gdold = gd; if (gd >= 0.0) { info = -4; break; } }
            task = dcsrch(f, gd, &stp, 1e-3, 0.9, 0.1, 0.0, stpmx, task, &ls);
            if (task == LS_ERROR) { info = -4; break; }          /* (dcsrch rejects its input: lnsrlb would loop on ERROR) */
            if (task != LS_CONV && task != LS_WARN) {
                ifun++; nfev++; iback = ifun - 1;
                if (iback >= o->maxls) { nfev--; break; }        /* (mainlb tests iback before the evaluation is made) */
                if (stp == 1.0) memcpy(x, z, sizeof(double) * n);
                else for (int i = 0; i < n; ++i) x[i] = stp * d[i] + t[i];
                if (fg(ctx, x, &f, g)) { rc = -2; goto done; }
                continue;
            }
            break;
        }
        if (info != 0 || iback >= o->maxls) {
            memcpy(x, t, sizeof(double) * n); memcpy(g, r, sizeof(double) * n); f = fold;
            if (col == 0) { abnormal = 1; }
            else { restart = 1; }
        }
        if (abnormal) { warnflag = 2; break; }
        if (restart) { info = 0; col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = 0; continue; }
        /* ---- NEW_X */
        iter++; nit = iter;
        sbgnrm = lb_projgr(n, l, u, w->nbd, x, g);
        /* SciPy's driver looks at the iteration / evaluation budgets at every NEW_X, before the convergence tests */
        if (nit >= o->maxiter) { warnflag = 1; break; }
        if (nfev > o->maxfun) { warnflag = 1; break; }
        if (sbgnrm <= pgtol) { warnflag = 0; break; }
        {
            const double ddum = fmax(fmax(fabs(fold), fabs(f)), 1.0);
            if (fold - f <= tol * ddum) { warnflag = 0; break; }
        }
        /* ---- BFGS pair */
        for (int i = 0; i < n; ++i) r[i] = g[i] - r[i];
        const double rr = ddot(n, r, r);
        double dr, ddum;
        if (stp == 1.0) { dr = gd - gdold; ddum = -gdold; }
        else { dr = (gd - gdold) * stp; for (int i = 0; i < n; ++i) d[i] *= stp; ddum = -gdold * stp; }
        if (dr <= epsmch * ddum) { updatd = 0; continue; }
        updatd = 1; iupdat++;
        lb_matupd(w, d, r, &itail, iupdat, &col, &head, &theta, rr, dr, stp, dtd);
        info = lb_formt(w, col, theta);
        if (info != 0) { info = 0; col = 0; head = 1; theta = 1.0; iupdat = 0; updatd = 0; }
    }
done:
    if (Amin) *Amin = f;
    if (status) *status = warnflag;
    if (nit_out) *nit_out = nit;
    if (nfev_out) *nfev_out = nfev;
    free(w->ws); free(w->wy); free(w->sy); free(w->ss); free(w->wt); free(w->wn); free(w->wn1);
    free(w->z); free(w->r); free(w->d); free(w->t); free(w->xp); free(w->wa);
    free(w->index); free(w->iwhere); free(w->indx2); free(w->nbd);
    free(g); free(l); free(u);
    return rc;
}

#undef WS
#undef WY
#undef SY
#undef SS
#undef WT
#undef WN
#undef WN1
