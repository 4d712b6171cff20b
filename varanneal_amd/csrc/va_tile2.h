// va_tile2.h -- column-mapped tile evaluation (the fast path of k_eval).
//
// Thread (ty, tx) owns state column tx and walks time rows ty, ty+RY, ty+2RY, ...
// of the staged tile, so that
//   * the cyclic neighbour columns of the RHS stencil (i-2, i-1, i+1, i+2), the
//     observed-component lookup and every division are loop-invariant;
//   * each thread keeps its own x, d, y(obs) in registers from the (fully unrolled,
//     all-loads-up-front) staging phase to the gradient phase: nothing is re-read;
//   * consecutive lanes still touch consecutive addresses of the time-major path,
//     so HBM loads/stores stay coalesced and LDS accesses stay conflict-free.
// Two-row discretisations (euler/trapezoid/forwardmap) fuse f into the residual
// phase: load | barrier | f+q | barrier | gradient -- two workgroup barriers.
// Simpson-Hermite keeps a separate f phase (its residual needs f at three rows).
//
// Same arithmetic as the flat-mapped phases in va_core.h (which remain the generic
// fallback and an independent cross-check); host/device code shared with
// tests/cpu_emul.
#pragma once
#include "va_core.h"

namespace va {

struct Cols { int i, im1, im2, ip1, ip2; };

VA_HD Cols make_cols(int i, int D)
{
    Cols c;
    c.i = i;
    c.im1 = i == 0 ? D - 1 : i - 1;
    c.im2 = c.im1 == 0 ? D - 1 : c.im1 - 1;
    c.ip1 = i == D - 1 ? 0 : i + 1;
    c.ip2 = c.ip1 == D - 1 ? 0 : c.ip1 + 1;
    return c;
}

// Lorenz-96 in column form (neighbour columns precomputed, own value passed in a register)
struct RhsL96c {
    static constexpr int NP = 1;
    static constexpr bool CHEAP_F = true;
    static VA_HD double f(const double *xr, const Cols &c, double xi, const double *p)
    {
        return xr[c.im1] * (xr[c.ip1] - xr[c.im2]) - xi + p[0];
    }
    static VA_HD double vjp(const double *xr, const Cols &c, double s_ip1, double s_im1, double s_ip2,
                            double s_own, const double *)
    {
        return s_ip1 * (xr[c.ip2] - xr[c.im1]) + s_im1 * xr[c.im2] - s_ip2 * xr[c.ip1] - s_own;
    }
    static VA_HD void pgrad(double s_own, double *acc) { acc[0] += s_own; }
};

template <int MAXR> struct TRegs {
    double xv[MAXR], dv[MAXR], yv[MAXR], wv[MAXR];
    unsigned meas;
};

struct Tile2 {
    int n0, R, RY, ty, use_d, l;
    Cols col;
    double stp, c;
    double *xs, *fs, *qs;
    const double *xg, *dg;
    double *gtg;
    double p[RHS_MAX_NP];
};

template <class RHS>
VA_HD void tile2_params(const Dims &dm, const ProblemPtrs &pp, int b, Tile2 &t)
{
#pragma unroll
    for (int k = 0; k < RHS::NP; ++k) t.p[k] = pp.Pfull[(size_t)b * dm.NP + k];
    for (int k = 0; k < dm.NPest; ++k) {
        double v = t.xg[dm.ND + k];
        if (t.use_d) v = trial(v, t.stp, t.dg[dm.ND + k]);
        const int dst = pp.Pidx[k];
#pragma unroll
        for (int j = 0; j < RHS::NP; ++j) t.p[j] = (dst == j) ? v : t.p[j];
    }
}

// phase 1: stage x (or x + stp*d) and prefetch the observations of the owned rows.
template <int DISC, int MAXR>
VA_HD void tile2_load(const Dims &dm, const ProblemPtrs &pp, Tile2 &t, TRegs<MAXR> &rg)
{
    constexpr int HL = Halo<DISC>::HL;
    const int D = dm.D, i = t.col.i;
    rg.meas = 0u;
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
        const int lr = t.ty + k * t.RY;
        double xv = 0.0, dv = 0.0, yv = 0.0, wv = 0.0;
        if (lr < t.R) {
            const int row = t.n0 - HL + lr;
            if (row >= 0 && row < dm.N) {
                const long gi = (long)row * D + i;
                xv = t.xg[gi];
                if (t.use_d) { dv = t.dg[gi]; xv = trial(xv, t.stp, dv); }
                if (t.l >= 0 && lr >= HL && lr < HL + dm.T) {
                    int nd = row;
                    bool ok = true;
                    if (dm.nskip != 1) { nd = row / dm.nskip; ok = (nd * dm.nskip == row); }
                    if (ok && nd < dm.N_data) {
                        yv = pp.Y[(size_t)nd * dm.L + t.l];
                        wv = pp.rm_arr ? pp.rm_arr[(size_t)nd * dm.L + t.l] : dm.rm;
                        rg.meas |= 1u << k;
                    }
                }
            }
            t.xs[lr * D + i] = xv;
        }
        rg.xv[k] = xv; rg.dv[k] = dv; rg.yv[k] = yv; rg.wv[k] = wv;
    }
}

// phase 2 (only when f is not fused into the residual phase)
template <class RHS, int DISC, int MAXR>
VA_HD void tile2_f(const Dims &dm, Tile2 &t, const TRegs<MAXR> &rg)
{
    constexpr int HL = Halo<DISC>::HL;
    const int D = dm.D, i = t.col.i;
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
        const int lr = t.ty + k * t.RY;
        if (lr < t.R) {
            const int row = t.n0 - HL + lr;
            t.fs[lr * D + i] = (row >= 0 && row < dm.N) ? RHS::f(t.xs + lr * D, t.col, rg.xv[k], t.p) : 0.0;
        }
    }
}

// phase 3: weighted residual adjoints q (rows [n0-HL, n0+T)), model-error partial sum.
template <class RHS, int DISC, int MAXR, bool FUSE>
VA_HD void tile2_q(const Dims &dm, const ProblemPtrs &pp, Tile2 &t, const TRegs<MAXR> &rg, ThreadAcc &acc)
{
    constexpr int HL = Halo<DISC>::HL;
    const int D = dm.D, N = dm.N, i = t.col.i;
    const double dt = dm.dt;
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
        const int lr = t.ty + k * t.RY;
        if (lr < dm.T + HL) {
            const int row = t.n0 - HL + lr;
            const double *x0 = t.xs + lr * D;
            const double xi = rg.xv[k];
            double q = 0.0, r = 0.0;
            bool have = false;
            if (row >= 0) {
                if (DISC == DISC_SH) {
                    const double *f0 = t.fs + lr * D;
                    if ((row & 1) == 0) {
                        if (row + 2 <= N - 1) {
                            r = x0[2 * D + i] - xi - (f0[i] + 4.0 * f0[D + i] + f0[2 * D + i]) * (2.0 * dt) / 6.0;
                            have = true;
                        }
                    } else if (row + 1 <= N - 1) {
                        r = xi - ((x0[i - D] + x0[i + D]) / 2.0 + (f0[i - D] - f0[i + D]) * (2.0 * dt) / 8.0);
                        have = true;
                    }
                } else if (row <= N - 2) {
                    const double x1 = x0[D + i];
                    const double f0 = FUSE ? RHS::f(x0, t.col, xi, t.p) : t.fs[lr * D + i];
                    if (DISC == DISC_TRAPEZOID) {
                        const double f1 = FUSE ? RHS::f(x0 + D, t.col, x1, t.p) : t.fs[(lr + 1) * D + i];
                        r = x1 - xi - dt * (f0 + f1) / 2.0;
                    } else if (DISC == DISC_EULER) r = x1 - xi - dt * f0;
                    else r = x1 - f0;
                    have = true;
                }
            }
            if (have) {
                const double w = pp.rf0_arr ? pp.rf0_arr[(size_t)row * D + i] : dm.rf0;
                q = t.c * w * r;
                if (lr >= HL && row < N) acc.v[EP_FE] += w * r * r;
            }
            t.qs[lr * D + i] = q;
        }
    }
}

// phase 4: gradient of the owned rows, measurement term, parameter-gradient and
// line-search partial sums.  s at the neighbour columns is formed on the fly from q.
template <class RHS, int DISC, int MAXR>
VA_HD void tile2_g(const Dims &dm, Tile2 &t, const TRegs<MAXR> &rg, ThreadAcc &acc)
{
    constexpr int HL = Halo<DISC>::HL;
    const int D = dm.D, i = t.col.i;
#pragma unroll
    for (int k = 0; k < MAXR; ++k) {
        const int lr = t.ty + k * t.RY, lt = lr - HL, m = t.n0 + lt;
        if (lt >= 0 && lt < dm.T && m < dm.N) {
            const double *qm = t.qs + lr * D;
            double direct, s_own, dmy, s_ip1, s_im1, s_ip2;
            disc_direct_s<DISC>(qm + i, D, m, dm.dt, direct, s_own);
            disc_direct_s<DISC>(qm + t.col.ip1, D, m, dm.dt, dmy, s_ip1);
            disc_direct_s<DISC>(qm + t.col.im1, D, m, dm.dt, dmy, s_im1);
            disc_direct_s<DISC>(qm + t.col.ip2, D, m, dm.dt, dmy, s_ip2);
            double g = direct + RHS::vjp(t.xs + lr * D, t.col, s_ip1, s_im1, s_ip2, s_own, t.p);
            RHS::pgrad(s_own, acc.v + EP_GP);
            if (rg.meas & (1u << k)) {
                const double diff = rg.xv[k] - rg.yv[k];
                acc.v[EP_ME] += rg.wv[k] * diff * diff;
                g += 2.0 * dm.cme * rg.wv[k] * diff;
            }
            t.gtg[(long)m * D + i] = g;
            if (t.use_d) acc.v[EP_GTD] += g * rg.dv[k];
            acc.v[EP_GN2] += g * g;
            acc.v[EP_GMAX] = fmax(acc.v[EP_GMAX], fabs(g));
        }
    }
}

// geometry of the column-mapped kernel for a given D
VA_HD constexpr int tile2_RY(int D) { return D >= 256 ? 1 : 256 / D; }
VA_HD constexpr int tile2_threads(int D) { return ((D * tile2_RY(D) + 63) / 64) * 64; }

}  // namespace va
