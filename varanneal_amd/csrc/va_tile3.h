// va_tile3.h -- "column-run" tile evaluation: the production mapping of k_eval.
//
// A workgroup owns T = RY*K consecutive time rows of one seed.  Lane (ty, tx) owns
// state column tx and the CONTIGUOUS run of K rows r0 = n0 + ty*K .. r0+K-1:
//   phase A  flat, fully coalesced 16-byte staging of rows [n0-HL, n0+T+HR) of x (or
//            the trial point x + stp*d) -- and of d when a line search needs g.d -- into LDS;
//   barrier
//   phase B  each lane pulls its run (+halo rows) of its column and of the stencil's
//            neighbour columns out of LDS ONCE, evaluates f for K+HL+HR rows, the
//            residuals/q for K+HL rows and direct_m, s_m for its K rows entirely in
//            registers (no f or q ever goes through LDS), and publishes only s_m;
//   barrier
//   phase C  J_m^T s_m from the three neighbour columns of s (LDS), measurement term,
//            gradient store, parameter-gradient / line-search partial sums.
// Interior tiles run a variant with every row-range predicate compiled out; with D
// fixed at compile time every LDS address is one base register + an immediate.
//
// LDS layout: staged row R (0 = row n0-HL) lives at  R*D + P*run(R),  run(R) =
// (R-HL+K)/K, i.e. P doubles of padding after every K-row run, with P chosen so that
// (K*D + P) == D (mod 32).  The lanes of a wave then hit LDS double-word banks
// ty*D + tx + const = linear lane id (mod 32): conflict-free ds_read_b64, where the
// unpadded layout is 4-way conflicted for D=20, K=8 (measured: 70% of LDS cycles).
//
// Arithmetic restated from the reference: see va_core.h.  Shared with tests/cpu_emul.
#pragma once
#include "va_tile2.h"

namespace va {

template <int K> struct T3Regs {
    double direct[K], sown[K], xown[K], yv[K], wv[K];
};

struct Tile3 {
    int n0, ty, use_d, l, r0;     // r0 = first owned row of this lane
    Cols col;
    double stp, c;
    double *xs, *ds, *ss;         // LDS: staged x, staged d (line search only), s rows
    const double *xg, *dg;
    double *gtg;
    double p[RHS_MAX_NP];
};

// run padding in doubles: (K*D + P) == D (mod 32), P even when D is even
VA_HD constexpr int tile3_pad(int K, int D) { return (((1 - K) * D) % 32 + 32) % 32; }
// doubles of LDS for the staged arrays (x and d) and for s
VA_HD constexpr int tile3_stage_elems(int K, int D, int RY, int HLR) { return (RY * K + HLR) * D + tile3_pad(K, D) * (RY + 2); }
VA_HD constexpr int tile3_s_elems(int K, int D, int RY) { return RY * K * D + tile3_pad(K, D) * RY; }

// 16-byte accesses (addresses are even-element offsets of 128-byte aligned rows)
VA_HD void ld2(const double *p, double &a, double &b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double2 v = *reinterpret_cast<const double2 *>(p);
    a = v.x; b = v.y;
#else
    a = p[0]; b = p[1];
#endif
}
VA_HD void st2(double *p, double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    *reinterpret_cast<double2 *>(p) = make_double2(a, b);
#else
    p[0] = a; p[1] = b;
#endif
}

// phase A: flat staging.  rows outside [0,N) read as 0.  `tid`/`nt` are linear.
template <int DISC, int K, int DC, bool EDGE, bool USE_D>
VA_HD void tile3_stage(const Dims &dm, const Tile3 &t, int tid, int nt)
{
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR;
    const int D = DC > 0 ? DC : dm.D;
    const int P = tile3_pad(K, D);
    const long base = (long)(t.n0 - HL) * D;
    const int tot = (dm.T + HL + HR) * D;
    const double *xsrc = t.xg + base, *dsrc = t.dg + base;
    if ((D & 1) == 0) {
        // rows are 16-byte aligned when D is even: move two doubles per lane per access
        for (int e = 2 * tid; e < tot; e += 2 * nt) {
            double x0 = 0.0, x1 = 0.0, d0 = 0.0, d1 = 0.0;
            if (dm.dbg & 4) { x0 = 1e-3 * e; x1 = 1e-3 * (e + 1); }      // ablation: no global loads
            else if (!EDGE || (base + e >= 0 && base + e + 1 < dm.ND)) {
                ld2(xsrc + e, x0, x1);
                if (USE_D) {
                    ld2(dsrc + e, d0, d1);
                    x0 = trial(x0, t.stp, d0); x1 = trial(x1, t.stp, d1);
                }
            }
            const int a = e + P * ((e / D - HL + K) / K);
            st2(t.xs + a, x0, x1);
            if (USE_D) st2(t.ds + a, d0, d1);
        }
    } else {
        for (int e = tid; e < tot; e += nt) {
            double x0 = 0.0, d0 = 0.0;
            if (!EDGE || (base + e >= 0 && base + e < dm.ND)) {
                x0 = xsrc[e];
                if (USE_D) { d0 = dsrc[e]; x0 = trial(x0, t.stp, d0); }
            }
            const int a = e + P * ((e / D - HL + K) / K);
            t.xs[a] = x0;
            if (USE_D) t.ds[a] = d0;
        }
    }
}

// observations of the lane's own rows (issued before the first barrier so that the
// global loads overlap the staging).  Unobserved entries get weight 0.
template <int K>
VA_HD void tile3_obs(const Dims &dm, const ProblemPtrs &pp, const Tile3 &t, T3Regs<K> &rg)
{
    if (dm.nskip == 1) {
        // common case (dt_model == dt_data): every row is an observation time.  Branch-free:
        // unobserved lanes read entry 0.
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int m = t.r0 + k;
            const bool ok = t.l >= 0 && m < dm.N;
            const size_t idx = ok ? (size_t)m * dm.L + t.l : 0;
            rg.yv[k] = pp.Y[idx];
            rg.wv[k] = ok ? dm.rm : 0.0;
        }
        if (pp.rm_arr) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int m = t.r0 + k;
                const bool ok = t.l >= 0 && m < dm.N;
                rg.wv[k] = ok ? pp.rm_arr[(size_t)m * dm.L + t.l] : 0.0;
            }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        rg.yv[k] = 0.0; rg.wv[k] = 0.0;
        const int m = t.r0 + k;
        if (t.l >= 0 && m < dm.N) {
            const int nd = m / dm.nskip;
            if (nd * dm.nskip == m && nd < dm.N_data) {
                rg.yv[k] = pp.Y[(size_t)nd * dm.L + t.l];
                rg.wv[k] = pp.rm_arr ? pp.rm_arr[(size_t)nd * dm.L + t.l] : dm.rm;
            }
        }
    }
}

// phase B: f, residuals, q, direct, s for the lane's run -- registers only.
template <class RHS, int DISC, int K, bool EDGE, int DC>
VA_HD void tile3_rows(const Dims &dm, const ProblemPtrs &pp, const Tile3 &t, T3Regs<K> &rg, ThreadAcc &acc)
{
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR, NR = K + HL + HR, NQ = K + HL;
    const int D = DC > 0 ? DC : dm.D, N = dm.N, i = t.col.i;
    const int P = tile3_pad(K, D);
    const double dt = dm.dt;
    // the lane's first needed row r0-HL is staged row ty*K: LDS offset ty*(K*D+P)
    const double *xbase = t.xs + t.ty * (K * D + P);
    double xo[NR], fo[NR], q[NQ], w[NQ];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const double *xr = xbase + j * D + P * ((j - HL + K) / K);
        xo[j] = xr[i];
        const int row = t.r0 - HL + j;
        if (!EDGE || (row >= 0 && row < N)) fo[j] = RHS::f(xr, t.col, xo[j], t.p);
        else fo[j] = 0.0;
    }
    // model-error weights RF0[n, i] (va_ode.py:203-209) or the scalar; one uniform branch
    if (pp.rf0_arr) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            int row = t.r0 - HL + j;
            if (EDGE) row = row < 0 ? 0 : (row > N - 2 ? N - 2 : row);   // clamped; r = 0 there anyway
            w[j] = pp.rf0_arr[(size_t)row * D + i];
        }
    } else {
#pragma unroll
        for (int j = 0; j < NQ; ++j) w[j] = dm.rf0;
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int row = t.r0 - HL + j;              // residual (or SH half-residual) index
        double r = 0.0;
        bool have = true;
        if constexpr (DISC == DISC_SH) {
            // r0 and HL are even, so the parity of the row is the parity of j
            const int je = (j & 1) ? j - 1 : j;     // start of the interval this (half-)residual belongs to
            if ((j & 1) == 0) {
                if (EDGE) have = row >= 0 && row + 2 <= N - 1;
                r = xo[je + 2] - xo[je] - (fo[je] + 4.0 * fo[je + 1] + fo[je + 2]) * (dt / 3.0);
            } else {
                if (EDGE) have = row >= 1 && row + 1 <= N - 1;
                r = xo[je + 1] - (0.5 * (xo[je] + xo[je + 2]) + (fo[je] - fo[je + 2]) * (dt / 4.0));
            }
        } else {
            if (EDGE) have = row >= 0 && row <= N - 2;
            if constexpr (DISC == DISC_TRAPEZOID) r = xo[j + 1] - xo[j] - (0.5 * dt) * (fo[j] + fo[j + 1]);
            else if constexpr (DISC == DISC_EULER) r = xo[j + 1] - xo[j] - dt * fo[j];
            else r = xo[j + 1] - fo[j];
        }
        if (EDGE && !have) r = 0.0;                 // select, not a branch
        const double wr = w[j] * r;
        q[j] = t.c * wr;
        if (j >= HL) acc.v[EP_FE] += wr * r;        // rows of this lane's own run
    }
    // direct_m, s_m (same linear combinations as disc_direct_s, on registers)
    double *sbase = t.ss + t.ty * (K * D + P) + i;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int j = k + HL;
        double direct, s;
        if constexpr (DISC == DISC_TRAPEZOID) { direct = q[j - 1] - q[j]; s = -0.5 * dt * (q[j - 1] + q[j]); }
        else if constexpr (DISC == DISC_EULER) { direct = q[j - 1] - q[j]; s = -dt * q[j]; }
        else if constexpr (DISC == DISC_FWDMAP) { direct = q[j - 1]; s = -q[j]; }
        else {
            // Simpson-Hermite (HL = 2, K even): q1 of interval n sits at q[n], q2 at q[n+1]
            const int je = (k & 1) ? j - 1 : j;     // even row of the pair this k belongs to
            if ((k & 1) == 0) {
                direct = -q[je] - 0.5 * q[je + 1] + q[je - 2] - 0.5 * q[je - 1];
                s = -(dt / 3.0) * (q[je] + q[je - 2]) - (dt / 4.0) * (q[je + 1] - q[je - 1]);
            } else { direct = q[je + 1]; s = -(4.0 * dt / 3.0) * q[je]; }
        }
        if (EDGE && t.r0 + k >= N) { direct = 0.0; s = 0.0; }
        rg.direct[k] = direct; rg.sown[k] = s; rg.xown[k] = xo[j];
        sbase[k * D] = s;
    }
}

// phase C: gradient rows of the lane's run.
template <class RHS, int DISC, int K, bool EDGE, int DC>
VA_HD void tile3_grad(const Dims &dm, const Tile3 &t, const T3Regs<K> &rg, ThreadAcc &acc)
{
    constexpr int HL = Halo<DISC>::HL;
    const int D = DC > 0 ? DC : dm.D, i = t.col.i;
    const int P = tile3_pad(K, D);
    const int runoff = t.ty * (K * D + P);
    // own rows k = 0..K-1 are staged rows ty*K + HL + k, all in run ty+1
    const double *xrun = t.xs + runoff + HL * D + P;
    const double *drun = t.ds + runoff + HL * D + P;
    const double *srun = t.ss + runoff;
    double dval[K];
    if (t.use_d) {                                  // workgroup-uniform
#pragma unroll
        for (int k = 0; k < K; ++k) dval[k] = drun[k * D + i];
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) dval[k] = 0.0;
    }
    const double two_cme = 2.0 * dm.cme;
    double gmax = acc.v[EP_GMAX];
    double *gout = t.gtg + (long)t.r0 * D + i;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double *sr = srun + k * D;
        const double *xr = xrun + k * D;
        double g = rg.direct[k] + RHS::vjp(xr, t.col, sr[t.col.ip1], sr[t.col.im1], sr[t.col.ip2], rg.sown[k], t.p);
        RHS::pgrad(rg.sown[k], acc.v + EP_GP);
        // measurement term: wv = 0 on unobserved entries, so no branch
        const double diff = rg.xown[k] - rg.yv[k];
        const double wd = rg.wv[k] * diff;
        acc.v[EP_ME] += wd * diff;
        g += two_cme * wd;
        if (!EDGE || t.r0 + k < dm.N) { if (!(dm.dbg & 1)) gout[k * D] = g; } else g = 0.0;
        acc.v[EP_GTD] += g * dval[k];
        acc.v[EP_GN2] += g * g;
        const double ag = fabs(g);
        gmax = ag > gmax ? ag : gmax;               // compare-select: fmax() would canonicalise twice
    }
    acc.v[EP_GMAX] = gmax;
}

}  // namespace va
