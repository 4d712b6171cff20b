// va_tile3.h -- "column-run" tile evaluation: the production mapping of k_eval.
//
// A workgroup owns T = RY*K consecutive time rows of one seed.  Lane (ty, tx) owns
// state column tx and the CONTIGUOUS run of K rows r0 = n0 + ty*K .. r0+K-1:
//   phase A  flat, fully coalesced 16-byte staging of rows [n0-HL, n0+T+HR) of x (or
//            the trial point x + stp*d) into LDS.  The global loads are issued into
//            registers BEFORE the seed's state is read, so the two round trips overlap;
//   barrier
//   phase B  each lane pulls its run (+halo rows) of its column and of the stencil's
//            neighbour columns out of LDS ONCE, evaluates f for K+HL+HR rows, the
//            residuals/q for K+HL rows and direct_m, s_m for its K rows entirely in
//            registers (no f or q ever goes through LDS), and publishes only s_m;
//   barrier
//   phase C  J_m^T s_m from the neighbour columns of s (LDS), measurement term,
//            gradient store, parameter-gradient / line-search partial sums.
// Interior tiles run a variant with every row-range predicate compiled out; with D
// fixed at compile time every LDS address is one base register + an immediate.
//
// LDS layout.  Rows carry G = RHS::GHOST ghost columns on each side holding the cyclic
// neighbours (x_{D-2}, x_{D-1} | x_0 .. x_{D-1} | x_0, x_1), so the stencil reads
// x[c-2..c+2] are plain offsets from ONE per-lane base address -- no wrap selects, and
// adjacent columns pair into ds_read2_b64.  Row pitch DP = D + 2G; after every K-row
// run there are P doubles of padding with (K*DP + P) == D (mod 32): the lanes of a wave
// then hit double-word banks ty*D + tx + const = linear lane id (mod 32), i.e.
// conflict-free (the unpadded, unghosted layout measured 70% of its LDS cycles in bank
// conflicts for D=20, K=8).
//
// Arithmetic restated from the reference: see va_core.h.  Shared with tests/cpu_emul.
#pragma once
#include "va_tile2.h"

namespace va {

// Lorenz-96 on a ghosted row: xc points at the lane's own column
struct RhsL96g {
    static constexpr int NP = 1;
    static constexpr int GHOST = 2;
    static VA_HD double f(const double *xc, double xi, const double *p)
    {
        return xc[-1] * (xc[1] - xc[-2]) - xi + p[0];
    }
    // (J^T s)_j = s_{j+1}(x_{j+2} - x_{j-1}) + s_{j-1} x_{j-2} - s_{j+2} x_{j+1} - s_j
    static VA_HD double vjp(const double *xc, const double *sc, double s_own, const double *)
    {
        return sc[1] * (xc[2] - xc[-1]) + sc[-1] * xc[-2] - sc[2] * xc[1] - s_own;
    }
    static VA_HD void pgrad(const double *, double, double s_own, const double *, double *acc) { acc[0] += s_own; }
};

template <int K> struct T3Regs {
    double direct[K], sown[K], xown[K], yv[K], wv[K], dval[K];
};

struct Tile3 {
    int n0, ty, tx, use_d, l, r0;   // r0 = first owned row of this lane
    double stp, c;
    double *xs, *ss;                // LDS: staged x rows, s rows
    const double *xg, *dg;
    double *gtg;
    double p[RHS_MAX_NP];
};

// workgroup geometry: RY lanes per state column.  Small groups win: what limits these kernels is
// how many INDEPENDENT workgroups a CU holds (their load / compute / store phases overlap), not
// the share of halo rows.  256-thread groups up to D = 64 (12 lanes per column at D = 20) and
// again for 128 < D <= 256 (ONE lane per column at D = 200, K = 8: four groups per CU; measured
// at C4: 314 us, against 329 us for 512-thread groups with two lanes per column and 378 us for a
// 1024-thread group with five, which is alone on its CU above 64 VGPRs); 512-thread groups for
// 64 < D <= 128 (D = 100: 156 us against 179 us with 256); one lane per column in groups of up to
// 512 threads for 256 < D <= 512 (D = 500: 295 us against 410 us for a 1024-thread group), of up to
// 1024 threads for 512 < D <= 1024 (still 3x the flat kernel, which takes over beyond that)
VA_HD constexpr int tile3_ntmax(int D) { return D <= 64 ? 256 : (D <= 128 ? 512 : (D <= 256 ? 256 : (D <= 512 ? 512 : 1024))); }
VA_HD constexpr int tile3_RY(int D) { return tile3_ntmax(D) / D > 0 ? tile3_ntmax(D) / D : 1; }
VA_HD constexpr int tile3_threads(int D) { return ((D * tile3_RY(D) + 63) / 64) * 64; }

// staged double2 per lane when D is only known at run time: rows R = RY*K + HL + HR are fetched
// RP = floor(2*NT/D) >= 2*RY at a time, so ceil(R / RP) <= K/2 + ceil((HL+HR) / 2) passes
// (K/2 + 2: with ONE lane per column, RY = 1, Simpson-Hermite needs the second extra pass)
VA_HD constexpr int tile3_ns_runtime(int K) { return (K + 1) / 2 + 2; }

// geometry helpers (G = ghost columns per side)
VA_HD constexpr int tile3_dp(int D, int G) { return D + 2 * G; }
VA_HD constexpr int tile3_pad(int K, int D, int G) { return ((D - K * tile3_dp(D, G)) % 32 + 32) % 32; }
VA_HD constexpr int tile3_runpitch(int K, int D, int G) { return K * tile3_dp(D, G) + tile3_pad(K, D, G); }
VA_HD constexpr int tile3_stage_elems(int K, int D, int G, int RY, int HLR)
{
    return (RY * K + HLR) * tile3_dp(D, G) + tile3_pad(K, D, G) * (RY + 2);
}
VA_HD constexpr int tile3_s_elems(int K, int D, int G, int RY) { return RY * tile3_runpitch(K, D, G); }
// staged-row address: row R (0 = n0-HL), column c
VA_HD constexpr int tile3_addr(int R, int c, int K, int D, int G, int HL)
{
    return R * tile3_dp(D, G) + tile3_pad(K, D, G) * ((R - HL + K) / K) + G + c;
}
// observed-component lookup without a load when D <= 64: index among the set bits
VA_HD int obs_index(unsigned long long mask, int tx)
{
    if (!((mask >> tx) & 1ull)) return -1;
    const unsigned long long below = mask & ((1ull << tx) - 1ull);
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(below);
#else
    return __builtin_popcountll(below);
#endif
}

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

// write one staged value pair (even D) and its ghost copies
template <int G>
VA_HD void tile3_put2(double *base, int a, int col, int D, double v0, double v1)
{
    st2(base + a, v0, v1);
    if (col < G) st2(base + a + D, v0, v1);             // x_0, x_1 also right of x_{D-1}
    if (col >= D - G) st2(base + a - D, v0, v1);        // x_{D-2}, x_{D-1} also left of x_0
}

// phase A (even D).  Staging lanes are laid out (sr, cp) = (tid / (D/2), tid % (D/2)): lane
// handles the column pair cp of rows sr, sr+RP, sr+2RP, ... with RP = 2*nt/D rows per pass.
// Consecutive lanes still read consecutive 16-byte pieces of the time-major path (fully
// coalesced), the column is loop-invariant (no div/mod per element, ghost-copy predicates
// hoisted) and lanes >= RP*D/2 idle during staging only.
// step 1: issue the global loads of x into registers.  NS double2 per lane.  Branch-free:
// every lane loads from a clamped (always valid) row, so the NS loads issue back to back;
// rows that do not exist are zeroed by a select (edge tiles) or never stored (row >= R).
template <int DISC, int K, int DC, bool EDGE, int NS>
VA_HD void tile3_stage_load(const Dims &dm, int n0, const double *xg, int tid, int nt, double (&xr)[NS][2])
{
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR;
    const int D = DC > 0 ? DC : dm.D, H = D / 2;
    const int RP = (2 * nt) / D, R = dm.T + HL + HR;
    const int sr = tid / H, cp = tid - sr * H;
    const double *xcol = xg + 2 * cp;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        int row = sr + u * RP;
        row = row < R ? row : R - 1;
        int grow = n0 - HL + row;
        bool ok = true;
        if (EDGE) {
            const int gc = grow < 0 ? 0 : (grow > dm.N - 1 ? dm.N - 1 : grow);
            ok = gc == grow; grow = gc;
        }
        ld2(xcol + (long)grow * D, xr[u][0], xr[u][1]);
        if (EDGE && !ok) { xr[u][0] = 0.0; xr[u][1] = 0.0; }
    }
}

// step 2: combine with d (line search) and write LDS incl. ghost copies.
template <class RHS, int DISC, int K, int DC, bool EDGE, bool USE_D, int NS>
VA_HD void tile3_stage_store(const Dims &dm, const Tile3 &t, int tid, int nt, double (&xr)[NS][2])
{
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR, G = RHS::GHOST;
    const int D = DC > 0 ? DC : dm.D, H = D / 2;
    const int DP = tile3_dp(D, G), P = tile3_pad(K, D, G);
    const int RP = (2 * nt) / D, R = dm.T + HL + HR;
    const int sr = tid / H, cp = tid - sr * H, col = 2 * cp;
    const double *dcol = t.dg + col;
    const bool gr = col < G, gl = col >= D - G;          // loop-invariant ghost predicates
    double dd[NS][2];
    if (USE_D) {                                         // same clamped, branch-free loads for d
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            int row = sr + u * RP;
            row = row < R ? row : R - 1;
            int grow = t.n0 - HL + row;
            if (EDGE) grow = grow < 0 ? 0 : (grow > dm.N - 1 ? dm.N - 1 : grow);
            ld2(dcol + (long)grow * D, dd[u][0], dd[u][1]);
        }
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int row = sr + u * RP;
        double x0 = xr[u][0], x1 = xr[u][1];
        if (USE_D) { x0 = trial(x0, t.stp, dd[u][0]); x1 = trial(x1, t.stp, dd[u][1]); }
        if (sr < RP && row < R) {
            double *dst = t.xs + row * DP + P * ((row - HL + K) / K) + G + col;
            st2(dst, x0, x1);
            if (gr) st2(dst + D, x0, x1);                // x_0, x_1 also right of x_{D-1}
            if (gl) st2(dst - D, x0, x1);                // x_{D-2}, x_{D-1} also left of x_0
        }
    }
}

// phase A for odd D (scalar accesses; rare): loads and stores in one loop
template <class RHS, int DISC, int K, int DC, bool EDGE, bool USE_D>
VA_HD void tile3_stage_odd(const Dims &dm, const Tile3 &t, int tid, int nt)
{
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR, G = RHS::GHOST;
    const int D = DC > 0 ? DC : dm.D;
    const long base = (long)(t.n0 - HL) * D;
    const int tot = (dm.T + HL + HR) * D;
    for (int e = tid; e < tot; e += nt) {
        double x0 = 0.0, d0 = 0.0;
        if (!EDGE || (base + e >= 0 && base + e < dm.ND)) {
            x0 = t.xg[base + e];
            if (USE_D) { d0 = t.dg[base + e]; x0 = trial(x0, t.stp, d0); }
        }
        const int row = e / D, col = e - row * D;
        const int a = tile3_addr(row, col, K, D, G, HL);
        t.xs[a] = x0;
        if (col < G) t.xs[a + D] = x0;
        if (col >= D - G) t.xs[a - D] = x0;
    }
}

// observations of the lane's own rows (issued before the first barrier so that the
// global loads overlap the staging).  Unobserved entries get weight 0.
template <int K>
VA_HD void tile3_obs(const Dims &dm, const ProblemPtrs &pp, const Tile3 &t, T3Regs<K> &rg)
{
    // the lane's own entries of d for the g.d partial (line search only; workgroup-uniform
    // branch).  Read again from global (L2-hot: the staging just fetched them) rather than
    // staged: a second LDS image of the tile would cost a workgroup per CU of occupancy.
    if (t.use_d) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            rg.dval[k] = (t.r0 + k < dm.N) ? t.dg[(long)(t.r0 + k) * dm.D + t.tx] : 0.0;
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) rg.dval[k] = 0.0;
    }
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
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR, NR = K + HL + HR, NQ = K + HL, G = RHS::GHOST;
    const int D = DC > 0 ? DC : dm.D, N = dm.N, i = t.tx;
    const int DP = tile3_dp(D, G), P = tile3_pad(K, D, G);
    const double dt = dm.dt;
    // the lane's first needed row r0-HL is staged row ty*K; its own column sits at +G+tx
    const double *xc0 = t.xs + t.ty * (K * DP + P) + G + i;
    double xo[NR], fo[NR], q[NQ], w[NQ];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const double *xc = xc0 + j * DP + P * ((j - HL + K) / K);
        xo[j] = xc[0];
        const int row = t.r0 - HL + j;
        if (!EDGE || (row >= 0 && row < N)) fo[j] = RHS::f(xc, xo[j], t.p);
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
    // direct_m, s_m (same linear combinations as disc_direct_s, on registers); s with ghosts
    double *sc0 = t.ss + t.ty * (K * DP + P) + G + i;
    // ghost copies for the cyclic neighbours: the first G columns also write D to the right,
    // the last G columns D to the left (D >= 2G, so a lane is in at most one of the two sets)
    const int goff = i < G ? D : (i >= D - G ? -D : 0);
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
        sc0[k * DP] = s;
    }
    if (goff != 0) {                                 // one divergent region for all K ghost copies
#pragma unroll
        for (int k = 0; k < K; ++k) sc0[k * DP + goff] = rg.sown[k];
    }
}

// phase C: gradient rows of the lane's run.
template <class RHS, int DISC, int K, bool EDGE, int DC>
VA_HD void tile3_grad(const Dims &dm, const Tile3 &t, const T3Regs<K> &rg, ThreadAcc &acc)
{
    constexpr int HL = Halo<DISC>::HL, G = RHS::GHOST;
    const int D = DC > 0 ? DC : dm.D, i = t.tx;
    const int DP = tile3_dp(D, G), P = tile3_pad(K, D, G);
    const int runoff = t.ty * (K * DP + P) + G + i;
    // own rows k = 0..K-1 are staged rows ty*K + HL + k, all in run ty+1
    const double *xrun = t.xs + runoff + HL * DP + P;
    const double *srun = t.ss + runoff;
    const double two_cme = 2.0 * dm.cme;
    double gmax = acc.v[EP_GMAX];
    double *gout = t.gtg + (long)t.r0 * D + i;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double g = rg.direct[k] + RHS::vjp(xrun + k * DP, srun + k * DP, rg.sown[k], t.p);
        if (!EDGE || t.r0 + k < dm.N) RHS::pgrad(xrun + k * DP, rg.xown[k], rg.sown[k], t.p, acc.v + EP_GP);   // (a row that does not exist sits at x = 0)
        // measurement term: wv = 0 on unobserved entries, so no branch
        const double diff = rg.xown[k] - rg.yv[k];
        const double wd = rg.wv[k] * diff;
        acc.v[EP_ME] += wd * diff;
        g += two_cme * wd;
        if (!EDGE || t.r0 + k < dm.N) gout[k * D] = g; else g = 0.0;
        acc.v[EP_GTD] += g * rg.dval[k];
        acc.v[EP_GN2] += g * g;
        gmax = __builtin_fmax(gmax, __builtin_fabs(g));      // one v_max_f64 with |g| as a source modifier
    }
    acc.v[EP_GMAX] = gmax;
}

}  // namespace va
