// va_tile4.h -- "wave-private column runs": the production mapping of the evaluation kernel for
// narrow states (even D <= 64 that fill a wave well, e.g. Lorenz-96 D = 20 of the reference's
// example and of BASELINE configs 1-3).
//
// A wave owns RW = floor(64 / D) column runs: lane = a*D + tx holds state column tx of the K
// consecutive time rows r0 = n0w + a*K .. r0+K-1.  Nothing is shared between the waves of a
// workgroup -- no workgroup barrier anywhere in the kernel:
//   stage    the wave copies ITS rows [n0w-HL, n0w+RW*K+HR) of x global -> LDS with direct-to-LDS
//            loads (global_load_lds_dwordx4: no VGPRs, no ds_write pass).  The LDS image is the
//            global row-major layout with P doubles of padding after every K-row run, placed by
//            the per-lane SOURCE address (the LDS side of such a load is lane-linear); with
//            (K*D + P) == D (mod 32) lane l reads LDS double-bank l (mod 32): conflict-free;
//   rows     each lane pulls its column and the stencil's neighbour columns (cyclic wrap folded
//            into three per-lane base addresses: no ghost columns) for K+HL+HR rows, evaluates f,
//            the residuals, q, direct_m and s_m in registers;
//   scatter  instead of handing s to the neighbours (who would need x again), the lane forms the
//            adjoint PRODUCTS s_i * df_i/dx_j of its own element and publishes those (va_core.h
//            header: dA/dx_m = direct_m + J_m^T s_m); Lorenz-96 needs two per element
//            (u = s (x_{i+1} - x_{i-2}), v = s x_{i-1});
//   gather   (J^T s)_j = -s_j + u_{j+1} + v_{j-1} - v_{j+2} from the same wave's products (LDS is
//            in order within a wave: no barrier), measurement term, gradient store, partial sums.
// Arithmetic restated from the reference: see va_core.h.  Shared with tests/cpu_emul.
#pragma once
#include "va_tile3.h"

namespace va {

// Lorenz-96 in scatter form: f_i = x_{i-1}(x_{i+1} - x_{i-2}) - x_i + k  (cyclic)
// (examples/Lorenz96_D20/Lorenz96_anneal.py:15-16)
struct RhsL96s {
    static constexpr int NP = 1;
    static constexpr int NB = 3;                       // neighbour columns f reads
    static VA_HD constexpr int nb_off(int k) { return k == 0 ? -2 : (k == 1 ? -1 : 1); }
    static constexpr int NE = 2;                       // products published per element
    static constexpr int NG = 3;                       // products gathered per element
    static VA_HD constexpr int g_e(int t) { return t == 0 ? 0 : 1; }                   // which product
    static VA_HD constexpr int g_off(int t) { return t == 0 ? 1 : (t == 1 ? -1 : 2); } // sender column - own column
    static constexpr bool USES_T = false;              // autonomous: no model time, no stimulus
    static constexpr bool GUARD_EDGE = false;          // polynomial in x: finite at the x = 0 of rows that do not exist
    static constexpr int NSTIM = 0;
    // (int col: the lane's state column -- unused by a translation-invariant right-hand side)
    static VA_HD double f(int, double x0, const double *xn, const double *p, double, const double *)
    {
        return xn[1] * (xn[2] - xn[0]) - x0 + p[0];
    }
    // e[0] = s df_i/dx_{i-1},  e[1] = s df_i/dx_{i+1} = -s df_i/dx_{i-2};  diag = s df_i/dx_i
    static VA_HD void scatter(int, double s, double, const double *xn, const double *, double, const double *, double *e, double &diag)
    {
        e[0] = s * (xn[2] - xn[0]);
        e[1] = s * xn[1];
        diag = -s;
    }
    // sum_{i != j} s_i df_i/dx_j for the own column j from the received products r[0..NG)
    static VA_HD double gather(const double *r) { return r[0] + (r[1] - r[2]); }
    static VA_HD void pgrad(int, double s, double, const double *, const double *, double, const double *, double *acc) { acc[0] += s; }
};

// ------------------------------------------------------------------ geometry
struct Geo4 {
    int D, K, RW, NW, SUB, T;     // SUB sub-tiles (RW runs of K rows each) per wave, one after the other; T = rows per workgroup = NW*SUB*RW*K
    int P, PITCH;                 // doubles of padding per run, run pitch
    int PP, KDP;                  // pieces (16 B) per run pitch / per K rows
    int XP;                       // pieces of one wave's x image
    int XW, EW1, R2;              // doubles: one x image, one product array, second region (products / d images)
    int WAVE;                     // doubles of LDS per wave (SUB x images + second region + reduction strip)
    unsigned magic;               // floor(q / PP) == (q * magic) >> 20 for q < 4096
};
constexpr int T4_STRIP = 4 * 32;  // [4 rows of 16 lanes][up to 32 values]
constexpr int T4_NI_MAX = 6;      // direct-to-LDS instructions per wave image when D is a run-time value

VA_HD constexpr bool tile4_ok(int D) { return D >= 4 && D <= 64 && (D & 1) == 0 && (64 / D) * D >= 48; }

template <int HLR>
VA_HD constexpr Geo4 tile4_geo(int D, int K, int NE, int SUB = 1, int NW = 4)
{
    Geo4 g{};
    g.D = D; g.K = K; g.RW = 64 / D; g.NW = NW; g.SUB = SUB; g.T = NW * SUB * g.RW * K;
    g.P = ((D - K * D) % 32 + 32) % 32;
    g.PITCH = K * D + g.P;
    g.PP = g.PITCH / 2; g.KDP = K * D / 2;
    g.XP = g.RW * g.PP + HLR * D / 2;
    g.XW = ((2 * g.XP + 15) / 16) * 16;
    g.EW1 = g.RW * g.PITCH;
    const int e = NE * g.EW1, dimg = SUB * g.XW;           // the d images of a line-search point die before the products are written
    g.R2 = (((e > dimg ? e : dimg) + 15) / 16) * 16;
    g.WAVE = SUB * g.XW + g.R2 + T4_STRIP;
    g.magic = (unsigned)((1u << 20) / (unsigned)g.PP + 1u);
    return g;
}
// (host) the magic multiplier must reproduce the division for every piece index the kernel forms
inline bool tile4_magic_ok(const Geo4 &g)
{
    for (unsigned q = 0; q < 4096; ++q)
        if (((q * g.magic) >> 20) != q / (unsigned)g.PP) return false;
    return true;
}

// LDS piece q of a wave's image -> source piece relative to the wave's first staged row
// (n0w - HL), or -1 beyond the image.  Pad pieces repeat the run's last piece.
VA_HD int tile4_src_piece(const Geo4 &g, int q)
{
    if (q >= g.XP) return -1;
    const int blk = (int)(((unsigned)q * g.magic) >> 20);
    const int rem = q - blk * g.PP;
    const int in = (blk < g.RW && rem > g.KDP - 1) ? g.KDP - 1 : rem;
    return blk * g.KDP + in;
}
// double offset of staged row R (0 = n0w - HL) of a wave's image, column 0
VA_HD constexpr int tile4_row(const Geo4 &g, int R) { return (R / g.K) * g.PITCH + (R % g.K) * g.D; }

template <int K, int NE> struct T4Regs {
    double direct[K], xown[K], yv[K], wv[K], dval[K];     // direct_m + the diagonal term of J^T s; wv unused when W_SCALAR
    unsigned has;                                         // WS == 2 only: bit k set <=> the lane's row k has an observation
};

struct Tile4 {
    int n0w, a, tx, r0, use_d, l;   // first owned row of the wave, run, column, first owned row of the lane
    double c;
    double wobs;                    // W_SCALAR: RM on observed columns, 0 elsewhere
    const double *xs;               // LDS: the wave's x image
    double *es;                     // LDS: the wave's product arrays [NE][RW*PITCH]
    double *gtg;
    double p[RHS_MAX_NP];
};

// LDS reads that must stay single ds_read_b64 (2 LDS cycles per wave): merged pairs (ds_read2_b64)
// take 8 cycles, half the bytes per clock, and the LDS pipe is what the rows / gather phases are
// bound by.  volatile keeps them apart; the explicit address space keeps them DS (not flat) loads.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const volatile __attribute__((address_space(3))) double *lds_cvp;
#define VA_LDS_CVP(p) ((lds_cvp)(p))
#else
typedef const volatile double *lds_cvp;
#define VA_LDS_CVP(p) ((lds_cvp)(p))
#endif

VA_HD int wrap_col(int c, int D) { c += c < 0 ? D : 0; c -= c >= D ? D : 0; return c; }

// observations / weights of the lane's own rows (general form: data every nskip-th row, weight
// arrays).  t.l = position of the lane's column in Lidx, or -1.
template <int K, int NE>
VA_HD void tile4_obs(const Dims &dm, const ProblemPtrs &pp, const Tile4 &t, T4Regs<K, NE> &rg)
{
    if (dm.nskip == 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int m = t.r0 + k;
            const bool ok = t.l >= 0 && m < dm.N;
            const size_t idx = ok ? (size_t)m * dm.L + t.l : 0;
            rg.yv[k] = ok ? pp.Y[idx] : 0.0;
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

// Per-row, per-column model-error weights (RF0 arrays) of the lane's K + HL residual rows.  They are needed after the
// f evaluations of tile4_rows, whose registers they must not occupy meanwhile, and fetched there they cost a round trip in
// the middle of the phase: the lane requests them early (tile4_rfw_load, with its observations), parks them in ITS OWN
// slots of the product arrays (tile4_rfw_store: slot (e, k) of the lane holds weight e * K + k -- nobody else touches
// those slots before the lane itself overwrites them with its products) and reads them back where the residuals are formed.
// (D a compile-time constant only: with D in a register the parked weights cost the generic instantiations 30+ registers)
template <int K, int NE, int HL, int DC> constexpr bool tile4_rfw_in_lds() { return DC > 0 && NE * K >= K + HL; }
// weight arrays, D a compile-time constant: the measurement terms are formed at the top of tile4_rows (same reason)
template <int WS, int DC> constexpr bool tile4_me_first() { return WS == 0 && DC > 0; }
template <int K, int HL>
VA_HD void tile4_rfw_load(const Dims &dm, const ProblemPtrs &pp, const Tile4 &t, int D, double (&wq)[K + HL])
{
#pragma unroll
    for (int j = 0; j < K + HL; ++j) {
        const int row = t.r0 - HL + j;
        const int rc = row < 0 ? 0 : (row > dm.N - 2 ? dm.N - 2 : row);    // clamped; the residual is 0 there anyway
        wq[j] = pp.rf0_arr[(size_t)rc * D + t.tx];
    }
}
template <int K, int HL>
VA_HD void tile4_rfw_store(const Geo4 &g, const Tile4 &t, int D, const double (&wq)[K + HL])
{
    double *ep = t.es + t.a * g.PITCH + t.tx;
#pragma unroll
    for (int j = 0; j < K + HL; ++j) ep[(j / K) * g.EW1 + (j % K) * D] = wq[j];
}

// rows + scatter: f, residuals, q, direct, s for the lane's run in registers; publishes the products.
// W_SCALAR: scalar RM and RF0, data at every row (the reference's Lorenz-96 example and every BASELINE
// config): the weights factor out of the sums and the loops carry no weight registers.  WS = 0: weight arrays,
// 1: W_SCALAR, 2: W_SCALAR with data at every merr_nskip-th row (rows without data masked by rg.has).
template <class RHS, int DISC, int K, bool EDGE, int DC, int WS>
VA_HD void tile4_rows(const Dims &dm, const ProblemPtrs &pp, const Geo4 &g, const Tile4 &t,
                      T4Regs<K, RHS::NE> &rg, ThreadAcc &acc)
{
    constexpr bool W_SCALAR = WS != 0;                   // (WS == 2: scalar weights, data every nskip-th row -- rg.has)
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR, NR = K + HL + HR, NQ = K + HL, NB = RHS::NB, NE = RHS::NE;
    const int D = DC > 0 ? DC : g.D, N = dm.N, PITCH = g.PITCH;
    const double dt = dm.dt;
    // the lane's first needed row r0-HL is staged row a*K of the wave's image
    lds_cvp x0p = VA_LDS_CVP(t.xs + t.a * PITCH + t.tx);
    lds_cvp xnp[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) xnp[k] = VA_LDS_CVP(t.xs + t.a * PITCH + wrap_col(t.tx + RHS::nb_off(k), D));
    if constexpr (tile4_me_first<WS, DC>()) {
        // weight arrays: the measurement terms first -- w (x - y) takes the data registers' place and the weights die
        // before the f evaluations below (whose registers are the kernel's peak); the gather phase adds 2 cme w (x - y)
        double me = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int j = k + HL;
            double xk = x0p[j < K ? j * D : PITCH + (j - K) * D];
            if (EDGE) xk = (t.r0 + k < N) ? xk : 0.0;
            const double diff = xk - rg.yv[k], dw = rg.wv[k] * diff;
            me = fma(dw, diff, me);
            rg.yv[k] = dw;
        }
        acc.v[EP_ME] += me;
    }
    double xo[NR], fo[NR], q[NQ];
    double xnb[K][NB];                                   // neighbour values of the own rows (for the products)
    double tm[NR];                                       // model time of the rows (non-autonomous right-hand sides only)
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int off = j < K ? j * D : PITCH + (j - K) * D;
        double xn[NB];
        xo[j] = x0p[off];
#pragma unroll
        for (int k = 0; k < NB; ++k) xn[k] = xnp[k][off];
        const int row = t.r0 - HL + j;
        const int rowc = row < 0 ? 0 : (row > N - 1 ? N - 1 : row);       // (clamped: only ever dereferenced, never used, outside [0, N))
        tm[j] = (RHS::USES_T && pp.tmodel) ? pp.tmodel[rowc] : 0.0;
        const double *st = RHS::NSTIM > 0 ? pp.stim + (size_t)rowc * pp.nstim : nullptr;
        if (EDGE) {
            // staged rows that do not exist hold whatever lies next to the path in memory: zero them
            // (selects), so that everything computed from them below is an exact zero
            const bool ok = row >= 0 && row < N;
            xo[j] = ok ? xo[j] : 0.0;
#pragma unroll
            for (int k = 0; k < NB; ++k) xn[k] = ok ? xn[k] : 0.0;
            fo[j] = RHS::f(t.tx, xo[j], xn, t.p, tm[j], st);
            fo[j] = ok ? fo[j] : 0.0;
        } else fo[j] = RHS::f(t.tx, xo[j], xn, t.p, tm[j], st);
        if (j >= HL && j < HL + K) {
#pragma unroll
            for (int k = 0; k < NB; ++k) xnb[j - HL][k] = xn[k];
        }
    }
    const double cw = t.c * dm.rf0;
    double fe = 0.0;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int row = t.r0 - HL + j;
        double r = 0.0;
        bool have = true;
        if constexpr (DISC == DISC_SH) {
            const int je = (j & 1) ? j - 1 : j;
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
        if (EDGE) r = have ? r : 0.0;
        if constexpr (W_SCALAR) {
            q[j] = cw * r;
            if (j >= HL) fe = fma(r, r, fe);
        } else {
            double w = dm.rf0;
            if (pp.rf0_arr) {
                // (tile4_rfw_store put it there; a volatile read: it stays below the f evaluations, where the registers are free)
                if constexpr (tile4_rfw_in_lds<K, NE, HL, DC>()) w = VA_LDS_CVP(t.es + t.a * PITCH + t.tx)[(j / K) * g.EW1 + (j % K) * D];
                else {
                    int rc = row < 0 ? 0 : (row > N - 2 ? N - 2 : row);    // clamped; r = 0 there anyway
                    w = pp.rf0_arr[(size_t)rc * D + t.tx];
                }
            }
            const double wr = w * r;
            q[j] = t.c * wr;
            if (j >= HL) fe = fma(wr, r, fe);
        }
    }
    acc.v[EP_FE] += W_SCALAR ? dm.rf0 * fe : fe;
    if constexpr (!tile4_me_first<WS, DC>()) {   // measurement error of the own rows (A needs nothing from the gather phase: a plain S1
        // evaluation publishes its partial sums right after this function)
        double me = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double diff = xo[k + HL] - rg.yv[k];
            if constexpr (WS == 2) diff = ((rg.has >> k) & 1u) ? diff : 0.0;
            if constexpr (W_SCALAR) me = fma(diff, diff, me);
            else me = fma(rg.wv[k] * diff, diff, me);
        }
        acc.v[EP_ME] += W_SCALAR ? t.wobs * me : me;
    }
    double *ep = t.es + t.a * PITCH + t.tx;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int j = k + HL;
        double direct, s;
        if constexpr (DISC == DISC_TRAPEZOID) { direct = q[j - 1] - q[j]; s = -0.5 * dt * (q[j - 1] + q[j]); }
        else if constexpr (DISC == DISC_EULER) { direct = q[j - 1] - q[j]; s = -dt * q[j]; }
        else if constexpr (DISC == DISC_FWDMAP) { direct = q[j - 1]; s = -q[j]; }
        else {
            const int je = (k & 1) ? j - 1 : j;
            if ((k & 1) == 0) {
                direct = -q[je] - 0.5 * q[je + 1] + q[je - 2] - 0.5 * q[je - 1];
                s = -(dt / 3.0) * (q[je] + q[je - 2]) - (dt / 4.0) * (q[je + 1] - q[je - 1]);
            } else { direct = q[je + 1]; s = -(4.0 * dt / 3.0) * q[je]; }
        }
        // (rows >= N: every q that enters is an exact zero already, see above)
        double e[NE], diag;
        const int rowk = t.r0 + k < N ? t.r0 + k : N - 1;
        const double *stk = RHS::NSTIM > 0 ? pp.stim + (size_t)rowk * pp.nstim : nullptr;
        RHS::scatter(t.tx, s, xo[j], xnb[k], t.p, tm[j], stk, e, diag);
        if (EDGE && RHS::GUARD_EDGE) {
            // a row that does not exist was evaluated at x = 0, where a right-hand side may be singular
            // (1/x, log x): its s = 0 does not make 0 * inf an exact zero, selects do
            const bool okk = t.r0 + k < N;
            double pa[RHS::NP > 0 ? RHS::NP : 1];
#pragma unroll
            for (int u = 0; u < RHS::NP; ++u) pa[u] = 0.0;
            RHS::pgrad(t.tx, s, xo[j], xnb[k], t.p, tm[j], stk, pa);
#pragma unroll
            for (int u = 0; u < RHS::NP; ++u) acc.v[EP_GP + u] += okk ? pa[u] : 0.0;
#pragma unroll
            for (int u = 0; u < NE; ++u) e[u] = okk ? e[u] : 0.0;
            diag = okk ? diag : 0.0;
        } else RHS::pgrad(t.tx, s, xo[j], xnb[k], t.p, tm[j], stk, acc.v + EP_GP);
        rg.direct[k] = direct + diag; rg.xown[k] = xo[j];
#pragma unroll
        for (int u = 0; u < NE; ++u) ep[u * g.EW1 + k * D] = e[u];
    }
}

// gather: gradient rows of the lane's run.  LSQ: also the sums the line search needs (g.d, g.g, max|g|).
// The K gradient values of the lane's run come back in gvv[]; the caller stores them (rows >= N of an
// edge tile hold exact zeros and must not be stored).
template <class RHS, int DISC, int K, bool EDGE, int DC, int WS, bool LSQ>
VA_HD void tile4_grad(const Dims &dm, const Geo4 &g, const Tile4 &t, const T4Regs<K, RHS::NE> &rg, ThreadAcc &acc,
                      double (&gvv)[K])
{
    constexpr bool W_SCALAR = WS != 0;
    constexpr int NG = RHS::NG;
    const int D = DC > 0 ? DC : g.D;
    lds_cvp rp[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) rp[u] = VA_LDS_CVP(t.es + RHS::g_e(u) * g.EW1 + t.a * g.PITCH + wrap_col(t.tx + RHS::g_off(u), D));
    const double two_cme = 2.0 * dm.cme;
    const double c2 = two_cme * t.wobs;
    double gmax = 0.0, gtd = 0.0, gn2 = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double r[NG];
#pragma unroll
        for (int u = 0; u < NG; ++u) r[u] = rp[u][k * D];
        double gv = rg.direct[k] + RHS::gather(r);
        if constexpr (tile4_me_first<WS, DC>()) gv = fma(two_cme, rg.yv[k], gv);          // (tile4_rows left w (x - y) there)
        else {
            double diff = rg.xown[k] - rg.yv[k];
            if constexpr (WS == 2) diff = ((rg.has >> k) & 1u) ? diff : 0.0;
            if constexpr (W_SCALAR) gv = fma(c2, diff, gv);
            else gv = fma(two_cme, rg.wv[k] * diff, gv);
        }
        // rows >= N of an edge tile: every term above is an exact zero (inputs zeroed in tile4_rows,
        // observation loads return 0 there)
        gvv[k] = gv;
        if constexpr (LSQ) {
            gtd = fma(gv, rg.dval[k], gtd);
            gn2 = fma(gv, gv, gn2);
            gmax = __builtin_fmax(gmax, __builtin_fabs(gv));
        }
    }
    if constexpr (LSQ) { acc.v[EP_GTD] += gtd; acc.v[EP_GN2] += gn2; acc.v[EP_GMAX] = __builtin_fmax(acc.v[EP_GMAX], gmax); }
}

}  // namespace va
