// tests/cpu_emul/va_emul.cpp -- TEST INFRASTRUCTURE.
// Serial driver for the __host__ __device__ logic in varanneal_amd/csrc/va_core.h:
// the same tile phases / line-search / coefficient-space two-loop code the HIP
// kernels execute, run phase by phase with an emulated thread loop.  Lets the CPU
// test-suite cover halo indexing, edge tiles and the per-seed state machine without
// a GPU.  Never linked into libvaranneal_amd.so and never used as a fallback.
#include <cstdio>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/varanneal_amd.h"
#include "../../varanneal_amd/csrc/va_core.h"
#include "../../varanneal_amd/csrc/va_tile2.h"
#include "../../varanneal_amd/csrc/va_tile3.h"
#include "../../varanneal_amd/csrc/va_tile4.h"
#ifdef VA_USER_RHS_HEADER
#include VA_USER_RHS_HEADER      // generated RhsUser (varanneal_amd/codegen.py)
#endif

using namespace va;

namespace {

struct Emul {
    Dims dm;
    Geo4 g4;
    ProblemPtrs pp;
    std::vector<int> lmap, pidx, lidx;
    std::vector<double> Y, rm, rf0, P, tm, stim;
    int rhs;
};

int setup(const va_problem_desc *d, int T, Emul &E)
{
    Dims &m = E.dm;
    m.D = d->D; m.N = d->N_model; m.ND = m.D * m.N; m.L = d->L; m.N_data = d->N_data;
    m.nskip = d->merr_nskip; m.NP = d->NP; m.NPest = d->NPest; m.B = d->batch;
    m.m = d->lbfgs_m > 0 ? d->lbfgs_m : 10; m.disc = d->disc;
    m.tdp = d->p_time_dependent ? 1 : 0; m.NPt = d->NP; m.NPe = d->NPest;
    if (m.tdp) { m.ND = m.N * (m.D + m.NPe); m.NP = 0; m.NPest = 0; }
    m.ld = ((m.ND + m.NPest + 15) / 16) * 16;
    if (m.disc == DISC_SH && (T & 1)) ++T;
    m.ghost = 2;
#ifdef VA_USER_GHOST
    if (d->rhs >= VA_RHS_USER_BASE) m.ghost = RhsUserG::GHOST;
#endif
    m.emode = (d->eval_kernel >= 1 && d->eval_kernel <= 4) ? d->eval_kernel : (tile4_ok(m.D) ? 4 : 3);
    if (m.emode == 2) m.emode = 3;           // (the row-strided kernel of round 1 is gone)
    if (m.emode == 4 && !tile4_ok(m.D)) m.emode = 3;
    m.RY = tile2_RY(m.D); m.NT = tile2_threads(m.D); m.maxr = 16;
    if (m.emode == 4) {                      // wave-private column runs: T = 4 waves x RW runs x K rows
        const int rows1 = 4 * (64 / m.D);
        int K = (T + rows1 - 1) / rows1;
        K = K < 4 ? 4 : (K > 8 ? 8 : K);
        if (m.disc == DISC_SH && (K & 1)) ++K;
        const char *se = getenv("VA_EMUL_SUB");            // sub-tiles per wave (the device picks 1..3, va_capi.hip)
        const int SUB = se ? atoi(se) : 1;
        int ne = RhsL96s::NE;
#ifdef VA_USER_COL
        if (d->rhs >= VA_RHS_USER_BASE) ne = RhsUserCol::NE;
#endif
        E.g4 = m.disc == DISC_SH ? tile4_geo<3>(m.D, K, ne, SUB) : tile4_geo<2>(m.D, K, ne, SUB);
        if ((E.g4.XP + 63) / 64 > T4_NI_MAX || !tile4_magic_ok(E.g4)) return VA_EUNSUPPORTED;
        m.RY = 4 * (64 / m.D); m.NT = 256; m.maxr = K; T = E.g4.T;
    }
    if (m.emode == 3) {                      // column-run: T = RY*K, K in {4,6,8}
        m.RY = tile3_RY(m.D); m.NT = tile3_threads(m.D);
        int K = (T + m.RY - 1) / m.RY;
        K = K < 4 ? 4 : (K > 8 ? 8 : K);
        if (m.disc == DISC_SH && (K & 1)) ++K;   // Simpson-Hermite runs start on even rows
        m.maxr = K; T = m.RY * K;
    }
    m.T = T; m.ntiles = (m.N + T - 1) / T;
    m.nprow = m.ntiles;
    m.chunk = 1000; m.nchunks = (m.ld + m.chunk - 1) / m.chunk;
    m.dt = d->dt_model; m.cme = 1.0 / ((double)m.L * m.N_data); m.cfe = 1.0 / ((double)m.D * (m.N - 1));
    m.rm = d->rm; m.rf0 = d->rf0;
    E.lmap.assign(m.D, -1);
    for (int l = 0; l < m.L; ++l) E.lmap[d->Lidx[l]] = l;
    E.Y.assign(d->Y, d->Y + (size_t)m.N_data * m.L);
    if (d->rm_kind) E.rm.assign(d->rm_array, d->rm_array + (size_t)m.N_data * m.L * (d->rm_kind == 2 ? m.L : 1));
    E.lidx.assign(d->Lidx, d->Lidx + m.L);
    if (d->rf_kind) E.rf0.assign(d->rf0_array, d->rf0_array + (size_t)(m.N - 1) * m.D * (d->rf_kind == 2 ? m.D : 1));
    E.pidx.assign(d->Pidx, d->Pidx + m.NPe);
    E.P.assign(d->P, d->P + (size_t)m.B * (m.tdp ? (size_t)m.N * m.NPt : (size_t)m.NPt));
    E.pp.lmap = E.lmap.data(); E.pp.Y = E.Y.data();
    E.pp.rm_arr = d->rm_kind == 1 ? E.rm.data() : nullptr;
    E.pp.rm_full = d->rm_kind == 2 ? E.rm.data() : nullptr; E.pp.Lidx = E.lidx.data();
    E.pp.rf0_arr = d->rf_kind == 1 ? E.rf0.data() : nullptr;
    E.pp.rf0_full = d->rf_kind == 2 ? E.rf0.data() : nullptr;
    E.pp.Pidx = E.pidx.data(); E.pp.Pfull = E.P.data();
    E.pp.lo = E.pp.hi = nullptr; m.bounded = 0;
    if (d->t_model) E.tm.assign(d->t_model, d->t_model + m.N);
    if (d->n_stim > 0) E.stim.assign(d->stim, d->stim + (size_t)m.N * d->n_stim);
    E.pp.tmodel = d->t_model ? E.tm.data() : nullptr;
    E.pp.stim = d->n_stim > 0 ? E.stim.data() : nullptr; E.pp.nstim = d->n_stim;
    bool user_flat = d->rhs >= VA_RHS_USER_BASE;
#ifdef VA_USER_COL
    // a generated model with a column form runs the wave-private kernel's phases when asked for by name
    if (d->eval_kernel == 4 && m.emode == 4) user_flat = false;
#endif
#ifdef VA_USER_GHOST
    if (d->eval_kernel == 3 && m.emode == 3) user_flat = false;      // likewise the ghosted form / workgroup kernel
#endif
    if (user_flat || m.tdp || d->rm_kind == 2 || d->rf_kind == 2) m.emode = 1;
    E.rhs = d->rhs;
    if (m.disc == DISC_SH && (m.N % 2) == 0) return VA_EINVAL;
    return VA_OK;
}

// K1 for one seed: all tiles, emulated NT threads per tile; returns summed partials.
template <class RHS, int DISC>
void eval_seed(const Emul &E, int b, const double *x, const double *d, int use_d, double stp,
               double rf_scale, double *gt, double *ev)
{
    const Dims &dm = E.dm;
    const int NT = 48;                         // deliberately not a divisor of anything
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR;
    const int R = dm.T + HL + HR;
    std::vector<double> xs(R * dm.D), fs(R * dm.D), qs(R * dm.D), ps((size_t)R * (dm.NPt > 0 ? dm.NPt : 1));
    for (int k = 0; k < EP_BIG; ++k) ev[k] = 0.0;        // (flat form: any number of parameters up to RHS_BIG_NP)
    for (int tile = 0; tile < dm.ntiles; ++tile) {
        TileCtx c;
        c.ps = ps.data();
        c.n0 = tile * dm.T; c.R = R; c.use_d = use_d; c.stp = stp; c.c = 2.0 * rf_scale * dm.cfe;
        c.xs = xs.data(); c.fs = fs.data(); c.qs = qs.data();
        c.xg = x; c.dg = d; c.gtg = gt;
        c.tmodel = E.pp.tmodel; c.stim = E.pp.stim; c.nstim = E.pp.nstim;
        if (!dm.tdp) tile_params<RHS>(dm, E.pp, b, c);
        std::vector<ThreadAccT<EP_BIG>> acc(NT);
        for (auto &a : acc) a.clear();
        for (int t = 0; t < NT; ++t) tile_load<DISC>(dm, E.pp, c, t, NT);
        if (dm.tdp) for (int t = 0; t < NT; ++t) tile_load_p<DISC>(dm, E.pp, b, c, t, NT);
        for (int t = 0; t < NT; ++t) tile_f<RHS, DISC>(dm, c, t, NT);
        for (int t = 0; t < NT; ++t) tile_q<DISC>(dm, E.pp, c, acc[t], t, NT);
        if (E.pp.rf0_full) {
            for (int t = 0; t < NT; ++t) tile_qfull<DISC>(dm, E.pp, c, acc[t], t, NT);
            std::swap(c.qs, c.fs);
        }
        for (int t = 0; t < NT; ++t) tile_s<DISC>(dm, c, t, NT);
        for (int t = 0; t < NT; ++t) tile_g<RHS, DISC>(dm, E.pp, c, acc[t], t, NT);
        if (dm.tdp) for (int t = 0; t < NT; ++t) tile_gp<RHS, DISC>(dm, E.pp, c, acc[t], t, NT);
        for (int t = 0; t < NT; ++t)
            for (int k = 0; k < EP_BIG; ++k) {
                if (k == EP_GMAX) ev[k] = fmax(ev[k], acc[t].v[k]);
                else ev[k] += acc[t].v[k];
            }
    }
}

// K1, column-run variant (va_tile3.h): ghosted + padded LDS layout, split staging
template <class RHS, int DISC, int K>
void eval_seed3(const Emul &E, int b, const double *x, const double *d, int use_d, double stp,
                double rf_scale, double *gt, double *ev)
{
    const Dims &dm = E.dm;
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR, G = RHS::GHOST, NS = tile3_ns_runtime(K);
    const int D = dm.D, T = dm.T, RY = dm.RY, NTH = D * RY, NT = dm.NT;
    std::vector<double> xs(tile3_stage_elems(K, D, G, RY, HL + HR)), ss(tile3_s_elems(K, D, G, RY));
    for (int k = 0; k < EP_N; ++k) ev[k] = 0.0;
    for (int tile = 0; tile < dm.ntiles; ++tile) {
        std::vector<Tile3> th(NT);
        std::vector<T3Regs<K>> rg(NT);
        std::vector<ThreadAcc> acc(NT);
        for (int t = 0; t < NT; ++t) {
            Tile3 &c = th[t];
            c.n0 = tile * T; c.ty = t / D; c.tx = t % D; c.r0 = c.n0 + c.ty * K; c.use_d = use_d;
            c.l = E.lmap[c.tx];
            c.stp = stp; c.c = 2.0 * rf_scale * dm.cfe;
            c.xs = xs.data(); c.ss = ss.data();
            c.xg = x; c.dg = d; c.gtg = gt;
            Tile2 tmp; tmp.xg = x; tmp.dg = d; tmp.use_d = use_d; tmp.stp = stp;
            tile2_params<RHS>(dm, E.pp, b, tmp);          // (only RHS::NP is used)
            for (int k = 0; k < RHS_MAX_NP; ++k) c.p[k] = tmp.p[k];
            acc[t].clear();
        }
        const bool edge = (tile * T - HL < 0) || (tile * T + T + HR > dm.N);
        for (int t = 0; t < NT; ++t) {
            if ((D & 1) == 0) {
                double xr[NS][2];
                if (edge) tile3_stage_load<DISC, K, 0, true, NS>(dm, th[t].n0, x, t, NT, xr);
                else tile3_stage_load<DISC, K, 0, false, NS>(dm, th[t].n0, x, t, NT, xr);
                if (edge) {
                    if (use_d) tile3_stage_store<RHS, DISC, K, 0, true, true, NS>(dm, th[t], t, NT, xr);
                    else tile3_stage_store<RHS, DISC, K, 0, true, false, NS>(dm, th[t], t, NT, xr);
                } else {
                    if (use_d) tile3_stage_store<RHS, DISC, K, 0, false, true, NS>(dm, th[t], t, NT, xr);
                    else tile3_stage_store<RHS, DISC, K, 0, false, false, NS>(dm, th[t], t, NT, xr);
                }
            } else {
                if (use_d) tile3_stage_odd<RHS, DISC, K, 0, true, true>(dm, th[t], t, NT);
                else tile3_stage_odd<RHS, DISC, K, 0, true, false>(dm, th[t], t, NT);
            }
        }
        for (int t = 0; t < NTH; ++t) tile3_obs<K>(dm, E.pp, th[t], rg[t]);
        for (int t = 0; t < NTH; ++t) {
            if (edge) tile3_rows<RHS, DISC, K, true, 0>(dm, E.pp, th[t], rg[t], acc[t]);
            else tile3_rows<RHS, DISC, K, false, 0>(dm, E.pp, th[t], rg[t], acc[t]);
        }
        for (int t = 0; t < NTH; ++t) {
            if (edge) tile3_grad<RHS, DISC, K, true, 0>(dm, th[t], rg[t], acc[t]);
            else tile3_grad<RHS, DISC, K, false, 0>(dm, th[t], rg[t], acc[t]);
        }
        for (int t = 0; t < NTH; ++t)
            for (int k = 0; k < EP_N; ++k) {
                if (k == EP_GMAX) ev[k] = fmax(ev[k], acc[t].v[k]);
                else ev[k] += acc[t].v[k];
            }
    }
}

// K1, wave-private column runs (va_tile4.h): every wave stages its own image through the piece
// map the direct-to-LDS loads use (bounds-checked here), then rows / scatter / gather
// DC, WS: the instantiation the device runs (va_kernels.hip: eval4_d) -- D = 20 as a compile-time constant (weight arrays:
// RF weights parked in the product arrays, measurement terms formed first), WS = 0 weight arrays / 1 scalar weights /
// 2 scalar weights with data at every nskip-th row (row mask)
template <class RHS, int DISC, int K, int DC, int WS>
void eval_seed4_v(const Emul &E, int b, const double *x, const double *d, int use_d, double stp,
                  double rf_scale, double *gt, double *ev)
{
    const Dims &dm = E.dm;
    const Geo4 &g = E.g4;
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR, NE = RHS::NE;
    const int D = dm.D;
    // the device keeps zero-filled guards around x and d (va_capi.hip: alloc_solver_state)
    const long guard = (((long)(dm.T + 8) * D + 15) / 16) * 16;
    std::vector<double> xg(guard + dm.ld + guard, 0.0), dgv(guard + dm.ld + guard, 0.0);
    memcpy(&xg[guard], x, sizeof(double) * (dm.ND + dm.NPest));
    if (use_d) memcpy(&dgv[guard], d, sizeof(double) * (dm.ND + dm.NPest));
    // poison what lies next to the path: the kernel must mask rows that do not exist
    for (long i = 0; i < guard; ++i) { xg[i] = 1e300; xg[guard + dm.ld + i] = -1e300; }
    for (long i = dm.ND + dm.NPest; i < dm.ld; ++i) xg[guard + i] = 3e299;
    std::vector<double> xs(g.XW), r2(g.R2);
    for (int k = 0; k < EP_N; ++k) ev[k] = 0.0;
    for (int tile = 0; tile < dm.ntiles; ++tile)
        for (int ws_ = 0; ws_ < g.NW * g.SUB; ++ws_) {    // (wave, sub-tile): a wave's sub-tiles are consecutive
            const int n0w = tile * g.T + ws_ * g.RW * K;
            const long src0 = guard + (long)(n0w - HL) * D;
            for (int q = 0; q < ((g.XP + 63) / 64) * 64; ++q) {
                const int sp = tile4_src_piece(g, q);
                if (sp < 0) continue;
                const long si = src0 + 2L * sp;
                if (si < 0 || si + 1 >= (long)xg.size() || 2 * q + 1 >= g.XW) { fprintf(stderr, "emul: image piece %d out of range\n", q); abort(); }      // a fault on the device
                xs[2 * q] = xg[si]; xs[2 * q + 1] = xg[si + 1];
                if (use_d) { xs[2 * q] = trial(xg[si], stp, dgv[si]); xs[2 * q + 1] = trial(xg[si + 1], stp, dgv[si + 1]); }
            }
            const bool edge = (n0w - HL < 0) || (n0w + g.RW * K + HR > dm.N);
            const int NL = g.RW * D;
            std::vector<Tile4> th(NL);
            std::vector<T4Regs<K, NE>> rg(NL);
            std::vector<ThreadAcc> acc(NL);
            for (int l = 0; l < NL; ++l) {
                Tile4 &c = th[l];
                c.n0w = n0w; c.a = l / D; c.tx = l % D; c.r0 = n0w + c.a * K; c.use_d = use_d;
                c.l = E.lmap[c.tx]; c.c = 2.0 * rf_scale * dm.cfe;
                c.wobs = c.l >= 0 ? dm.rm : 0.0;
                c.xs = xs.data(); c.es = r2.data(); c.gtg = gt;
                Tile2 tmp; tmp.xg = x; tmp.dg = d; tmp.use_d = use_d; tmp.stp = stp;
                tile2_params<RHS>(dm, E.pp, b, tmp);           // (only RHS::NP is used)
                for (int k = 0; k < RHS_MAX_NP; ++k) c.p[k] = tmp.p[k];
                acc[l].clear();
                tile4_obs<K, NE>(dm, E.pp, c, rg[l]);
                if (WS == 2) {
                    // the row-mask variant: observations as k_eval4 loads them (0 where there is none) and one bit per row with data
                    unsigned has = 0u;
                    for (int k = 0; k < K; ++k) {
                        const int m = c.r0 + k, nd = m / dm.nskip;
                        const bool h = c.l >= 0 && nd * dm.nskip == m && nd < dm.N_data && m < dm.N;
                        rg[l].yv[k] = h ? E.pp.Y[(size_t)nd * dm.L + c.l] : 0.0;
                        has |= h ? (1u << k) : 0u;
                    }
                    rg[l].has = has;
                }
                for (int k = 0; k < K; ++k) {
                    const long gi = (long)(c.r0 + k) * D + c.tx;
                    rg[l].dval[k] = (use_d && c.r0 + k < dm.N) ? d[gi] : 0.0;
                }
            }
            if (WS == 0 && E.pp.rf0_arr) {
                // RF0 arrays: every lane parks its weights in its own slots of the product arrays (k_eval4 does so after
                // the trial point is formed)
                if constexpr (tile4_rfw_in_lds<K, NE, HL, DC>()) {
                    for (int l = 0; l < NL; ++l) {
                        double wq[K + HL];
                        tile4_rfw_load<K, HL>(dm, E.pp, th[l], D, wq);
                        tile4_rfw_store<K, HL>(g, th[l], D, wq);
                    }
                }
            }
            for (int l = 0; l < NL; ++l) {
                if (edge) tile4_rows<RHS, DISC, K, true, DC, WS>(dm, E.pp, g, th[l], rg[l], acc[l]);
                else tile4_rows<RHS, DISC, K, false, DC, WS>(dm, E.pp, g, th[l], rg[l], acc[l]);
            }
            for (int l = 0; l < NL; ++l) {
                double gvv[K];
                if (edge) tile4_grad<RHS, DISC, K, true, DC, WS, true>(dm, g, th[l], rg[l], acc[l], gvv);
                else tile4_grad<RHS, DISC, K, false, DC, WS, true>(dm, g, th[l], rg[l], acc[l], gvv);
                for (int k = 0; k < K; ++k) {
                    if (th[l].r0 + k < dm.N) gt[(long)(th[l].r0 + k) * D + th[l].tx] = gvv[k];
                    else if (gvv[k] != 0.0) { fprintf(stderr, "emul: row %d col %d beyond N has gradient %g\n", th[l].r0 + k, th[l].tx, gvv[k]); abort(); }          // rows that do not exist must come out as exact zeros
                }
            }
            for (int l = 0; l < NL; ++l)
                for (int k = 0; k < EP_N; ++k) {
                    if (k == EP_GMAX) ev[k] = fmax(ev[k], acc[l].v[k]);
                    else ev[k] += acc[l].v[k];
                }
        }
}

template <class RHS, int DISC, int K>
void eval_seed4(const Emul &E, int b, const double *x, const double *d, int use_d, double stp,
                double rf_scale, double *gt, double *ev)
{
    // the instantiation the device picks (va_kernels.hip: eval4_d; generated modules: va_user_rhs.hip)
    const bool sw = !E.pp.rm_arr && !E.pp.rf0_arr, ws = sw && E.dm.nskip == 1;
    const bool builtin = E.rhs < VA_RHS_USER_BASE;
    if (E.dm.D == 20) {
        if (ws) eval_seed4_v<RHS, DISC, K, 20, 1>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (sw && builtin) eval_seed4_v<RHS, DISC, K, 20, 2>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else eval_seed4_v<RHS, DISC, K, 20, 0>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
    } else if (ws) eval_seed4_v<RHS, DISC, K, 0, 1>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
    else eval_seed4_v<RHS, DISC, K, 0, 0>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
}

#if defined(VA_USER_RHS_HEADER) && defined(VA_USER_COL)
// a generated module compiles k_eval4 with ITS D as the constant, scalar weights or weight arrays (va_user_rhs.hip)
template <int DISC, int K>
void eval_seed4_user(const Emul &E, int b, const double *x, const double *d, int use_d, double stp,
                     double rf_scale, double *gt, double *ev)
{
    const bool ws = !E.pp.rm_arr && !E.pp.rf0_arr && E.dm.nskip == 1;
    if (ws) eval_seed4_v<RhsUserCol, DISC, K, RhsUserCol::D, 1>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
    else eval_seed4_v<RhsUserCol, DISC, K, RhsUserCol::D, 0>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
}
#endif

template <int DISC>
void eval_seed_rhs(const Emul &E, int b, const double *x, const double *d, int use_d, double stp,
                   double rf_scale, double *gt, double *ev)
{
#ifdef VA_USER_RHS_HEADER
#ifdef VA_USER_COL
    if (E.rhs >= VA_RHS_USER_BASE && E.dm.emode == 4) {
        if (E.dm.maxr == 4) eval_seed4_user<DISC, 4>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 5) eval_seed4_user<DISC, 5>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 6) eval_seed4_user<DISC, 6>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 7) eval_seed4_user<DISC, 7>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else eval_seed4_user<DISC, 8>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        return;
    }
#endif
#ifdef VA_USER_GHOST
    if (E.rhs >= VA_RHS_USER_BASE && E.dm.emode == 3) {
        if (E.dm.maxr == 4) eval_seed3<RhsUserG, DISC, 4>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 5) eval_seed3<RhsUserG, DISC, 5>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 6) eval_seed3<RhsUserG, DISC, 6>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 7) eval_seed3<RhsUserG, DISC, 7>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else eval_seed3<RhsUserG, DISC, 8>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        return;
    }
#endif
    if (E.rhs >= VA_RHS_USER_BASE) { eval_seed<RhsUser, DISC>(E, b, x, d, use_d, stp, rf_scale, gt, ev); return; }
#endif
    if (E.dm.emode == 4) {
        if (E.dm.maxr == 4) eval_seed4<RhsL96s, DISC, 4>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 5) eval_seed4<RhsL96s, DISC, 5>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 6) eval_seed4<RhsL96s, DISC, 6>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 7) eval_seed4<RhsL96s, DISC, 7>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 12) eval_seed4<RhsL96s, DISC, 12>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else eval_seed4<RhsL96s, DISC, 8>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
    } else if (E.dm.emode == 3) {
        if (E.dm.maxr == 4) eval_seed3<RhsL96g, DISC, 4>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 5) eval_seed3<RhsL96g, DISC, 5>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 6) eval_seed3<RhsL96g, DISC, 6>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else if (E.dm.maxr == 7) eval_seed3<RhsL96g, DISC, 7>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
        else eval_seed3<RhsL96g, DISC, 8>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
    } else eval_seed<RhsL96, DISC>(E, b, x, d, use_d, stp, rf_scale, gt, ev);
}

void eval_dispatch(const Emul &E, int b, const double *x, const double *d, int use_d, double stp,
                   double rf_scale, double *gt, double *ev)
{
    switch (E.dm.disc) {
    case DISC_EULER: eval_seed_rhs<DISC_EULER>(E, b, x, d, use_d, stp, rf_scale, gt, ev); break;
    case DISC_TRAPEZOID: eval_seed_rhs<DISC_TRAPEZOID>(E, b, x, d, use_d, stp, rf_scale, gt, ev); break;
    case DISC_SH: eval_seed_rhs<DISC_SH>(E, b, x, d, use_d, stp, rf_scale, gt, ev); break;
    default: eval_seed_rhs<DISC_FWDMAP>(E, b, x, d, use_d, stp, rf_scale, gt, ev); break;
    }
}

// parameter tail of the gradient and its contributions to the line-search sums
void finish_tail(const Emul &E, const double *d, int use_d, double *gt, double *ev)
{
    const Dims &dm = E.dm;
    for (int k = 0; k < dm.NPest; ++k) {
        double g = ev[EP_GP + E.pidx[k]];
        gt[dm.ND + k] = g;
        if (use_d) ev[EP_GTD] += g * d[dm.ND + k];
        ev[EP_GN2] += g * g;
        ev[EP_GMAX] = fmax(ev[EP_GMAX], fabs(g));
    }
}

}  // namespace

extern "C" int emul_action_grad(const va_problem_desc *desc, int T, const double *XP, double rf_scale,
                                double *A, double *me, double *fe, double *grad)
{
    Emul E;
    int rc = setup(desc, T, E);
    if (rc) return rc;
    const Dims &dm = E.dm;
    const int nv = dm.ND + dm.NPest;
    std::vector<double> gt(dm.ld);
    for (int b = 0; b < dm.B; ++b) {
        double ev[EP_BIG] = {0.0};
        eval_dispatch(E, b, XP + (size_t)b * nv, nullptr, 0, 0.0, rf_scale, gt.data(), ev);
        finish_tail(E, nullptr, 0, gt.data(), ev);
        me[b] = ev[EP_ME] * dm.cme; fe[b] = ev[EP_FE] * dm.cfe * rf_scale; A[b] = me[b] + fe[b];
        if (grad) memcpy(grad + (size_t)b * nv, gt.data(), sizeof(double) * nv);
    }
    return VA_OK;
}

extern "C" int emul_anneal(const va_problem_desc *desc, int T, double *XP, const double *rf_scale,
                           int nbeta, const va_lbfgs_opts *o_, double *ame, double *pest,
                           int *status, int *nit, long long *nfev, double *minpaths,
                           long long *cycles_out)
{
    Emul E;
    int rc = setup(desc, T, E);
    if (rc) return rc;
    const Dims &dm = E.dm;
    const int nv = dm.ND + dm.NPest, ld = dm.ld, M = dm.m, wide = dm.ND + dm.NP;
    Opts o; o.m = o_->maxcor < M ? o_->maxcor : M; o.maxiter = o_->maxiter; o.maxls = o_->maxls;
    o.maxfun = o_->maxfun; o.ftol = o_->ftol; o.gtol = o_->gtol;
    const int B = dm.B;
    std::vector<double> x((size_t)B * ld, 0.0), g((size_t)B * ld, 0.0), gt((size_t)B * ld, 0.0),
        d((size_t)B * ld, 0.0), S((size_t)B * M * ld, 0.0), Yh((size_t)B * M * ld, 0.0);
    std::vector<SeedState> st(B);
    std::vector<double> dirp((size_t)B * DP_N, 0.0);
    int n_active = B;
    long long cycles = 0;
    for (int b = 0; b < B; ++b) {
        memcpy(&x[(size_t)b * ld], XP + (size_t)b * nv, sizeof(double) * nv);
        memset(&st[b], 0, sizeof(SeedState));
        st[b].phase = PH_START; st[b].beta_idx = 0; st[b].rf_scale = rf_scale[0]; st[b].theta = 1.0;
    }
    while (n_active > 0) {
        ++cycles;
        for (int b = 0; b < B; ++b) {
            SeedState &s = st[b];
            if (s.phase != PH_START && s.phase != PH_LS) continue;
            double *xb = &x[(size_t)b * ld], *gb = &g[(size_t)b * ld], *gtb = &gt[(size_t)b * ld],
                   *db = &d[(size_t)b * ld];
            // K1
            double ev[EP_BIG] = {0.0};
            const int use_d = s.phase == PH_LS;
            eval_dispatch(E, b, xb, db, use_d, s.stp, s.rf_scale, gtb, ev);
            // K2
            finish_tail(E, db, use_d, gtb, ev);
            SeedResults r;
            r.ame = ame + (size_t)b * nbeta * 3;
            r.pest = pest ? pest + (size_t)b * nbeta * dm.NPest : nullptr;
            r.status = status + (size_t)b * nbeta; r.nit = nit + (size_t)b * nbeta;
            r.nfev = nfev + (size_t)b * nbeta;
            int dec = 0;
            ls_step(s, ev, &dirp[(size_t)b * DP_N], o, rf_scale, nbeta, r, &dec, dm.cme, dm.cfe);
            n_active -= dec;
            // K3
            double up[UP_N];
            for (int k = 0; k < UP_N; ++k) up[k] = 0.0;
            if (s.upd || s.dir) {
                const bool hist = s.upd & UPD_HIST;
                double *Sn = hist ? &S[((size_t)b * M + s.slot) * ld] : nullptr;
                double *Yn = hist ? &Yh[((size_t)b * M + s.slot) * ld] : nullptr;
                for (int i = 0; i < ld; ++i) {
                    const double dv = db[i], gv = gb[i], gtv = gtb[i];
                    double xn = xb[i];
                    if (s.upd & UPD_X) { xn = trial(xn, s.stp_upd, dv); xb[i] = xn; }
                    double sv = 0.0, yv = 0.0;
                    if (hist) { sv = s.stp_upd * dv; yv = gtv - gv; Sn[i] = sv; Yn[i] = yv; }
                    if (s.upd & UPD_G) gb[i] = gtv;
                    if (s.dir) {
                        up[UP_YGT] += yv * gtv; up[UP_SGT] += sv * gtv; up[UP_YY] += yv * yv;
                        up[UP_SY] += sv * yv; up[UP_GTGT] += gtv * gtv;
                        for (int j = 0; j < s.nold; ++j) {
                            const double sj = S[((size_t)b * M + s.order[j]) * ld + i],
                                         yj = Yh[((size_t)b * M + s.order[j]) * ld + i];
                            up[UP_OLD + 4 * j + 0] += sj * gtv; up[UP_OLD + 4 * j + 1] += yj * gtv;
                            up[UP_OLD + 4 * j + 2] += sj * yv; up[UP_OLD + 4 * j + 3] += yj * yv;
                        }
                    }
                }
                if (s.upd & UPD_STORE) {
                    const int k = s.store_idx;
                    if (minpaths) {
                        double *mp = minpaths + ((size_t)b * nbeta + k) * wide;
                        memcpy(mp, xb, sizeof(double) * dm.ND);
                        for (int j = 0; j < dm.NP; ++j) mp[dm.ND + j] = E.P[(size_t)b * dm.NP + j];
                        for (int j = 0; j < dm.NPest; ++j) mp[dm.ND + E.pidx[j]] = xb[dm.ND + j];
                    }
                    if (r.pest) for (int j = 0; j < dm.NPest; ++j) r.pest[k * dm.NPest + j] = xb[dm.ND + j];
                }
            }
            if (s.dir) {
                // K4
                direction_coeffs(s, up, o);
                // K5
                double gd = 0.0, dd = 0.0;
                for (int i = 0; i < ld; ++i) {
                    double v = s.cg * gb[i];
                    for (int j = 0; j < s.col; ++j) {
                        const int sj = s.order[j];
                        v += s.cY[sj] * Yh[((size_t)b * M + sj) * ld + i];
                        v += s.cS[sj] * S[((size_t)b * M + sj) * ld + i];
                    }
                    db[i] = v; gd += gb[i] * v; dd += v * v;
                }
                dirp[(size_t)b * DP_N + DP_GD] = gd; dirp[(size_t)b * DP_N + DP_DD] = dd;
            }
        }
    }
    for (int b = 0; b < B; ++b) memcpy(XP + (size_t)b * nv, &x[(size_t)b * ld], sizeof(double) * nv);
    if (cycles_out) *cycles_out = cycles;
    return VA_OK;
}
