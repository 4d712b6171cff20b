// va_eval5.h -- the STREAMING evaluation kernel k_eval5<RHS, DISC, D, NSLOT, LSRUN> for wide states
// (BASELINE config 4: Lorenz-96 D = 200, N = 5000) and its launcher.  Geometry: va_tile5.h.
//
// Reference arithmetic (va_core.h header): fe_gaussian va_ode.py:160-234 with disc_trapezoid :358-380,
// disc_euler :341-356, disc_forwardmap :439-454 or disc_SimpsonHermite :404-437 (an interval of two rows per
// step instead of a row, see the walk), me_gaussian :138-158; gradient by the hand-coded adjoint of
// the stencil.  One wave = one strip of state columns of one seed over one segment of time rows:
//
//   ring      NSLOT slots of two staged rows each; slot k+NSLOT-1 is requested (global_load_lds_dwordx4:
//             x rows, the observation rows, at a line-search point the rows of d) before slot k is touched, and
//             a counted s_waitcnt vmcnt(N) retires exactly slot k.  No load in the walk returns to a register,
//             so nothing else ever waits on the vector-memory queue;
//   step(j)   reads row j (own column + the stencil's neighbours, ds_read_b64 at immediate offsets), evaluates
//             f_j, the residual of the interval (j-1, j), its adjoint weight q_{j-1}, then finishes row j-1:
//             direct term, s_{j-1}, the scatter products (published to the wave's LDS arrays, gathered back from
//             the neighbour lanes -- LDS is in order within a wave), the measurement term, the gradient store
//             (16 B per lane pair, joined by a DPP quad permutation) and the partial sums.
//
// Every row of x is read once per strip (+ the strip's ghost columns), every gradient row written once; the
// only recomputation is one row per segment (the row before it) and 3 ghost columns per 50.
#pragma once
#include "va_eval4.h"
#include "va_tile5.h"

namespace va {

#define VA_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// value of lane + O (|O| <= 2) through DPP whole-wave shifts (wave_shl:1 / wave_shr:1): lanes whose source lies
// outside the wave get 0 -- they are ghost lanes
template <int O>
__device__ __forceinline__ double lane_from(double x)
{
    static_assert(O >= -2 && O <= 2, "reach of the DPP exchange");
    int lo = __double2loint(x), hi = __double2hiint(x);
    constexpr int CTRL = O > 0 ? 0x130 : 0x138;               // wave_shl:1 : wave_shr:1
#pragma unroll
    for (int k = 0; k < (O > 0 ? O : -O); ++k) {
        lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    }
    return __hiloint2double(hi, lo);
}
template <class RHS> constexpr bool t5_dpp_ok() { return t5_gl<RHS>() <= 2 && t5_gr<RHS>() <= 2; }
template <class RHS, int U> struct T5Gather {
    static __device__ __forceinline__ void run(const double *e, double *r)
    {
        r[U] = lane_from<RHS::g_off(U)>(e[RHS::g_e(U)]);
        T5Gather<RHS, U - 1>::run(e, r);
    }
};
template <class RHS> struct T5Gather<RHS, -1> { static __device__ __forceinline__ void run(const double *, double *) {} };

// XDPP: the scatter products change lanes through DPP shifts instead of the wave's LDS arrays
template <class RHS, int DISC, int DC, int NSLOT, bool LSRUN, bool XDPP, bool WARR>
__global__ __launch_bounds__(64 * T5_WPG_MAX, 4) void k_eval5(const Dev dv)
{
    static_assert(!RHS::USES_T && RHS::NSTIM == 0, "autonomous right-hand sides only");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Dims &dm = dv.dm;
    const int nwork = dm.B * dm.ntiles;
    const int w = xcd_swizzle(blockIdx.x, nwork);
    if (w >= nwork) return;
    const int b = w / dm.ntiles, tile = w - b * dm.ntiles;

    constexpr int NE = RHS::NE, NB = RHS::NB, NG = RHS::NG, NV = EP_GP + RHS::NP;
    constexpr int P = NSLOT - 1;                              // slots requested ahead of the one in use
    // column geometry: constant when D is
    constexpr Geo5 gcc = tile5_cols_rhs<RHS>(DC > 0 ? DC : 128);
    const Geo5 gc = DC > 0 ? gcc : dv.g5;
    const int D = DC > 0 ? DC : dv.g5.D;
    const int PR = gc.PR, GL = gc.GL, XL = gc.XL, PW = gc.PW, PL = gc.PL;
    const int NSG = gc.NSG, WPG = gc.WPG;
    const int SEGL = dv.g5.SEGL, YPMAX = dv.g5.YPMAX;
    const int SLOTX = 4 * PR, SLOTY = 4 * YPMAX;              // doubles per ring slot
    const int sg = tile / NSG, grp = tile - sg * NSG;

    const auto *st = as_const(static_cast<const SeedHot *>(dv.st + b));
    const int phase = st->phase;
    if (phase != PH_START && phase != PH_LS) return;
    const bool use_d = LSRUN && phase == PH_LS;
    const double stp = st->stp;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = grp * WPG + wave;
    const bool wave_on = strip < gc.NS;                       // (the last workgroup of a row may have idle waves)
    const int WAVE = tile5_wave_doubles(dv.g5, NSLOT, LSRUN, WARR);
    double *xring = smem + wave * WAVE;
    double *dring = xring + NSLOT * SLOTX;
    double *yring = dring + (LSRUN ? NSLOT * SLOTX : 0);
    double *wfring = yring + NSLOT * SLOTY;                   // WARR: the model-error weights of the staged rows (image like x)
    double *wmring = wfring + (WARR ? NSLOT * SLOTX : 0);     // WARR: the measurement weights (image like the observations)
    double *prod = wmring + (WARR ? NSLOT * SLOTY : 0);
    double *outb = prod + (XDPP ? 0 : NE * PW);                          // [2 rows][64 lanes]: the gradient rows of a slot on their way out
    double *strip_red = xring;                                // after the walk

    ThreadAcc acc;
    acc.clear();
    if (wave_on) {
        const int N = dm.N, L = dv.g5.LY;                     // (L: the row pitch of Y / RM on the device, even)
        const int c0 = tile5_c0(D, gc.NS, strip);
        const int cws = tile5_c0(D, gc.NS, strip + 1) - c0;
        const int n0 = sg * SEGL, n1 = (n0 + SEGL) < N ? (n0 + SEGL) : N;
        // first and one-past-last row of the stream.  One-step rules: the row before the segment .. the row
        // after it.  Simpson-Hermite walks interval by interval (rows m, m+1, m+2, m even): the stream starts one
        // interval early, and one more row in front (never used) keeps the interval's rows (m+1, m+2) in ONE slot
        const int rs = DISC == DISC_SH ? (n0 > 0 ? n0 - 3 : -1) : (n0 > 0 ? n0 - 1 : 0), re = n1 < N ? n1 + 1 : N;
        const int SR = re - rs;                               // rows of the stream (>= 2; even for Simpson-Hermite)
        const int nslots = (SR + 1) >> 1;
        const double *xg = dv.x + (size_t)b * dm.ld;
        const double *dg = dv.d + (size_t)b * dm.ld;

        // ---- per-lane constants
        // staging: lane -> (row of the slot, piece of the image row); the image is the cyclic column window
        // [c0 - XL, c0 - XL + 2 PR)
        const int rr = lane >= PR ? 1 : 0, pc = lane - rr * PR;
        const bool dma_on = lane < 2 * PR && 2 * pc < XL + cws + ((t5_gr<RHS>() + t5_xr<RHS>() + 1) & ~1);     // (a narrower strip stages fewer pieces)
        const unsigned xoff = (unsigned)(rr * D + t5_wrap(c0 - XL + 2 * pc, D)) * 8u;
        const int l_start = as_const(dv.ystrip)[2 * strip], YP = as_const(dv.ystrip)[2 * strip + 1];
        const int yrr = lane >= YP ? 1 : 0, ypc = lane - yrr * YP;
        const bool ydma_on = lane < 2 * YP;
        const unsigned yoff = (unsigned)(yrr * L + l_start + 2 * ypc) * 8u;
        // arithmetic: lane -> column c0 - GL + lane
        const int col = t5_wrap(c0 - GL + lane, D);
        const bool own = lane >= GL && lane < GL + cws;
        const int lm = dv.pp.lmap[col];
        const bool obs = own && lm >= 0;
        const double wobs = obs ? dm.rm : 0.0;
        const int xlane = XL - GL + lane;                     // own column inside a staged row
        const int ylane = obs ? lm - l_start : 0;
        // gradient rows leave two at a time, 16 bytes per lane, lanes packed: lane -> (row of the pair, column pair)
        const int hp = cws >> 1;                              // column pairs of the strip
        const int srow = lane >= hp ? 1 : 0, spc = lane - srow * hp;
        const int gvoff2 = lane < 2 * hp ? (srow * D + c0 + 2 * spc) * 8 : 0x7ffffff0;     // both rows of a pair
        const int gvoff1 = lane < hp ? (c0 + 2 * lane) * 8 : 0x7ffffff0;                    // a single row
        const double *outrd = outb + srow * 64 + GL + 2 * spc;
        const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(dv.gt + (size_t)b * dm.ld), 0, (int)(sizeof(double) * N * D), 0x00020000);

        double p[RHS::NP > 0 ? RHS::NP : 1];
        {   // parameters (estimated ones from the trial point), all uniform
#pragma unroll
            for (int k = 0; k < RHS::NP; ++k) p[k] = as_const(dv.pp.Pfull)[(size_t)b * dm.NP + k];
            for (int k = 0; k < dm.NPest; ++k) {
                double v = as_const(xg)[dm.ND + k];
                if (use_d) v = trial(v, stp, as_const(dg)[dm.ND + k]);
                const int dst = as_const(dv.pp.Pidx)[k];
#pragma unroll
                for (int j = 0; j < RHS::NP; ++j) p[j] = (dst == j) ? v : p[j];
            }
        }
        const double dt = dm.dt, hdt = 0.5 * dm.dt;
        // scalar weights: q = cw r, the measurement term c2 (x - y).  Weight arrays / data every nskip-th row (WARR):
        // the row's own weights come through the ring, q = cw w r and the term is c2 w (x - y), w = 0 where no data
        const double cw = 2.0 * st->rf_scale * dm.cfe * (WARR ? 1.0 : dm.rf0);
        const double c2 = 2.0 * dm.cme * (WARR ? (obs ? 1.0 : 0.0) : wobs);
        const int nskip = WARR ? dm.nskip : 1;

        // ---- staging
        const char *xrow = reinterpret_cast<const char *>(xg + (ptrdiff_t)rs * D);   // uniform: first row of slot 0
        const char *drow = reinterpret_cast<const char *>(dg + (ptrdiff_t)rs * D);
        const char *yrow = reinterpret_cast<const char *>(dv.pp.Y + (ptrdiff_t)rs * L);
        const char *wfrow = WARR ? reinterpret_cast<const char *>(dv.pp.rf0_arr + (ptrdiff_t)rs * D) : nullptr;
        const size_t xstep = (size_t)2 * D * 8, ystep = (size_t)2 * L * 8;
        // WARR: the lane's data row of the next slot to be requested = its model row / nskip, kept as quotient and
        // remainder (slots are requested in order, two model rows apart)
        int ynd = 0, yrem = 0;
        if constexpr (WARR) {
            const int r0 = rs + yrr;
            if (r0 >= 0) { ynd = r0 / nskip; yrem = r0 - ynd * nskip; }
            else { ynd = -1; yrem = nskip - 1; }               // (the row in front of the path: one data row in front, never used)
        }
        const unsigned ycol = (unsigned)(l_start + 2 * ypc) * 8u;
        auto issue = [&](int k, int pos) {
            if (dma_on) __builtin_amdgcn_global_load_lds((glb_void_t *)(xrow + k * xstep + xoff), (lds_void_t *)(xring + pos * SLOTX), 16, 0, 0);
            if (LSRUN && use_d) {
                if (dma_on) __builtin_amdgcn_global_load_lds((glb_void_t *)(drow + k * xstep + xoff), (lds_void_t *)(dring + pos * SLOTX), 16, 0, 0);
            }
            if constexpr (WARR) {
                const ptrdiff_t yo = (ptrdiff_t)ynd * L * 8 + (ptrdiff_t)ycol;
                if (ydma_on) __builtin_amdgcn_global_load_lds((glb_void_t *)(reinterpret_cast<const char *>(dv.pp.Y) + yo), (lds_void_t *)(yring + pos * SLOTY), 16, 0, 0);
                if (dma_on) __builtin_amdgcn_global_load_lds((glb_void_t *)(wfrow + k * xstep + xoff), (lds_void_t *)(wfring + pos * SLOTX), 16, 0, 0);
                if (ydma_on) __builtin_amdgcn_global_load_lds((glb_void_t *)(reinterpret_cast<const char *>(dv.pp.rm_arr) + yo), (lds_void_t *)(wmring + pos * SLOTY), 16, 0, 0);
                yrem += 2;
                while (yrem >= nskip) { yrem -= nskip; ++ynd; }
            } else {
                if (ydma_on) __builtin_amdgcn_global_load_lds((glb_void_t *)(yrow + k * ystep + yoff), (lds_void_t *)(yring + pos * SLOTY), 16, 0, 0);
            }
        };
#pragma unroll
        for (int k = 0; k < P; ++k)
            if (k < nslots) issue(k, k);

        // ---- the walk
        double x0p = 0.0, fp = 0.0, qp = 0.0, yp = 0.0, dp = 0.0, wfp = 0.0, wmp = 0.0, xnp[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) xnp[k] = 0.0;
        double fe = 0.0, me = 0.0, gtd = 0.0, gn2 = 0.0, gmax = 0.0, gp[RHS::NP > 0 ? RHS::NP : 1];
#pragma unroll
        for (int k = 0; k < RHS::NP; ++k) gp[k] = 0.0;

        // finish a row: its direct term and s (the adjoint weight that multiplies df/dx of that row)
        // (WARR) the row about to be finished, modulo nskip: data exist where it is 0
        int mrem = 0;
        if constexpr (WARR) {
            const int first = DISC == DISC_SH ? rs + 1 : rs;
            mrem = first % nskip;
        }
        auto emit_row = [&](double x0, const double *xn, double yv, double dval, double wm, double direct, double s, int orow) {
            double e[NE], diag;
            RHS::scatter(col, s, x0, xn, p, 0.0, nullptr, e, diag);
            RHS::pgrad(col, s, x0, xn, p, 0.0, nullptr, gp);
            double r[NG];
            if constexpr (XDPP) T5Gather<RHS, NG - 1>::run(e, r);
            else {
#pragma unroll
                for (int u = 0; u < NE; ++u) prod[u * PW + PL + lane] = e[u];
                wave_sync_lds();
#pragma unroll
                for (int u = 0; u < NG; ++u) r[u] = VA_LDS_CVP(prod + RHS::g_e(u) * PW + PL + lane)[RHS::g_off(u)];
                wave_sync_lds();                              // (the next row overwrites the arrays)
            }
            const double diff = x0 - yv;
            double gv = (direct + diag) + RHS::gather(r);
            if constexpr (WARR) {
                const double w = mrem == 0 ? wm : 0.0;
                mrem = mrem + 1 == nskip ? 0 : mrem + 1;
                const double wd = w * diff;
                me = fma(wd, diff, me);
                gv = fma(c2, wd, gv);
            } else {
                me = fma(diff, diff, me);
                gv = fma(c2, diff, gv);
            }
            if constexpr (LSRUN) {
                gtd = fma(gv, dval, gtd);
                gn2 = fma(gv, gv, gn2);
                gmax = __builtin_fmax(gmax, __builtin_fabs(gv));
            }
            outb[orow * 64 + lane] = gv;
        };
        // one-step rules: finish the row of the state registers; q = adjoint weight of the residual that starts there
        auto emit = [&](double q, int orow) {
            double direct, s;
            if constexpr (DISC == DISC_TRAPEZOID) { direct = qp - q; s = -hdt * (qp + q); }
            else if constexpr (DISC == DISC_EULER) { direct = qp - q; s = -dt * q; }
            else { direct = qp; s = -q; }
            emit_row(x0p, xnp, yp, dp, wmp, direct, s, orow);
            qp = q;
        };
        // rows m0, m0 + 1 (or m0 alone) of the gradient: LDS -> 16-byte stores
        auto flush = [&](int m0, int voff) {
            wave_sync_lds();
            double a0, a1;
            ld2(outrd, a0, a1);
            wave_sync_lds();
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u v = {(unsigned)__double2loint(a0), (unsigned)__double2hiint(a0), (unsigned)__double2loint(a1), (unsigned)__double2hiint(a1)};
            if (dv.gaux) __builtin_amdgcn_raw_buffer_store_b128(v, grs, voff, m0 * D * 8, 16);
            else __builtin_amdgcn_raw_buffer_store_b128(v, grs, voff, m0 * D * 8, 0);
        };
        // one staged row in registers: own column, the stencil's neighbours, observation, own entry of d
        struct Row { double x0, xn[NB], yv, dval, wf, wm; };
        auto load_row = [&](int pos, int r01) {
            Row t;
            lds_cvp xr = VA_LDS_CVP(xring + pos * SLOTX + r01 * 2 * PR + xlane);
            t.x0 = xr[0];
#pragma unroll
            for (int k = 0; k < NB; ++k) t.xn[k] = xr[RHS::nb_off(k)];
            t.yv = VA_LDS_CVP(yring + pos * SLOTY + r01 * 2 * YP + ylane)[0];
            t.dval = 0.0;
            if (LSRUN && use_d) t.dval = VA_LDS_CVP(dring + pos * SLOTX + r01 * 2 * PR + xlane)[0];
            t.wf = 1.0; t.wm = 0.0;
            if constexpr (WARR) {
                t.wf = VA_LDS_CVP(wfring + pos * SLOTX + r01 * 2 * PR + xlane)[0];
                t.wm = obs ? VA_LDS_CVP(wmring + pos * SLOTY + r01 * 2 * YP + ylane)[0] : 0.0;
            }
            return t;
        };
        // row j enters: f_j, the residual of the interval (j-1, j), and row j-1 is finished; FIRST: only
        // load the state registers
        auto step = [&](const Row &t, bool first, int orow) {
            const double f = RHS::f(col, t.x0, t.xn, p, 0.0, nullptr);
            if (!first) {
                double r;
                if constexpr (DISC == DISC_TRAPEZOID) r = (t.x0 - x0p) - hdt * (fp + f);
                else if constexpr (DISC == DISC_EULER) r = (t.x0 - x0p) - dt * fp;
                else r = t.x0 - fp;
                if constexpr (WARR) {
                    const double wr = wfp * r;                // (the weight row of the residual that starts at the state row)
                    fe = fma(wr, r, fe);
                    emit(cw * wr, orow);
                } else {
                    fe = fma(r, r, fe);
                    emit(cw * r, orow);
                }
            }
            x0p = t.x0; fp = f; yp = t.yv; dp = t.dval; wfp = t.wf; wmp = t.wm;
#pragma unroll
            for (int k = 0; k < NB; ++k) xnp[k] = t.xn[k];
        };
        // retire slot k, request slot k + P, form the trial point in place
        bool drained = false;
        auto begin_slot = [&](int k, int pos, int ppos) {
            if (k + P > nslots) {
                if (!drained) { VA_WAIT_VM(0); drained = true; }
            } else if (LSRUN && use_d) {
                // per slot in flight: x, d, Y rows (+ the two weight images) + one gradient store
                if (k < P) VA_WAIT_VM((P - 1) * (WARR ? 5 : 3)); else VA_WAIT_VM((P - 1) * (WARR ? 6 : 4));
            } else {
                if (k < P) VA_WAIT_VM((P - 1) * (WARR ? 4 : 2)); else VA_WAIT_VM((P - 1) * (WARR ? 5 : 3));
            }
            __builtin_amdgcn_wave_barrier();
            if (k + P < nslots) issue(k + P, ppos);
            if (LSRUN && use_d) {
                if (dma_on) {
                    double a0, a1, d0, d1;
                    ld2(xring + pos * SLOTX + 2 * lane, a0, a1); ld2(dring + pos * SLOTX + 2 * lane, d0, d1);
                    st2(xring + pos * SLOTX + 2 * lane, trial(a0, stp, d0), trial(a1, stp, d1));
                }
                wave_sync_lds();
            }
        };

        if constexpr (DISC == DISC_SH) {
            // Simpson-Hermite: the state registers hold an even row m with the adjoint weights (q1p, q2p) of the
            // interval that ENDS there; rows (m+1, m+2) enter together, give the interval's two residuals
            //   r1 = x_{m+2} - x_m - (dt/3)(f_m + 4 f_{m+1} + f_{m+2})          (counted at row m)
            //   r2 = x_{m+1} - (x_m + x_{m+2})/2 - (dt/4)(f_m - f_{m+2})        (counted at row m+1)
            // and rows m and m+1 are finished (va_tile4.h tile4_rows has the same sums tile by tile)
            double q2p = 0.0;                                 // (qp is q1 of the interval behind)
            const double dt3 = dt / 3.0, dt4 = dt / 4.0;
            auto step_sh = [&](const Row &t0, const Row &t1) {
                const double f0 = RHS::f(col, t0.x0, t0.xn, p, 0.0, nullptr);
                const double f1 = RHS::f(col, t1.x0, t1.xn, p, 0.0, nullptr);
                const double r1 = t1.x0 - x0p - (fp + 4.0 * f0 + f1) * dt3;
                const double r2 = t0.x0 - (0.5 * (x0p + t1.x0) + (fp - f1) * dt4);
                const double w1 = WARR ? wfp * r1 : r1, w2 = WARR ? t0.wf * r2 : r2;     // (weight rows m and m+1, as va_tile4.h)
                fe = fma(w1, r1, fe);
                fe = fma(w2, r2, fe);
                const double q1 = cw * w1, q2 = cw * w2;
                emit_row(x0p, xnp, yp, dp, wmp, -q1 - 0.5 * q2 + qp - 0.5 * q2p, -dt3 * (q1 + qp) - dt4 * (q2 - q2p), 0);
                emit_row(t0.x0, t0.xn, t0.yv, t0.dval, t0.wm, q2, -(4.0 * dt3) * q1, 1);
                x0p = t1.x0; fp = f1; yp = t1.yv; dp = t1.dval; wfp = t1.wf; wmp = t1.wm; qp = q1; q2p = q2;
#pragma unroll
                for (int k = 0; k < NB; ++k) xnp[k] = t1.xn[k];
            };
            // slot 0: its second row is the even row the walk starts from (row 0, or two rows before the segment)
            begin_slot(0, 0, P % NSLOT);
            {
                const Row t1 = load_row(0, 1);
                step(t1, true, 0);
                flush(0, 0x7ffffff0);                         // (a store that goes nowhere: the queue counts of begin_slot stay those of the one-step walk)
            }
            int pos = 1 % NSLOT, ppos = (1 + P) % NSLOT;
            for (int k = 1; k < nslots; ++k) {
                begin_slot(k, pos, ppos);
                const Row t0 = load_row(pos, 0), t1 = load_row(pos, 1);
                step_sh(t0, t1);
                const bool lead = k == 1 && n0 > 0;           // the interval before the segment: finished into nowhere
                flush(rs + 2 * k - 1, lead ? 0x7ffffff0 : gvoff2);
                if (lead) {
                    fe = 0.0; me = 0.0; gtd = 0.0; gn2 = 0.0; gmax = 0.0;
#pragma unroll
                    for (int u = 0; u < RHS::NP; ++u) gp[u] = 0.0;
                }
                pos = pos + 1 == NSLOT ? 0 : pos + 1;
                ppos = ppos + 1 == NSLOT ? 0 : ppos + 1;
            }
            if (n1 == N) {                                    // the path's last row: no interval starts there
                emit_row(x0p, xnp, yp, dp, wmp, qp - 0.5 * q2p, -dt3 * qp + dt4 * q2p, 0);
                flush(N - 1, gvoff1);
            }
        } else {
        // slot 0: the row before the segment only loads the state; a segment that starts at row 0 has no
        // residual behind it (q_{-1} = 0) and its first row is finished for real, any other segment's
        // "row n0 - 1" is finished into nowhere and the sums it left are cleared
        begin_slot(0, 0, P % NSLOT);
        {
            const Row t0 = load_row(0, 0), t1 = load_row(0, 1);
            step(t0, true, 0);
            step(t1, false, 0);
            flush(rs, n0 > 0 ? 0x7ffffff0 : gvoff1);
        }
        if (n0 > 0) {
            fe = 0.0; me = 0.0; gtd = 0.0; gn2 = 0.0; gmax = 0.0;
#pragma unroll
            for (int k = 0; k < RHS::NP; ++k) gp[k] = 0.0;
        }
        const int nfull = SR >> 1;                            // slots with two rows of the stream
        int pos = 1 % NSLOT, ppos = (1 + P) % NSLOT;
        for (int k = 1; k < nfull; ++k) {
            begin_slot(k, pos, ppos);
            const Row t0 = load_row(pos, 0), t1 = load_row(pos, 1);      // both rows' reads in flight before the first product exchange
            step(t0, false, 0);
            step(t1, false, 1);
            flush(rs + 2 * k - 1, gvoff2);
            pos = pos + 1 == NSLOT ? 0 : pos + 1;
            ppos = ppos + 1 == NSLOT ? 0 : ppos + 1;
        }
        if (SR & 1) {
            begin_slot(nfull, pos, ppos);
            const Row t0 = load_row(pos, 0);
            step(t0, false, 0);
            flush(rs + 2 * nfull - 1, gvoff1);
        }
        if (n1 == N) { emit(0.0, 0); flush(N - 1, gvoff1); }    // the path's last row: no residual starts there
        }
        if (!drained) VA_WAIT_VM(0);

        // ---- the lane's sums (lanes outside the strip's own columns computed ghosts: not theirs to count)
        acc.v[EP_FE] = own ? (WARR ? fe : dm.rf0 * fe) : 0.0;
        acc.v[EP_ME] = own ? (WARR ? me : wobs * me) : 0.0;
        acc.v[EP_GTD] = own ? gtd : 0.0;
        acc.v[EP_GN2] = own ? gn2 : 0.0;
        acc.v[EP_GMAX] = own ? gmax : 0.0;
#pragma unroll
        for (int k = 0; k < RHS::NP; ++k) acc.v[EP_GP + k] = own ? gp[k] : 0.0;
    }

    // ---- ONE row of partial sums per workgroup (as k_eval4): matrix-pipe wave sums, LDS strip, wave 0 adds
    // the waves in wave order and stores the row write-through; the seed's last workgroup runs the tail
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if ((k == EP_GTD || k == EP_GN2 || k == EP_GMAX) && !LSRUN) continue;
        const double r = (k == EP_GMAX) ? wave_max(acc.v[k]) : wave_sum_mfma(acc.v[k]);
        if (lane == 0) strip_red[k] = r;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave != 0) return;
    if (lane < NV) {
        double v = 0.0;
        if (!((lane == EP_GTD || lane == EP_GN2 || lane == EP_GMAX) && !LSRUN)) {
            v = smem[lane];
            for (int w2 = 1; w2 < WPG; ++w2) {
                const double o = smem[w2 * WAVE + lane];
                v = (lane == EP_GMAX) ? fmax(v, o) : v + o;
            }
        }
        st_sc1(dv.evp + ((size_t)b * dm.ntiles + tile) * EP_N + lane, v);
    }
    if (dv.epi == EPI_NONE) return;
    if (arrive_last(dv.cnt_eval + (size_t)b * CNT_STRIDE, (unsigned)dm.ntiles, lane))
        eval_epilogue<true>(dv, b, lane, reinterpret_cast<SeedHot *>(smem + T4_STRIP), dv.epi);
}

inline size_t eval5_lds_bytes(const Dev &dv)
{
    return sizeof(double) * (size_t)dv.g5.WPG * tile5_wave_doubles(dv.g5, dv.lsrun ? dv.g5.nslot_ls : dv.g5.nslot, dv.lsrun != 0, dv.g5.warr != 0);
}

// launch (or, once per handle and device, opt in to > 64 KiB of LDS) the instantiation the handle's geometry and
// this launch's kind call for: ring depth and "line-search points possible" are template parameters
// weight arrays / data every nskip-th row (Geo5::warr): a three-slot ring only
template <class RHS, int DISC, int DC, bool XDPP>
inline hipError_t eval5_slots_w(const Dev &dv, bool prepare, hipStream_t s)
{
    const int threads = 64 * dv.g5.WPG;
    if (prepare) {
        hipError_t err = hipSuccess;
        Dev t = dv;
        for (int ls = 0; ls < 2; ++ls) {
            t.lsrun = ls;
            if (eval5_lds_bytes(t) <= 64 * 1024) continue;
            hipError_t e = ls ? hipFuncSetAttribute((const void *)k_eval5<RHS, DISC, DC, 3, true, XDPP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                              : hipFuncSetAttribute((const void *)k_eval5<RHS, DISC, DC, 3, false, XDPP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) err = e;
        }
        return err;
    }
    const size_t lds = eval5_lds_bytes(dv);
    const int grid = ((dv.dm.B * dv.dm.ntiles + 7) / 8) * 8;
    if (dv.lsrun) hipLaunchKernelGGL((k_eval5<RHS, DISC, DC, 3, true, XDPP, true>), dim3(grid), dim3(threads), lds, s, dv);
    else hipLaunchKernelGGL((k_eval5<RHS, DISC, DC, 3, false, XDPP, true>), dim3(grid), dim3(threads), lds, s, dv);
    return hipSuccess;
}
template <class RHS, int DISC, int DC, bool XDPP>
inline hipError_t eval5_slots(const Dev &dv, bool prepare, hipStream_t s)
{
    if (dv.g5.warr) return eval5_slots_w<RHS, DISC, DC, XDPP>(dv, prepare, s);
    const int threads = 64 * dv.g5.WPG;
    if (prepare) {
        hipError_t err = hipSuccess;
        Dev t = dv;
        for (int ls = 0; ls < 2; ++ls) {          // both launch kinds of the handle
            t.lsrun = ls;
            if (eval5_lds_bytes(t) <= 64 * 1024) continue;
            const int ns = ls ? t.g5.nslot_ls : t.g5.nslot;
            hipError_t e = hipSuccess;
#define VA_E5_ATTR(NSL, LS) e = hipFuncSetAttribute((const void *)k_eval5<RHS, DISC, DC, NSL, LS, XDPP, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
            if (ls) { if (ns == 3) VA_E5_ATTR(3, true); else VA_E5_ATTR(4, true); }
            else { if (ns == 3) VA_E5_ATTR(3, false); else VA_E5_ATTR(4, false); }
#undef VA_E5_ATTR
            if (e != hipSuccess) err = e;
        }
        return err;
    }
    const size_t lds = eval5_lds_bytes(dv);
    const int ns = dv.lsrun ? dv.g5.nslot_ls : dv.g5.nslot;
    const int grid = ((dv.dm.B * dv.dm.ntiles + 7) / 8) * 8;
#define VA_E5_GO(NSL, LS) hipLaunchKernelGGL((k_eval5<RHS, DISC, DC, NSL, LS, XDPP, false>), dim3(grid), dim3(threads), lds, s, dv)
    if (dv.lsrun) { if (ns == 3) VA_E5_GO(3, true); else VA_E5_GO(4, true); }
    else { if (ns == 3) VA_E5_GO(3, false); else VA_E5_GO(4, false); }
#undef VA_E5_GO
    return hipSuccess;
}
template <class RHS, int DISC, int DC>
inline hipError_t eval5_run(const Dev &dv, bool prepare, hipStream_t s)
{
    return dv.g5.xdpp ? eval5_slots<RHS, DISC, DC, t5_dpp_ok<RHS>()>(dv, prepare, s) : eval5_slots<RHS, DISC, DC, false>(dv, prepare, s);
}

}  // namespace va
