// va_persist.h -- k_seed<RHS, DISC>: the WHOLE RF ladder of a seed in ONE launch, for the reference's default use --
// one seed (or a few), a short path (examples/Lorenz96_D20/Lorenz96_anneal.py:84-86: 1 seed, N = 161, 101 rungs;
// BASELINE configs 1-2).  There the three-launch L-BFGS cycle (va_kernels.hip) is pure latency: each kernel costs a
// dependent graph node (~1.6 us) plus 4-8 us of round trips for a few kilobytes of work.
//
// Here a seed is owned by G co-resident 1024-thread workgroups (one per CU), workgroup w keeping ITS slice of T
// time rows of every vector of the minimisation ON CHIP for the whole ladder: x, g, the trial gradient, the direction
// and all 2m history vectors live in LDS (C2: 29-32 workgroups, ~150 KB each) -- each with the slice's halo rows, which
// every workgroup keeps up to date itself (all vector operations of L-BFGS are element-wise), so that the trial point
// x + stp*d of the halo rows never has to be fetched.  What a workgroup cannot form alone is exchanged ONCE per cycle:
//   evaluation   x (or x + stp*d) of the slice + halo into the staging rows; f, residuals, q, s, gradient rows by the
//                flat tile phases of va_core.h (any right-hand side, any discretisation, weight arrays, merr_nskip);
//   speculation  the inner products the update WOULD need if this point is accepted (k_update's: y.g_t, s.g_t, y.y, s.y,
//                S_j.g_t, Y_j.g_t, S_j.y, Y_j.y with y = g_t - g, s = stp*d) -- a history vector per wave;
//   all-gather   one row per workgroup: the evaluation's partial sums, those inner products, the g.d partial of the
//                direction in use, and the first / last gradient rows of the slice (the neighbours' halo), published as
//                self-validating granules -- each double as one 16-byte sc1 store {lo, tag, hi, tag}, tag = cycle number,
//                two alternating buffers -- and polled by EVERY workgroup with sc1 loads: no counter, no atomic, no
//                barrier (MI355X_MICROARCH.md, price list: handoff-1to1 / allgather; granules need no ordering);
//   decision     every workgroup adds the G rows in the same fixed order and runs the same More'-Thuente / L-BFGS-B /
//                ladder step (va_core.h ls_step; _autodiffmin.py:72-95, va_ode.py:707-789) on its own copy of the seed's
//                state: identical inputs, identical decisions, nothing to broadcast;
//   update       the compact-form coefficients (Byrd-Nocedal-Schnabel; pz_coeffs, the LDS twin of coeffs_wave), then ONE
//                element-wise pass: x += stp*d, (s, y) into their slot, g <- g_t, d = cg g + sum_j cY_j Y_j + cS_j S_j.
// Every poll is bounded (a seed whose workgroups are not all resident aborts the launch through a flag every poller
// reads: every wave reaches an exit).  Partial sums are added in a fixed order: results do not depend on timing.
// Not for bounded problems (clamps, k_lbfgsb_dir), time-dependent parameters or a dense linear part: those keep the
// three-launch cycle.
//
// Compiled into libvaranneal_amd.so for the built-in Lorenz-96 and into every generated right-hand-side module.
#pragma once
#include "va_eval_flat.h"
#include "va_measure.h"
#include "va_persist_geo.h"

namespace va {

typedef unsigned pz_v4u __attribute__((ext_vector_type(4)));

// one double as two self-validating 8-byte granules {lo, tag}, {hi, tag}, written by ONE 16-byte write-through store
__device__ __forceinline__ void pz_put(__amdgpu_buffer_rsrc_t r, int granule, double v, unsigned tag)
{
    const pz_v4u q = {(unsigned)__double2loint(v), tag, (unsigned)__double2hiint(v), tag};
    __builtin_amdgcn_raw_buffer_store_b128(q, r, granule * 16, 0, 16);
}

// poll the granule pair until both halves carry `tag`; false: the launch is being abandoned
__device__ __forceinline__ bool pz_poll(__amdgpu_buffer_rsrc_t r, int granule, unsigned tag, int *abort_flag, double &v)
{
    for (unsigned tries = 0;; ++tries) {
        const pz_v4u q = __builtin_amdgcn_raw_buffer_load_b128(r, granule * 16, 0, 16);
        if (q.y == tag && q.w == tag) { v = __hiloint2double((int)q.z, (int)q.x); return true; }
        asm volatile("" ::: "memory");                          // (the next load is a new load)
#ifndef PZ_POLL_SLEEP
#define PZ_POLL_SLEEP 1
#endif
        __builtin_amdgcn_s_sleep(PZ_POLL_SLEEP);
        if ((tries & 1023u) == 1023u) {
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (tries > PZ_POLL_LIMIT) { __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
        }
    }
}

// Gram update + compact-form direction coefficients by ONE wave, everything in LDS (the twin of coeffs_wave in
// va_kernels.hip; serial form: va_core.h direction_coeffs_view).  sSY / sYY: m x m, physical-slot indexed, persistent.
// up[]: the update totals in k_update's layout, the old pairs indexed as they were ordered BEFORE the line-search step:
// joff = 1 when that step evicted the oldest pair (new position j = old position j + 1).
__device__ __forceinline__ void pz_coeffs(SeedHot &s, const double *up, int joff, double *sSY, double *sYY, double *sp,
                                          double *cYs, double *cSs, int m, int lane)
{
    const int nold = s.nold, col = s.col, sn = s.slot;
    const bool hist = (s.upd & UPD_HIST) != 0;
    const double dr = s.dr;
    double theta = s.theta;
    const int myslot = lane < col ? s.order[lane] : 0;
    const double *upo = up + UP_OLD + 4 * ((lane < nold ? lane : 0) + joff);
    if (lane < MAX_M) { cYs[lane] = 0.0; cSs[lane] = 0.0; }
    if (hist) {
        if (lane < nold) {
            const double sjy = upo[2], yjy = upo[3];
            sSY[myslot * m + sn] = sjy; sYY[myslot * m + sn] = yjy; sYY[sn * m + myslot] = yjy;
        }
        if (lane == 0) { sSY[sn * m + sn] = dr; sYY[sn * m + sn] = up[UP_YY]; }   // s.y as the line search saw it
        theta = up[UP_YY] / dr;
    }
    wave_sync_lds();
    double aj = 0.0, bj = 0.0;
    if (lane < nold) { aj = upo[0]; bj = upo[1]; }
    if (hist && lane == col - 1) { aj = up[UP_SGT]; bj = up[UP_YGT]; }
    const double gamma = 1.0 / theta;
    const double rjj = lane < col ? sSY[myslot * m + myslot] : 1.0;
    const double rinv = 1.0 / rjj;
    double pj = 0.0, qj = 0.0, uj = 0.0;
    if (col <= PZ_MREG) {
        // the lane's row and column of R = S'Y and its row of Y'Y in registers (position order): the two triangular
        // solves then run on readlanes alone -- no LDS round trip inside their dependent chains
        // (unconditional loads at in-range addresses: entries of unused slots are finite and never enter a sum)
        double rrow[PZ_MREG], rcol[PZ_MREG], yrow[PZ_MREG];
        const int myrow = myslot * m;
#pragma unroll
        for (int i = 0; i < PZ_MREG; ++i) {
            const int si = s.order[i];                                  // (one LDS word, broadcast; always a slot number < m)
            rrow[i] = sSY[myrow + si];
            rcol[i] = sSY[si * m + myslot];
            yrow[i] = sYY[myrow + si];
        }
#pragma unroll
        for (int i = PZ_MREG - 1; i >= 0; --i) {                        // p = R^-1 a  (R_ji = S_j . Y_i for j <= i)
            if (i < col) {
                const double pi = lane_scalar(aj * rinv, i);
                if (lane == i) pj = pi;
                if (lane < i) aj -= rrow[i] * pi;
            }
        }
        double acc = 0.0;                                               // q = (D + gamma Y'Y) p - gamma b
#pragma unroll
        for (int k = 0; k < PZ_MREG; ++k)
            if (k < col) acc += yrow[k] * lane_scalar(pj, k);
        if (lane < col) qj = rjj * pj + gamma * acc - gamma * bj;
#pragma unroll
        for (int i = 0; i < PZ_MREG; ++i) {                             // u = R^-T q
            if (i < col) {
                const double ui = lane_scalar(qj * rinv, i);
                if (lane == i) uj = ui;
                if (lane > i && lane < col) qj -= rcol[i] * ui;
            }
        }
    } else {
        for (int i = col - 1; i >= 0; --i) {
            const int si = __builtin_amdgcn_readlane(myslot, i);
            const double pi = lane_scalar(aj * rinv, i);
            if (lane == i) pj = pi;
            if (lane < i) aj -= sSY[myslot * m + si] * pi;
        }
        if (lane < MAX_M) sp[lane] = pj;
        wave_sync_lds();
        if (lane < col) {
            double acc = 0.0;
            for (int k = 0; k < col; ++k) acc += sYY[myslot * m + __builtin_amdgcn_readlane(myslot, k)] * sp[k];
            qj = rjj * pj + gamma * acc - gamma * bj;
        }
        for (int i = 0; i < col; ++i) {
            const int si = __builtin_amdgcn_readlane(myslot, i);
            const double ui = lane_scalar(qj * rinv, i);
            if (lane == i) uj = ui;
            if (lane > i && lane < col) qj -= sSY[si * m + myslot] * ui;
        }
    }
    if (lane < col) { cYs[myslot] = gamma * pj; cSs[myslot] = -uj; }
    if (lane == 0) { s.cg = -gamma; s.theta = theta; }
    wave_sync_lds();
}

// rows 0 .. G-1 of one staged column, added in row order (EP_GMAX: the maximum); eight LDS loads in flight at a time
__device__ __forceinline__ double pz_col_sum(const double *col, int G, int stride, bool is_max)
{
    double v = is_max ? 0.0 : 0.0;
    for (int g0 = 0; g0 < G; g0 += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = g0 + u < G ? col[(g0 + u) * stride] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) v = is_max ? fmax(v, t[u]) : v + t[u];
    }
    return v;
}

// job r of the 2 col0 + 3 inner-product jobs of a trial point: which vector meets g_t and y, and where the two sums go
// (k_update's layout).  kind 0: a stored vector V, 1: y, 2: s, 3: g_t (o1 < 0: one sum only)
struct PzJob { const double *V; int kind, o0, o1; };
__device__ __forceinline__ PzJob pz_job(int r, int col0, const SeedHot *hot, const double *S, const double *Y, int nvh)
{
    PzJob j{nullptr, 0, 0, -1};
    if (r < col0) { j.V = S + (size_t)hot->order[r] * nvh; j.o0 = UP_OLD + 4 * r + 0; j.o1 = UP_OLD + 4 * r + 2; }
    else if (r < 2 * col0) { j.V = Y + (size_t)hot->order[r - col0] * nvh; j.o0 = UP_OLD + 4 * (r - col0) + 1; j.o1 = UP_OLD + 4 * (r - col0) + 3; }
    else if (r == 2 * col0) { j.kind = 1; j.o0 = UP_YGT; j.o1 = UP_YY; }
    else if (r == 2 * col0 + 1) { j.kind = 2; j.o0 = UP_SGT; j.o1 = UP_SY; }
    else { j.kind = 3; j.o0 = UP_GTGT; j.o1 = -1; }
    return j;
}

template <class RHS, int DISC>
__global__ __launch_bounds__(PZ_THREADS) void k_seed(const Dev dv)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Dims &dm = dv.dm;                 // the persistent image: dm.T = rows per slice, dm.ntiles = G
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR;
    constexpr int K = EP_GP + RHS::NP;      // eval partial columns in use
    constexpr int NT = PZ_THREADS;
    const int G = dm.ntiles, T = dm.T, D = dm.D, m = dm.m, NPe = dm.NPest;
    const int b = blockIdx.x / G, w = blockIdx.x - b * G;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = w * T;
    const int rows = (dm.N - n0 < T) ? dm.N - n0 : T;
    const int R = T + HL + HR, RD = R * D;  // staged rows: the slice and its halo
    const int nvh = RD + NPe;               // LDS pitch of a vector: [R rows | estimated parameters]
    const int own0 = HL * D, ne = rows * D; // own path elements: [own0, own0 + ne)
    const bool count_p = (w == 0);          // the parameter block enters the g.d partial once

    double *X = smem, *Gv = X + nvh, *Dd = Gv + nvh, *GT = Dd + nvh;
    double *S = GT + nvh, *Y = S + (size_t)m * nvh;
    double *xs = Y + (size_t)m * nvh, *fs = xs + RD, *qs = fs + RD;
    double *red = qs + RD;                          // [PZ_WAVES * EP_N]
    double *part = red + PZ_WAVES * EP_N;           // [PZ_HALO] this workgroup's sums, exchange-row layout
    double *tot = part + PZ_HALO;                   // [PZ_HALO] the seed's totals
    double *sSY = tot + PZ_HALO, *sYY = sSY + m * m;
    double *sp = sYY + m * m, *cYs = sp + MAX_M, *cSs = cYs + MAX_M;
    SeedHot *hot = reinterpret_cast<SeedHot *>(cSs + MAX_M);
    int *lds_ok = reinterpret_cast<int *>(reinterpret_cast<double *>(hot) + sizeof(SeedHot) / 8);
    double *stg = reinterpret_cast<double *>(lds_ok) + 1;              // [G * pz_sum_cols] the all-gather's staging rows
    // read-only rows the tile phases look up every cycle, copied once: the slice's observations and weight rows, the
    // column -> data-column map, the estimated parameters' slots
    double *ylds = stg + (size_t)G * pz_sum_cols(RHS::NP, m), *rmlds = ylds + (T + 1) * dm.L, *rflds = rmlds + (T + 1) * dm.L;
    int *lmap_l = reinterpret_cast<int *>(rflds + RD), *pidx_l = lmap_l + D;

    const int PD = pz_row_granules(D);
    // the seed's exchange area: [2 buffers][G rows][PD granule pairs of 16 bytes]
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(dv.pz.xch + (size_t)b * 2 * G * PD * 2), 0, (int)(2 * G * PD * 16), 0x00020000);

    // ---- the seed's state and its start point (halo rows included; rows that do not exist stay zero for good)
    {
        const double *xg = dv.x + (size_t)b * dm.ld;
        for (int e = tid; e < nvh; e += NT) {
            double v = 0.0;
            if (e < RD) {
                const int lr = e / D, row = n0 - HL + lr;
                if (row >= 0 && row < dm.N) v = xg[(size_t)row * D + (e - lr * D)];
            } else v = xg[dm.ND + (e - RD)];
            X[e] = v; Gv[e] = 0.0; Dd[e] = 0.0; GT[e] = 0.0;
        }
        for (int e = tid; e < 2 * m * nvh; e += NT) S[e] = 0.0;              // (S and Y are adjacent)
        for (int e = tid; e < 2 * m * m; e += NT) sSY[e] = 0.0;
        for (int e = tid; e < 2 * PZ_HALO; e += NT) part[e] = 0.0;           // (part and tot are adjacent)
        const double *gst = reinterpret_cast<const double *>(static_cast<const SeedHot *>(&dv.st[b]));
        if (tid < (int)(sizeof(SeedHot) / 8)) reinterpret_cast<double *>(hot)[tid] = gst[tid];
        if (tid == 0) *lds_ok = 1;
    }
    ProblemPtrs ppl = dv.pp;
    {
        const int L = dm.L;
        const int nd_lo = (n0 + dm.nskip - 1) / dm.nskip;                  // data rows that fall on own model rows
        int nd_hi = (n0 + rows - 1) / dm.nskip;
        if (nd_hi > dm.N_data - 1) nd_hi = dm.N_data - 1;
        const int ndn = nd_hi >= nd_lo ? nd_hi - nd_lo + 1 : 0;
        for (int e = tid; e < ndn * L; e += NT) {
            ylds[e] = dv.pp.Y[(size_t)nd_lo * L + e];
            if (dv.pp.rm_arr) rmlds[e] = dv.pp.rm_arr[(size_t)nd_lo * L + e];
        }
        const int r_lo = n0 - HL > 0 ? n0 - HL : 0, r_hi = n0 + T < dm.N - 1 ? n0 + T : dm.N - 1;      // rows of RF0: residuals (r, r + 1)
        if (dv.pp.rf0_arr)
            for (int e = tid; e < (r_hi - r_lo) * D; e += NT) rflds[e] = dv.pp.rf0_arr[(size_t)r_lo * D + e];
        for (int e = tid; e < D; e += NT) lmap_l[e] = dv.pp.lmap[e];
        for (int e = tid; e < NPe; e += NT) pidx_l[e] = dv.pp.Pidx[e];
        ppl.Y = ylds - (long)nd_lo * L;
        if (dv.pp.rm_arr) ppl.rm_arr = rmlds - (long)nd_lo * L;
        if (dv.pp.rf0_arr) ppl.rf0_arr = rflds - (long)r_lo * D;
        ppl.lmap = lmap_l; ppl.Pidx = pidx_l;
    }
    double pfix[RHS::NP > 0 ? RHS::NP : 1];
#pragma unroll
    for (int k = 0; k < RHS::NP; ++k) pfix[k] = dv.pp.Pfull[(size_t)b * dm.NP + k];
    __syncthreads();

    TileCtx c;
    c.n0 = n0; c.R = R; c.xs = xs; c.fs = fs; c.qs = qs;
    c.xg = nullptr; c.dg = Dd; c.gtg = GT; c.goff = (long)(n0 - HL) * D;
    c.tmodel = dv.pp.tmodel; c.stim = dv.pp.stim; c.nstim = dv.pp.nstim;
    c.ps = nullptr;

    long long cyc = 0;
    PZ_MARK_SETUP();               // (measurement builds only: va_measure.h)
    for (;; ++cyc) {
        PZ_MARK(12);
        const int phase = hot->phase;
        if (phase != PH_START && phase != PH_LS) break;
        if (cyc >= dv.pz.max_cycles) {
            if (tid == 0) __hip_atomic_store(dv.pz.abort_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        const unsigned gen = (unsigned)cyc + 1u;
        const int par = (int)(gen & 1u);
        const int use_d = phase == PH_LS;
        const double stp = hot->stp;
        const int col0 = use_d ? hot->col : 0;        // pairs in the history while this point is evaluated
        c.use_d = use_d; c.stp = stp; c.c = 2.0 * hot->rf_scale * dm.cfe;

        // ---- evaluation, 1: the trial point of the slice and its halo
        for (int e = tid; e < RD; e += NT) xs[e] = use_d ? trial(X[e], stp, Dd[e]) : X[e];
        // (parameters of the right-hand side: fixed ones from the table, estimated ones from the trial point)
#pragma unroll
        for (int k = 0; k < RHS::NP; ++k) c.p[k] = pfix[k];
        for (int k = 0; k < NPe; ++k) {
            double v = X[RD + k];
            if (use_d) v = trial(v, stp, Dd[RD + k]);
            const int dst = pidx_l[k];
#pragma unroll
            for (int j = 0; j < RHS::NP; ++j) c.p[j] = (dst == j) ? v : c.p[j];
        }
        __syncthreads();
        PZ_MARK(0);

        // ---- evaluation, 2: the flat tile phases (va_core.h) on the staged rows
        ThreadAcc acc;
        acc.clear();
        tile_f<RHS, DISC>(dm, c, tid, NT);
        __syncthreads();
        tile_q<DISC>(dm, ppl, c, acc, tid, NT);
        __syncthreads();
        if (dv.pp.rf0_full) {
            tile_qfull<DISC>(dm, ppl, c, acc, tid, NT);
            double *t = c.qs; c.qs = c.fs; c.fs = t;
            __syncthreads();
        }
        tile_s<DISC>(dm, c, tid, NT);
        __syncthreads();
        tile_g<RHS, DISC>(dm, ppl, c, acc, tid, NT);
        if (dv.pp.rf0_full) { double *t = c.qs; c.qs = c.fs; c.fs = t; }
        PZ_MARK(1);
        // the workgroup's evaluation sums: wave totals -- of the waves that held elements (thread t works on element t of the
        // staged rows: the waves beyond them have nothing to add, and a wave reduction costs the same with or without data)
        if (wave * 64 < RD) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double v = (k == EP_GMAX) ? wave_max(acc.v[k]) : wave_sum(acc.v[k]);
                if (lane == 0) red[wave * K + k] = v;
            }
        } else if (lane < K) red[wave * K + lane] = 0.0;
        __syncthreads();                                  // (the slice's gradient rows are complete)
        PZ_MARK(2);

        // ---- this workgroup's row goes out in two parts.  The last wave: evaluation sums, the g.d partial of the direction
        // in use, the slice's edge gradient rows.  The other waves meanwhile: SPECULATION -- the inner products the update
        // will need if this point is accepted, a job per wave, its lanes striding the slice's own path elements (the
        // parameter block joins after the all-gather, when its gradient is known), each wave publishing its own two sums.
        const int KU = use_d ? UP_OLD + 4 * col0 : 0;
        const int njobs = use_d ? 2 * col0 + 3 : 0;
        const int mine = (par * G + w) * PD;
        PzJob myjob{nullptr, 3, 0, -1};
        if (tid < njobs) myjob = pz_job(tid, col0, hot, S, Y, nvh);           // (for the parameter block, after the step below has reordered the history)
        if (wave == PZ_WAVES - 1) {
            if (lane < K) {
                double v = red[lane];
#pragma unroll
                for (int ww = 1; ww < PZ_WAVES; ++ww) v = (lane == EP_GMAX) ? fmax(v, red[ww * K + lane]) : v + red[ww * K + lane];
                part[lane] = v;
                if (G > 1) pz_put(xr, mine + lane, v, gen);
            }
            if (G > 1) {
                if (lane == K) pz_put(xr, mine + PZ_GDO, part[PZ_GDO], gen);
                for (int h = lane; h < PZ_EDGE_ROWS * D; h += 64) {
                    const int r = h / D, j = h - r * D;
                    const int lr = r < 2 ? HL + rows - 2 + r : HL;       // (every slice holds >= 2 rows)
                    pz_put(xr, mine + PZ_HALO + h, GT[lr * D + j], gen);
                }
            }
        }
        {
            for (int r = wave; r < njobs; r += PZ_WAVES) {
                const PzJob jb = pz_job(r, col0, hot, S, Y, nvh);
                double a0 = 0.0, a1 = 0.0;
                const double *V = jb.kind == 0 ? jb.V : (jb.kind == 2 ? Dd : GT);      // (wave-uniform)
                for (int i0 = lane; i0 < ne; i0 += 256) {
                    double tv[4], gv[4], vv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {                                      // (all twelve loads before the first use)
                        const int i = i0 + 64 * u, e = own0 + (i < ne ? i : 0);
                        tv[u] = GT[e]; gv[u] = Gv[e]; vv[u] = V[e];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (i0 + 64 * u < ne) {
                            const double yv = tv[u] - gv[u];
                            const double v = jb.kind == 0 ? vv[u] : (jb.kind == 1 ? yv : (jb.kind == 2 ? stp * vv[u] : tv[u]));
                            a0 += v * tv[u]; a1 += v * yv;
                        }
                    }
                }
                a0 = wave_sum(a0); a1 = wave_sum(a1);
                if (lane == 0) {
                    part[EP_N + jb.o0] = a0;
                    if (jb.o1 >= 0) part[EP_N + jb.o1] = a1;
                    if (G > 1) {
                        pz_put(xr, mine + EP_N + jb.o0, a0, gen);
                        if (jb.o1 >= 0) pz_put(xr, mine + EP_N + jb.o1, a1, gen);
                    }
                }
            }
        }
        PZ_MARK(3);

        // ---- all-gather, first part: every workgroup's evaluation sums, the neighbours' gradient rows
        const int nev = K + 1;
        bool ok = true;
        if (G > 1) {
            if (w > 0) {
                const int src = (par * G + w - 1) * PD + PZ_HALO + (2 - HL) * D;
                for (int e = tid; e < HL * D; e += NT) { double v = 0.0; ok = pz_poll(xr, src + e, gen, dv.pz.abort_flag, v) && ok; GT[e] = v; }
            }
            if (w + 1 < G) {
                const int src = (par * G + w + 1) * PD + PZ_HALO + 2 * D;
                for (int e = tid; e < HR * D; e += NT) { double v = 0.0; ok = pz_poll(xr, src + e, gen, dv.pz.abort_flag, v) && ok; GT[(HL + T) * D + e] = v; }
            }
            for (int it = tid; it < nev * G; it += NT) {
                const int g = it / nev, ci = it - g * nev;
                double v = 0.0;
                ok = pz_poll(xr, (par * G + g) * PD + (ci < K ? ci : PZ_GDO), gen, dv.pz.abort_flag, v) && ok;
                stg[it] = v;
            }
            if (!ok) *lds_ok = 0;
            __syncthreads();
            if (*lds_ok == 0) return;
            if (tid < nev) tot[tid < K ? tid : PZ_GDO] = pz_col_sum(stg + tid, G, nev, tid == EP_GMAX);
        } else {
            __syncthreads();
            if (tid < nev) tot[tid < K ? tid : PZ_GDO] = part[tid < K ? tid : PZ_GDO];
        }
        __syncthreads();
        PZ_MARK(4);

        // ---- decision (thread 0; every workgroup, identically): the parameter tail of the gradient, then one line-search /
        // ladder step.  Beside it the other waves bring in the second part of the rows: the speculated inner products.
        if (tid == 0) {
            for (int k = 0; k < NPe; ++k) {
                const double g = tot[EP_GP + pidx_l[k]];
                GT[RD + k] = g;
                if (use_d) tot[EP_GTD] += g * Dd[RD + k];
                tot[EP_GN2] += g * g;
                tot[EP_GMAX] = fmax(tot[EP_GMAX], fabs(g));
            }
            double ev[EP_N];
#pragma unroll
            for (int k = 0; k < EP_N; ++k) ev[k] = k < K ? tot[k] : 0.0;
            if (w == 0) atomicAdd(dv.n_evals, 1ULL);
            SeedResults r;
            r.ame = dv.ame + (size_t)b * dv.max_beta * 3;
            r.pest = nullptr;
            r.status = dv.status + (size_t)b * dv.max_beta;
            r.nit = dv.nit + (size_t)b * dv.max_beta;
            r.nfev = dv.nfev + (size_t)b * dv.max_beta;
            int dec = 0;
            double dirp[DP_N];
            dirp[DP_GD] = tot[PZ_GDO]; dirp[DP_DD] = 0.0;
            hot->gd_dir = tot[PZ_GDO];
            hot->pad0 = 0;
            ls_step(*hot, ev, dirp, dv.o, dv.rf_ladder, dv.nbeta, r, &dec, dm.cme, dm.cfe, false);
            if (dec && w == 0) atomicSub(dv.n_active, 1);
        } else if (G > 1 && tid >= 64) {
            double *stg2 = stg + nev * G;
            for (int it = tid - 64; it < KU * G; it += NT - 64) {
                const int g = it / KU, ci = it - g * KU;
                double v = 0.0;
                ok = pz_poll(xr, (par * G + g) * PD + EP_N + ci, gen, dv.pz.abort_flag, v) && ok;
                stg2[it] = v;
            }
            if (!ok) *lds_ok = 0;
        }
        __syncthreads();
        if (*lds_ok == 0) return;
        PZ_MARK(5);
        if (use_d) {
            // the seed's inner products: rows added in order, then the parameter block's share (the job's own vector
            // against g_t's and y's parameter entries, now that the tail has formed them)
            if (tid < KU) {
                double v;
                if (G > 1) v = pz_col_sum(stg + nev * G + tid, G, KU, false);
                else v = part[EP_N + tid];
                tot[EP_N + tid] = v;
            }
            __syncthreads();
            if (NPe > 0) {
                if (tid < njobs) {
                    double a0 = 0.0, a1 = 0.0;
                    for (int k = 0; k < NPe; ++k) {
                        const int e = RD + k;
                        const double tv = GT[e], yv = tv - Gv[e];
                        const double v = myjob.kind == 0 ? myjob.V[e] : (myjob.kind == 1 ? yv : (myjob.kind == 2 ? stp * Dd[e] : tv));
                        a0 += v * tv; a1 += v * yv;
                    }
                    tot[EP_N + myjob.o0] += a0;
                    if (myjob.o1 >= 0) tot[EP_N + myjob.o1] += a1;
                }
                __syncthreads();
            }
        }
        PZ_MARK(6);

        // ---- update and direction
        const int upd = hot->upd, dir = hot->dir;
        if (!upd && !dir) continue;
        const bool hist = (upd & UPD_HIST) != 0;
        const int joff = (hist && (hot->pad0 & 2)) ? 1 : 0;      // the step evicted the oldest pair: positions moved down by one
        if (dir) {
            if (wave == 0) pz_coeffs(*hot, tot + EP_N, joff, sSY, sYY, sp, cYs, cSs, m, lane);
            __syncthreads();
        }
        PZ_MARK(7);
        {
            const int slot = hot->slot, col = hot->col;
            const double stpu = hot->stp_upd, cg = hot->cg;
            double *Sn = S + (size_t)slot * nvh, *Yn = Y + (size_t)slot * nvh;
            double *mp = nullptr;
            if ((upd & UPD_STORE) && dv.minpaths) mp = dv.minpaths + ((size_t)b * dv.max_beta + hot->store_idx) * (dm.ND + dm.NP);
            double gd = 0.0;
            for (int e = tid; e < nvh; e += NT) {
                const double dv2 = Dd[e], gv = Gv[e], tv = GT[e];
                double xv = X[e];
                if (upd & UPD_X) { xv = trial(xv, stpu, dv2); X[e] = xv; }
                if (upd & UPD_STORE) {
                    if (e >= own0 && e < own0 + ne) { if (mp) mp[(size_t)n0 * D + (e - own0)] = xv; }
                    else if (e >= RD && w == 0) {
                        const int k = e - RD;
                        dv.pest[((size_t)b * dv.max_beta + hot->store_idx) * NPe + k] = xv;
                        if (mp) mp[dm.ND + dv.pp.Pidx[k]] = xv;
                    }
                }
                if (hist) { Sn[e] = stpu * dv2; Yn[e] = tv - gv; }
                double gn = gv;
                if (upd & UPD_G) { gn = tv; Gv[e] = tv; }
                if (dir) {
                    double a = cg * gn;
                    // (five pairs at a time: their slots and coefficients first, then the ten vector entries -- two levels of LDS
                    // latency per batch instead of two per pair)
                    for (int j0 = 0; j0 < col; j0 += 5) {
                        int off[5]; double cy[5], cs[5], yv[5], sv[5];
#pragma unroll
                        for (int u = 0; u < 5; ++u) {
                            const int sj = hot->order[j0 + u < col ? j0 + u : 0];
                            off[u] = sj * nvh + e; cy[u] = j0 + u < col ? cYs[sj] : 0.0; cs[u] = j0 + u < col ? cSs[sj] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 5; ++u) { yv[u] = Y[off[u]]; sv[u] = S[off[u]]; }
#pragma unroll
                        for (int u = 0; u < 5; ++u) { a += cy[u] * yv[u]; a += cs[u] * sv[u]; }
                    }
                    Dd[e] = a;
                    if ((e >= own0 && e < own0 + ne) || (e >= RD && count_p)) gd += gn * a;
                }
            }
            if (mp && w == 0 && tid == 0) {
                for (int j = 0; j < dm.NP; ++j) {         // fixed (non-estimated) parameters of the stored step come from P
                    bool est = false;
                    for (int k = 0; k < NPe; ++k) est = est || (dv.pp.Pidx[k] == j);
                    if (!est) mp[dm.ND + j] = dv.pp.Pfull[(size_t)b * dm.NP + j];
                }
            }
            if (dir) {
                gd = wave_sum(gd);
                if (lane == 0) red[wave] = gd;
            }
            __syncthreads();
            if (tid == 0) {
                if (dir) {
                    double v = red[0];
#pragma unroll
                    for (int ww = 1; ww < PZ_WAVES; ++ww) v += red[ww];
                    part[PZ_GDO] = v;                     // (travels with the next evaluation's row)
                }
                hot->upd = 0; hot->dir = 0;
            }
            __syncthreads();
        }
        PZ_MARK(8);
    }

    // ---- the final iterate back to the seed's global vector; state for the host's bookkeeping
    __syncthreads();
    {
        double *xg = dv.x + (size_t)b * dm.ld;
        for (int e = tid; e < nvh; e += NT) {
            if (e >= own0 && e < own0 + ne) xg[(size_t)n0 * D + (e - own0)] = X[e];
            else if (e >= RD && w == 0) xg[dm.ND + (e - RD)] = X[e];
        }
        if (w == 0) {
            double *gst = reinterpret_cast<double *>(static_cast<SeedHot *>(&dv.st[b]));
            if (tid < (int)(sizeof(SeedHot) / 8)) gst[tid] = reinterpret_cast<double *>(hot)[tid];
            if (tid == 0) atomicAdd(dv.pz.cycles, (unsigned long long)cyc);
        }
    }
    PZ_MARK_FLUSH(dv.pz.stamps, cyc);
}

inline const void *seed_kernel_of(const void *const k[4], int disc)
{
    return k[disc == DISC_EULER ? 0 : disc == DISC_TRAPEZOID ? 1 : disc == DISC_SH ? 2 : 3];
}

// launch == false: opt the instantiation in to the LDS it needs on the current device (once per handle);
// launch == true: dv.dm.B * dv.dm.ntiles workgroups, one per CU (each takes more than half of a CU's LDS, and the host
// never asks for more workgroups than the device has CUs).  A PLAIN launch: it has the residency of a cooperative one
// (MI355X_MICROARCH.md, "Residency and cooperative launch"), costs 15-19 us less on the host, and -- measured in round 4 --
// a process that made a cooperative launch under rocprofv3 --kernel-trace dies in the tool's exit handler.  Should the
// workgroups not all be resident after all (another process holding CUs), the bounded polls end the launch with
// abort_flag = 1 and the host falls back to the three-launch cycle.
template <class RHS>
inline hipError_t seed_kernel_op(const Dev &dv, bool launch, hipStream_t s)
{
    const void *const ks[4] = {(const void *)k_seed<RHS, DISC_EULER>, (const void *)k_seed<RHS, DISC_TRAPEZOID>,
                               (const void *)k_seed<RHS, DISC_SH>, (const void *)k_seed<RHS, DISC_FWDMAP>};
    const void *k = seed_kernel_of(ks, dv.dm.disc);
    const int HL = dv.dm.disc == DISC_SH ? 2 : 1;
    const size_t lds = 8 * persist_lds_doubles(dv.dm.T, dv.dm.D, dv.dm.L, RHS::NP, dv.dm.NPest, dv.dm.m, HL, dv.dm.ntiles);
    if (!launch) return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    void *args[1] = {(void *)&dv};
    return hipLaunchKernel(k, dim3(dv.dm.B * dv.dm.ntiles), dim3(PZ_THREADS), args, lds, s);
}

}  // namespace va
