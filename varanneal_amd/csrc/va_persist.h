// va_persist.h -- k_seed<RHS, DISC>: the WHOLE RF ladder of a seed in ONE launch, for the reference's default use --
// one seed (or a few), a short path (examples/Lorenz96_D20/Lorenz96_anneal.py:84-86: 1 seed, N = 161, 101 rungs;
// BASELINE configs 1-2).  There the three-launch L-BFGS cycle (va_kernels.hip) is pure latency: each kernel costs a
// dependent graph node (~1.6 us) plus 4-8 us of round trips for a few kilobytes of work.
//
// Here a seed is owned by G co-resident workgroups (cooperative launch), workgroup w keeping ITS slice of T time rows
// of every vector of the minimisation ON CHIP for the whole ladder: x, g, the trial gradient, the direction and all
// 2m history vectors live in LDS (C2: 32 workgroups x 32 rows, 149 KB each).  A cycle is what the three launches do,
// in the same order, between workgroup barriers:
//   evaluation   x (or x + stp*d) of the own rows into the staging rows; the first / last rows of the slice are
//                published for the neighbours' halo (sc1 stores);  GRID BARRIER A;  f, residuals, q, s, gradient rows
//                by the flat tile phases of va_core.h (any right-hand side, any discretisation, weights, merr_nskip);
//                the slice's partial sums are published;  GRID BARRIER B;  EVERY workgroup adds the G partial rows in
//                the same fixed order and runs the same More'-Thuente / L-BFGS-B / ladder step (va_core.h ls_step,
//                _autodiffmin.py:72-95, va_ode.py:707-789) on its own copy of the seed's state: identical inputs,
//                identical decisions, nothing to broadcast;
//   update       x += stp*d, (s, y) into their history slot, g <- g_t, all inner products of the compact form; partial
//                sums published;  GRID BARRIER C;  every workgroup solves for the direction coefficients
//                (Byrd-Nocedal-Schnabel compact form: pz_coeffs, the LDS-resident twin of coeffs_wave);
//   direction    d = cg g + sum_j cY_j Y_j + cS_j S_j on the own slice; its g.d partial travels with barrier A of the
//                next evaluation.
// Three grid barriers per accepted iterate, two per extra line-search trial, no launch, no atomics on data.  A grid
// barrier = own stores drained (s_waitcnt vmcnt(0)) + workgroup barrier + one agent-scope atomic add on the seed's
// monotonic counter + a bounded spin (wall clock: a seed whose workgroups are not all resident aborts the launch through
// a flag every spinning workgroup polls -- every wave reaches an exit).  Partial sums are added in a fixed order: results
// do not depend on timing.
//
// Compiled into libvaranneal_amd.so for the built-in Lorenz-96 and into every generated right-hand-side module.
#pragma once
#include "va_eval_flat.h"

namespace va {

// measurement build (-DVA_PZ_STAMPS, never the product): thread 0 of workgroup 0 of seed 0 accumulates the wall-clock
// ticks (100 MHz) between consecutive marks of the cycle and leaves the sums in the first words of pz.upp
#ifdef VA_PZ_STAMPS
#define PZ_MARK(i) do { if (tid == 0) { const long long t_ = wall_clock64(); pz_acc[i] += t_ - pz_prev; pz_prev = t_; } } while (0)
#else
#define PZ_MARK(i) do {} while (0)
#endif

constexpr int PZ_THREADS = 256;
constexpr int PZ_WAVES = PZ_THREADS / 64;
constexpr int PZ_EDGE_ROWS = 3;                 // per workgroup: its last two rows (right neighbour's left halo), its first row
constexpr long long PZ_SPIN_LIMIT = 400000000;  // wall_clock64() ticks (100 MHz): 4 s

// LDS doubles of one workgroup (T rows per slice, history length m)
VA_HD size_t persist_lds_doubles(int T, int D, int NPest, int m, int HL)
{
    const size_t nv = (size_t)T * D + NPest, RD = (size_t)(T + HL + 1) * D;
    return (4 + 2 * (size_t)m) * nv + 3 * RD + (size_t)PZ_WAVES * UP_N + UP_N + EP_N + 2 * (size_t)m * m + 3 * MAX_M
           + sizeof(SeedHot) / 8 + 8;
}

// slice geometry: the fewest workgroups whose slices fit the CU's LDS; every slice holds at least two rows;
// Simpson-Hermite slices start on even rows (an interval's three rows then reach one row past the slice: HR = 1)
inline bool persist_geometry(int N, int D, int NPest, int m, int disc, size_t lds_bytes, int *G, int *T)
{
    const bool sh = disc == DISC_SH;
    const int HL = sh ? 2 : 1;
    int Tmax = 0;
    for (int t = 2; t <= N; ++t) {
        if (persist_lds_doubles(t, D, NPest, m, HL) * 8 > lds_bytes) break;
        Tmax = t;
    }
    if (Tmax < 2) return false;
    for (int g = (N + Tmax - 1) / Tmax; g <= N / 2 + 1; ++g) {
        int t = (N + g - 1) / g;
        if (sh && (t & 1)) ++t;
        if (t > Tmax || t < 2) continue;
        if ((long)(g - 1) * t < N && N - (g - 1) * t >= 2) { *G = g; *T = t; return true; }
    }
    return false;
}

struct PzBarrier {
    unsigned long long *cnt;      // the seed's monotonic arrival counter
    unsigned long long gen;       // barriers passed so far
    int *abort_flag;
    int G;
};

// returns false (workgroup-uniform) when the launch is being abandoned
__device__ __forceinline__ bool pz_grid_barrier(PzBarrier &bar, int tid, int *lds_ok)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own published stores acknowledged
    __syncthreads();
    bar.gen += 1;
    if (tid == 0) {
        const unsigned long long target = bar.gen * (unsigned long long)bar.G;
        __hip_atomic_fetch_add(bar.cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        unsigned spins = 0;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(bar.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255u) == 0u) {
                if (__hip_atomic_load(bar.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                if (wall_clock64() - t0 > PZ_SPIN_LIMIT) {
                    __hip_atomic_store(bar.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0; break;
                }
            }
        }
        *lds_ok = ok;
    }
    __syncthreads();
    return *lds_ok != 0;
}

// the K leading entries of acc.v, summed (EP_GMAX: maximum) over the workgroup: red [PZ_WAVES * K] scratch, out[K]
template <int K>
__device__ __forceinline__ void pz_block_reduce_ev(const ThreadAcc &acc, double *red, double *out, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double v = (k == EP_GMAX) ? wave_max(acc.v[k]) : wave_sum(acc.v[k]);
        if (lane == 0) red[wave * K + k] = v;
    }
    __syncthreads();
    if (tid < K) {
        double v = red[tid];
#pragma unroll
        for (int ww = 1; ww < PZ_WAVES; ++ww) v = (tid == EP_GMAX) ? fmax(v, red[ww * K + tid]) : v + red[ww * K + tid];
        out[tid] = v;
    }
    __syncthreads();
}

// Gram update + compact-form direction coefficients by ONE wave, everything in LDS (the twin of coeffs_wave in
// va_kernels.hip; serial form: va_core.h direction_coeffs_view).  sSY / sYY: m x m, physical-slot indexed, persistent.
__device__ __forceinline__ void pz_coeffs(SeedHot &s, const double *up, double *sSY, double *sYY, double *sp,
                                          double *cYs, double *cSs, int m, int lane)
{
    const int nold = s.nold, col = s.col, sn = s.slot;
    const bool hist = (s.upd & UPD_HIST) != 0;
    const double dr = s.dr;
    double theta = s.theta;
    const int myslot = lane < col ? s.order[lane] : 0;
    if (lane < MAX_M) { cYs[lane] = 0.0; cSs[lane] = 0.0; }
    if (hist) {
        if (lane < nold) {
            const double sjy = up[UP_OLD + 4 * lane + 2], yjy = up[UP_OLD + 4 * lane + 3];
            sSY[myslot * m + sn] = sjy; sYY[myslot * m + sn] = yjy; sYY[sn * m + myslot] = yjy;
        }
        if (lane == 0) { sSY[sn * m + sn] = dr; sYY[sn * m + sn] = up[UP_YY]; }   // s.y as the line search saw it
        theta = up[UP_YY] / dr;
    }
    wave_sync_lds();
    double aj = 0.0, bj = 0.0;
    if (lane < nold) { aj = up[UP_OLD + 4 * lane + 0]; bj = up[UP_OLD + 4 * lane + 1]; }
    if (hist && lane == col - 1) { aj = up[UP_SGT]; bj = up[UP_YGT]; }
    const double gamma = 1.0 / theta;
    const double rjj = lane < col ? sSY[myslot * m + myslot] : 1.0;
    const double rinv = 1.0 / rjj;
    double pj = 0.0;                                                   // p = R^-1 a  (R_ji = S_j . Y_i for j <= i)
    for (int i = col - 1; i >= 0; --i) {
        const int si = __builtin_amdgcn_readlane(myslot, i);
        const double pi = lane_scalar(aj * rinv, i);
        if (lane == i) pj = pi;
        if (lane < i) aj -= sSY[myslot * m + si] * pi;
    }
    if (lane < MAX_M) sp[lane] = pj;
    wave_sync_lds();
    double qj = 0.0;                                                   // q = (D + gamma Y'Y) p - gamma b
    if (lane < col) {
        double acc = 0.0;
        for (int k = 0; k < col; ++k) acc += sYY[myslot * m + __builtin_amdgcn_readlane(myslot, k)] * sp[k];
        qj = rjj * pj + gamma * acc - gamma * bj;
    }
    double uj = 0.0;                                                   // u = R^-T q
    for (int i = 0; i < col; ++i) {
        const int si = __builtin_amdgcn_readlane(myslot, i);
        const double ui = lane_scalar(qj * rinv, i);
        if (lane == i) uj = ui;
        if (lane > i && lane < col) qj -= sSY[si * m + myslot] * ui;
    }
    if (lane < col) { cYs[myslot] = gamma * pj; cSs[myslot] = -uj; }
    if (lane == 0) { s.cg = -gamma; s.theta = theta; }
    wave_sync_lds();
}

template <class RHS, int DISC>
__global__ __launch_bounds__(PZ_THREADS) void k_seed(const Dev dv)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Dims &dm = dv.dm;                 // the persistent image: dm.T = rows per slice, dm.ntiles = G
    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR;
    constexpr int K = EP_GP + RHS::NP;      // eval partial columns in use
    const int G = dm.ntiles, T = dm.T, D = dm.D, m = dm.m, NPe = dm.NPest;
    const int b = blockIdx.x / G, w = blockIdx.x - b * G;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = w * T;
    const int rows = (dm.N - n0 < T) ? dm.N - n0 : T;
    const int ne = rows * D;                // own path elements
    const int TD = T * D, nv = TD + NPe;    // LDS pitch of a vector: [T rows | estimated parameters]
    const int R = T + HL + HR, RD = R * D;

    double *X = smem, *Gv = X + nv, *Dd = Gv + nv, *GT = Dd + nv;
    double *S = GT + nv, *Y = S + (size_t)m * nv;
    double *xs = Y + (size_t)m * nv, *fs = xs + RD, *qs = fs + RD;
    double *red = qs + RD;                          // [PZ_WAVES * UP_N]
    double *tot = red + PZ_WAVES * UP_N;            // [UP_N] update totals
    double *evt = tot + UP_N;                       // [EP_N] evaluation totals
    double *sSY = evt + EP_N, *sYY = sSY + m * m;
    double *sp = sYY + m * m, *cYs = sp + MAX_M, *cSs = cYs + MAX_M;
    SeedHot *hot = reinterpret_cast<SeedHot *>(cSs + MAX_M);
    int *lds_ok = reinterpret_cast<int *>(reinterpret_cast<double *>(hot) + sizeof(SeedHot) / 8);

    PzBarrier bar{dv.pz.bar + (size_t)b * 32, 0ull, dv.pz.abort_flag, G};
    double *edge = dv.pz.edge + ((size_t)b * G + w) * (PZ_EDGE_ROWS * D);
    double *my_evp = dv.pz.evp + ((size_t)b * G + w) * EP_N;
    double *my_upp = dv.pz.upp + ((size_t)b * G + w) * UP_N;
    const double *all_evp = dv.pz.evp + (size_t)b * G * EP_N, *all_upp = dv.pz.upp + (size_t)b * G * UP_N;
    const double *all_gdp = dv.pz.gdp + (size_t)b * G;
    const bool count_p = (w == 0);          // the parameter block enters inner products once

    // ---- the seed's state and its start point
    {
        const double *xg = dv.x + (size_t)b * dm.ld;
        for (int e = tid; e < nv; e += PZ_THREADS) {
            double v = 0.0;
            if (e < ne) v = xg[(size_t)n0 * D + e];
            else if (e >= TD) v = xg[dm.ND + (e - TD)];
            X[e] = v; Gv[e] = 0.0; Dd[e] = 0.0; GT[e] = 0.0;
        }
        for (int e = tid; e < 2 * m * nv; e += PZ_THREADS) S[e] = 0.0;        // (S and Y are adjacent)
        for (int e = tid; e < 2 * m * m; e += PZ_THREADS) sSY[e] = 0.0;
        const double *gst = reinterpret_cast<const double *>(static_cast<const SeedHot *>(&dv.st[b]));
        if (tid < (int)(sizeof(SeedHot) / 8)) reinterpret_cast<double *>(hot)[tid] = gst[tid];
        if (tid == 0) *lds_ok = 1;
    }
    __syncthreads();

    TileCtx c;
    c.n0 = n0; c.R = R; c.xs = xs; c.fs = fs; c.qs = qs;
    c.xg = nullptr; c.dg = Dd; c.gtg = GT; c.goff = (long)n0 * D;
    c.tmodel = dv.pp.tmodel; c.stim = dv.pp.stim; c.nstim = dv.pp.nstim;
    c.ps = nullptr;

    bool pending_gd = false;
    long long cyc = 0;
#ifdef VA_PZ_STAMPS
    long long pz_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pz_prev = wall_clock64();
#endif
    for (;; ++cyc) {
        PZ_MARK(11);
        const int phase = hot->phase;
        if (phase != PH_START && phase != PH_LS) break;
        if (cyc >= dv.pz.max_cycles) {
            if (tid == 0) __hip_atomic_store(dv.pz.abort_flag, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        const int use_d = phase == PH_LS;
        const double stp = hot->stp;
        c.use_d = use_d; c.stp = stp; c.c = 2.0 * hot->rf_scale * dm.cfe;

        // ---- evaluation, 1: the trial point of the own rows; rows of the slice that do not exist stay zero
        for (int e = tid; e < RD; e += PZ_THREADS) {
            const int o = e - HL * D;
            double v = 0.0;
            if (o >= 0 && o < ne) {
                v = X[o];
                if (use_d) v = clampb(trial(v, stp, Dd[o]), dv.pp, (long)n0 * D + o);
            }
            xs[e] = v;
        }
        // (parameters of the right-hand side: fixed ones from the table, estimated ones from the trial point)
#pragma unroll
        for (int k = 0; k < RHS::NP; ++k) c.p[k] = dv.pp.Pfull[(size_t)b * dm.NP + k];
        for (int k = 0; k < NPe; ++k) {
            double v = X[TD + k];
            if (use_d) v = clampb(trial(v, stp, Dd[TD + k]), dv.pp, dm.ND + k);
            const int dst = dv.pp.Pidx[k];
#pragma unroll
            for (int j = 0; j < RHS::NP; ++j) c.p[j] = (dst == j) ? v : c.p[j];
        }
        __syncthreads();
        PZ_MARK(0);
        if (G > 1) {
            // halo exchange: my last two rows / my first row, as the neighbours will stage them
            for (int e = tid; e < PZ_EDGE_ROWS * D; e += PZ_THREADS) {
                const int r = e / D, j = e - r * D;
                const int lr = r < 2 ? HL + rows - 2 + r : HL;           // (every slice holds >= 2 rows)
                st_sc1(edge + e, xs[lr * D + j]);
            }
            if (!pz_grid_barrier(bar, tid, lds_ok)) return;               // ---- A
            if (w > 0) {
                const double *le = dv.pz.edge + ((size_t)b * G + w - 1) * (PZ_EDGE_ROWS * D);
                for (int e = tid; e < HL * D; e += PZ_THREADS) xs[e] = ld_sc1(le + (2 - HL) * D + e);
            }
            if (w + 1 < G) {
                const double *re = dv.pz.edge + ((size_t)b * G + w + 1) * (PZ_EDGE_ROWS * D);
                for (int e = tid; e < HR * D; e += PZ_THREADS) xs[(HL + T) * D + e] = ld_sc1(re + 2 * D + e);
            }
            if (pending_gd && tid == 0) {
                double v = 0.0;
                for (int t = 0; t < G; ++t) v += ld_sc1(all_gdp + t);
                hot->gd_dir = v;
            }
        }
        pending_gd = false;
        __syncthreads();
        PZ_MARK(1);

        // ---- evaluation, 2: the flat tile phases (va_core.h) on the staged rows
        ThreadAcc acc;
        acc.clear();
        tile_f<RHS, DISC>(dm, c, tid, PZ_THREADS);
        __syncthreads();
        tile_q<DISC>(dm, dv.pp, c, acc, tid, PZ_THREADS);
        __syncthreads();
        if (dv.pp.rf0_full) {
            tile_qfull<DISC>(dm, dv.pp, c, acc, tid, PZ_THREADS);
            double *t = c.qs; c.qs = c.fs; c.fs = t;
            __syncthreads();
        }
        tile_s<DISC>(dm, c, tid, PZ_THREADS);
        __syncthreads();
        tile_g<RHS, DISC>(dm, dv.pp, c, acc, tid, PZ_THREADS);
        if (dv.pp.rf0_full) { double *t = c.qs; c.qs = c.fs; c.fs = t; }
        PZ_MARK(2);
        pz_block_reduce_ev<K>(acc, red, evt, tid);
        PZ_MARK(3);
        if (G > 1) {
            if (tid < K) st_sc1(my_evp + tid, evt[tid]);
            if (!pz_grid_barrier(bar, tid, lds_ok)) return;               // ---- B
            if (tid < K) evt[tid] = col_reduce<true>(all_evp + tid, G, EP_N, 0, 1, tid == EP_GMAX);
            __syncthreads();
        }
        PZ_MARK(4);

        // ---- evaluation, 3: parameter tail of the gradient, then one line-search / ladder step (every workgroup, identically)
        if (tid == 0) {
            double ev[EP_N];
#pragma unroll
            for (int k = 0; k < EP_N; ++k) ev[k] = k < K ? evt[k] : 0.0;
            for (int k = 0; k < NPe; ++k) {
                const int idx = dv.pp.Pidx[k];
                double g = 0.0;
#pragma unroll
                for (int j = 0; j < RHS::NP; ++j) g = (idx == j) ? ev[EP_GP + j] : g;
                GT[TD + k] = g;
                if (use_d) ev[EP_GTD] += g * Dd[TD + k];
                ev[EP_GN2] += g * g;
                ev[EP_GMAX] = fmax(ev[EP_GMAX], fabs(g));
            }
            if (w == 0) atomicAdd(dv.n_evals, 1ULL);
            SeedResults r;
            r.ame = dv.ame + (size_t)b * dv.max_beta * 3;
            r.pest = nullptr;
            r.status = dv.status + (size_t)b * dv.max_beta;
            r.nit = dv.nit + (size_t)b * dv.max_beta;
            r.nfev = dv.nfev + (size_t)b * dv.max_beta;
            int dec = 0;
            double dirp[DP_N];
            dirp[DP_GD] = hot->gd_dir; dirp[DP_DD] = 0.0;
            ls_step(*hot, ev, dirp, dv.o, dv.rf_ladder, dv.nbeta, r, &dec, dm.cme, dm.cfe, false);
            if (dec && w == 0) atomicSub(dv.n_active, 1);
        }
        __syncthreads();
        PZ_MARK(5);

        // ---- update: x += stp*d, the new pair, g <- g_t, inner products (k_update's arithmetic on the own slice)
        const int upd = hot->upd, dir = hot->dir;
        if (!upd && !dir) continue;
        const bool hist = (upd & UPD_HIST) != 0;
        const int slot = hot->slot, nold = dir ? hot->nold : 0;
        const double stpu = hot->stp_upd;
        double *Sn = S + (size_t)slot * nv, *Yn = Y + (size_t)slot * nv;
        double *mp = nullptr;
        if ((upd & UPD_STORE) && dv.minpaths) mp = dv.minpaths + ((size_t)b * dv.max_beta + hot->store_idx) * (dm.ND + dm.NP);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
        for (int e = tid; e < nv; e += PZ_THREADS) {
            const double dv2 = Dd[e], gv = Gv[e], tv = GT[e];
            double xv = X[e];
            if (upd & UPD_X) {
                xv = clampb(trial(xv, stpu, dv2), dv.pp, e < TD ? (long)n0 * D + e : (long)dm.ND + (e - TD));
                X[e] = xv;
            }
            if (upd & UPD_STORE) {
                if (e < ne) { if (mp) mp[(size_t)n0 * D + e] = xv; }
                else if (e >= TD && w == 0) {
                    const int k = e - TD;
                    dv.pest[((size_t)b * dv.max_beta + hot->store_idx) * NPe + k] = xv;
                    if (mp) mp[dm.ND + dv.pp.Pidx[k]] = xv;
                }
            }
            double sv = 0.0, yv = 0.0;
            if (hist) { sv = stpu * dv2; yv = tv - gv; Sn[e] = sv; Yn[e] = yv; }
            if (upd & UPD_G) Gv[e] = tv;
            if (e < TD || count_p) {
                a0 += yv * tv; a1 += sv * tv; a2 += yv * yv; a3 += sv * yv; a4 += tv * tv;
            }
        }
        if (mp && w == 0 && tid == 0) {
            for (int j = 0; j < dm.NP; ++j) {         // fixed (non-estimated) parameters of the stored step come from P
                bool est = false;
                for (int k = 0; k < NPe; ++k) est = est || (dv.pp.Pidx[k] == j);
                if (!est) mp[dm.ND + j] = dv.pp.Pfull[(size_t)b * dm.NP + j];
            }
        }
        if (!dir) {
            if (tid == 0) hot->upd = 0;
            __syncthreads();
            continue;
        }
        {
            double v;
            v = wave_sum(a0); if (lane == 0) red[wave * UP_N + UP_YGT] = v;
            v = wave_sum(a1); if (lane == 0) red[wave * UP_N + UP_SGT] = v;
            v = wave_sum(a2); if (lane == 0) red[wave * UP_N + UP_YY] = v;
            v = wave_sum(a3); if (lane == 0) red[wave * UP_N + UP_SY] = v;
            v = wave_sum(a4); if (lane == 0) red[wave * UP_N + UP_GTGT] = v;
        }
        __syncthreads();                                  // (the new pair is in its slot)
        // inner products with the old pairs: wave v takes pairs v, v + 4, ...; its lanes stride the slice
        for (int j = wave; j < nold; j += PZ_WAVES) {
            const int sj = hot->order[j];
            const double *Sj = S + (size_t)sj * nv, *Yj = Y + (size_t)sj * nv;
            double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
            for (int e = lane; e < nv; e += 64) {
                if (e >= TD && !count_p) break;
                const double sv = Sj[e], yj = Yj[e], tv = GT[e], yv = hist ? Yn[e] : 0.0;
                b0 += sv * tv; b1 += yj * tv; b2 += sv * yv; b3 += yj * yv;
            }
            double v;
            v = wave_sum(b0); if (lane == 0) tot[UP_OLD + 4 * j + 0] = v;
            v = wave_sum(b1); if (lane == 0) tot[UP_OLD + 4 * j + 1] = v;
            v = wave_sum(b2); if (lane == 0) tot[UP_OLD + 4 * j + 2] = v;
            v = wave_sum(b3); if (lane == 0) tot[UP_OLD + 4 * j + 3] = v;
        }
        if (tid < UP_OLD) {
            double v = red[tid];
#pragma unroll
            for (int ww = 1; ww < PZ_WAVES; ++ww) v += red[ww * UP_N + tid];
            tot[tid] = v;
        }
        __syncthreads();
        const int KU = UP_OLD + 4 * nold;
        PZ_MARK(6);
        if (G > 1) {
            for (int k = tid; k < KU; k += PZ_THREADS) st_sc1(my_upp + k, tot[k]);
            if (!pz_grid_barrier(bar, tid, lds_ok)) return;               // ---- C
            for (int k = tid; k < KU; k += PZ_THREADS) tot[k] = col_reduce<true>(all_upp + k, G, UP_N, 0, 1, false);
            __syncthreads();
        }
        PZ_MARK(7);
        if (wave == 0) pz_coeffs(*hot, tot, sSY, sYY, sp, cYs, cSs, m, lane);
        __syncthreads();
        PZ_MARK(8);

        // ---- direction (k_direction's arithmetic on the own slice) and its g.d partial
        {
            const int col = hot->col;
            const double cg = hot->cg;
            double gd = 0.0;
            for (int e = tid; e < nv; e += PZ_THREADS) {
                const double gv = Gv[e];
                double a = cg * gv;
                for (int j = 0; j < col; ++j) {
                    const int sj = hot->order[j];
                    a += cYs[sj] * Y[(size_t)sj * nv + e];
                    a += cSs[sj] * S[(size_t)sj * nv + e];
                }
                Dd[e] = a;
                if (e < TD || count_p) gd += gv * a;
            }
            gd = wave_sum(gd);
            if (lane == 0) red[wave] = gd;
            __syncthreads();
            if (tid == 0) {
                const double v = ((red[0] + red[1]) + red[2]) + red[3];
                if (G > 1) st_sc1(dv.pz.gdp + (size_t)b * G + w, v);       // (travels with barrier A of the next evaluation)
                else hot->gd_dir = v;
                hot->upd = 0; hot->dir = 0;
            }
            pending_gd = true;
            __syncthreads();
        }
        PZ_MARK(9);
    }

    // ---- the final iterate back to the seed's global vector; state for the host's bookkeeping
    __syncthreads();
    {
        double *xg = dv.x + (size_t)b * dm.ld;
        for (int e = tid; e < nv; e += PZ_THREADS) {
            if (e < ne) xg[(size_t)n0 * D + e] = X[e];
            else if (e >= TD && w == 0) xg[dm.ND + (e - TD)] = X[e];
        }
        if (w == 0) {
            double *gst = reinterpret_cast<double *>(static_cast<SeedHot *>(&dv.st[b]));
            if (tid < (int)(sizeof(SeedHot) / 8)) gst[tid] = reinterpret_cast<double *>(hot)[tid];
            if (tid == 0) atomicAdd(dv.pz.cycles, (unsigned long long)cyc);
        }
    }
#ifdef VA_PZ_STAMPS
    if (blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 12; ++i) dv.pz.upp[i] = (double)pz_acc[i];
        dv.pz.upp[12] = (double)cyc;
    }
#endif
}

inline const void *seed_kernel_of(const void *const k[4], int disc)
{
    return k[disc == DISC_EULER ? 0 : disc == DISC_TRAPEZOID ? 1 : disc == DISC_SH ? 2 : 3];
}

// launch == false: opt the instantiation in to the LDS it needs on the current device (once per handle);
// launch == true: cooperative launch of dv.dm.B * dv.dm.ntiles workgroups (all resident, or the launch fails)
template <class RHS>
inline hipError_t seed_kernel_op(const Dev &dv, bool launch, hipStream_t s)
{
    const void *const ks[4] = {(const void *)k_seed<RHS, DISC_EULER>, (const void *)k_seed<RHS, DISC_TRAPEZOID>,
                               (const void *)k_seed<RHS, DISC_SH>, (const void *)k_seed<RHS, DISC_FWDMAP>};
    const void *k = seed_kernel_of(ks, dv.dm.disc);
    const int HL = dv.dm.disc == DISC_SH ? 2 : 1;
    const size_t lds = 8 * persist_lds_doubles(dv.dm.T, dv.dm.D, dv.dm.NPest, dv.dm.m, HL);
    if (!launch) return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    void *args[1] = {(void *)&dv};
    return hipLaunchCooperativeKernel(k, dim3(dv.dm.B * dv.dm.ntiles), dim3(PZ_THREADS), args, (unsigned)lds, s);
}

}  // namespace va
