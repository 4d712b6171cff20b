// va_capi.hip -- host side of libvaranneal_amd.so: the C-ABI of include/varanneal_amd.h.
//
// Owns the device image of one annealing problem (B seeds resident in HBM), moves
// paths in/out, and drives the three-launch kernel cycle
//     k_eval (+ line-search / ladder step) -> k_update (+ direction coefficients) -> k_direction
// until every seed has climbed its whole RF ladder.  No per-iteration host sync:
// the host only polls a device counter of unfinished seeds every few cycles.
#include <dlfcn.h>
#include <link.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/varanneal_amd.h"
#include "va_device.h"
#include "va_nnet.h"
#include "va_eval_flat.h"
#include "va_persist.h"

using namespace va;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(e_ == hipErrorOutOfMemory ? VA_ENOMEM : VA_EHIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                 \
    } while (0)

// generated right-hand-side modules (va_rhs_load_module); ids are VA_RHS_USER_BASE + index
struct UserRhs {
    // the ONE column-run instantiation the module carries besides its flat kernel, if any (va_user_rhs.hip):
    // (eval kernel 0 / 3 / 4 / 5, DISC, K, W_SCALAR [4] or threads [3], NE [4, 5], GHOST [3])
    int var[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (+ the column form's reaches xl, xr, gl, gr [5]; [10]: the flat kernel carries a dense linear part)
    void (*launch_var)(const Dev *, void *) = nullptr;
    int (*prepare_var)(const Dev *) = nullptr;
    std::string path;
    void *dl = nullptr;
    void (*launch)(const Dev *, void *) = nullptr;
    int (*prepare)(const Dev *) = nullptr;
    int (*seed_kernel)(const Dev *, int, void *) = nullptr;      // the persistent per-seed ladder kernel (va_persist.h), if the module carries it
    int NP = 0, D = 0, NSTIM = 0;
};
struct UserAct {
    std::string path;
    void *dl = nullptr;
    NnetActLaunch launch = nullptr;
};
std::vector<UserAct> g_user_act;      // (guarded by g_user_rhs_mutex, like the right-hand-side registry)
std::vector<UserRhs> g_user_rhs;
std::mutex g_user_rhs_mutex;            // the registry is process-wide; handles are not shared

}  // namespace

struct va_problem_s {
    Dev dv;
    int device = 0, rhs = 0, keep_paths = 0;
    void (*user_launch)(const Dev *, void *) = nullptr;
    int (*user_prepare)(const Dev *) = nullptr;
    int (*user_seed)(const Dev *, int, void *) = nullptr;
    NnetActLaunch user_act = nullptr;  // generated activation module's launcher (nn.act >= NNET_USER)
    // few seeds, short paths: the whole ladder in ONE launch, every vector of the minimisation resident in
    // the LDS of pz_G workgroups per seed (va_persist.h); chosen at create when the slices fit and all are co-resident
    bool persist = false, tune_persist = true;
    int pz_G = 0, pz_T = 0, pz_maxG = 0;
    void *pz_misc = nullptr;           // device: [abort flag (int), pad, cycles (unsigned long long), stamps (PZ_NSTAMP doubles)]
    size_t pz_xch_bytes = 0;
    bool is_nnet = false;              // feed-forward-network action (va_nnet.hip) instead of an ODE path
    bool fold = false;                 // the evaluation kernel runs the tail itself (last arriver of each seed)
    bool tune_graph = true;            // ladder cycles / timed evaluations replayed from a hipGraph (va_problem_tune)
    hipGraphExec_t timed_gexec = nullptr;   // va_eval_timed's chunk of launches
    int timed_chunk = 0;
    Dev timed_dv;                      // the device image the chunk was captured with (kernels take it by value):
    NnetDev timed_nn;                  //   the graph is replayed only while h->dv / h->nn still equal these bytes
    double timed_armed_rf = -1.0;      // >= 0: every seed sits in PH_START at this rf_scale (S1 launches leave it so)
    NnetDev nn;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<void *> allocs;
    int *h_nactive = nullptr;          // pinned: [0] live seeds; bytes 8..15: evaluation counter
    double *d_rf = nullptr;            // ladder on device [max_beta]
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int64_t n_eval_launch = 0, n_seed_evals = 0, n_seed_evals_direct = 0, n_cycles = 0;
    int last_nbeta = 0;

    template <class T> int alloc(T **p, size_t n, bool zero = true)
    {
        void *q = nullptr;
        size_t bytes = sizeof(T) * (n ? n : 1);
        HIPCHK(hipMalloc(&q, bytes));
        allocs.push_back(q);
        if (zero) HIPCHK(hipMemsetAsync(q, 0, bytes, stream));
        *p = (T *)q;
        return VA_OK;
    }
};

namespace {

// one batched evaluation.  epi: what the last-arriving wave of each seed does with the partial sums
// (EPI_FINALIZE: S1 outputs; EPI_LS: one line-search / ladder step).  The network action's
// evaluation is several kernels, so its tail stays a launch of its own.
void run_eval(va_handle h, int epi)
{
    if (epi != EPI_FINALIZE) h->timed_armed_rf = -1.0;      // (line-search launches move the seeds' states)
    h->dv.lsrun = epi == EPI_LS ? 1 : 0;       // (S1 evaluations put every seed in PH_START: no line-search points)
    if (h->is_nnet ? !h->nn.small : !h->fold) {
        // (large grids: a workgroup that waits for its arrival to come back holds its LDS and wave
        // slots ~1 us longer, which costs more than the 64-wave tail kernel it saves)
        h->dv.epi = EPI_NONE;
        if (h->is_nnet) launch_nnet_eval(h->dv, h->nn, h->stream, h->user_act);
        else if (h->user_launch) h->user_launch(&h->dv, (void *)h->stream);
        else launch_eval(h->dv, h->rhs, h->stream);
        if (epi == EPI_FINALIZE) launch_finalize_eval(h->dv, h->stream);
        else if (epi == EPI_LS) launch_ls(h->dv, h->stream);
        return;
    }
    h->dv.epi = epi;
    if (h->is_nnet) { launch_nnet_eval(h->dv, h->nn, h->stream, h->user_act); return; }   // (small nets: k_nnet_small carries the tail)
    if (h->user_launch) h->user_launch(&h->dv, (void *)h->stream);
    else launch_eval(h->dv, h->rhs, h->stream);
}

// Tile geometry of the eval kernel: which mapping, rows per workgroup, threads.
// ne: products per element of the model's column form (RhsL96s::NE for the built-in; a generated
// module's RhsUserCol::NE), or 0 when the model has none; ghost: ghost columns per side of its ghosted
// form for the workgroup column-run kernel (RhsL96g::GHOST; a module's RhsUserG::GHOST), or 0.  A model
// with neither runs the flat kernel.
// reach5: {xl, xr, gl, gr} of a column form that the streaming kernel (emode 5, va_tile5.h) may run, or NULL.
void pick_eval_geometry(const va_problem_desc *d, Dims &dm, Geo4 &g4, int ne, int ghost,
                        const int *reach5 = nullptr, Geo5 *g5 = nullptr, std::vector<int> *ystrip = nullptr)
{
    bool user_rhs = ne <= 0 && ghost <= 0;
    dm.ghost = ghost > 0 ? ghost : 2;
    user_rhs = user_rhs || d->p_time_dependent || d->rm_kind == 2 || d->rf_kind == 2;   // per-row parameters / full RM or RF matrices: flat kernel only
    user_rhs = user_rhs || (d->lower && d->upper);                   // box bounds: the flat kernel carries the clamp / projected gradient
    const int D = d->D, N = d->N_model;
    const bool sh = d->disc == VA_DISC_SIMPSON_HERMITE;
    const int HLR = sh ? 3 : 2;
    dm.emode = d->eval_kernel;
    // auto: wave-private column runs for narrow states that fill a wave, workgroup column runs up
    // to 1024 columns, flat mapping beyond
    if (dm.emode == 2) dm.emode = 3;                      // (the row-strided kernel of round 1 is gone)
    // streaming column strips: wide even states, a column form, scalar or per-row weights with
    // data at every model time (what every BASELINE config has); anything else keeps the tile kernels
    const bool ws5 = d->rm_kind <= 1 && d->rf_kind <= 1 && d->merr_nskip >= 1 && d->L >= 1;     // (full matrices: flat kernel)
    const bool can5 = reach5 && g5 && ystrip && !user_rhs && ne > 0 && ws5 && D > 64 && (!sh || (N & 1)) &&
                      tile5_ok(D, reach5[0], reach5[1], reach5[2], reach5[3]);
    if (dm.emode == 5 && !can5) dm.emode = 0;
    if (dm.emode < 1 || dm.emode > 5) dm.emode = (tile4_ok(D) && ne > 0) ? 4 : (can5 ? 5 : ((D <= 1024 && ghost > 0) ? 3 : 1));
    if (user_rhs) dm.emode = 1;                           // no column form (or a case only the flat kernel carries)
    if (dm.emode == 5) {
        Geo5 g = tile5_cols(D, reach5[0], reach5[1], reach5[2], reach5[3]);
        g.ne = ne;
        // segments: as many workgroups as the chip holds at once (four 4-wave groups per CU: 128 registers, 40 KiB
        // of LDS each), every one with the same number of rows; at least 32 rows per segment
        const long per_row = (long)d->batch * g.NSG;
        long nseg = (4L * 256) / per_row;
        if (d->tile_rows > 0) nseg = (N + d->tile_rows - 1) / d->tile_rows;
        if (nseg > N / 32) nseg = N / 32;
        if (nseg < 1) nseg = 1;
        g.SEGL = (int)((N + nseg - 1) / nseg);
        g.SEGL = (g.SEGL + 1) & ~1;
        g.NSEG = (N + g.SEGL - 1) / g.SEGL;
        // observation rows per strip: the data columns of the strip's own state columns (Lidx ascending on the device)
        std::vector<int> ls(d->Lidx, d->Lidx + d->L);
        std::sort(ls.begin(), ls.end());
        ystrip->assign(2 * g.NS, 0);
        g.YPMAX = 1;
        for (int s5 = 0; s5 < g.NS; ++s5) {
            const int c0 = tile5_c0(D, g.NS, s5), c1 = tile5_c0(D, g.NS, s5 + 1);
            const int l0 = (int)(std::lower_bound(ls.begin(), ls.end(), c0) - ls.begin());
            const int l1 = (int)(std::lower_bound(ls.begin(), ls.end(), c1) - ls.begin());
            int start = l0 & ~1;
            int yp = (l1 - start + 1) / 2;
            if (yp < 1) yp = 1;                           // (every staging instruction has an active lane: the queue counts are exact)
            (*ystrip)[2 * s5] = start; (*ystrip)[2 * s5 + 1] = yp;
            if (yp > g.YPMAX) g.YPMAX = yp;
        }
        // ring depth: four slots (three requested ahead) while four workgroups still share a CU's 160 KiB, else three.
        // Measured at C4 (profiles/r03_e5_experiments.txt): 3 and 4 slots equal; 6 slots drop a workgroup per CU (+33 %)
        g.xdpp = (reach5[2] <= 2 && reach5[3] <= 2) ? 1 : 0;
        auto fits = [&](int nslot, bool lsr) { return (size_t)g.WPG * tile5_wave_doubles(g, nslot, lsr) * sizeof(double) <= 40 * 1024; };
        g.nslot = fits(4, false) ? 4 : 3;
        g.nslot_ls = fits(4, true) ? 4 : 3;
        g.warr = (d->rm_kind == 1 || d->rf_kind == 1 || d->merr_nskip > 1) ? 1 : 0;
        if (g.warr) g.nslot = g.nslot_ls = 3;          // (two more images per slot; only the three-slot instantiations exist)
        g.LY = (d->L + 1) & ~1;                        // (data rows are staged by 16-byte pieces: an odd L gets a pad column on the device)
        if (g.YPMAX <= 32) {
            *g5 = g;
            dm.RY = 0; dm.NT = 64 * g.WPG; dm.maxr = 2; dm.T = g.SEGL;
            dm.ntiles = g.NSEG * g.NSG;
            return;
        }
        dm.emode = (D <= 1024 && ghost > 0) ? 3 : 1;
    }
    if (dm.emode == 4 && (!tile4_ok(D) || ne <= 0)) dm.emode = 3;
    if (dm.emode == 3 && ghost <= 0) dm.emode = 1;
    if (dm.emode == 3 && D > 1024) dm.emode = 1;          // column runs: a lane per column
    int tmin, tmax;
    if (dm.emode == 4) {
        // wave-private column runs: T = 4 waves x RW runs x K rows.  K is the run length that gives
        // every CU the same number of workgroups when the grid is only a few per CU (C3: 64 seeds,
        // N = 1000: K = 7 -> 12 tiles x 64 = 768 = 3 x 256), 6 otherwise
        const int RW = 64 / D, rows1 = 4 * RW;
        int K = 6;
        auto ntl = [&](int k) { return (long)d->batch * ((N + rows1 * k - 1) / (rows1 * k)); };
        if (ntl(K) < 256) K = 4;
        else if (ntl(K) >= 8 * 256) {
            // many rounds of workgroups: the run length that stages the fewest rows (own + halo) per seed,
            // longest on ties, up to 7 (8 drops the kernel to two waves per SIMD).  N = 1000, D = 20: K = 7
            // (12 tiles x 9 rows per lane against 14 x 8 for K = 6): 370 vs 409 us at 4096 seeds
            long best = -1;
            for (int k = 5; k <= 7; ++k) {
                if (sh && (k & 1)) continue;
                const long cost = (long)((N + rows1 * k - 1) / (rows1 * k)) * (k + (sh ? 3 : 2));
                if (best < 0 || cost <= best) { best = cost; K = k; }
            }
        } else {
            // a few workgroups per CU, all resident at once: the busiest CU sets the time.  Pick the K whose (workgroups per
            // CU, rounded up) x (rows per lane + fixed per-workgroup cost) is smallest, e.g. C3 (64 seeds, N = 1000):
            // K = 7 gives 12 tiles x 64 = 768 workgroups = exactly 3 per CU.  Simpson-Hermite (even K; 8 runs at two waves
            // per SIMD, so only two workgroups per CU are resident): rounds of resident workgroups x rows per lane --
            // N = 1001: K = 4 -> 2 rounds x 6 rows (11.4 us) against 2 x 8 at K = 6 (11.7) and 2 x 10 at K = 8 (12.1)
            long best = -1;
            for (int k = sh ? 4 : 5; k <= 8; ++k) {
                if (sh && (k & 1)) continue;              // Simpson-Hermite runs start on even rows
                const long per_round = sh ? 256L * (k <= 7 ? 3 : 2) : 256L;
                const long cost = ((ntl(k) + per_round - 1) / per_round) * (k + 2) * 4 + (k == 6 ? 0 : 1);     // ties go to 6
                if (best < 0 || cost < best) { best = cost; K = k; }
            }
            // Simpson-Hermite, D = 20, scalar weights: runs of 12 rows (two workgroups per CU, no spill) when they put the whole
            // grid in ONE round of resident workgroups -- a launch of this size is a chain of latencies, not of rows: N = 1001,
            // 64 seeds: K = 4 -> 21 tiles, 1.75 rounds, 11.3 us; K = 12 -> 7 tiles, 448 workgroups, 9.4 us (trapezoid at K = 7: 8.6)
            // (measured for the built-in right-hand side: a generated model keeps the chooser's K unless asked)
            if (sh && D == 20 && d->rhs == VA_RHS_LORENZ96 && d->rm_kind == 0 && d->rf_kind == 0 && ntl(12) <= 2 * 256 && ntl(K) > 3 * 256) K = 12;
        }
        if (d->tile_rows > 0) {
            K = (d->tile_rows + rows1 - 1) / rows1;
            const bool k12 = D == 20 && d->rm_kind == 0 && d->rf_kind == 0 && (d->merr_nskip == 1 || d->rhs == VA_RHS_LORENZ96);     // (the one longer run compiled)
            K = K < 4 ? 4 : (K >= 12 && k12 ? 12 : (K > 8 ? 8 : K));
            if (sh && (K & 1)) ++K;
        }
        // weight arrays / data every nskip-th row: runs of 6 and 7 rows do not fit three waves per SIMD's 168 registers
        // with their weight registers (35-58 spilled; measured at the C3 shape, profiles/r03_f3_variants.txt: 21 us
        // against 13 us for K = 5): those problems run K <= 5, or 8 at two waves per SIMD -- except the built-in
        // right-hand side at D = 20, whose instantiations park the RF weights in LDS, fold the RM weights into the data
        // registers before the f evaluations (va_tile4.h) and mask merr_nskip's rows by a bit each
        {
            const bool ws4 = (d->rm_kind == 0 && d->rf_kind == 0 && d->merr_nskip == 1) || (D == 20 && d->rhs == VA_RHS_LORENZ96);
            if (!ws4 && (K == 6 || K == 7)) K = sh ? 4 : 5;
        }
        // column forms with many products per element (a ring of coupled units: 8): the product arrays grow with the run
        // length; keep at least two workgroups per CU (measured, five 4-state units at the C3 shape: K = 4 23.4 us, K = 7 31.5)
        if (d->tile_rows <= 0) {
            while (K > 4) {
                const Geo4 gt = sh ? tile4_geo<3>(D, K, ne, 1) : tile4_geo<2>(D, K, ne, 1);
                if (sizeof(double) * (size_t)gt.NW * gt.WAVE <= 80 * 1024) break;
                K -= (sh || K == 5) ? (K == 5 ? 1 : 2) : 1;
            }
        }
        // (one wave per SIMD walking SUB sub-tiles in turn was measured slower than co-resident waves -- DESIGN.md section 7;
        // only SUB = 1 is instantiated, and the host never asks for anything else)
        const int SUB = 1;
        g4 = sh ? tile4_geo<3>(D, K, ne, SUB) : tile4_geo<2>(D, K, ne, SUB);
        // (D = 20 is compiled with its geometry constant: the kernel sizes its staging loop exactly)
        if ((D == 20 || (g4.XP + 63) / 64 <= T4_NI_MAX) && tile4_magic_ok(g4)) {
            dm.RY = 4 * RW; dm.NT = 256; dm.maxr = K; dm.T = g4.T;
            dm.ntiles = (N + dm.T - 1) / dm.T;
            return;
        }
        dm.emode = ghost > 0 ? 3 : 1;
    }
    if (dm.emode == 3) {
        // column-run kernel: T = RY*K exactly, K rows per lane in {4, 6, 8}
        dm.RY = tile3_RY(D); dm.NT = tile3_threads(D);
        // K = 6 keeps the kernel at 128 VGPRs (4 waves/SIMD) and measured best from 64 to 4096
        // seeds (profiles/r01_sweep_*.txt); drop to 4 when that leaves CUs without a workgroup
        int K = 6;
        if ((long)d->batch * ((N + dm.RY * K - 1) / (dm.RY * K)) < 256) K = 4;
        else if ((long)d->batch * ((N + dm.RY * K - 1) / (dm.RY * K)) < 8 * 256 && dm.NT == 256) {
            // small grids run as a handful of workgroups per CU, all resident at once: the busiest
            // CU sets the time.  Pick the K whose (workgroups per CU, rounded up) x (rows per lane +
            // fixed per-workgroup cost) is smallest, e.g. C3 (64 seeds, N = 1000): K = 7 gives
            // 12 tiles x 64 = 768 workgroups = exactly 3 per CU (10.2 us) against 3.5 for K = 6 (10.7 us).
            long best = -1;
            for (int k = 5; k <= 8; ++k) {
                if (sh && (k & 1)) continue;              // Simpson-Hermite runs start on even rows
                const long wgs = (long)d->batch * ((N + dm.RY * k - 1) / (dm.RY * k));
                const long cost = ((wgs + 255) / 256) * (k + 2) * 4 + (k == 6 ? 0 : 1);     // ties go to 6
                if (best < 0 || cost < best) { best = cost; K = k; }
            }
        }
        if (d->tile_rows > 0) {
            K = (d->tile_rows + dm.RY - 1) / dm.RY;
            K = K < 4 ? 4 : (K > 8 ? 8 : K);
            if (sh && (K & 1)) ++K;
        }
        if (D > 64 && d->tile_rows <= 0 && (long)d->batch * ((N + dm.RY * 8 - 1) / (dm.RY * 8)) >= 256)
            K = 8;                                         // few lanes per column: long runs keep the halo share down
        if (dm.NT == 1024 && K > 6) K = 6;                 // (1024-thread groups live on 128 registers: runs of 8 rows spill 20-88 of them)
        for (;;) {                                        // shrink until the staging arrays fit in LDS
            const size_t elems = (size_t)tile3_stage_elems(K, D, dm.ghost, dm.RY, HLR) + tile3_s_elems(K, D, dm.ghost, dm.RY);
            if (sizeof(double) * elems <= (D <= 64 ? 60 : (D <= 512 ? 78 : 150)) * 1024 || K <= 4) break;   // two groups per CU (one beyond D = 512)
            K -= (sh || K == 5) ? (K == 5 ? 1 : 2) : 1;  // Simpson-Hermite keeps K even; never below 4
        }
        dm.maxr = K; dm.T = dm.RY * K;
        dm.ntiles = (N + dm.T - 1) / dm.T;
        return;
    }
    if (dm.emode == 2) {
        dm.RY = tile2_RY(D); dm.NT = tile2_threads(D);
        const int narr = sh ? 3 : 2;
        const int lds_rows = (int)((60 * 1024) / (narr * sizeof(double) * D));
        // 8 rows per lane keeps the register tile small; go to 16 when that would leave
        // tiles so short that the halo rows dominate (large D)
        dm.maxr = (8 * dm.RY - HLR >= 24) ? 8 : 16;
        int rmax = dm.maxr * dm.RY;
        if (rmax > lds_rows) rmax = lds_rows;
        tmax = rmax - HLR;
        tmin = dm.RY;
    } else {
        dm.RY = 0; dm.NT = EVAL_THREADS; dm.maxr = 0;
        // LDS: 3 staged arrays of (T+halo) rows (4 when the right-hand side has a dense linear part: J^T s of it)
        // ~24 KiB per workgroup (six per CU) measured best (D = 100: T = 8, 349 us against 403 us at
        // T = 18); wider states take 48 KiB, then whatever still gives two owned rows
        const size_t narr = 3 + (dm.lin ? 1 : 0);
        auto rows_in = [&](size_t kib) { return (int)((kib * 1024) / (narr * sizeof(double) * D)) - HLR; };
        tmax = rows_in(24);
        if (dm.lin) {
            // the matrix cores take 16 staged rows at a time and every workgroup reads the whole table of the linear
            // part per product: the smallest budget that stages >= 16 rows, up to 80 KiB (two workgroups per CU)
            for (size_t kib : {24, 48, 80}) { tmax = rows_in(kib); if (tmax + HLR >= 16) break; }
        }
        if (tmax < 2) tmax = rows_in(48);
        if (tmax < 2) tmax = rows_in(150);
        if (dm.lin && d->tile_rows > tmax && d->tile_rows <= rows_in(150)) tmax = d->tile_rows;     // (an explicit run length may take the CU's whole LDS)
        tmin = (EVAL_THREADS + D - 1) / D;               // >= one element per lane
    }
    if (tmax < 2) tmax = 2;
    if (tmin > tmax) tmin = tmax;
    int T;
    if (d->tile_rows > 0) T = d->tile_rows < tmax ? d->tile_rows : tmax;
    else {
        // enough workgroups to cover 256 CUs a few times over
        int want = (1024 + d->batch - 1) / d->batch;     // tiles per seed
        T = N / (want > 0 ? want : 1);
        if (T < tmin) T = tmin;
        if (T > tmax) T = tmax;
    }
    if (T > N) T = N;
    if (sh && (T & 1)) T += (T + 1 <= tmax) ? 1 : -1;
    if (T < 2) T = 2;
    dm.T = T;
    dm.ntiles = (N + T - 1) / T;
}

// per-seed vectors, L-BFGS history, partial tables and result tables: the part of the device
// image that does not depend on which action is being minimised
int alloc_solver_state(va_handle h, int max_beta, int keep_paths)
{
    Dev &dv = h->dv;
    const Dims &dm = dv.dm;
    const size_t B = dm.B, ld = dm.ld, m = dm.m;
    int rc;
#define TRYA(x) do { rc = (x); if (rc) return rc; } while (0)
    // x and d carry a zero-filled guard in front and behind: the evaluation kernels stage whole
    // tiles (+ halo rows) without clamping, so the first / last tile of the first / last seed reads
    // up to one tile beyond its path (such rows are masked out of the arithmetic)
    const size_t guard = (((size_t)(dm.T + 8) * dm.D + 15) / 16) * 16;
    TRYA(h->alloc(&dv.x, B * ld + 2 * guard)); TRYA(h->alloc(&dv.g, B * ld));
    TRYA(h->alloc(&dv.gt, B * ld)); TRYA(h->alloc(&dv.d, B * ld + 2 * guard));
    dv.x += guard; dv.d += guard;
    TRYA(h->alloc(&dv.S, B * m * ld)); TRYA(h->alloc(&dv.Y, B * m * ld));
    TRYA(h->alloc(&dv.st, B));
    TRYA(h->alloc(&dv.evp, B * dm.nprow * EP_N));
    dv.npbig = (!h->is_nnet && dm.NPt > RHS_MAX_NP) ? dm.NPt - RHS_MAX_NP : 0;
    if (dv.npbig) TRYA(h->alloc(&dv.evp_big, B * dm.nprow * dv.npbig));
    TRYA(h->alloc(&dv.upp, B * dm.nchunks * dv.ups));
    TRYA(h->alloc(&dv.dpp, B * dm.nchunks * DP_N));
    TRYA(h->alloc(&h->d_rf, (size_t)max_beta));
    TRYA(h->alloc(&dv.ame, B * max_beta * 3));
    TRYA(h->alloc(&dv.pest, B * max_beta * (dm.NPest ? dm.NPest : 1)));
    TRYA(h->alloc(&dv.status, B * max_beta)); TRYA(h->alloc(&dv.nit, B * max_beta));
    TRYA(h->alloc(&dv.nfev, B * max_beta));
    if (keep_paths) TRYA(h->alloc(&dv.minpaths, B * max_beta * (size_t)(dm.ND + dm.NP), false));
    if (dm.bounded) {
        TRYA(h->alloc(&dv.lb_z, B * ld)); TRYA(h->alloc(&dv.lb_r, B * ld)); TRYA(h->alloc(&dv.lb_xp, B * ld));
        TRYA(h->alloc(&dv.lb_t, B * ld)); TRYA(h->alloc(&dv.lb_iwhere, B * ld));
        TRYA(h->alloc(&dv.lb_mat, B * 3 * m * m)); TRYA(h->alloc(&dv.lb_dtd, B));
    }
    TRYA(h->alloc(&dv.cnt_eval, B * CNT_STRIDE)); TRYA(h->alloc(&dv.cnt_upd, B * CNT_STRIDE)); TRYA(h->alloc(&dv.cnt_dir, B * CNT_STRIDE));
    TRYA(h->alloc(&dv.n_active, 1));
    TRYA(h->alloc(&dv.n_evals, 1));
    TRYA(h->alloc(&dv.outA, B)); TRYA(h->alloc(&dv.outme, B)); TRYA(h->alloc(&dv.outfe, B));
#undef TRYA
    dv.rf_ladder = h->d_rf;
    return VA_OK;
}

// pinned poll word, timing events, initial seed states
int finish_create(va_handle h)
{
    hipError_t e = hipHostMalloc((void **)&h->h_nactive, 4 * sizeof(int), hipHostMallocDefault);
    if (e != hipSuccess) return fail(VA_ENOMEM, "hipHostMalloc: %s", hipGetErrorString(e));
    e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) return fail(VA_EHIP, "hipEventCreate: %s", hipGetErrorString(e));
    e = hipStreamSynchronize(h->stream);     // host staging buffers must be consumed before we return
    if (e != hipSuccess) return fail(VA_EHIP, "create sync: %s", hipGetErrorString(e));
    launch_init_states(h->dv, PH_IDLE, 1.0, h->stream);
    return VA_OK;
}

int copy_in(va_handle h, const double *XP, int64_t ld, int32_t mem)
{
    const Dims &dm = h->dv.dm;
    const size_t w = sizeof(double) * (dm.ND + dm.NPest);
    HIPCHK(hipMemcpy2DAsync(h->dv.x, sizeof(double) * dm.ld, XP, sizeof(double) * ld, w, dm.B,
                            mem == VA_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                            h->stream));
    return VA_OK;
}

int copy_out(va_handle h, const double *src, double *dst, int64_t ld, int32_t mem)
{
    const Dims &dm = h->dv.dm;
    const size_t w = sizeof(double) * (dm.ND + dm.NPest);
    HIPCHK(hipMemcpy2DAsync(dst, sizeof(double) * ld, src, sizeof(double) * dm.ld, w, dm.B,
                            mem == VA_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                            h->stream));
    return VA_OK;
}

int check_xp(va_handle h, const void *XP, int64_t ld, int32_t mem)
{
    if (!h) return fail(VA_EINVAL, "null handle");
    if (!XP) return fail(VA_EINVAL, "XP is NULL");
    if (ld < h->dv.dm.ND + h->dv.dm.NPest) return fail(VA_EINVAL, "ld (%lld) < n_var (%d)", (long long)ld, h->dv.dm.ND + h->dv.dm.NPest);
    if (mem != VA_MEM_HOST && mem != VA_MEM_DEVICE) return fail(VA_EINVAL, "bad mem kind %d", mem);
    return VA_OK;
}

int set_opts(va_handle h, const va_lbfgs_opts *o)
{
    if (!o) return fail(VA_EINVAL, "opts is NULL");
    if (o->maxls <= 0) return fail(VA_EINVAL, "maxls must be positive");
    if (o->maxcor <= 0) return fail(VA_EINVAL, "maxcor must be positive");
    Opts &d = h->dv.o;
    d.m = o->maxcor < h->dv.dm.m ? o->maxcor : h->dv.dm.m;
    d.maxiter = o->maxiter; d.maxls = o->maxls; d.maxfun = o->maxfun;
    d.ftol = o->ftol; d.gtol = o->gtol;
    return VA_OK;
}

// Is a rocprofiler-sdk tool (rocprofv3, rocprof-compute) loaded into this process?  Replaying ONE hipGraphExec of 192
// kernel nodes ~900 times under rocprofv3 --kernel-trace ends in a SIGSEGV 13 frames below hipGraphLaunch, inside the
// runtime / tool libraries (round 4: tools/ex1_probe.py --nbeta 101 --graph 1 under the profiler, at a 1 MiB-aligned
// address; the same ladder with plain launches under the profiler, and with the graph without it, runs).  A fault
// inside a HIP call cannot be turned into an error code, so the ladder does not replay graphs while such a tool is
// attached -- the profiler then also sees every kernel as a dispatch of its own.  (HIP itself links only
// librocprofiler-register.so; the sdk and its tool library arrive with the profiler.)
bool profiler_attached()
{
    static const bool attached = [] {
        bool found = false;
        dl_iterate_phdr([](struct dl_phdr_info *info, size_t, void *out) -> int {
            const char *n = info->dlpi_name;
            if (n && (strstr(n, "librocprofiler-sdk.so") || strstr(n, "librocprofiler-sdk-tool"))) { *(bool *)out = true; return 1; }
            return 0;
        }, &found);
        return found;
    }();
    return attached;
}

// the kernel cycle until no seed is left (or the evaluation budget bound is hit)
int run_ladder_persist(va_handle h, const double *rf_scale, int nbeta, bool *fell_back);

int run_ladder(va_handle h, const double *rf_scale, int nbeta)
{
    Dev &dv = h->dv;
    if (nbeta < 1 || nbeta > dv.max_beta) return fail(VA_EINVAL, "nbeta=%d outside [1, max_beta=%d]", nbeta, dv.max_beta);
    if (h->persist && h->tune_persist) {
        bool fell_back = false;
        const int rc = run_ladder_persist(h, rf_scale, nbeta, &fell_back);
        if (!fell_back) return rc;
    }
    HIPCHK(hipMemcpyAsync(h->d_rf, rf_scale, sizeof(double) * nbeta, hipMemcpyHostToDevice, h->stream));
    dv.nbeta = nbeta; h->last_nbeta = nbeta;
    *h->h_nactive = dv.dm.B;
    HIPCHK(hipMemcpyAsync(dv.n_active, h->h_nactive, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(dv.dpp, 0, sizeof(double) * dv.dm.B * dv.dm.nchunks * DP_N, h->stream));
    launch_init_states(dv, PH_START, -1.0, h->stream);
    h->timed_armed_rf = -1.0;
    if (dv.dm.bounded) launch_clamp_x(dv, h->stream);
    // every cycle costs each live seed at least one evaluation
    const double per_step = (double)dv.o.maxfun + dv.o.maxls + 4.0;
    const double bound = per_step * nbeta;
    long long max_cycles = bound > 4e18 ? (long long)4e18 : (long long)bound;
    long long cyc = 0;
    int poll = 4;
    // Long ladders on small problems are bound by the host's launch rate (3 launches per cycle at
    // ~4 us each against ~25 us of device time): once the polling interval has grown to 64 cycles,
    // that batch of 192 launches is captured ONCE into a hipGraph and replayed.  The kernels take the device image
    // by value, so the graph is private to this call (ladder length, options).
    hipGraphExec_t gexec = nullptr;
    bool use_graph = h->tune_graph && !profiler_attached();
    auto enqueue = [&](int n) {
        for (int k = 0; k < n; ++k) {
            run_eval(h, EPI_LS);
            launch_update(dv, h->stream);
            if (dv.dm.bounded) launch_lbfgsb_dir(dv, h->stream);      // L-BFGS-B: Cauchy point + subspace minimisation
            else launch_direction(dv, h->stream);
        }
    };
    struct GraphGuard { hipGraphExec_t &g; ~GraphGuard() { if (g) (void)hipGraphExecDestroy(g); } } guard{gexec};
    for (;;) {
        if (poll == 64 && use_graph) {
            if (!gexec) {
                // any failure here just means plain launches for the rest of this call
                hipGraph_t g = nullptr;
                if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    enqueue(poll);
                    if (hipStreamEndCapture(h->stream, &g) != hipSuccess || !g ||
                        hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0) != hipSuccess) gexec = nullptr;
                    if (g) (void)hipGraphDestroy(g);
                }
                if (!gexec) { (void)hipGetLastError(); use_graph = false; }
            }
            if (gexec) HIPCHK(hipGraphLaunch(gexec, h->stream));
            else enqueue(poll);
        } else enqueue(poll);
        cyc += poll;
        HIPCHK(hipGetLastError());            // a failed launch surfaces here, with its cause, not as a stalled ladder
        HIPCHK(hipMemcpyAsync(h->h_nactive, dv.n_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(h->h_nactive + 2, dv.n_evals, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (*h->h_nactive <= 0) break;
        if (cyc > max_cycles) return fail(VA_ESTATE, "ladder did not finish within %lld cycles", max_cycles);
        if (poll < 64) poll *= 2;
    }
    h->n_cycles += cyc; h->n_eval_launch += cyc;
    h->n_seed_evals = h->n_seed_evals_direct + (int64_t)*(unsigned long long *)(h->h_nactive + 2);
    HIPCHK(hipGetLastError());
    return VA_OK;
}

// The same ladder as ONE launch of the persistent per-seed kernel.  *fell_back: the launch was refused, or its workgroups
// turned out not to be all resident (e.g. another process holds CUs): nothing was kept, take the three-launch cycle.
int run_ladder_persist(va_handle h, const double *rf_scale, int nbeta, bool *fell_back)
{
    Dev &dv = h->dv;
    *fell_back = false;
    const double per_step = (double)dv.o.maxfun + dv.o.maxls + 4.0;
    const double bound = per_step * nbeta;
    const long long max_cycles = bound > 4e18 ? (long long)4e18 : (long long)bound;
    Dev dvp = dv;
    dvp.dm.T = h->pz_T; dvp.dm.ntiles = h->pz_G; dvp.dm.nprow = h->pz_G;
    dvp.nbeta = nbeta; dvp.pz.max_cycles = max_cycles;
    HIPCHK(hipMemsetAsync(dv.pz.xch, 0, h->pz_xch_bytes, h->stream));           // (tags restart at 1 with every launch)
    HIPCHK(hipMemsetAsync(h->pz_misc, 0, 16, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_rf, rf_scale, sizeof(double) * nbeta, hipMemcpyHostToDevice, h->stream));
    dv.nbeta = nbeta; h->last_nbeta = nbeta;
    *h->h_nactive = dv.dm.B;
    HIPCHK(hipMemcpyAsync(dv.n_active, h->h_nactive, sizeof(int), hipMemcpyHostToDevice, h->stream));
    launch_init_states(dv, PH_START, -1.0, h->stream);
    h->timed_armed_rf = -1.0;
    const hipError_t e = h->user_seed ? (hipError_t)h->user_seed(&dvp, 1, (void *)h->stream) : seed_kernel_builtin(dvp, true, h->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        h->persist = false;           // (for the rest of this handle's life)
        *fell_back = true;
        return VA_OK;
    }
    struct { int abort_flag, pad; unsigned long long cycles; } misc = {0, 0, 0ull};
    HIPCHK(hipMemcpyAsync(h->h_nactive, dv.n_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_nactive + 2, dv.n_evals, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&misc, h->pz_misc, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    const long long cyc = (long long)(misc.cycles / (unsigned long long)(dv.dm.B > 0 ? 1 : 1));
    h->n_cycles += cyc; h->n_eval_launch += cyc;
    h->n_seed_evals = h->n_seed_evals_direct + (int64_t)*(unsigned long long *)(h->h_nactive + 2);
    if (misc.abort_flag == 1) {
        // a poll timed out: the seed's workgroups were not all resident.  Nothing was written back (x still holds the start
        // point, the result tables are rewritten from rung 0): this handle takes the three-launch cycle from now on
        h->persist = false;
        *fell_back = true;
        return VA_OK;
    }
    if (misc.abort_flag == 2) return fail(VA_ESTATE, "ladder did not finish within %lld cycles", max_cycles);
    if (*h->h_nactive > 0) return fail(VA_ESTATE, "persistent ladder ended with %d live seeds", *h->h_nactive);
    return VA_OK;
}

template <class T>
int fetch_table(va_handle h, const T *dev, T *host, int nbeta, int per)
{
    if (!host) return VA_OK;
    const Dev &dv = h->dv;
    HIPCHK(hipMemcpy2DAsync(host, sizeof(T) * nbeta * per, dev, sizeof(T) * dv.max_beta * per,
                            sizeof(T) * nbeta * per, dv.dm.B, hipMemcpyDeviceToHost, h->stream));
    return VA_OK;
}

}  // namespace

extern "C" {

int32_t va_abi_version(void) { return VA_ABI_VERSION; }
const char *va_last_error(void) { return g_err.c_str(); }

int va_device_count(int32_t *count)
{
    if (!count) return fail(VA_EINVAL, "count is NULL");
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    *count = n;
    return VA_OK;
}

int va_eval_plan_reach(const va_problem_desc *d, int32_t ne, int32_t ghost, const int32_t *reach, int32_t *out)
{
    if (!d || !out) return fail(VA_EINVAL, "null argument");
    if (d->struct_size != (int32_t)sizeof(va_problem_desc)) return fail(VA_EINVAL, "struct_size %d != %zu", d->struct_size, sizeof(va_problem_desc));
    out[0] = out[1] = out[2] = out[3] = 0;
    if ((ne <= 0 && ghost <= 0) || d->D < 1 || d->N_model < 2 || d->batch < 1) return VA_OK;
    Dims dm{};
    Geo4 g4{};
    Geo5 g5{};
    std::vector<int> ys;
    int r5[4] = {0, 0, 0, 0};
    if (reach) for (int k = 0; k < 4; ++k) r5[k] = reach[k];
    if (d->L > 0 && !d->Lidx) return fail(VA_EINVAL, "Lidx is NULL");
    pick_eval_geometry(d, dm, g4, ne, ghost, reach ? r5 : nullptr, &g5, &ys);
    if (dm.emode != 3 && dm.emode != 4 && dm.emode != 5) return VA_OK;
    out[0] = dm.emode; out[1] = d->disc; out[2] = dm.emode == 5 ? 0 : dm.maxr;
    out[3] = dm.emode == 4 ? ((d->rm_kind == 0 && d->rf_kind == 0 && d->merr_nskip == 1) ? 1 : 0) : (dm.emode == 5 ? 0 : dm.NT);
    return VA_OK;
}

int va_eval_plan(const va_problem_desc *d, int32_t ne, int32_t ghost, int32_t *out)
{
    return va_eval_plan_reach(d, ne, ghost, nullptr, out);
}

int va_rhs_load_module(const char *path, int32_t *rhs_id)
{
    if (!path || !rhs_id) return fail(VA_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(g_user_rhs_mutex);
    for (size_t i = 0; i < g_user_rhs.size(); ++i)
        if (g_user_rhs[i].path == path) { *rhs_id = VA_RHS_USER_BASE + (int32_t)i; return VA_OK; }
    UserRhs u;
    u.path = path;
    u.dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!u.dl) return fail(VA_EINVAL, "dlopen(%s): %s", path, dlerror());
    typedef void (*info_fn)(int *);
    info_fn info = (info_fn)dlsym(u.dl, "va_user_rhs_info");
    u.launch = (void (*)(const Dev *, void *))dlsym(u.dl, "va_user_launch_eval");
    u.prepare = (int (*)(const Dev *))dlsym(u.dl, "va_user_prepare_eval");
    u.seed_kernel = (int (*)(const Dev *, int, void *))dlsym(u.dl, "va_user_seed_kernel");
    if (!info || !u.launch || !u.prepare) { dlclose(u.dl); return fail(VA_EINVAL, "%s lacks va_user_rhs_info / va_user_launch_eval / va_user_prepare_eval", path); }
    int v[5] = {0, 0, 0, 0, 0};
    info(v);
    if (v[3] != (int)sizeof(Dev) || v[4] != (int)sizeof(SeedState)) {
        dlclose(u.dl);
        return fail(VA_EINVAL, "%s was built against different headers (Dev %d vs %zu bytes): rebuild it", path, v[3], sizeof(Dev));
    }
    if (v[0] < 0 || v[0] > RHS_BIG_NP) { dlclose(u.dl); return fail(VA_EUNSUPPORTED, "%s: NP=%d > %d", path, v[0], RHS_BIG_NP); }
    u.NP = v[0]; u.D = v[1]; u.NSTIM = v[2];
    if (info_fn vinfo = (info_fn)dlsym(u.dl, "va_user_variant_info")) {        // (writes 12 ints)
        vinfo(u.var);
        u.launch_var = (void (*)(const Dev *, void *))dlsym(u.dl, "va_user_launch_variant");
        u.prepare_var = (int (*)(const Dev *))dlsym(u.dl, "va_user_prepare_variant");
        if (!u.launch_var || !u.prepare_var) u.var[0] = 0;
    }
    g_user_rhs.push_back(u);
    *rhs_id = VA_RHS_USER_BASE + (int32_t)g_user_rhs.size() - 1;
    return VA_OK;
}

int va_act_load_module(const char *path, int32_t *act_id)
{
    if (!path || !act_id) return fail(VA_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(g_user_rhs_mutex);
    for (size_t i = 0; i < g_user_act.size(); ++i)
        if (g_user_act[i].path == path) { *act_id = VA_ACT_USER_BASE + (int32_t)i; return VA_OK; }
    UserAct u;
    u.path = path;
    u.dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!u.dl) return fail(VA_EINVAL, "dlopen(%s): %s", path, dlerror());
    typedef void (*info_fn)(int *);
    info_fn info = (info_fn)dlsym(u.dl, "va_user_act_info");
    u.launch = (NnetActLaunch)dlsym(u.dl, "va_user_act_launch");
    if (!info || !u.launch) { dlclose(u.dl); return fail(VA_EINVAL, "%s lacks va_user_act_info / va_user_act_launch", path); }
    int v[3] = {0, 0, 0};
    info(v);
    if (v[0] != (int)sizeof(Dev) || v[1] != (int)sizeof(NnetDev) || v[2] != (int)sizeof(SeedState)) {
        dlclose(u.dl);
        return fail(VA_EINVAL, "%s was built against different headers: rebuild it", path);
    }
    g_user_act.push_back(u);
    *act_id = VA_ACT_USER_BASE + (int32_t)g_user_act.size() - 1;
    return VA_OK;
}

int va_problem_create(const va_problem_desc *d, va_handle *out)
{
    if (!d || !out) return fail(VA_EINVAL, "null argument");
    *out = nullptr;
    if (d->struct_size != (int32_t)sizeof(va_problem_desc)) return fail(VA_EINVAL, "struct_size %d != %zu", d->struct_size, sizeof(va_problem_desc));
    if (d->batch < 1 || d->D < 1 || d->N_model < 2 || d->N_data < 1 || d->L < 0 || d->merr_nskip < 1)
        return fail(VA_EINVAL, "bad sizes (batch=%d D=%d N_model=%d N_data=%d L=%d nskip=%d)", d->batch, d->D, d->N_model, d->N_data, d->L, d->merr_nskip);
    if ((int64_t)(d->N_data - 1) * d->merr_nskip + 1 != d->N_model)      /* va_ode.py:557 */
        return fail(VA_EINVAL, "N_model (%d) must equal (N_data-1)*merr_nskip+1 (%lld)", d->N_model,
                    (long long)(d->N_data - 1) * d->merr_nskip + 1);
    if (d->disc < VA_DISC_EULER || d->disc > VA_DISC_FORWARDMAP) return fail(VA_EINVAL, "unknown disc %d", d->disc);
    if (d->disc == VA_DISC_SIMPSON_HERMITE && (d->N_model % 2) == 0)
        return fail(VA_EINVAL, "SimpsonHermite needs an odd number of time points (N_model=%d)", d->N_model);
    UserRhs user_copy;
    const UserRhs *user = nullptr;
    if (d->rhs >= VA_RHS_USER_BASE) {
        std::lock_guard<std::mutex> lock(g_user_rhs_mutex);
        if ((size_t)(d->rhs - VA_RHS_USER_BASE) >= g_user_rhs.size()) return fail(VA_EINVAL, "rhs module id %d was never registered", d->rhs);
        user_copy = g_user_rhs[d->rhs - VA_RHS_USER_BASE];      // (the vector may grow under another thread)
        user = &user_copy;
        if (user->NP != d->NP || user->D != d->D || user->NSTIM != d->n_stim)
            return fail(VA_EINVAL, "rhs module %s was generated for D=%d NP=%d n_stim=%d, problem has D=%d NP=%d n_stim=%d",
                        user->path.c_str(), user->D, user->NP, user->NSTIM, d->D, d->NP, d->n_stim);
    } else if (d->rhs != VA_RHS_LORENZ96) return fail(VA_EUNSUPPORTED, "unknown built-in rhs %d", d->rhs);
    if (d->rhs == VA_RHS_LORENZ96 && (d->NP != RhsL96::NP || d->D < 4))
        return fail(VA_EINVAL, "Lorenz-96 needs NP=1 and D>=4 (NP=%d D=%d)", d->NP, d->D);
    if (d->n_stim < 0 || (d->n_stim > 0 && !d->stim)) return fail(VA_EINVAL, "n_stim=%d without a stimulus array", d->n_stim);
    if (d->NPest < 0 || d->NPest > d->NP || d->NP > RHS_BIG_NP) return fail(VA_EINVAL, "bad NP/NPest (%d/%d)", d->NP, d->NPest);
    const bool tdp = d->p_time_dependent != 0;
    // more than RHS_MAX_NP parameters: the flat kernel carries them (their gradient partials in a table of their own)
    const bool bigp = d->NP > RHS_MAX_NP;
    if (bigp && tdp) return fail(VA_EUNSUPPORTED, "time-dependent parameters: at most %d of them", RHS_MAX_NP);
    if (tdp && d->disc != VA_DISC_TRAPEZOID && d->disc != VA_DISC_SIMPSON_HERMITE)
        return fail(VA_EUNSUPPORTED, "time-dependent parameters: trapezoid and SimpsonHermite only (upstream's euler/forwardmap "
                                     "branches are inconsistent, va_ode.py:345-349)");
    if (tdp && (int64_t)d->N_model * (d->D + d->NPest) > 2000000000LL) return fail(VA_EUNSUPPORTED, "n_var does not fit 32-bit indexing");
    if (!d->Y || (d->L > 0 && !d->Lidx) || !d->P || (d->NPest > 0 && !d->Pidx)) return fail(VA_EINVAL, "null array in desc");
    if ((d->rm_kind && !d->rm_array) || (d->rf_kind && !d->rf0_array)) return fail(VA_EINVAL, "rm/rf array kind without array");
    if ((d->lower != nullptr) != (d->upper != nullptr)) return fail(VA_EINVAL, "lower and upper bounds come together");
    {
        std::vector<char> seen(d->D, 0);
        for (int l = 0; l < d->L; ++l) {
            if (d->Lidx[l] < 0 || d->Lidx[l] >= d->D) return fail(VA_EINVAL, "Lidx[%d]=%d outside [0,D)", l, d->Lidx[l]);
            // any order is fine (data column l pairs with state column Lidx[l], va_ode.py:141); a state
            // column observed twice has no slot in the column -> data-column map the kernels use
            if (seen[d->Lidx[l]] && d->rm_kind != 2) return fail(VA_EUNSUPPORTED, "Lidx lists state column %d twice", d->Lidx[l]);
            seen[d->Lidx[l]] = 1;
        }
    }
    for (int k = 0; k < d->NPest; ++k)
        if (d->Pidx[k] < 0 || d->Pidx[k] >= d->NP) return fail(VA_EINVAL, "Pidx[%d]=%d outside [0,NP)", k, d->Pidx[k]);
    const int m = d->lbfgs_m > 0 ? d->lbfgs_m : 10;
    if (m > MAX_M) return fail(VA_EINVAL, "lbfgs_m=%d > %d", m, MAX_M);
    const int max_beta = d->max_beta > 0 ? d->max_beta : 1;

    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (d->device < 0 || d->device >= ndev) return fail(VA_EINVAL, "device %d of %d", d->device, ndev);
    HIPCHK(hipSetDevice(d->device));

    va_handle h = new va_problem_s();
    h->device = d->device; h->rhs = d->rhs; h->keep_paths = d->keep_paths;
    h->user_launch = user ? user->launch : nullptr;
    h->user_prepare = user ? user->prepare : nullptr;
    h->user_seed = user ? user->seed_kernel : nullptr;
    if (d->stream) h->stream = (hipStream_t)d->stream;
    else {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete h; return fail(VA_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        h->own_stream = true;
    }
    Dev &dv = h->dv;
    memset(&dv, 0, sizeof dv);
    Dims &dm = dv.dm;
    dm.D = d->D; dm.N = d->N_model; dm.ND = dm.D * dm.N; dm.L = d->L; dm.N_data = d->N_data;
    dm.nskip = d->merr_nskip; dm.NP = d->NP; dm.NPest = d->NPest; dm.B = d->batch; dm.m = m;
    dm.disc = d->disc;
    dm.tdp = tdp ? 1 : 0; dm.NPt = d->NP; dm.NPe = d->NPest;
    dm.bounded = (d->lower && d->upper) ? 1 : 0;
    if (tdp) { dm.ND = dm.N * (dm.D + dm.NPe); dm.NP = 0; dm.NPest = 0; }   // one flat run for the L-BFGS kernels
    dm.ld = ((dm.ND + dm.NPest + 15) / 16) * 16;
    std::vector<int> ystrip_h;
    if (!user) {
        const int reach5[4] = {t5_xl<RhsL96s>(), t5_xr<RhsL96s>(), t5_gl<RhsL96s>(), t5_gr<RhsL96s>()};
        pick_eval_geometry(d, dm, dv.g4, RhsL96s::NE, RhsL96g::GHOST, reach5, &dv.g5, &ystrip_h);
    }
    else {
        // the module holds ONE instantiation of a column-run kernel (va_eval_plan named it when the module
        // was generated); a problem that calls for any other geometry runs the module's flat kernel
        const int *v = user->var;
        dm.lin = v[10] ? 1 : 0;
        if (bigp) pick_eval_geometry(d, dm, dv.g4, 0, 0);
        else pick_eval_geometry(d, dm, dv.g4, (v[0] == 4 || v[0] == 5) ? v[4] : 0, v[0] == 3 ? v[5] : 0, v[0] == 5 ? v + 6 : nullptr, &dv.g5, &ystrip_h);
        const bool ws = d->rm_kind == 0 && d->rf_kind == 0 && d->merr_nskip == 1;
        const bool fits = dm.emode == v[0] && v[1] == d->disc &&
                          (dm.emode == 5 ? true : (v[2] == dm.maxr && (dm.emode == 4 ? (v[3] != 0) == ws : v[3] == dm.NT)));
        if (dm.emode != 1 && fits) { h->user_launch = user->launch_var; h->user_prepare = user->prepare_var; }
        else if (dm.emode != 1) pick_eval_geometry(d, dm, dv.g4, 0, 0);
    }
    if (dm.emode == 4 && (unsigned long long)dm.B * dm.ntiles * dm.ntiles >= (1ull << 32)) {
        va_problem_destroy(h);        // (umulhi by ntiles_magic would no longer be an exact division)
        return fail(VA_EUNSUPPORTED, "batch x tiles too large for the wave-private kernel: pass eval_kernel=3");
    }
    dv.ntiles_magic = (unsigned)(((1ull << 32) + dm.ntiles - 1) / dm.ntiles);
    // fold the tail into the evaluation kernel while the whole grid is resident at once (<= 8 workgroups per CU)
    h->fold = (long)dm.B * dm.ntiles <= 8L * 256;
    dm.nprow = dm.ntiles;                                                    // one partial row per workgroup
    dm.chunk = VEC_CHUNK; dm.nchunks = (dm.ld + VEC_CHUNK - 1) / VEC_CHUNK;
    dm.dt = d->dt_model;
    dm.cme = d->L > 0 ? 1.0 / ((double)dm.L * dm.N_data) : 0.0;
    dm.cfe = 1.0 / ((double)dm.D * (dm.N - 1));
    dm.rm = d->rm; dm.rf0 = d->rf0;
    dv.ups = UP_OLD + 4 * m; dv.max_beta = max_beta; dv.nbeta = 1;
    dv.evcols = EP_GP + d->NP <= 8 ? 8 : (EP_GP + d->NP <= 16 ? 16 : 32);
    // write-through gradient stores pay where the grid is one resident round and the end-of-kernel write-back
    // of 10 MB is on the critical path (C3: -1.3 us); on large grids they cost 10 % (4096 seeds: 446 vs 404 us)
    dv.gaux = h->fold ? 1 : 0;
    dv.prio = 1;
    dv.o.m = m; dv.o.maxiter = 15000; dv.o.maxls = 20; dv.o.maxfun = 15000; dv.o.ftol = 2.2204460492503131e-09; dv.o.gtol = 1e-5;

    {
        // the flat kernel keeps 3 staged arrays of (T + halo) rows: up to the CU's 160 KiB
        size_t need = eval_lds_bytes(dv);
        if (dm.emode == 5) { dv.lsrun = 1; need = std::max(need, eval_lds_bytes(dv)); dv.lsrun = 0; }
        const size_t cap = 160 * 1024;
        if (need > cap) {
            const int T = dm.T, D = dm.D;
            va_problem_destroy(h);
            return fail(VA_EUNSUPPORTED, "a tile of %d rows x D=%d needs %zu B of LDS (> %zu): state too wide for this kernel",
                        T, D, need, cap);
        }
        // more than 64 KiB of dynamic LDS is an opt-in per kernel AND per device: once per handle
        hipError_t e = h->user_prepare ? (hipError_t)h->user_prepare(&dv) : prepare_eval(dv, h->rhs);
        if (e == hipSuccess && dm.bounded) e = prepare_lbfgsb(dv);
        if (e != hipSuccess) {
            va_problem_destroy(h);
            return fail(VA_EHIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(e));
        }
    }

    int rc = VA_OK;
    const size_t B = dm.B;
    int *lmap_d = nullptr, *pidx_d = nullptr;
    double *Y_d = nullptr, *rm_d = nullptr, *rf_d = nullptr, *P_d = nullptr, *t_d = nullptr, *st_d = nullptr;
#define TRY(x) do { rc = (x); if (rc) { va_problem_destroy(h); return rc; } } while (0)
    TRY(h->alloc(&lmap_d, dm.D));
    // (two rows + a line of padding: the streaming kernel stages observation rows by whole 16-byte pieces, two rows at a time)
    const size_t LY = dm.emode == 5 ? (size_t)dv.g5.LY : (size_t)dm.L;        // row pitch of Y (and of the RM image of k_eval5) on the device
    TRY(h->alloc(&Y_d, (size_t)dm.N_data * LY + 4 * LY + 16));   // (k_eval5 stages row pairs: one pair before the first row, one past the last)
    Y_d += 2 * (LY / 2) + (LY & 1) * 2;                   // an even number of doubles >= L: the data keep their 16-byte alignment
    int *ystrip_d = nullptr;
    if (dm.emode == 5) TRY(h->alloc(&ystrip_d, ystrip_h.size()));
    const size_t np_seed = tdp ? (size_t)dm.N * dm.NPt : (size_t)dm.NPt;       // parameters stored per seed
    TRY(h->alloc(&pidx_d, dm.NPe));
    TRY(h->alloc(&P_d, B * np_seed));
    if (d->rm_kind < 0 || d->rm_kind > 2) { va_problem_destroy(h); return fail(VA_EINVAL, "rm_kind %d", d->rm_kind); }
    const size_t rm_elems = (size_t)dm.N_data * dm.L * (d->rm_kind == 2 ? dm.L : 1);
    int *lidx_d = nullptr;
    const bool warr5 = dm.emode == 5 && dv.g5.warr;      // k_eval5 streams both weight images: scalar weights are spread out into arrays
    if (warr5) {
        // (as Y: one row pair in front and behind; L is even on this path)
        TRY(h->alloc(&rm_d, (size_t)dm.N_data * LY + 4 * LY + 16));
        rm_d += LY;
    } else if (d->rm_kind) TRY(h->alloc(&rm_d, rm_elems));
    if (d->rm_kind == 2) TRY(h->alloc(&lidx_d, dm.L));
    if (warr5) {
        TRY(h->alloc(&rf_d, (size_t)(dm.N + 3) * dm.D + 16));
        rf_d += dm.D;
    } else if (d->rf_kind) TRY(h->alloc(&rf_d, (size_t)(dm.N - 1) * dm.D * (d->rf_kind == 2 ? dm.D : 1)));
    double *lo_d = nullptr, *hi_d = nullptr;
    std::vector<double> lo_h, hi_h;
    if (dm.bounded) {
        const int nv = dm.ND + dm.NPest;
        lo_h.assign(dm.ld, -HUGE_VAL); hi_h.assign(dm.ld, HUGE_VAL);
        for (int i = 0; i < nv; ++i) {
            if (!(d->lower[i] <= d->upper[i])) { va_problem_destroy(h); return fail(VA_EINVAL, "lower[%d] > upper[%d] (or NaN)", i, i); }
            lo_h[i] = d->lower[i]; hi_h[i] = d->upper[i];
        }
        bool boxed = true;
        for (int i = 0; i < nv; ++i) boxed = boxed && lo_h[i] > -HUGE_VAL && hi_h[i] < HUGE_VAL;
        if (boxed) { dm.bounded |= 2; dv.dm.bounded |= 2; }
        TRY(h->alloc(&lo_d, (size_t)dm.ld)); TRY(h->alloc(&hi_d, (size_t)dm.ld));
    }
    if (d->t_model) TRY(h->alloc(&t_d, (size_t)dm.N));
    if (d->n_stim > 0) TRY(h->alloc(&st_d, (size_t)dm.N * d->n_stim));
    TRY(alloc_solver_state(h, max_beta, d->keep_paths));

    // On the device Lidx is ascending: data column l pairs with state column Lidx[l] in any order
    // (va_ode.py:141), so sorting Lidx and permuting the columns of Y (and of a weight array) the same
    // way changes nothing, and the kernels of narrow states find a column's data by counting the
    // observed columns below it (obsmask) instead of loading a map.
    std::vector<int> perm(dm.L), lidx_sorted(dm.L);
    for (int l = 0; l < dm.L; ++l) perm[l] = l;
    if (d->rm_kind != 2) std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return d->Lidx[a] < d->Lidx[b]; });
    for (int l = 0; l < dm.L; ++l) lidx_sorted[l] = d->Lidx[perm[l]];
    std::vector<double> Ys((size_t)dm.N_data * LY, 0.0), rms;
    for (int n = 0; n < dm.N_data; ++n)
        for (int l = 0; l < dm.L; ++l) Ys[(size_t)n * LY + l] = d->Y[(size_t)n * dm.L + perm[l]];
    if (warr5 && d->rm_kind == 0) rms.assign((size_t)dm.N_data * LY, d->rm);
    if (d->rm_kind == 1) {
        rms.assign((size_t)dm.N_data * LY, 0.0);
        for (int n = 0; n < dm.N_data; ++n)
            for (int l = 0; l < dm.L; ++l) rms[(size_t)n * LY + l] = d->rm_array[(size_t)n * dm.L + perm[l]];
    }
    std::vector<int> lmap(dm.D, -1);
    for (int l = 0; l < dm.L; ++l) lmap[lidx_sorted[l]] = l;
    dm.obsmask = 0ull;
    if (dm.D <= 64) for (int l = 0; l < dm.L; ++l) dm.obsmask |= 1ull << lidx_sorted[l];
    dv.dm.obsmask = dm.obsmask;
#define H2D(dst, src, n, T) do { hipError_t e_ = hipMemcpyAsync(dst, src, sizeof(T) * (n), hipMemcpyHostToDevice, h->stream); \
        if (e_ != hipSuccess) { va_problem_destroy(h); return fail(VA_EHIP, "H2D %s: %s", #dst, hipGetErrorString(e_)); } } while (0)
    H2D(lmap_d, lmap.data(), dm.D, int);
    if (dm.emode == 5) { H2D(ystrip_d, ystrip_h.data(), ystrip_h.size(), int); dv.ystrip = ystrip_d; }
    H2D(Y_d, Ys.data(), (size_t)dm.N_data * LY, double);
    if (dm.NPe) H2D(pidx_d, d->Pidx, dm.NPe, int);
    H2D(P_d, d->P, B * np_seed, double);
    if (d->rm_kind || warr5) H2D(rm_d, d->rm_kind != 2 ? rms.data() : d->rm_array, d->rm_kind != 2 ? rms.size() : rm_elems, double);
    if (d->rm_kind == 2) H2D(lidx_d, d->Lidx, dm.L, int);
    std::vector<double> rf_fill;
    if (warr5 && d->rf_kind == 0) rf_fill.assign((size_t)(dm.N - 1) * dm.D, d->rf0);
    if (d->rf_kind || warr5) H2D(rf_d, d->rf_kind ? d->rf0_array : rf_fill.data(), (size_t)(dm.N - 1) * dm.D * (d->rf_kind == 2 ? dm.D : 1), double);
    if (dm.bounded) { H2D(lo_d, lo_h.data(), (size_t)dm.ld, double); H2D(hi_d, hi_h.data(), (size_t)dm.ld, double); }
    dv.pp.lo = lo_d; dv.pp.hi = hi_d;
    if (d->t_model) H2D(t_d, d->t_model, (size_t)dm.N, double);
    if (d->n_stim > 0) H2D(st_d, d->stim, (size_t)dm.N * d->n_stim, double);
    dv.pp.lmap = lmap_d; dv.pp.Y = Y_d; dv.pp.rf0_arr = (d->rf_kind == 1 || warr5) ? rf_d : nullptr;
    dv.pp.rf0_full = d->rf_kind == 2 ? rf_d : nullptr;
    dv.pp.rm_arr = (d->rm_kind == 1 || warr5) ? rm_d : nullptr;
    dv.pp.rm_full = d->rm_kind == 2 ? rm_d : nullptr; dv.pp.Lidx = lidx_d;
    dv.pp.Pidx = pidx_d; dv.pp.Pfull = P_d;
    dv.pp.tmodel = t_d; dv.pp.stim = st_d; dv.pp.nstim = d->n_stim;

    // few seeds, short paths: can the whole minimisation live in LDS?  (flat tile phases: any right-hand side, any
    // discretisation, weight arrays, merr_nskip, full weight matrices; not bounds, time-dependent parameters, a dense
    // linear part, or the padded observation rows of the streaming kernel)
    if (!dm.bounded && !tdp && !dm.lin && !bigp && dm.emode != 5 && (!user || user->seed_kernel)) {
        int G = 0, T = 0, ncu = 0;
        HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, d->device));
        h->pz_maxG = ncu / (int)B;
        if (h->pz_maxG >= 1 && persist_geometry(dm.N, dm.D, dm.L, dm.NP, dm.NPest, m, dm.disc, PZ_LDS_BYTES, h->pz_maxG, 0, &G, &T)) {
            Dev dvp = dv;
            dvp.dm.T = T; dvp.dm.ntiles = G;
            const hipError_t e = h->user_seed ? (hipError_t)h->user_seed(&dvp, 0, nullptr) : seed_kernel_builtin(dvp, false, nullptr);
            if (e == hipSuccess) {
                h->persist = true; h->pz_G = G; h->pz_T = T;
                // (sized for the most workgroups a seed may get: va_problem_tune may choose other slices)
                const size_t units = B * 2 * (size_t)h->pz_maxG * (size_t)pz_row_granules(dm.D) * 2;     // 8-byte units: 16 per granule pair
                TRY(h->alloc(&dv.pz.xch, units));
                h->pz_xch_bytes = units * 8;
                unsigned long long *misc = nullptr;
                TRY(h->alloc(&misc, 2 + PZ_NSTAMP));
                h->pz_misc = misc;
                dv.pz.abort_flag = (int *)misc; dv.pz.cycles = misc + 1; dv.pz.stamps = (double *)(misc + 2);
            } else (void)hipGetLastError();
        }
    }
    TRY(finish_create(h));
#undef TRY
#undef H2D
    *out = h;
    return VA_OK;
}

int va_nnet_problem_create(const va_nnet_desc *d, va_handle *out)
{
    if (!d || !out) return fail(VA_EINVAL, "null argument");
    *out = nullptr;
    if (d->struct_size != (int32_t)sizeof(va_nnet_desc)) return fail(VA_EINVAL, "struct_size %d != %zu", d->struct_size, sizeof(va_nnet_desc));
    if (d->batch < 1 || d->n_layers < 2 || d->M < 1 || !d->structure) return fail(VA_EINVAL, "bad sizes (batch=%d n_layers=%d M=%d)", d->batch, d->n_layers, d->M);
    if ((d->rm_in_matrix != nullptr) != (d->rm_out_matrix != nullptr)) return fail(VA_EINVAL, "rm_in_matrix and rm_out_matrix come together");
    NnetActLaunch user_act = nullptr;
    if (d->activation >= VA_ACT_USER_BASE) {
        std::lock_guard<std::mutex> lock(g_user_rhs_mutex);
        if ((size_t)(d->activation - VA_ACT_USER_BASE) >= g_user_act.size()) return fail(VA_EINVAL, "activation module id %d was never registered", d->activation);
        user_act = g_user_act[d->activation - VA_ACT_USER_BASE].launch;
    } else if (d->activation < VA_ACT_SIGMOID || d->activation > VA_ACT_SOFTPLUS) return fail(VA_EUNSUPPORTED, "unknown activation %d", d->activation);
    const int NL = d->n_layers;
    std::vector<int> s(d->structure, d->structure + NL), off(NL + 1, 0), woff(NL - 1), boff(NL - 1);
    long long np = 0;
    for (int n = 0; n < NL; ++n) {
        if (s[n] < 1) return fail(VA_EINVAL, "structure[%d]=%d", n, s[n]);
        off[n + 1] = off[n] + s[n];
    }
    for (int n = 0; n < NL - 1; ++n) { woff[n] = (int)np; np += (long long)s[n + 1] * s[n]; boff[n] = (int)np; np += s[n + 1]; }
    if (np != d->NP) return fail(VA_EINVAL, "NP=%d but the structure holds %lld weights and biases (va_nnet.py:194-207)", d->NP, np);
    if (d->NPest < 0 || d->NPest > d->NP || !d->P || (d->NPest > 0 && !d->Pidx)) return fail(VA_EINVAL, "bad NPest/P/Pidx");
    if (d->L_in < 0 || d->L_out < 0 || (d->L_in > 0 && (!d->Lidx_in || !d->data_in)) || (d->L_out > 0 && (!d->Lidx_out || !d->data_out)))
        return fail(VA_EINVAL, "observed-neuron arrays missing");
    if (d->L_in + d->L_out < 1) return fail(VA_EINVAL, "no observed neurons: the measurement error divides by Ltot*M (va_nnet.py:173)");
    const int NDnet = off[NL];
    const long long nvar = (long long)NDnet * d->M + d->NPest;
    if (nvar > 2000000000LL) return fail(VA_EUNSUPPORTED, "n_var=%lld does not fit 32-bit indexing", nvar);
    std::vector<int> lin(s[0], -1), lout(s[NL - 1], -1), pmap(d->NP, -1);
    for (int l = 0; l < d->L_in; ++l) {
        if (d->Lidx_in[l] < 0 || d->Lidx_in[l] >= s[0]) return fail(VA_EINVAL, "Lidx_in[%d]=%d outside the input layer", l, d->Lidx_in[l]);
        lin[d->Lidx_in[l]] = l;
    }
    for (int l = 0; l < d->L_out; ++l) {
        if (d->Lidx_out[l] < 0 || d->Lidx_out[l] >= s[NL - 1]) return fail(VA_EINVAL, "Lidx_out[%d]=%d outside the output layer", l, d->Lidx_out[l]);
        lout[d->Lidx_out[l]] = l;
    }
    for (int k = 0; k < d->NPest; ++k) {
        if (d->Pidx[k] < 0 || d->Pidx[k] >= d->NP) return fail(VA_EINVAL, "Pidx[%d]=%d outside [0,NP)", k, d->Pidx[k]);
        if (pmap[d->Pidx[k]] >= 0) return fail(VA_EINVAL, "Pidx[%d]=%d listed twice", k, d->Pidx[k]);
        pmap[d->Pidx[k]] = k;
    }
    const int m = d->lbfgs_m > 0 ? d->lbfgs_m : 10;
    if (m > MAX_M) return fail(VA_EINVAL, "lbfgs_m=%d > %d", m, MAX_M);
    const int max_beta = d->max_beta > 0 ? d->max_beta : 1;

    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (d->device < 0 || d->device >= ndev) return fail(VA_EINVAL, "device %d of %d", d->device, ndev);
    HIPCHK(hipSetDevice(d->device));

    va_handle h = new va_problem_s();
    h->device = d->device; h->rhs = -1; h->keep_paths = d->keep_paths; h->is_nnet = true; h->user_act = user_act;
    if (d->stream) h->stream = (hipStream_t)d->stream;
    else {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete h; return fail(VA_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
        h->own_stream = true;
    }
    Dev &dv = h->dv;
    memset(&dv, 0, sizeof dv);
    NnetDev &nn = h->nn;
    memset(&nn, 0, sizeof nn);
    Dims &dm = dv.dm;
    // to the L-BFGS kernels the unknown vector is one flat run of ND doubles with no tail
    dm.D = NDnet; dm.N = d->M; dm.ND = (int)nvar; dm.NP = 0; dm.NPest = 0; dm.B = d->batch; dm.m = m;
    dm.L = d->L_in + d->L_out; dm.N_data = d->M; dm.nskip = 1; dm.disc = VA_DISC_FORWARDMAP;
    dm.ld = ((dm.ND + 15) / 16) * 16;
    dm.chunk = VEC_CHUNK; dm.nchunks = (dm.ld + VEC_CHUNK - 1) / VEC_CHUNK;
    dm.cme = 1.0 / ((double)(d->L_in + d->L_out) * d->M);                 /* va_nnet.py:173 */
    dm.cfe = d->rf0 / ((double)(NDnet - s[0]) * d->M);                    /* va_nnet.py:255 */
    dm.rm = d->rm_in; dm.rf0 = d->rf0;
    dv.ups = UP_OLD + 4 * m; dv.max_beta = max_beta; dv.nbeta = 1;
    dv.evcols = 8;                 // the network kernels fill EP_ME .. EP_GMAX only
    dv.o.m = m; dv.o.maxiter = 15000; dv.o.maxls = 20; dv.o.maxfun = 15000; dv.o.ftol = 2.2204460492503131e-09; dv.o.gtol = 1e-5;

    nn.NL = NL; nn.M = d->M; nn.NDnet = NDnet; nn.NDens = NDnet * d->M; nn.NP = d->NP; nn.NPest = d->NPest;
    nn.act = d->activation; nn.Lin = d->L_in; nn.Lout = d->L_out; nn.rm_in = d->rm_in; nn.rm_out = d->rm_out;
    // job tables: one entry per 32x32 output tile
    std::vector<NnetTile> t1, t2, t3;
    // examples per chunk of the weight-gradient product: 256, doubled while the launch keeps >= 6 workgroups per CU
    // (each chunk writes a partial of the whole parameter gradient that k_nnet_pred reads back)
    nn.mch = d->M <= 256 ? ((d->M + NN_KC - 1) / NN_KC) * NN_KC : 256;
    {
        long long tiles = 0;
        int ncu = 256;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, d->device) != hipSuccess) ncu = 256;
        for (int n = 0; n < NL - 1; ++n) tiles += (long long)((s[n + 1] + NN_TILE - 1) / NN_TILE) * ((s[n] + NN_TILE - 1) / NN_TILE);
        while (nn.mch * 2 <= d->M && tiles * d->batch * ((d->M + 2 * nn.mch - 1) / (2 * nn.mch)) >= 6LL * ncu) nn.mch *= 2;
    }
    nn.nmch = (d->M + nn.mch - 1) / nn.mch;
    auto tile = [&](int n, int r0, int c0, int c) {
        NnetTile t;
        memset(&t, 0, sizeof t);
        t.layer = n; t.r0 = r0; t.c0 = c0; t.chunk = c; t.sn = s[n]; t.offn = off[n];
        if (n < NL - 1) { t.sn1 = s[n + 1]; t.offn1 = off[n + 1]; t.woff = woff[n]; t.boff = boff[n]; }
        return t;
    };
    // Order of the jobs: workgroup i of a launch runs on XCD i % 8, each with an L2 of its own.  The jobs that read the
    // same operand rows (the column tiles of one row block; the four tiles of one example chunk) are placed 8 apart --
    // eight such families at a time, member by member -- so that they meet in ONE L2 and the rows come from HBM once
    auto place = [&](std::vector<NnetTile> &out, std::vector<std::vector<NnetTile>> &fam) {
        size_t f0 = 0;
        while (f0 < fam.size()) {
            const size_t nf = std::min<size_t>(8, fam.size() - f0);
            size_t width = 0;
            for (size_t f = 0; f < nf; ++f) width = std::max(width, fam[f0 + f].size());
            bool uniform = nf == 8;
            for (size_t f = 0; f < nf; ++f) uniform = uniform && fam[f0 + f].size() == width;
            if (uniform)
                for (size_t k = 0; k < width; ++k)
                    for (size_t f = 0; f < nf; ++f) out.push_back(fam[f0 + f][k]);
            else
                for (size_t f = 0; f < nf; ++f) out.insert(out.end(), fam[f0 + f].begin(), fam[f0 + f].end());
            f0 += nf;
        }
        fam.clear();
    };
    std::vector<std::vector<NnetTile>> fam;
    for (int n = 0; n < NL - 1; ++n) {
        for (int m0 = 0; m0 < d->M; m0 += NN_TILE) {
            fam.emplace_back();
            for (int i0 = 0; i0 < s[n + 1]; i0 += NN_TILE) fam.back().push_back(tile(n, m0, i0, 0));
        }
        place(t1, fam);
    }
    for (int n = 0; n < NL; ++n) {
        for (int m0 = 0; m0 < d->M; m0 += NN_TILE) {
            fam.emplace_back();
            for (int j0 = 0; j0 < s[n]; j0 += NN_TILE) fam.back().push_back(tile(n, m0, j0, 0));
        }
        place(t2, fam);
    }
    for (int n = 0; n < NL - 1; ++n) {
        for (int c = 0; c < nn.nmch; ++c) {
            fam.emplace_back();
            for (int i0 = 0; i0 < s[n + 1]; i0 += NN_TILE)
                for (int j0 = 0; j0 < s[n]; j0 += NN_TILE) fam.back().push_back(tile(n, i0, j0, c));
        }
        place(t3, fam);
    }
    nn.n1 = (int)t1.size(); nn.n2 = (int)t2.size(); nn.n3 = (int)t3.size();
    nn.n4 = (d->NP + NN_THREADS - 1) / NN_THREADS;
    nn.n0 = (nn.NDens + d->NP + NN_THREADS * NN_PACK - 1) / (NN_THREADS * NN_PACK);
    nn.nraw = nn.n1 + nn.n2 + nn.n4;
    // small networks: one workgroup per layer does the whole evaluation (k_nnet_small)
    {
        int widest = d->M;
        for (int n = 0; n < NL; ++n) widest = s[n] > widest ? s[n] : widest;
        nn.small = (widest <= NN_SMALL && NL <= NN_ROWS_DIRECT) ? (widest <= 16 ? 16 : 32) : 0;
    }
    if (nn.small) nn.nraw = NL;
    // layers up to NN_FB_W wide with scalar measurement weights: forward and state-gradient products in one kernel
    // (k_nnet_fb); its workgroups write the first nfb of the n1 + n2 rows, k_nnet_wfrag zeroes the others
    std::vector<int> wfoff(2 * (NL - 1), 0);
    bool fb_ok = false;
    {
        int widest = 0, wfsz = 0;
        for (int n = 0; n < NL; ++n) widest = s[n] > widest ? s[n] : widest;
        // fragment tables of the two products of every transition: NN_FB_W / 16 column blocks of nn_fb_steps(K) k-steps
        for (int n = 0; n < NL - 1; ++n) { wfoff[n] = wfsz; wfsz += (NN_FB_W / 16) * nn_fb_steps(s[n]) * 64; }
        for (int n = 0; n < NL - 1; ++n) { wfoff[NL - 1 + n] = wfsz; wfsz += (NN_FB_W / 16) * nn_fb_steps(s[n + 1]) * 64; }
        nn.nfb = (d->M + NN_FB_R - 1) / NN_FB_R;
        nn.wfsz = wfsz;
        fb_ok = !nn.small && widest <= NN_FB_W && NL <= NN_FB_LAYERS && !d->rm_in_matrix && nn.nfb <= nn.n1 + nn.n2 &&
                (long long)NDnet * d->M < (1LL << 31);
        {
            int ncu = 256;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, d->device) != hipSuccess) ncu = 256;
            nn.fb_slots = ncu * NN_FB_WGS; nn.fb_stagger = 0;
        }
        // on when its blocks of NN_FB_R examples fill the chip at least once (a workgroup walks ALL layers of its block:
        // with few blocks the separate kernels, one workgroup per layer and tile, have more in flight); va_problem_tune
        // switches it either way (c5x, 1024 workgroups: 637 against 709 us per evaluation, profiles/r04_nnet_fused.txt)
        // (not for softplus -- log1p / exp / expm1 spill 28 registers there: 830 against 809 us -- nor, unasked, for a generated
        // activation, whose register needs nobody has looked at)
        nn.fused = fb_ok && (long long)nn.nfb * d->batch >= nn.fb_slots && d->activation != NNET_SOFTPLUS && d->activation < NNET_USER;
    }
    const bool fold_rows = nn.nraw > NN_ROWS_DIRECT;      // k_ls sums the rows with one wave: keep them few
    dm.nprow = fold_rows ? NN_RED_ROWS : nn.nraw;
    dm.ntiles = dm.nprow; dm.T = NN_TILE; dm.emode = 0;

    int rc = VA_OK;
    const size_t B = dm.B;
    int *s_d = nullptr, *off_d = nullptr, *woff_d = nullptr, *boff_d = nullptr, *lin_d = nullptr, *lout_d = nullptr, *pmap_d = nullptr;
    double *din_d = nullptr, *dout_d = nullptr, *P_d = nullptr;
    NnetTile *t1_d = nullptr, *t2_d = nullptr, *t3_d = nullptr;
#define TRY(x) do { rc = (x); if (rc) { va_problem_destroy(h); return rc; } } while (0)
#define H2D(dst, src, n, T) do { hipError_t e_ = hipMemcpyAsync(dst, src, sizeof(T) * (n), hipMemcpyHostToDevice, h->stream); \
        if (e_ != hipSuccess) { va_problem_destroy(h); return fail(VA_EHIP, "H2D %s: %s", #dst, hipGetErrorString(e_)); } } while (0)
    TRY(h->alloc(&s_d, NL)); TRY(h->alloc(&off_d, NL + 1)); TRY(h->alloc(&woff_d, NL - 1)); TRY(h->alloc(&boff_d, NL - 1));
    TRY(h->alloc(&lin_d, s[0])); TRY(h->alloc(&lout_d, s[NL - 1])); TRY(h->alloc(&pmap_d, d->NP));
    TRY(h->alloc(&din_d, (size_t)d->M * d->L_in)); TRY(h->alloc(&dout_d, (size_t)d->M * d->L_out));
    TRY(h->alloc(&P_d, B * d->NP)); TRY(h->alloc(&nn.Pw, B * d->NP));
    TRY(h->alloc(&nn.delta, B * dm.ld));
    TRY(h->alloc(&nn.Xw, B * dm.ld));
    if (fold_rows) TRY(h->alloc(&nn.raw, B * nn.nraw * EP_GP));
    TRY(h->alloc(&nn.gpart, B * nn.nmch * (size_t)d->NP));
    int *wfoff_d = nullptr;
    if (fb_ok) {
        TRY(h->alloc(&nn.Wf, B * (size_t)nn.wfsz));
        TRY(h->alloc(&wfoff_d, 2 * (NL - 1)));
        TRY(h->alloc(&dv.pz.stamps, (size_t)PZ_NSTAMP));       // (measurement builds of k_nnet_fb: va_measure.h)
        TRY(h->alloc(&nn.fb_cu, (size_t)16));
    }
    TRY(h->alloc(&t1_d, t1.size())); TRY(h->alloc(&t2_d, t2.size())); TRY(h->alloc(&t3_d, t3.size()));
    TRY(alloc_solver_state(h, max_beta, d->keep_paths));
    H2D(s_d, s.data(), NL, int); H2D(off_d, off.data(), NL + 1, int);
    H2D(woff_d, woff.data(), NL - 1, int); H2D(boff_d, boff.data(), NL - 1, int);
    H2D(lin_d, lin.data(), s[0], int); H2D(lout_d, lout.data(), s[NL - 1], int); H2D(pmap_d, pmap.data(), d->NP, int);
    if (d->L_in) H2D(din_d, d->data_in, (size_t)d->M * d->L_in, double);
    if (d->L_out) H2D(dout_d, d->data_out, (size_t)d->M * d->L_out, double);
    H2D(P_d, d->P, B * d->NP, double);
    if (fb_ok) { H2D(wfoff_d, wfoff.data(), 2 * (NL - 1), int); nn.wfoff = wfoff_d; }
    H2D(t1_d, t1.data(), t1.size(), NnetTile); H2D(t2_d, t2.data(), t2.size(), NnetTile);
    if (!t3.empty()) H2D(t3_d, t3.data(), t3.size(), NnetTile);
    nn.s = s_d; nn.off = off_d; nn.woff = woff_d; nn.boff = boff_d; nn.lmap_in = lin_d; nn.lmap_out = lout_d;
    nn.pmap = pmap_d; nn.din = din_d; nn.dout = dout_d; nn.Pfix = P_d; nn.t1 = t1_d; nn.t2 = t2_d; nn.t3 = t3_d;
    if (d->rm_in_matrix) {
        // full measurement matrices (va_nnet.py:136-139): the kernels walk the layer's observed neurons
        double *ri = nullptr, *ro = nullptr; int *li = nullptr, *lo = nullptr;
        TRY(h->alloc(&ri, (size_t)d->L_in * d->L_in + 1)); TRY(h->alloc(&ro, (size_t)d->L_out * d->L_out + 1));
        TRY(h->alloc(&li, d->L_in + 1)); TRY(h->alloc(&lo, d->L_out + 1));
        if (d->L_in) { H2D(ri, d->rm_in_matrix, (size_t)d->L_in * d->L_in, double); H2D(li, d->Lidx_in, d->L_in, int); }
        if (d->L_out) { H2D(ro, d->rm_out_matrix, (size_t)d->L_out * d->L_out, double); H2D(lo, d->Lidx_out, d->L_out, int); }
        nn.rmm_in = ri; nn.rmm_out = ro; nn.lidx_in = li; nn.lidx_out = lo;
    }
    if (fb_ok) {
        const hipError_t e = prepare_nnet_fb(nn, user_act);
        if (e != hipSuccess) { (void)hipGetLastError(); nn.Wf = nullptr; }      // (the tune knob then refuses)
    }
    TRY(finish_create(h));
#undef TRY
#undef H2D
    *out = h;
    return VA_OK;
}

void va_problem_destroy(va_handle h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->timed_gexec) (void)hipGraphExecDestroy(h->timed_gexec);
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->h_nactive) (void)hipHostFree(h->h_nactive);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int va_problem_info(va_handle h, int64_t *n_var, int64_t *ld_internal, int32_t *tile_rows, int32_t *ntiles)
{
    if (!h) return fail(VA_EINVAL, "null handle");
    if (n_var) *n_var = h->dv.dm.ND + h->dv.dm.NPest;
    if (ld_internal) *ld_internal = h->dv.dm.ld;
    if (tile_rows) *tile_rows = h->dv.dm.T;
    if (ntiles) *ntiles = h->dv.dm.ntiles;
    return VA_OK;
}

int va_problem_eval_kernel(va_handle h, int32_t *eval_kernel, int32_t *run_rows)
{
    if (!h) return fail(VA_EINVAL, "null handle");
    if (eval_kernel) *eval_kernel = h->is_nnet ? 0 : h->dv.dm.emode;
    if (run_rows) *run_rows = h->is_nnet ? 0 : h->dv.dm.maxr;
    return VA_OK;
}

int va_problem_persistent(va_handle h, int32_t *workgroups_per_seed, int32_t *rows_per_workgroup)
{
    if (!h) return 0;
    const bool on = h->persist && h->tune_persist;
    if (workgroups_per_seed) *workgroups_per_seed = on ? h->pz_G : 0;
    if (rows_per_workgroup) *rows_per_workgroup = on ? h->pz_T : 0;
    return on ? 1 : 0;
}

int va_action_grad(va_handle h, const double *XP, int64_t ld, int32_t mem, double rf_scale,
                   double *A, double *me, double *fe, double *grad, int64_t ldg)
{
    int rc = check_xp(h, XP, ld, mem);
    if (rc) return rc;
    if (!A || !me || !fe) return fail(VA_EINVAL, "A/me/fe must not be NULL");
    if (grad && ldg < h->dv.dm.ND + h->dv.dm.NPest) return fail(VA_EINVAL, "ldg < n_var");
    HIPCHK(hipSetDevice(h->device));
    Dev &dv = h->dv;
    if ((rc = copy_in(h, XP, ld, mem))) return rc;
    launch_init_states(dv, PH_START, rf_scale, h->stream);
    h->timed_armed_rf = rf_scale;
    run_eval(h, EPI_FINALIZE);
    const hipMemcpyKind k = mem == VA_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    HIPCHK(hipMemcpyAsync(A, dv.outA, sizeof(double) * dv.dm.B, k, h->stream));
    HIPCHK(hipMemcpyAsync(me, dv.outme, sizeof(double) * dv.dm.B, k, h->stream));
    HIPCHK(hipMemcpyAsync(fe, dv.outfe, sizeof(double) * dv.dm.B, k, h->stream));
    if (grad && (rc = copy_out(h, dv.gt, grad, ldg, mem))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    h->n_eval_launch += 1; h->n_seed_evals += dv.dm.B; h->n_seed_evals_direct += dv.dm.B;
    return VA_OK;
}

int va_anneal(va_handle h, double *XP, int64_t ld, int32_t mem, const double *rf_scale, int32_t nbeta,
              const va_lbfgs_opts *opts, double *ame, double *pest, int32_t *status, int32_t *nit,
              int64_t *nfev, double *minpaths)
{
    int rc = check_xp(h, XP, ld, mem);
    if (rc) return rc;
    if (!rf_scale) return fail(VA_EINVAL, "rf_scale is NULL");
    if (minpaths && !h->dv.minpaths) return fail(VA_ESTATE, "minpaths requested but the problem was created with keep_paths=0");
    if ((rc = set_opts(h, opts))) return rc;
    HIPCHK(hipSetDevice(h->device));
    Dev &dv = h->dv;
    if ((rc = copy_in(h, XP, ld, mem))) return rc;
    if ((rc = run_ladder(h, rf_scale, nbeta))) return rc;
    if ((rc = copy_out(h, dv.x, XP, ld, mem))) return rc;
    if ((rc = fetch_table(h, dv.ame, ame, nbeta, 3))) return rc;
    if (dv.dm.NPest && (rc = fetch_table(h, dv.pest, pest, nbeta, dv.dm.NPest))) return rc;
    if ((rc = fetch_table(h, dv.status, status, nbeta, 1))) return rc;
    if ((rc = fetch_table(h, dv.nit, nit, nbeta, 1))) return rc;
    if ((rc = fetch_table(h, (const int64_t *)dv.nfev, nfev, nbeta, 1))) return rc;
    if ((rc = fetch_table(h, dv.minpaths, minpaths, nbeta, dv.dm.ND + dv.dm.NP))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return VA_OK;
}

int va_minimize_lbfgs(va_handle h, double *XP, int64_t ld, int32_t mem, double rf_scale,
                      const va_lbfgs_opts *opts, double *Amin, double *me, double *fe,
                      int32_t *status, int32_t *nit, int64_t *nfev)
{
    if (!h) return fail(VA_EINVAL, "null handle");
    const int B = h->dv.dm.B;
    std::vector<double> ame((size_t)B * 3);
    int rc = va_anneal(h, XP, ld, mem, &rf_scale, 1, opts, ame.data(), nullptr, status, nit, nfev, nullptr);
    if (rc) return rc;
    for (int b = 0; b < B; ++b) {
        if (Amin) Amin[b] = ame[3 * b];
        if (me) me[b] = ame[3 * b + 1];
        if (fe) fe[b] = ame[3 * b + 2];
    }
    return VA_OK;
}

int va_get_minpath(va_handle h, int32_t seed, int32_t beta_idx, double *out)
{
    if (!h || !out) return fail(VA_EINVAL, "null argument");
    const Dev &dv = h->dv;
    if (!dv.minpaths) return fail(VA_ESTATE, "problem was created with keep_paths=0");
    if (seed < 0 || seed >= dv.dm.B || beta_idx < 0 || beta_idx >= h->last_nbeta)
        return fail(VA_EINVAL, "seed/beta index out of range");
    HIPCHK(hipSetDevice(h->device));
    const size_t wide = dv.dm.ND + dv.dm.NP;
    HIPCHK(hipMemcpyAsync(out, dv.minpaths + ((size_t)seed * dv.max_beta + beta_idx) * wide,
                          sizeof(double) * wide, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return VA_OK;
}

namespace {
// The graph of `chunk` S1 evaluations va_eval_timed replays, built (or kept) OUTSIDE any timed region.  A host thread
// issues launches ~3.7 us apart, which would be the number measured for any kernel shorter than that (the kernel
// boundary is the same either way: MI355X_MICROARCH.md, "boundary: eager = hipGraph").  The kernels take the device
// image by value, so the graph is keyed on the bytes of h->dv / h->nn as run_eval(EPI_FINALIZE) leaves them.
int timed_chunk_of(int iters) { return iters < 250 ? iters : 250; }

bool timed_graph_current(va_handle h, int chunk)
{
    return h->timed_gexec && h->timed_chunk == chunk && h->tune_graph &&
           memcmp(&h->timed_dv, &h->dv, sizeof(Dev)) == 0 && memcmp(&h->timed_nn, &h->nn, sizeof(NnetDev)) == 0;
}

void timed_graph_drop(va_handle h)
{
    if (h->timed_gexec) { (void)hipGraphExecDestroy(h->timed_gexec); h->timed_gexec = nullptr; }
}

int timed_prepare(va_handle h, double rf_scale, int iters)
{
    Dev &dv = h->dv;
    if (h->timed_armed_rf != rf_scale) {
        launch_init_states(dv, PH_START, rf_scale, h->stream);
        h->timed_armed_rf = rf_scale;
    }
    if (!h->tune_graph || iters < 8) { timed_graph_drop(h); return VA_OK; }
    const int chunk = timed_chunk_of(iters);
    // (the fields run_eval writes, as it will leave them: the comparison below must not see a stale line-search launch)
    h->dv.lsrun = 0;
    h->dv.epi = (h->is_nnet ? !h->nn.small : !h->fold) ? EPI_NONE : EPI_FINALIZE;
    if (timed_graph_current(h, chunk)) return VA_OK;
    timed_graph_drop(h);
    hipGraph_t g = nullptr;
    if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        for (int i = 0; i < chunk; ++i) run_eval(h, EPI_FINALIZE);
        if (hipStreamEndCapture(h->stream, &g) != hipSuccess || !g ||
            hipGraphInstantiate(&h->timed_gexec, g, nullptr, nullptr, 0) != hipSuccess) h->timed_gexec = nullptr;
        if (g) (void)hipGraphDestroy(g);
    }
    if (!h->timed_gexec) { (void)hipGetLastError(); return VA_OK; }      // plain launches then
    h->timed_chunk = chunk;
    memcpy(&h->timed_dv, &h->dv, sizeof(Dev)); memcpy(&h->timed_nn, &h->nn, sizeof(NnetDev));
    (void)hipGraphUpload(h->timed_gexec, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    return VA_OK;
}
}  // namespace

int va_eval_timed_prepare(va_handle h, double rf_scale, int32_t iters)
{
    if (!h || iters < 1 || !(rf_scale >= 0.0)) return fail(VA_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    return timed_prepare(h, rf_scale, iters);
}

int va_eval_timed(va_handle h, double rf_scale, int32_t iters, float *elapsed_ms)
{
    if (!h || !elapsed_ms || iters < 1 || !(rf_scale >= 0.0)) return fail(VA_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    Dev &dv = h->dv;
    // Each launch forms A, me, fe and the full gradient.  After va_eval_timed_prepare(h, rf_scale, iters) nothing
    // below but the launches themselves is issued: the seeds are armed and the chunk's graph is instantiated and
    // uploaded; without it (or after anything changed the handle) the same preparation happens here first.
    int rc = timed_prepare(h, rf_scale, iters);
    if (rc) return rc;
    const int chunk = timed_chunk_of(iters);
    hipGraphExec_t gexec = (iters >= 8 && timed_graph_current(h, chunk)) ? h->timed_gexec : nullptr;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    int done = 0;
    if (gexec)
        for (; done + chunk <= iters; done += chunk) HIPCHK(hipGraphLaunch(gexec, h->stream));
    for (; done < iters; ++done) run_eval(h, EPI_FINALIZE);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    {
        // (polled: a blocking wait hands the thread to the kernel's scheduler, and its wake-up would be part of
        // whatever wall clock the caller keeps around this call)
        hipError_t q;
        while ((q = hipEventQuery(h->ev1)) == hipErrorNotReady) {}
        if (q != hipSuccess) return fail(VA_EHIP, "hipEventQuery: %s", hipGetErrorString(q));
    }
    HIPCHK(hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
    HIPCHK(hipGetLastError());
    h->n_eval_launch += iters; h->n_seed_evals += (int64_t)iters * dv.dm.B;
    h->n_seed_evals_direct += (int64_t)iters * dv.dm.B;
    return VA_OK;
}

int va_problem_tune(va_handle h, int32_t what, int32_t value)
{
    if (!h) return fail(VA_EINVAL, "null handle");
    timed_graph_drop(h);      // (captured with the old settings)
    switch (what) {
    case VA_TUNE_FOLD:
        if (h->is_nnet) return fail(VA_EUNSUPPORTED, "the network action chooses its tail by the net's size");
        h->fold = value != 0; break;
    case VA_TUNE_GRAD_SC1: h->dv.gaux = value != 0 ? 1 : 0; break;
    case VA_TUNE_PRIO: h->dv.prio = value < 0 ? 0 : (value > 2 ? 2 : value); break;
    case VA_TUNE_GRAPH: h->tune_graph = value != 0; break;
    case VA_TUNE_NNET_FUSED:
        if (!h->is_nnet || !h->nn.Wf) return fail(VA_ESTATE, "not a network handle whose layers fit the fused kernel");
        // value 1: fused; value 2 + t: fused, the first workgroups' starts spread over t microseconds (measurement)
        h->nn.fused = value != 0 ? 1 : 0;
        h->nn.fb_stagger = value >= 2 ? (value - 2) * 100 : h->nn.fb_stagger; break;
    case VA_TUNE_PERSIST: h->tune_persist = value != 0; break;
    case VA_TUNE_PERSIST_ROWS: {
        if (!h->persist) return fail(VA_ESTATE, "the handle does not run the persistent kernel");
        int G = 0, T = 0;
        const Dims &dm = h->dv.dm;
        if (!persist_geometry(dm.N, dm.D, dm.L, dm.NP, dm.NPest, dm.m, dm.disc, PZ_LDS_BYTES, h->pz_maxG, value, &G, &T))
            return fail(VA_EINVAL, "%d rows per workgroup: not an admissible slice (>= 2 rows everywhere, even for SimpsonHermite, "
                                   "at most %d workgroups per seed, LDS)", value, h->pz_maxG);
        h->pz_G = G; h->pz_T = T;
        break;
    }
    default: return fail(VA_EINVAL, "unknown tuning knob %d", what);
    }
    return VA_OK;
}

// ---------------------------------------------------------------- the job's one collective (RCCL)
namespace {

typedef struct ncclComm *ncclComm_t;
struct NcclId { char internal[VA_COMM_ID_BYTES]; };
struct Rccl {
    void *dl = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, NcclId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

int rccl_load()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.dl) return VA_OK;
    void *dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) dl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!dl) return fail(VA_EUNSUPPORTED, "librccl.so not found: %s", dlerror());
    Rccl r;
    r.GetUniqueId = (int (*)(NcclId *))dlsym(dl, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(ncclComm_t *, int, NcclId, int))dlsym(dl, "ncclCommInitRank");
    r.AllGather = (int (*)(const void *, void *, size_t, int, ncclComm_t, hipStream_t))dlsym(dl, "ncclAllGather");
    r.CommDestroy = (int (*)(ncclComm_t))dlsym(dl, "ncclCommDestroy");
    r.GetErrorString = (const char *(*)(int))dlsym(dl, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString)
        return fail(VA_EUNSUPPORTED, "librccl.so lacks an expected symbol");
    r.dl = dl;
    g_rccl = r;
    return VA_OK;
}

// [seed][step] rows of (A, me, fe, p_est..., status) in one buffer: what a rank contributes
__global__ void k_pack_results(const Dev dv, int nbeta, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int w = 4 + dv.dm.NPest;
    if (i >= dv.dm.B * nbeta) return;
    const int b = i / nbeta, k = i - b * nbeta;
    double *o = out + (size_t)i * w;
    const double *a = dv.ame + ((size_t)b * dv.max_beta + k) * 3;
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
    for (int j = 0; j < dv.dm.NPest; ++j) o[3 + j] = dv.pest[((size_t)b * dv.max_beta + k) * dv.dm.NPest + j];
    o[3 + dv.dm.NPest] = (double)dv.status[(size_t)b * dv.max_beta + k];
}

}  // namespace

struct va_comm_s {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
};

int va_comm_unique_id(char id[VA_COMM_ID_BYTES])
{
    if (!id) return fail(VA_EINVAL, "id is NULL");
    int rc = rccl_load();
    if (rc) return rc;
    NcclId nid;
    const int e = g_rccl.GetUniqueId(&nid);
    if (e) return fail(VA_EHIP, "ncclGetUniqueId: %s", g_rccl.GetErrorString(e));
    memcpy(id, nid.internal, VA_COMM_ID_BYTES);
    return VA_OK;
}

int va_comm_create(const char id[VA_COMM_ID_BYTES], int32_t world, int32_t rank, int32_t device, va_comm *out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(VA_EINVAL, "bad argument");
    *out = nullptr;
    int rc = rccl_load();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    NcclId nid;
    memcpy(nid.internal, id, VA_COMM_ID_BYTES);
    va_comm c = new va_comm_s();
    c->world = world; c->rank = rank; c->device = device;
    const int e = g_rccl.CommInitRank(&c->comm, world, nid, rank);
    if (e) { delete c; return fail(VA_EHIP, "ncclCommInitRank: %s", g_rccl.GetErrorString(e)); }
    *out = c;
    return VA_OK;
}

void va_comm_destroy(va_comm c)
{
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

int va_gather_results(va_handle h, va_comm c, int32_t nbeta, double *table, int32_t *status)
{
    if (!h || !c || !table) return fail(VA_EINVAL, "null argument");
    const Dev &dv = h->dv;
    if (nbeta < 1 || nbeta > dv.max_beta) return fail(VA_EINVAL, "nbeta=%d outside [1, max_beta=%d]", nbeta, dv.max_beta);
    if (c->device != h->device) return fail(VA_EINVAL, "communicator lives on device %d, the problem on %d", c->device, h->device);
    HIPCHK(hipSetDevice(h->device));
    const int B = dv.dm.B, w = 4 + dv.dm.NPest;
    const size_t mine = (size_t)B * nbeta * w;
    double *send = nullptr, *recv = nullptr;
    HIPCHK(hipMalloc((void **)&send, sizeof(double) * mine));
    hipError_t e1 = hipMalloc((void **)&recv, sizeof(double) * mine * c->world);
    if (e1 != hipSuccess) { (void)hipFree(send); return fail(VA_ENOMEM, "hipMalloc: %s", hipGetErrorString(e1)); }
    hipLaunchKernelGGL(k_pack_results, dim3((B * nbeta + 255) / 256), dim3(256), 0, h->stream, dv, nbeta, send);
    const int e = g_rccl.AllGather(send, recv, mine, /* ncclFloat64 */ 8, c->comm, h->stream);      // the single collective
    std::vector<double> host(mine * c->world);
    hipError_t e2 = e ? hipSuccess : hipMemcpyAsync(host.data(), recv, sizeof(double) * host.size(), hipMemcpyDeviceToHost, h->stream);
    hipError_t e3 = hipStreamSynchronize(h->stream);
    (void)hipFree(send); (void)hipFree(recv);
    if (e) return fail(VA_EHIP, "ncclAllGather: %s", g_rccl.GetErrorString(e));
    if (e2 != hipSuccess || e3 != hipSuccess) return fail(VA_EHIP, "gather copy: %s", hipGetErrorString(e2 != hipSuccess ? e2 : e3));
    const size_t rows = (size_t)c->world * B * nbeta;
    for (size_t i = 0; i < rows; ++i) {
        memcpy(table + i * (w - 1), host.data() + i * w, sizeof(double) * (w - 1));
        if (status) status[i] = (int32_t)host[i * w + w - 1];
    }
    return VA_OK;
}

int va_lbfgs_timed(va_handle h, int32_t iters, float *ms_update, float *ms_direction)
{
    if (!h || !ms_update || !ms_direction || iters < 1) return fail(VA_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->device));
    Dev &dv = h->dv;
    // (a bounded handle's third launch is k_lbfgsb_dir, whose cost depends on the active set and the breakpoints
    // crossed: it has no "steady state" to arm -- profile it inside a real bounded ladder instead)
    if (dv.dm.bounded) return fail(VA_EUNSUPPORTED, "va_lbfgs_timed times k_direction; a bounded handle runs k_lbfgsb_dir");
    dv.sticky = 1;
    h->timed_armed_rf = -1.0;
    auto dir = [&]() { launch_direction(dv, h->stream); };
    launch_arm_full_history(dv, h->stream);
    launch_update(dv, h->stream); dir();           // warm-up
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < iters; ++i) launch_update(dv, h->stream);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(ms_update, h->ev0, h->ev1));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int i = 0; i < iters; ++i) dir();
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(ms_direction, h->ev0, h->ev1));
    dv.sticky = 0;
    launch_init_states(dv, PH_IDLE, 1.0, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    return VA_OK;
}

int va_eval_ls_timed(va_handle h, double rf_scale, int32_t iters, float *ms_eval)
{
    if (!h || !ms_eval || iters < 1) return fail(VA_EINVAL, "bad argument");
    if (h->is_nnet) return fail(VA_EUNSUPPORTED, "ODE problems only");
    HIPCHK(hipSetDevice(h->device));
    Dev &dv = h->dv;
    float both = 0.f, arm = 0.f;
    h->timed_armed_rf = -1.0;
    for (int pass = 0; pass < 2; ++pass) {
        launch_arm_ls(dv, rf_scale, h->stream);
        if (pass == 0) run_eval(h, EPI_LS);                                  // warm-up
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        for (int i = 0; i < iters; ++i) {
            launch_arm_ls(dv, rf_scale, h->stream);
            if (pass == 0) run_eval(h, EPI_LS);
        }
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        HIPCHK(hipEventSynchronize(h->ev1));
        HIPCHK(hipEventElapsedTime(pass == 0 ? &both : &arm, h->ev0, h->ev1));
    }
    *ms_eval = both - arm;
    launch_init_states(dv, PH_IDLE, 1.0, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    return VA_OK;
}

int va_read_eval_outputs(va_handle h, double *A, double *me, double *fe, double *grad, int64_t ldg)
{
    if (!h) return fail(VA_EINVAL, "null handle");
    const Dev &dv = h->dv;
    if (grad && ldg < dv.dm.ND + dv.dm.NPest) return fail(VA_EINVAL, "ldg < n_var");
    HIPCHK(hipSetDevice(h->device));
    const size_t nb = sizeof(double) * dv.dm.B;
    if (A) HIPCHK(hipMemcpyAsync(A, dv.outA, nb, hipMemcpyDeviceToHost, h->stream));
    if (me) HIPCHK(hipMemcpyAsync(me, dv.outme, nb, hipMemcpyDeviceToHost, h->stream));
    if (fe) HIPCHK(hipMemcpyAsync(fe, dv.outfe, nb, hipMemcpyDeviceToHost, h->stream));
    int rc;
    if (grad && (rc = copy_out(h, dv.gt, grad, ldg, VA_MEM_HOST))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    return VA_OK;
}

int va_debug_read_partials(va_handle h, double *out, int64_t n)
{
    if (!h || !out || n < 0) return fail(VA_EINVAL, "bad argument");
    const Dev &dv = h->dv;
    const int64_t have = (int64_t)dv.dm.B * dv.dm.nchunks * dv.ups;
    if (n > have) return fail(VA_EINVAL, "n=%lld > %lld partials", (long long)n, (long long)have);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(out, dv.upp, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return VA_OK;
}

int va_debug_read_persist(va_handle h, double *out, int64_t n)
{
    if (!h || !out || n < 0) return fail(VA_EINVAL, "bad argument");
    if (!h->dv.pz.stamps) return fail(VA_ESTATE, "the handle has no persistent-kernel buffers");
    if (n > PZ_NSTAMP) return fail(VA_EINVAL, "n > %d", PZ_NSTAMP);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(out, h->dv.pz.stamps, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return VA_OK;
}

int va_get_counters(va_handle h, int64_t *eval_launches, int64_t *seed_evals, int64_t *cycles)
{
    if (!h) return fail(VA_EINVAL, "null handle");
    if (eval_launches) *eval_launches = h->n_eval_launch;
    if (seed_evals) *seed_evals = h->n_seed_evals;
    if (cycles) *cycles = h->n_cycles;
    return VA_OK;
}

}  // extern "C"
