// va_device.h -- device-side problem image shared by va_kernels.hip and va_capi.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "va_core.h"
#include "va_tile2.h"
#include "va_tile3.h"
#include "va_tile4.h"
#include "va_tile5.h"

namespace va {

constexpr int EVAL_THREADS = 256;    // 4 waves per workgroup
constexpr int VEC_THREADS = 256;
constexpr int VEC_CHUNK = 1024;      // elements of a seed's vector per workgroup (2 x double2 per lane)

// Arrival counters sit one per 256-byte line: returning atomics on words of one line serialise
// (~10 ns each: 3072 arrivals on 64 adjacent words cost 25 us), on different lines they do not.
constexpr int CNT_STRIDE = 64;

// what the last-arriving wave of an evaluation does with the seed's partial sums (va_epilogue.h)
enum { EPI_NONE = 0, EPI_FINALIZE = 1, EPI_LS = 2 };

// The persistent per-seed kernel's exchange area (va_persist.h): G workgroups per seed
struct Persist {
    unsigned long long *xch;          // [B][2 buffers][G rows][granule pairs of 16 bytes]: one row per workgroup and cycle
    unsigned long long *cycles;       // cycles run by the last launch (summed over seeds)
    int *abort_flag;                  // 0; 1 a poll timed out (workgroups not co-resident); 2 cycle budget exceeded
    double *stamps;                   // measurement builds (-DVA_PZ_STAMPS) leave their per-phase tick sums here
    long long max_cycles;
};

// Everything a kernel needs, passed by value as the kernel argument.
struct Dev {
    Dims dm;
    Geo4 g4;                       // wave-private column-run geometry (emode 4)
    Geo5 g5;                       // streaming column-strip geometry (emode 5)
    const int *ystrip;             // [NS][2] per strip: first data column staged (even), 16-byte pieces per observation row
    int lsrun;                     // this launch may hold line-search points (seeds in PH_LS): stage d as well
    int epi;                       // EPI_*: tail folded into the evaluation kernel
    unsigned ntiles_magic;         // floor(w / ntiles) == umulhi(w, ntiles_magic) for w < B*ntiles
    int gaux;                      // 1: gradient stores write through (sc1)
    int prio;                      // 1: later-dispatched workgroups of a CU issue at higher priority
    int sticky;                    // measurement only (va_lbfgs_timed): last arrivers leave upd / dir set
    int evcols;                    // 8, 16 or 32 >= EP_GP + NP: columns of the eval partial rows in use
    ProblemPtrs pp;
    Opts o;
    // per-seed vectors, stride dm.ld (multiple of 16 doubles -> 128-byte aligned rows)
    double *x, *g, *gt, *d;
    double *S, *Y;                 // [B][m][ld]
    SeedState *st;                 // [B]
    double *evp;                   // [B][ntiles][EP_N]       eval partials
    double *evp_big;               // NULL or [B][ntiles][npbig]: parameter-gradient partials of parameters RHS_MAX_NP, RHS_MAX_NP + 1, ... (flat kernel)
    int npbig;                     // = NP - RHS_MAX_NP when positive
    double *upp;                   // [B][nchunks][ups]       update-kernel dot partials
    double *dpp;                   // [B][nchunks][DP_N]      direction partials
    int ups;                       // = UP_OLD + 4*m
    // ladder + results (device copies; host reads them back once at the end)
    const double *rf_ladder;       // [nbeta]
    int nbeta, max_beta;
    double *ame;                   // [B][max_beta][3]
    double *pest;                  // [B][max_beta][NPest]
    int *status, *nit;             // [B][max_beta]
    long long *nfev;               // [B][max_beta]
    double *minpaths;              // NULL or [B][max_beta][ND+NP]
    int *n_active;
    unsigned *cnt_eval, *cnt_upd, *cnt_dir;   // [B][CNT_STRIDE] arrival counters of the three kernels of a cycle (zero between launches)
    unsigned long long *n_evals;   // seed-evaluations consumed by k_ls since create
    // S1 outputs
    double *outA, *outme, *outfe;  // [B]
    // bounded problems (va_lbfgsb.hip): Cauchy point / subspace minimiser z, reduced gradient r, the copy xp,
    // breakpoints t, variable status iwhere ([B][ld] each); S'Y, S'S, T ([B][3 m m]); d'd of the direction in use ([B])
    double *lb_z, *lb_r, *lb_xp, *lb_t, *lb_mat, *lb_dtd;
    int *lb_iwhere;
    Persist pz;
};

// launch wrappers (va_kernels.hip); all asynchronous on `s`
void launch_eval(const Dev &dv, int rhs, hipStream_t s);
void launch_ls(const Dev &dv, hipStream_t s);
void launch_update(const Dev &dv, hipStream_t s);
void launch_direction(const Dev &dv, hipStream_t s);
void launch_init_states(const Dev &dv, int phase, double rf_scale_or_neg, hipStream_t s);
void launch_arm_full_history(const Dev &dv, hipStream_t s);
void launch_arm_ls(const Dev &dv, double rf, hipStream_t s);
void launch_clamp_x(const Dev &dv, hipStream_t s);
void launch_finalize_eval(const Dev &dv, hipStream_t s);
size_t eval_lds_bytes(const Dev &dv);
size_t update_lds_bytes(const Dims &dm);
hipError_t prepare_eval(const Dev &dv, int rhs);   // once per handle: opt the kernel in to > 64 KiB of LDS on this device
int eval_grid(const Dims &dm);
// bounded problems: L-BFGS-B's direction step in k_direction's place (va_lbfgsb.hip)
void launch_lbfgsb_dir(const Dev &dv, hipStream_t s);
hipError_t prepare_lbfgsb(const Dev &dv);
// the persistent per-seed ladder kernel for the built-in right-hand side (va_persist.h): launch == false opts the
// instantiation in to its LDS on the current device, launch == true launches B * ntiles workgroups, one per CU
hipError_t seed_kernel_builtin(const Dev &dv, bool launch, hipStream_t s);
// streaming column strips (va_eval5.hip)
void launch_eval5(const Dev &dv, hipStream_t s);
hipError_t prepare_eval5(const Dev &dv);
size_t eval5_lds(const Dev &dv);

}  // namespace va
