// va_nnet.h -- device image of the feed-forward-network action (reference:
// varanneal/va_nnet.py:111-255).  The unknown vector of one seed is
//     [ X: M examples x NDnet neuron states (example-major, layers in order) | estimated params ]
// (va_nnet.py:440) and rides in the same x/g/gt/d/S/Y arrays and the same L-BFGS kernels as
// the ODE path (Dims.ND = M*NDnet + NPest, Dims.NPest = 0); only the evaluator differs.
#pragma once
#include "va_device.h"

namespace va {

enum { NNET_SIGMOID = 0, NNET_TANH = 1, NNET_LINEAR = 2, NNET_RELU = 3, NNET_SOFTPLUS = 4,
       NNET_USER = 1000 };   // >= NNET_USER: a generated activation module (va_act_load_module)

constexpr int NN_TILE = 64;      // workgroup output tile: 2 x 2 waves, each 2 x 2 MFMA blocks of 16x16
constexpr int NN_KC = 32;        // K elements staged in LDS per step
constexpr int NN_THREADS = 256;
constexpr int NN_PACK = 8;       // elements per thread of the trial-point kernel (k_nnet_pack)

// layers up to NN_FB_W wide: the forward product and the state-gradient product of every transition in ONE kernel
// (k_nnet_fb, va_nnet_kernels.h): a workgroup marches a block of NN_FB_R examples through the layers
#ifndef NN_FB_ROWS
#define NN_FB_ROWS 32
#endif
#ifndef NN_FB_THR
#define NN_FB_THR 256
#endif
// NN_FB_THREADS / 64 waves share the NN_FB_W / 16 column blocks of a layer; NN_FB_WGS workgroups per CU (their LDS: two
// operand images of NN_FB_R rows)
constexpr int NN_FB_R = NN_FB_ROWS, NN_FB_W = 128, NN_FB_THREADS = NN_FB_THR, NN_FB_PITCH = NN_FB_W + 2;
constexpr int NN_FB_WGS = NN_FB_R == 32 ? 2 : 1;
constexpr int NN_FB_LAYERS = 64; // most layers its per-layer table in LDS holds
constexpr int NN_FB_PF = 8;      // k-steps the B fragments of its products are requested ahead
// k-steps a fragment table holds per column block for a product over K: whole groups of NN_FB_PF, one group of zeros behind
constexpr int nn_fb_steps(int K) { return (((K + 3) / 4 + NN_FB_PF - 1) / NN_FB_PF) * NN_FB_PF + NN_FB_PF; }

constexpr int NN_SMALL = 32;         // widest layer / most examples the single-kernel path handles
constexpr int NN_ROWS_DIRECT = 64;   // partial rows per seed the line-search kernel reduces itself
constexpr int NN_RED_ROWS = 32;      // rows left by k_nnet_rows when there are more
// one workgroup's job: rows [r0, r0+64) x columns [c0, c0+64) of a layer's product
// (layer metadata rides in the entry so a workgroup needs ONE dependent load before its data)
struct NnetTile { int layer, r0, c0, chunk, sn, sn1, offn, offn1, woff, boff, pad0, pad1; };

struct NnetDev {
    int NL, M, NDnet, NDens, NP, NPest, act;
    int Lin, Lout;
    double rm_in, rm_out;          // measurement weights (va_nnet.py:132-147)
    const double *rmm_in, *rmm_out; // NULL, or full matrices [Lin x Lin], [Lout x Lout]: diff . (RM . diff) per example (:136-139)
    const int *lidx_in, *lidx_out; // [Lin], [Lout] observed neuron indices (matrix RM only)
    const int *s, *off;            // [NL] layer widths, [NL+1] offsets inside one example
    const int *woff, *boff;        // [NL-1] offsets of W_n (s[n+1] x s[n], row-major) and b_n in P
    const int *lmap_in, *lmap_out; // [s[0]], [s[NL-1]] -> observed index or -1
    const double *din, *dout;      // [M][Lin], [M][Lout]
    const int *pmap;               // [NP] -> index among the estimated parameters or -1
    const double *Pfix;            // [B][NP] fixed values (entries of estimated parameters unused)
    double *Pw;                    // [B][NP] full parameter vector at the trial point
    double *Xw;                    // [B][ld] neuron states at the trial point x + stp*d
    double *delta;                 // [B][ld]  dA/dz, indexed like X (layer-0 slots unused)
    double *gpart;                 // [B][nmch][NP] parameter-gradient partials per example chunk
    const NnetTile *t1, *t2, *t3;  // job tables of the three product kernels
    int n1, n2, n3, n4;            // workgroups per seed (n4: parameter reduce)
    int mch, nmch;                 // examples per chunk of the weight-gradient product, chunks
    int small;                     // 0, or rows staged per operand (16 / 32) when every layer and M fit one
                                   // 32x32 tile and k_nnet_small does the whole evaluation
    int n0;                        // workgroups per seed of the pack kernel
    int nraw;                      // partial rows per seed written by the kernels (n1 + n2 + n4)
    double *raw;                   // NULL (rows go straight to Dev::evp) or [B][nraw][EP_GP]
    int fused;                     // 1: k_nnet_fb in place of k_nnet_fwd + k_nnet_bwd_x (every layer <= NN_FB_W wide, scalar RM); off unless
                                   // asked for (va_problem_tune): measured SLOWER than the two kernels at c5x (profiles/r04_nnet_fused.txt)
    int fb_stagger, fb_slots;      // ticks of the 100 MHz clock over which the first fb_slots workgroups of k_nnet_fb spread their starts (0: none)
    int *fb_cu;                    // [0] workgroups of k_nnet_fb arrived in this launch (zeroed by k_nnet_wfrag)
    int nfb;                       // its workgroups per seed: blocks of NN_FB_R examples
    double *Wf;                    // [B][wfsz] the weights in MFMA B-fragment order (k_nnet_wfrag, per evaluation)
    const int *wfoff;              // [2 (NL-1)] offset in Wf of transition n's fragments for the first product, [NL-1 + n] for the second
    int wfsz;
};

// launches k_nnet_small (which = 1), k_nnet_fwd (0) or k_nnet_fb (2) of a generated activation module (va_user_act.hip)
typedef void (*NnetActLaunch)(const Dev *, const NnetDev *, void *stream, int which);
hipError_t prepare_nnet_fb(const NnetDev &nn, NnetActLaunch user);

void launch_nnet_eval(const Dev &dv, const NnetDev &nn, hipStream_t s, NnetActLaunch user);

}  // namespace va
