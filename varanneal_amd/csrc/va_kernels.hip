// va_kernels.hip -- gfx950 kernels of the variational-annealing hot path.
//
// One L-BFGS cycle is THREE launches; each ends with a "last arriver" tail (va_epilogue.h):
//   k_eval4 / k_eval3 / k_eval   (A, me, fe, grad A) of every live seed at x (or x + stp*d);
//                the wave that completes a seed's partial sums forms A and either writes the S1
//                outputs or runs one More'-Thuente line-search / beta-ladder step (va_core.h ls_step).
//     k_eval4    narrow states (D <= 64): wave-private column runs, direct-to-LDS staging, no
//                workgroup barrier (va_tile4.h) -- the production kernel of BASELINE configs 1-3.
//     k_eval3    wider states: one workgroup = T = RY*K time rows of one seed, rows (+halo, +ghost
//                columns) staged in LDS (va_tile3.h) -- BASELINE config 4 (D = 200).
//     k_eval     flat mapping, any D, any right-hand side (va_eval_flat.h): generated right-hand
//                sides, time-dependent parameters, full RM matrices; independent cross-check.
//   k_update     x += stp*d, history pair (s, y) into its slot, g <- g_t, and all the inner
//                products the direction needs in ONE sweep over S and Y; the last workgroup of a
//                seed updates the Gram matrices and solves for the compact-form
//                (Byrd-Nocedal-Schnabel) direction coefficients.
//   k_direction  d = cg*g + sum_j cY_j*Y_j + cS_j*S_j; the last workgroup of a seed leaves g.d.
// k_ls / k_finalize_eval remain for the network action, whose evaluation is several kernels.
//
// Reference arithmetic: see va_core.h header.  All fp64.  HBM-bound streaming kernels: nothing
// here is GEMM-shaped, so no MFMA (the network action in va_nnet.hip is, and uses it).
#include "va_device.h"
#include "va_eval_flat.h"
#include "va_eval3.h"
#include "va_eval4.h"
#include "va_tile5.h"
#include "va_persist.h"

namespace va {

size_t eval_lds_bytes(const Dev &dv)
{
    const Dims &dm = dv.dm;
    if (dm.emode == 4) return sizeof(double) * (size_t)dv.g4.NW * dv.g4.WAVE;
    if (dm.emode == 5) return eval5_lds(dv);
    if (dm.emode == 1) return eval_flat_lds_bytes(dm);
    const int HL = dm.disc == DISC_SH ? 2 : 1;
    const size_t elems = (size_t)tile3_stage_elems(dm.maxr, dm.D, dm.ghost, dm.RY, HL + 1) + tile3_s_elems(dm.maxr, dm.D, dm.ghost, dm.RY);
    const size_t strips = (size_t)(dm.NT / 64) * 5 * t3_strip_stride(dm.NP);      // (>= 512 B: the tail stages the seed's state there)
    return sizeof(double) * (elems + (strips < 64 ? 64 : strips));
}

int eval_grid(const Dims &dm) { return ((dm.B * dm.ntiles + 7) / 8) * 8; }

// launch, or (prepare) opt the instantiation in to the LDS it needs on the current device
struct EvalOp { bool prepare; hipStream_t s; hipError_t err; };

template <class KERNEL>
static void eval_op(KERNEL kern, const Dev &dv, int threads, EvalOp &op)
{
    const size_t lds = eval_lds_bytes(dv);
    if (op.prepare) {
        if (lds > 64 * 1024)
            op.err = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return;
    }
    hipLaunchKernelGGL(kern, dim3(eval_grid(dv.dm)), dim3(threads), lds, op.s, dv);
}

template <class RHS, int K, int DC, int NTMAX>
static void eval3_rhs(const Dev &dv, EvalOp &op)
{
    switch (dv.dm.disc) {
    case DISC_EULER: eval_op(k_eval3<RHS, DISC_EULER, K, DC, NTMAX>, dv, dv.dm.NT, op); break;
    case DISC_TRAPEZOID: eval_op(k_eval3<RHS, DISC_TRAPEZOID, K, DC, NTMAX>, dv, dv.dm.NT, op); break;
    case DISC_SH: eval_op(k_eval3<RHS, DISC_SH, K, DC, NTMAX>, dv, dv.dm.NT, op); break;
    default: eval_op(k_eval3<RHS, DISC_FWDMAP, K, DC, NTMAX>, dv, dv.dm.NT, op); break;
    }
}

// D fixed at compile time for the state size of BASELINE config 4 (D = 200, 256-thread groups); any
// other D runs the same kernel with D in a register.
template <class RHS, int K>
static void eval3_d(const Dev &dv, EvalOp &op)
{
    if (dv.dm.D == 200) eval3_rhs<RHS, K, 200, 256>(dv, op);
    else if (dv.dm.D <= 64) eval3_rhs<RHS, K, 0, 256>(dv, op);
    else if (dv.dm.D <= 128) eval3_rhs<RHS, K, 0, 512>(dv, op);
    else if (dv.dm.D <= 256) eval3_rhs<RHS, K, 0, 256>(dv, op);
    else if (dv.dm.D <= 512) eval3_rhs<RHS, K, 0, 512>(dv, op);
    else eval3_rhs<RHS, K, 0, 1024>(dv, op);
}

template <class RHS, int K, int DC, int WS, int SUB>
static void eval4_disc(const Dev &dv, EvalOp &op)
{
    switch (dv.dm.disc) {
    case DISC_EULER: eval_op(k_eval4<RHS, DISC_EULER, K, DC, WS, SUB>, dv, 256, op); break;
    case DISC_TRAPEZOID: eval_op(k_eval4<RHS, DISC_TRAPEZOID, K, DC, WS, SUB>, dv, 256, op); break;
    case DISC_SH: eval_op(k_eval4<RHS, DISC_SH, K, DC, WS, SUB>, dv, 256, op); break;
    default: eval_op(k_eval4<RHS, DISC_FWDMAP, K, DC, WS, SUB>, dv, 256, op); break;
    }
}

// sub-tiles per wave: 1, or 2 / 3 for grids of a few wave-tiles per SIMD (compiled for the
// scalar-weight variant with D fixed only: that is where such grids are the BASELINE configs)
template <class RHS, int K, int DC, int WS>
static void eval4_rhs(const Dev &dv, EvalOp &op)
{
    // (measured at C3, SUB = 3 -- one wave per SIMD, 256 workgroups -- against SUB = 1 -- three waves
    // per SIMD: 8.7 vs 8.0 us.  A lone wave has nothing to hide its LDS and matrix-pipe latencies
    // behind, so only SUB = 1 is compiled; the kernel keeps the loop.)
    eval4_disc<RHS, K, DC, WS, 1>(dv, op);
}

// D = 20 (examples/Lorenz96_D20, BASELINE configs 1-3) is compiled with D as a constant; scalar RM /
// RF0 with data at every model time (the same configs) run the variant without weight registers
template <class RHS, int K>
static void eval4_d(const Dev &dv, EvalOp &op)
{
    const bool sw = !dv.pp.rm_arr && !dv.pp.rf0_arr, ws = sw && dv.dm.nskip == 1;
    if (dv.dm.D == 20) {
        if (ws) eval4_rhs<RHS, K, 20, 1>(dv, op);
        else if (sw) eval4_rhs<RHS, K, 20, 2>(dv, op);        // scalar weights, data at every nskip-th row
        else eval4_rhs<RHS, K, 20, 0>(dv, op);
    } else { if (ws) eval4_rhs<RHS, K, 0, 1>(dv, op); else eval4_rhs<RHS, K, 0, 0>(dv, op); }
}

static void eval_dispatch(const Dev &dv, int rhs, EvalOp &op)
{
    (void)rhs;                 // VA_RHS_LORENZ96 is the only built-in RHS
    if (dv.dm.emode == 5) {
        if (op.prepare) op.err = prepare_eval5(dv);
        else launch_eval5(dv, op.s);
    } else if (dv.dm.emode == 4) {
        switch (dv.dm.maxr) {
        case 4: eval4_d<RhsL96s, 4>(dv, op); break;
        case 5: eval4_d<RhsL96s, 5>(dv, op); break;
        case 6: eval4_d<RhsL96s, 6>(dv, op); break;
        case 7: eval4_d<RhsL96s, 7>(dv, op); break;
        // runs of 12 rows at two workgroups per CU (D = 20, scalar weights, data at every row or every nskip-th: Simpson-Hermite at
        // the C3 shape in ONE round of resident workgroups, see the chooser in va_capi.hip)
        case 12: if (dv.dm.nskip == 1) eval4_rhs<RhsL96s, 12, 20, 1>(dv, op); else eval4_rhs<RhsL96s, 12, 20, 2>(dv, op); break;
        default: eval4_d<RhsL96s, 8>(dv, op); break;
        }
    } else if (dv.dm.emode == 3) {
        switch (dv.dm.maxr) {
        case 4: eval3_d<RhsL96g, 4>(dv, op); break;
        case 5: eval3_d<RhsL96g, 5>(dv, op); break;
        case 6: eval3_d<RhsL96g, 6>(dv, op); break;
        case 7: eval3_d<RhsL96g, 7>(dv, op); break;
        default: eval3_d<RhsL96g, 8>(dv, op); break;
        }
    } else {
        if (op.prepare) op.err = prepare_eval_rhs<RhsL96>(dv);
        else launch_eval_rhs<RhsL96>(dv, op.s);
    }
}

void launch_eval(const Dev &dv, int rhs, hipStream_t s)
{
    EvalOp op{false, s, hipSuccess};
    eval_dispatch(dv, rhs, op);
}

hipError_t prepare_eval(const Dev &dv, int rhs)
{
    EvalOp op{true, nullptr, hipSuccess};
    eval_dispatch(dv, rhs, op);
    return op.err;
}

hipError_t seed_kernel_builtin(const Dev &dv, bool launch, hipStream_t s) { return seed_kernel_op<RhsL96>(dv, launch, s); }

// ------------------------------------------------------------------ K2: tails as kernels of their own
// (the network action, whose evaluation is several kernels, and grids too large to fold the tail
// into the evaluation kernel).  One workgroup per seed; up to 16 waves share the row reduction
// (C4: 2500 partial rows per seed), wave 0 runs the tail.
constexpr int TAIL_MAX_WAVES = 16;
static int tail_waves(const Dims &dm) { const int w = (dm.nprow + 127) / 128; return w < 1 ? 1 : (w > TAIL_MAX_WAVES ? TAIL_MAX_WAVES : w); }

__global__ __launch_bounds__(64 * TAIL_MAX_WAVES) void k_ls(const Dev dv)
{
    __shared__ SeedHot sh;
    __shared__ double part[TAIL_MAX_WAVES * 32];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int phase = dv.st[b].phase;
    if (phase != PH_START && phase != PH_LS) return;
    double ev[EP_N];
    reduce_eval_block(dv, b, lane, wave, nw, part, ev);
    if (wave == 0) eval_epilogue<false>(dv, b, lane, &sh, EPI_LS, ev);
}
void launch_ls(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_ls, dim3(dv.dm.B), dim3(64 * tail_waves(dv.dm)), 0, s, dv);
}

// S1 tail: A, me, fe and the parameter tail for a plain evaluation.
__global__ __launch_bounds__(64 * TAIL_MAX_WAVES) void k_finalize_eval(const Dev dv)
{
    __shared__ double part[TAIL_MAX_WAVES * 32];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    double ev[EP_N];
    reduce_eval_block(dv, b, lane, wave, nw, part, ev);
    if (wave == 0) eval_epilogue<false>(dv, b, lane, nullptr, EPI_FINALIZE, ev);
}
void launch_finalize_eval(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_finalize_eval, dim3(dv.dm.B), dim3(64 * tail_waves(dv.dm)), 0, s, dv);
}

// reset every seed: phase, ladder position, RF.  rf < 0 -> take rf_ladder[0].
__global__ void k_init_states(const Dev dv, int phase, double rf)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= dv.dm.B) return;
    SeedState &s = dv.st[b];
    s.phase = phase; s.beta_idx = 0; s.iter = 0; s.col = 0; s.head = 0; s.ifun = 0; s.iback = 0;
    s.ls_task = LS_START; s.upd = 0; s.slot = 0; s.dir = 0; s.store_idx = -1; s.nold = 0;
    s.nfev = 0; s.f = 0.0; s.fold = 0.0; s.me = 0.0; s.fe = 0.0; s.theta = 1.0; s.stp = 0.0;
    s.stp_upd = 0.0; s.gd = 0.0; s.gdold = 0.0; s.gn2 = 0.0; s.dr = 0.0; s.cg = -1.0; s.gd_dir = 0.0; s.stpmx = 1e10;
    s.rf_scale = rf < 0.0 ? dv.rf_ladder[0] : rf;
    dv.cnt_eval[(size_t)b * CNT_STRIDE] = 0u; dv.cnt_upd[(size_t)b * CNT_STRIDE] = 0u; dv.cnt_dir[(size_t)b * CNT_STRIDE] = 0u;
}
void launch_init_states(const Dev &dv, int phase, double rf, hipStream_t s)
{
    hipLaunchKernelGGL(k_init_states, dim3((dv.dm.B + 63) / 64), dim3(64), 0, s, dv, phase, rf);
}

// bounded problems: the start point is projected onto the box (as L-BFGS-B does with x0)
__global__ void k_clamp_x(const Dev dv)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dv.dm.ld) return;
    double *x = dv.x + (size_t)blockIdx.y * dv.dm.ld;
    x[i] = fmin(fmax(x[i], dv.pp.lo[i]), dv.pp.hi[i]);
}
void launch_clamp_x(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_clamp_x, dim3((dv.dm.ld + 255) / 256, dv.dm.B), dim3(256), 0, s, dv);
}

// measurement only (va_lbfgs_timed): every seed as if in the middle of a long minimisation -- history
// full, a pair to store, a direction to form -- so that k_update / k_direction move the bytes they
// move in the steady state of a ladder
__global__ void k_arm_full_history(const Dev dv)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= dv.dm.B) return;
    SeedState &s = dv.st[b];
    const int m = dv.o.m;
    s.phase = PH_IDLE; s.col = m; s.head = 1 % m; s.nold = m - 1; s.slot = 0;
    for (int j = 0; j < m; ++j) s.order[j] = (j + 1) % m;
    s.upd = UPD_G | UPD_HIST; s.dir = 1; s.stp_upd = 1.0; s.dr = 1.0; s.theta = 1.0; s.store_idx = -1;
}
// measurement only (va_eval_ls_timed): every seed at a line-search trial point whose outcome is "another trial"
// (the sufficient-decrease test cannot pass against finit = -1e300), so that an evaluation launch does what it does
// inside a ladder cycle -- x + stp*d formed from the two staged images, g.d / g.g / max|g|, one More'-Thuente step in
// the tail -- and can be repeated after re-arming
__global__ void k_arm_ls(const Dev dv, double rf)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= dv.dm.B) return;
    SeedState &s = dv.st[b];
    s.phase = PH_LS; s.ifun = 1; s.iback = 0; s.stp = 1e-3; s.ls_task = LS_FG; s.rf_scale = rf;
    s.iter = 1; s.nfev = 2; s.f = 1.0; s.fold = 1.0; s.gd = -1.0; s.gdold = -1.0; s.stpmx = 1e10;
    LsState &l = s.ls;
    l.brackt = 0; l.stage = 1; l.ginit = -1.0; l.gtest = -1e-3; l.gx = -1.0; l.gy = -1.0;
    l.finit = -1e300; l.fx = -1e300; l.fy = -1e300; l.stx = 0.0; l.sty = 0.0; l.stmin = 0.0; l.stmax = 5e-3;
    l.width = 1e10; l.width1 = 2e10;
}
void launch_arm_ls(const Dev &dv, double rf, hipStream_t s)
{
    hipLaunchKernelGGL(k_arm_ls, dim3((dv.dm.B + 63) / 64), dim3(64), 0, s, dv, rf);
}
void launch_arm_full_history(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_arm_full_history, dim3((dv.dm.B + 63) / 64), dim3(64), 0, s, dv);
}

// ------------------------------------------------------------------ K3: update + inner products + coefficients
// wave-uniform broadcast of lane i's double through SGPRs (v_readlane), i uniform
__device__ __forceinline__ double bcast_lane(double x, int i)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), i);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), i);
    return __hiloint2double(hi, lo);
}

// doubles of LDS the coefficient solve needs for history length m
__host__ __device__ inline int coeffs_lds_doubles(int m) { return UP_N + 2 * m * m + 3 * MAX_M; }

// ONE wave, lane j <-> j-th oldest history pair: Gram update, then the compact form of the
// two-loop recursion (va_core.h: direction_coeffs_view, which the CPU emulator runs serially):
//   p = R^-1 a,  q = (D + gamma Y'Y) p - gamma b,  u = R^-T q,  d = -gamma g - S u + gamma Y p.
// Both triangular solves are column-oriented, so a step is one SGPR broadcast + one fused update
// per lane -- no cross-lane reductions and no serial O(m^2) loop.  Run by the last workgroup of
// the seed in k_update; `lds`: coeffs_lds_doubles(m) doubles private to the calling wave.
__device__ __forceinline__ void coeffs_wave(const Dev &dv, int b, int lane, double *lds)
{
    const Dims &dm = dv.dm;
    const int M = MAX_M, m = dm.m;
    double *up = lds, *sSY = up + UP_N, *sYY = sSY + m * m, *sp = sYY + m * m, *cYs = sp + M, *cSs = cYs + M;
    SeedState &s = dv.st[b];
    const int nold = s.nold, col = s.col, sn = s.slot;
    const bool hist = (s.upd & UPD_HIST) != 0;
    const double dr = s.dr;
    double theta = s.theta;
    const int K = UP_OLD + 4 * nold;
    for (int k = lane; k < K; k += 64)
        up[k] = col_reduce<true>(dv.upp + (size_t)b * dm.nchunks * dv.ups + k, dm.nchunks, dv.ups, 0, 1, false);
    const int myslot = lane < col ? s.order[lane] : 0;
    for (int e0 = 0; e0 < m * m; e0 += 256) {     // the m x m blocks of the Gram matrices: loads first, LDS stores after
        double t0[4], t1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + lane + 64 * u;
            const int i = e / m, j = e - i * m;
            t0[u] = e < m * m ? s.SY[i * M + j] : 0.0;
            t1[u] = e < m * m ? s.YY[i * M + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + lane + 64 * u;
            if (e < m * m) { sSY[e] = t0[u]; sYY[e] = t1[u]; }
        }
    }
    if (lane < M) { cYs[lane] = 0.0; cSs[lane] = 0.0; }
    wave_sync_lds();
    if (hist) {
        // new column/row of the Gram matrices (LDS copy and the persistent one)
        if (lane < nold) {
            const double sjy = up[UP_OLD + 4 * lane + 2], yjy = up[UP_OLD + 4 * lane + 3];
            sSY[myslot * m + sn] = sjy; sYY[myslot * m + sn] = yjy; sYY[sn * m + myslot] = yjy;
            s.SY[myslot * M + sn] = sjy; s.YY[myslot * M + sn] = yjy; s.YY[sn * M + myslot] = yjy;
        }
        if (lane == 0) {
            sSY[sn * m + sn] = dr; sYY[sn * m + sn] = up[UP_YY];   // s.y as the line search saw it
            s.SY[sn * M + sn] = dr; s.YY[sn * M + sn] = up[UP_YY];
        }
        theta = up[UP_YY] / dr;
    }
    wave_sync_lds();
    double aj = 0.0, bj = 0.0;
    if (lane < nold) { aj = up[UP_OLD + 4 * lane + 0]; bj = up[UP_OLD + 4 * lane + 1]; }
    if (hist && lane == col - 1) { aj = up[UP_SGT]; bj = up[UP_YGT]; }
    const double gamma = 1.0 / theta;
    const double rjj = lane < col ? sSY[myslot * m + myslot] : 1.0;
    const double rinv = 1.0 / rjj;
    // p = R^-1 a  (R_ji = S_j . Y_i for j <= i)
    double pj = 0.0;
    for (int i = col - 1; i >= 0; --i) {
        const int si = __builtin_amdgcn_readlane(myslot, i);
        const double pi = bcast_lane(aj * rinv, i);
        if (lane == i) pj = pi;
        if (lane < i) aj -= sSY[myslot * m + si] * pi;
    }
    if (lane < M) sp[lane] = pj;
    wave_sync_lds();
    // q = (D + gamma Y'Y) p - gamma b
    double qj = 0.0;
    if (lane < col) {
        double acc = 0.0;
        for (int k = 0; k < col; ++k) acc += sYY[myslot * m + __builtin_amdgcn_readlane(myslot, k)] * sp[k];
        qj = rjj * pj + gamma * acc - gamma * bj;
    }
    // u = R^-T q  ((R^T)_ji = R_ij = S_i . Y_j for i <= j)
    double uj = 0.0;
    for (int i = 0; i < col; ++i) {
        const int si = __builtin_amdgcn_readlane(myslot, i);
        const double ui = bcast_lane(qj * rinv, i);
        if (lane == i) uj = ui;
        if (lane > i && lane < col) qj -= sSY[si * m + myslot] * ui;
    }
    if (lane < col) { cYs[myslot] = gamma * pj; cSs[myslot] = -uj; }
    wave_sync_lds();
    if (lane < M) { s.cY[lane] = cYs[lane]; s.cS[lane] = cSs[lane]; }
    if (lane == 0) { s.cg = -gamma; s.theta = theta; }
}

size_t update_lds_bytes(const Dims &dm)
{
    const int red = (VEC_THREADS / 64) * UP_N, co = coeffs_lds_doubles(dm.m);
    return sizeof(double) * (size_t)(red > co ? red : co);
}

__global__ __launch_bounds__(VEC_THREADS) void k_update(const Dev dv)
{
    extern __shared__ __attribute__((aligned(16))) double red[];      // [(VEC_THREADS/64)*UP_N], then the coefficient solve
    const Dims &dm = dv.dm;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const SeedState &s = dv.st[b];
    const int upd = s.upd, dir = s.dir;
    if (!upd && !dir) return;
    const int nold = dir ? s.nold : 0;
    const bool hist = (upd & UPD_HIST) != 0;
    const double stp = s.stp_upd;
    const size_t vo = (size_t)b * dm.ld;
    double *x = dv.x + vo, *g = dv.g + vo;
    const double *gt = dv.gt + vo, *d = dv.d + vo;
    double *Sn = dv.S + ((size_t)b * dm.m + s.slot) * dm.ld;
    double *Yn = dv.Y + ((size_t)b * dm.m + s.slot) * dm.ld;
    double *mp = nullptr;
    if ((upd & UPD_STORE) && dv.minpaths)
        mp = dv.minpaths + ((size_t)b * dv.max_beta + s.store_idx) * (dm.ND + dm.NP);

    constexpr int E = VEC_CHUNK / VEC_THREADS / 2;      // double2 per lane
    const int tid = threadIdx.x;
    double2 gtv[E], yv[E];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0;
    int idx[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = chunk * VEC_CHUNK + (e * VEC_THREADS + tid) * 2;
        idx[e] = i;
        gtv[e] = make_double2(0.0, 0.0); yv[e] = make_double2(0.0, 0.0);
        if (i >= dm.ld) continue;
        const double2 dv2 = *reinterpret_cast<const double2 *>(d + i);
        const double2 gv = *reinterpret_cast<const double2 *>(g + i);
        const double2 tv = *reinterpret_cast<const double2 *>(gt + i);
        double2 xv = *reinterpret_cast<const double2 *>(x + i);
        gtv[e] = tv;
        if (upd & UPD_X) {
            // (the same expression as the evaluated trial point, clamp into the box included)
            const double *zb = dv.lb_z ? dv.lb_z + vo : nullptr;            // (bounded: step 1 lands on z itself, as the evaluation did)
            xv.x = trial_b(xv.x, stp, dv2.x, zb, dv.pp, i); xv.y = trial_b(xv.y, stp, dv2.y, zb, dv.pp, i + 1);
            *reinterpret_cast<double2 *>(x + i) = xv;
        }
        if (upd & UPD_STORE) {
            // the lane that owns an element stores it (va_ode.py:776): path entries as they
            // are, estimated parameters scattered to their Pidx slot of the full vector.
            const double vals[2] = {xv.x, xv.y};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int ii = i + u;
                if (ii < dm.ND) { if (mp) mp[ii] = vals[u]; }
                else if (ii < dm.ND + dm.NPest) {
                    const int k = ii - dm.ND;
                    dv.pest[((size_t)b * dv.max_beta + s.store_idx) * dm.NPest + k] = vals[u];
                    if (mp) mp[dm.ND + dv.pp.Pidx[k]] = vals[u];
                }
            }
        }
        double2 sv = make_double2(0.0, 0.0);
        if (hist) {
            sv = make_double2(stp * dv2.x, stp * dv2.y);
            yv[e] = make_double2(tv.x - gv.x, tv.y - gv.y);
            *reinterpret_cast<double2 *>(Sn + i) = sv;
            *reinterpret_cast<double2 *>(Yn + i) = yv[e];
        }
        if (upd & UPD_G) *reinterpret_cast<double2 *>(g + i) = tv;
        a0 += yv[e].x * tv.x + yv[e].y * tv.y;          // y.gt
        a1 += sv.x * tv.x + sv.y * tv.y;                // s.gt
        a2 += yv[e].x * yv[e].x + yv[e].y * yv[e].y;    // y.y
        a3 += sv.x * yv[e].x + sv.y * yv[e].y;          // s.y
        a4 += tv.x * tv.x + tv.y * tv.y;                // gt.gt
    }
    if (mp && chunk == 0 && tid == 0) {
        // fixed (non-estimated) parameters of the stored step come from P
        for (int j = 0; j < dm.NP; ++j) {
            bool est = false;
            for (int k = 0; k < dm.NPest; ++k) est = est || (dv.pp.Pidx[k] == j);
            if (!est) mp[dm.ND + j] = dv.pp.Pfull[(size_t)b * dm.NP + j];
        }
    }
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = UP_OLD + 4 * nold;
    if (dir) {
        {
            double v;
            v = wave_sum(a0); if (lane == 0) red[wave * UP_N + UP_YGT] = v;
            v = wave_sum(a1); if (lane == 0) red[wave * UP_N + UP_SGT] = v;
            v = wave_sum(a2); if (lane == 0) red[wave * UP_N + UP_YY] = v;
            v = wave_sum(a3); if (lane == 0) red[wave * UP_N + UP_SY] = v;
            v = wave_sum(a4); if (lane == 0) red[wave * UP_N + UP_GTGT] = v;
        }
        for (int j = 0; j < nold; ++j) {
            const int sj = s.order[j];
            const double *Sj = dv.S + ((size_t)b * dm.m + sj) * dm.ld;
            const double *Yj = dv.Y + ((size_t)b * dm.m + sj) * dm.ld;
            double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (idx[e] >= dm.ld) continue;
                const double2 sv = *reinterpret_cast<const double2 *>(Sj + idx[e]);
                const double2 yj = *reinterpret_cast<const double2 *>(Yj + idx[e]);
                b0 += sv.x * gtv[e].x + sv.y * gtv[e].y;    // S_j . gt
                b1 += yj.x * gtv[e].x + yj.y * gtv[e].y;    // Y_j . gt
                b2 += sv.x * yv[e].x + sv.y * yv[e].y;      // S_j . y
                b3 += yj.x * yv[e].x + yj.y * yv[e].y;      // Y_j . y
            }
            double v;
            v = wave_sum(b0); if (lane == 0) red[wave * UP_N + UP_OLD + 4 * j + 0] = v;
            v = wave_sum(b1); if (lane == 0) red[wave * UP_N + UP_OLD + 4 * j + 1] = v;
            v = wave_sum(b2); if (lane == 0) red[wave * UP_N + UP_OLD + 4 * j + 2] = v;
            v = wave_sum(b3); if (lane == 0) red[wave * UP_N + UP_OLD + 4 * j + 3] = v;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    if (dir) {
        for (int k = lane; k < K; k += 64) {
            double v = red[k];
#pragma unroll
            for (int ww = 1; ww < VEC_THREADS / 64; ++ww) v += red[ww * UP_N + k];
            st_sc1(dv.upp + ((size_t)b * dm.nchunks + chunk) * dv.ups + k, v);
        }
    }
    // the seed's last workgroup: direction coefficients (the former k_coeffs), then the update
    // request is consumed -- a finished seed must not have it applied again next cycle
    if (!arrive_last(dv.cnt_upd + (size_t)b * CNT_STRIDE, (unsigned)dm.nchunks, lane)) return;
    if (dir) coeffs_wave(dv, b, lane, red);
    if (lane == 0 && !dv.sticky) dv.st[b].upd = 0;
}
void launch_update(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_update, dim3(dv.dm.nchunks, dv.dm.B), dim3(VEC_THREADS), update_lds_bytes(dv.dm), s, dv);
}

// ------------------------------------------------------------------ K5: direction
__global__ __launch_bounds__(VEC_THREADS) void k_direction(const Dev dv)
{
    __shared__ double red[(VEC_THREADS / 64) * DP_N];
    const Dims &dm = dv.dm;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const SeedState &s = dv.st[b];
    if (!s.dir) return;
    const size_t vo = (size_t)b * dm.ld;
    const double *g = dv.g + vo;
    double *d = dv.d + vo;
    const int col = s.col;
    const double cg = s.cg;
    constexpr int E = VEC_CHUNK / VEC_THREADS / 2;
    const int tid = threadIdx.x;
    double2 acc[E], gv[E];
    int idx[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        idx[e] = chunk * VEC_CHUNK + (e * VEC_THREADS + tid) * 2;
        gv[e] = make_double2(0.0, 0.0);
        if (idx[e] < dm.ld) gv[e] = *reinterpret_cast<const double2 *>(g + idx[e]);
        acc[e] = make_double2(cg * gv[e].x, cg * gv[e].y);
    }
    for (int j = 0; j < col; ++j) {
        const int sj = s.order[j];
        const double cy = s.cY[sj], cs = s.cS[sj];
        const double *Sj = dv.S + ((size_t)b * dm.m + sj) * dm.ld;
        const double *Yj = dv.Y + ((size_t)b * dm.m + sj) * dm.ld;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (idx[e] >= dm.ld) continue;
            const double2 yv = *reinterpret_cast<const double2 *>(Yj + idx[e]);
            const double2 sv = *reinterpret_cast<const double2 *>(Sj + idx[e]);
            acc[e].x += cy * yv.x; acc[e].y += cy * yv.y;
            acc[e].x += cs * sv.x; acc[e].y += cs * sv.y;
        }
    }
    double gd = 0.0, dd = 0.0, smx = 1e10;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (idx[e] >= dm.ld) continue;
        *reinterpret_cast<double2 *>(d + idx[e]) = acc[e];
        gd += gv[e].x * acc[e].x + gv[e].y * acc[e].y;
        dd += acc[e].x * acc[e].x + acc[e].y * acc[e].y;
    }
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    gd = wave_sum(gd); dd = wave_sum(dd); smx = -wave_max(-smx);
    if (lane == 0) { red[wave * DP_N + DP_GD] = gd; red[wave * DP_N + DP_DD] = dd; red[wave * DP_N + DP_STPMX] = smx; }
    __syncthreads();
    if (wave != 0) return;
    if (lane < DP_N) {
        double v = red[lane];
#pragma unroll
        for (int ww = 1; ww < VEC_THREADS / 64; ++ww) v = (lane == DP_STPMX) ? fmin(v, red[ww * DP_N + lane]) : v + red[ww * DP_N + lane];
        st_sc1(dv.dpp + ((size_t)b * dm.nchunks + chunk) * DP_N + lane, v);
    }
    // the seed's last workgroup leaves g.d (and the step limits of a bounded problem) for the line
    // search and consumes the request
    if (!arrive_last(dv.cnt_dir + (size_t)b * CNT_STRIDE, (unsigned)dm.nchunks, lane)) return;
    double tot[DP_N];
#pragma unroll
    for (int c = 0; c < DP_N; ++c) {
        // lane r takes rows r, r+64, ...; the 64 lane totals meet through shuffles in a fixed order
        double v = c == DP_STPMX ? 1e10 : 0.0;
        for (int t = lane; t < dm.nchunks; t += 64) {
            const double q = ld_sc1(dv.dpp + ((size_t)b * dm.nchunks + t) * DP_N + c);
            v = c == DP_STPMX ? fmin(v, q) : v + q;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double q = __shfl_down(v, o, 64);
            v = c == DP_STPMX ? fmin(v, q) : v + q;
        }
        tot[c] = v;
    }
    if (lane == 0) {
        SeedState &s = dv.st[b];
        s.gd_dir = tot[DP_GD];
        if (!dv.sticky) s.dir = 0;
    }
}
void launch_direction(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_direction, dim3(dv.dm.nchunks, dv.dm.B), dim3(VEC_THREADS), 0, s, dv);
}

}  // namespace va
