// va_eval_flat.h -- the flat-mapped tile kernel k_eval<RHS, DISC> (any D, any right-hand
// side) and its launcher, as a header: compiled into libvaranneal_amd.so for the built-in
// RHS (generic fallback / cross-check of the column kernels) and into every generated
// right-hand-side module (csrc/va_user_rhs.hip) for user models.
#pragma once
#include "va_device.h"

namespace va {

// ------------------------------------------------------------------ helpers
// order LDS accesses of ONE wave (its lanes exchange data through LDS; LDS is in order within a wave)
__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cross-lane reductions without LDS round trips.  Inside a row of 16 lanes the exchanges are DPP
// moves (row_mirror: i <-> 15-i, row_half_mirror: i <-> 7-i, quad_perm for xor 2 / xor 1): every
// step pairs groups that were disjoint so far, so after four steps each lane holds its row's
// total.  The four rows are then combined through v_readlane (wave-uniform scalars), in a fixed
// order: every lane returns the same, deterministic total.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x)
{
    // every lane has a valid source under these row permutations: `old` is never used, so the
    // value itself stands in for it and no register has to be initialised
    const int xl = __double2loint(x), xh = __double2hiint(x);
    const int lo = __builtin_amdgcn_update_dpp(xl, xl, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(xh, xh, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141, DPP_QUAD_XOR2 = 0x4E, DPP_QUAD_XOR1 = 0xB1;
template <bool MAX>
__device__ __forceinline__ double row16_reduce(double v)
{
    double o;
    o = dpp_mov<DPP_ROW_MIRROR>(v); v = MAX ? fmax(v, o) : v + o;
    o = dpp_mov<DPP_ROW_HALF_MIRROR>(v); v = MAX ? fmax(v, o) : v + o;
    o = dpp_mov<DPP_QUAD_XOR2>(v); v = MAX ? fmax(v, o) : v + o;
    o = dpp_mov<DPP_QUAD_XOR1>(v); v = MAX ? fmax(v, o) : v + o;
    return v;
}
__device__ __forceinline__ double lane_scalar(double x, int i)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), i);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), i);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v = row16_reduce<false>(v);
    return ((lane_scalar(v, 0) + lane_scalar(v, 16)) + lane_scalar(v, 32)) + lane_scalar(v, 48);
}
// The same total through the matrix pipe: two v_mfma_f64_16x16x4_f64 against a matrix of ones.
// With A[i = l&15][k = l>>4] = v the first leaves D[i][*] = sum of the four lanes i, i+16, i+32, i+48;
// lane l holds rows (l>>4) + 4r of it, adds its four, and the second product sums the four lane
// groups.  6 vector instructions instead of 12 + their DPP wait states, fixed order, and the matrix
// pipe is otherwise idle in these kernels.
__device__ __forceinline__ double wave_sum_mfma(double v)
{
    typedef double v4d __attribute__((ext_vector_type(4)));
    const v4d z = {0.0, 0.0, 0.0, 0.0};
    const v4d d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v, 1.0, z, 0, 0, 0);
    const double p = (d1[0] + d1[1]) + (d1[2] + d1[3]);
    const v4d d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(p, 1.0, z, 0, 0, 0);
    return d2[0];
}
__device__ __forceinline__ double wave_max(double v)
{
    v = row16_reduce<true>(v);
    return fmax(fmax(lane_scalar(v, 0), lane_scalar(v, 16)), fmax(lane_scalar(v, 32), lane_scalar(v, 48)));
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one):
// give each XCD a contiguous range of work items so that neighbouring tiles of a
// seed (which share halo rows) meet in the same L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_swizzle(int bid, int nwork)
{
    const int per = (nwork + 7) >> 3;
    return (bid & 7) * per + (bid >> 3);
}

}  // namespace va
#include "va_epilogue.h"
namespace va {

// ------------------------------------------------------------------ K1: eval
// ------------------------------------------------------------------ dense linear part on the matrix cores
// dst[r][c] = sum_k src[r][k] Bt[k][c] for r < rows, c < D (row pitch D in LDS); Bt is a zero-padded DP x DP table in
// global memory (DP a multiple of 16).  v_mfma_f64_16x16x4_f64: lane l feeds A[row = l & 15][k = l >> 4] and
// B[k = l >> 4][col = l & 15]; result register i of lane l is C[row = (l >> 4) + 4 i][col = l & 15].  A wave takes
// 16 x 32 output blocks (two accumulators share the A fragment); the B fragments are 128-byte runs of the table
// (L1 / L2 resident: it is the same for every workgroup).
__device__ __forceinline__ void lin_gemm(const double *src, int rows, double *dst, int D, const double *Bt, int DP, int tid, int nt)
{
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6, lo = lane & 15, hi = lane >> 4;
    const int nRB = (rows + 15) >> 4, nCB2 = (DP + 31) >> 5;
    for (int blk = wave; blk < nRB * nCB2; blk += nw) {
        const int rb = blk / nCB2, cb = blk - rb * nCB2;
        const int row = rb * 16 + lo, c0 = cb * 32 + lo, c1 = c0 + 16;
        const bool two = cb * 32 + 16 < DP;                           // (wave-uniform)
        const double *ar = src + row * D;
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        // K in chunks of KU steps of 4: the fragments of the next chunk (LDS reads of the rows, 128-byte runs of the
        // table from L2) are requested before the 2 KU products of this one are issued
        constexpr int KU = 4;
        const double *bp = Bt + hi * DP + c0;
        double a[KU], b0[KU], b1[KU], an[KU] = {}, b0n[KU] = {}, b1n[KU] = {};
        auto fetch = [&](int k0, double *a_, double *b0_, double *b1_) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int ks = k0 + 4 * u;                            // (wave-uniform)
                a_[u] = 0.0; b0_[u] = 0.0; b1_[u] = 0.0;
                if (ks < D) {
                    a_[u] = (row < rows && ks + hi < D) ? ar[ks + hi] : 0.0;     // (the table is zero beyond D, the staged rows are not)
                    b0_[u] = bp[ks * DP];
                    if (two) b1_[u] = bp[ks * DP + 16];
                }
            }
        };
        fetch(0, a, b0, b1);
        for (int k0 = 0; k0 < D; k0 += 4 * KU) {
            if (k0 + 4 * KU < D) fetch(k0 + 4 * KU, an, b0n, b1n);
#pragma unroll
            for (int u = 0; u < KU; ++u)
                if (k0 + 4 * u < D) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b0[u], acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b1[u], acc1, 0, 0, 0);
                }
#pragma unroll
            for (int u = 0; u < KU; ++u) { a[u] = an[u]; b0[u] = b0n[u]; b1[u] = b1n[u]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rb * 16 + hi + 4 * i;
            if (r < rows) {
                if (c0 < D) dst[r * D + c0] = acc0[i];
                if (two && c1 < D) dst[r * D + c1] = acc1[i];
            }
        }
    }
}

template <class RHS, int DISC>
__global__ __launch_bounds__(EVAL_THREADS) void k_eval(const Dev dv)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Dims &dm = dv.dm;
    const int nwork = dm.B * dm.ntiles;
    const int w = xcd_swizzle(blockIdx.x, nwork);
    if (w >= nwork) return;
    const int b = w / dm.ntiles, tile = w - b * dm.ntiles;
    const SeedState &st = dv.st[b];
    const int phase = st.phase;
    if (phase != PH_START && phase != PH_LS) return;

    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR;
    constexpr int K = EP_GP + RHS::NP;            // partial columns in use (beyond EP_N: parameters RHS_MAX_NP..., a table of their own)
    constexpr int KR = K > EP_N ? K : EP_N;
    const int R = dm.T + HL + HR, RD = R * dm.D;
    TileCtx c;
    c.n0 = tile * dm.T; c.R = R; c.use_d = (phase == PH_LS);
    c.stp = st.stp; c.c = 2.0 * st.rf_scale * dm.cfe;
    c.xs = smem; c.fs = smem + RD; c.qs = smem + 2 * RD;
    constexpr bool LIN = rhs_linear<RHS>::value;
    if constexpr (LIN) c.js = smem + 3 * RD;
    double *red = smem + 3 * RD + (LIN ? dm.T * dm.D : 0);
    c.ps = red + (EVAL_THREADS / 64) * KR;       // [R * NPt], time-dependent parameters only
    c.xg = dv.x + (size_t)b * dm.ld; c.dg = dv.d + (size_t)b * dm.ld;
    c.gtg = dv.gt + (size_t)b * dm.ld;
    c.zg = dv.lb_z ? dv.lb_z + (size_t)b * dm.ld : nullptr;
    c.tmodel = dv.pp.tmodel; c.stim = dv.pp.stim; c.nstim = dv.pp.nstim;
    if (!dm.tdp) tile_params<RHS>(dm, dv.pp, b, c);

    const int tid = threadIdx.x, nt = blockDim.x;
    ThreadAccT<KR> acc;
    acc.clear();
    tile_load<DISC>(dm, dv.pp, c, tid, nt);
    if (dm.tdp) tile_load_p<DISC>(dm, dv.pp, b, c, tid, nt);
    __syncthreads();
    if constexpr (LIN) {                           // f = X A0^T (matrix cores) + rest (tile_f adds it)
        lin_gemm(c.xs, R, c.fs, dm.D, RHS::lin_A0T(), RHS::LIN_DP, tid, nt);
        __syncthreads();
    }
    tile_f<RHS, DISC>(dm, c, tid, nt);
    __syncthreads();
    tile_q<DISC>(dm, dv.pp, c, acc, tid, nt);
    __syncthreads();
    if (dv.pp.rf0_full) {
        tile_qfull<DISC>(dm, dv.pp, c, acc, tid, nt);
        double *t = c.qs; c.qs = c.fs; c.fs = t;
        __syncthreads();
    }
    tile_s<DISC>(dm, c, tid, nt);
    __syncthreads();
    if constexpr (LIN) {                           // J^T s of the linear part: S A0 for the owned rows
        lin_gemm(c.fs + HL * dm.D, dm.T, c.js, dm.D, RHS::lin_A0(), RHS::LIN_DP, tid, nt);
        __syncthreads();
    }
    tile_g<RHS, DISC>(dm, dv.pp, c, acc, tid, nt);
    if (dm.tdp) tile_gp<RHS, DISC>(dm, dv.pp, c, acc, tid, nt);

    // wave64 shuffle reduction, then across the workgroup's waves through LDS
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double v = (k == EP_GMAX) ? wave_max(acc.v[k]) : wave_sum(acc.v[k]);
        if (lane == 0) red[wave * K + k] = v;
    }
    __syncthreads();
    if (wave != 0) return;
    if (tid < K) {
        double v = red[tid];
        for (int ww = 1; ww < nw; ++ww)
            v = (tid == EP_GMAX) ? fmax(v, red[ww * K + tid]) : v + red[ww * K + tid];
        if (tid < EP_N) st_sc1(dv.evp + ((size_t)b * dm.ntiles + tile) * EP_N + tid, v);
        else st_sc1(dv.evp_big + ((size_t)b * dm.ntiles + tile) * dv.npbig + (tid - EP_N), v);
    }
    // the workgroup that completes the seed's partial rows forms A / runs the line-search step
    if (dv.epi != EPI_NONE && arrive_last(dv.cnt_eval + (size_t)b * CNT_STRIDE, (unsigned)dm.ntiles, lane))
        eval_epilogue<true>(dv, b, lane, reinterpret_cast<SeedHot *>(red), dv.epi);
}

// LDS bytes / grid of the flat-mapped kernel
inline size_t eval_flat_lds_bytes(const Dims &dm)
{
    const int HL = dm.disc == DISC_SH ? 2 : 1;
    const int KR = EP_GP + dm.NPt > EP_N ? EP_GP + dm.NPt : EP_N;
    return sizeof(double) * ((size_t)3 * (dm.T + HL + 1) * dm.D + (dm.lin ? (size_t)dm.T * dm.D : 0) + (EVAL_THREADS / 64) * KR
                             + (dm.tdp ? (size_t)(dm.T + HL + 1) * dm.NPt : 0));
}
inline int eval_flat_grid(const Dims &dm) { return ((dm.B * dm.ntiles + 7) / 8) * 8; }

// wide states: opt the kernel in to the CU's full LDS on the CURRENT device (the attribute is per
// device; called once per problem handle, va_problem_create)
template <class RHS>
inline hipError_t prepare_eval_rhs(const Dev &dv)
{
    if (eval_flat_lds_bytes(dv.dm) <= 64 * 1024) return hipSuccess;
    const void *k = nullptr;
    switch (dv.dm.disc) {
    case DISC_EULER: k = (const void *)k_eval<RHS, DISC_EULER>; break;
    case DISC_TRAPEZOID: k = (const void *)k_eval<RHS, DISC_TRAPEZOID>; break;
    case DISC_SH: k = (const void *)k_eval<RHS, DISC_SH>; break;
    default: k = (const void *)k_eval<RHS, DISC_FWDMAP>; break;
    }
    return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <class RHS>
inline void launch_eval_rhs(const Dev &dv, hipStream_t s)
{
    const dim3 grid(eval_flat_grid(dv.dm)), block(EVAL_THREADS);
    const size_t lds = eval_flat_lds_bytes(dv.dm);
    switch (dv.dm.disc) {
    case DISC_EULER: hipLaunchKernelGGL((k_eval<RHS, DISC_EULER>), grid, block, lds, s, dv); break;
    case DISC_TRAPEZOID: hipLaunchKernelGGL((k_eval<RHS, DISC_TRAPEZOID>), grid, block, lds, s, dv); break;
    case DISC_SH: hipLaunchKernelGGL((k_eval<RHS, DISC_SH>), grid, block, lds, s, dv); break;
    default: hipLaunchKernelGGL((k_eval<RHS, DISC_FWDMAP>), grid, block, lds, s, dv); break;
    }
}


}  // namespace va
