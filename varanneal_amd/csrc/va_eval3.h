// va_eval3.h -- the workgroup column-run evaluation kernel (wide states: 64 < D <= 1024) as a template over
// the right-hand side, so that the library (built-in Lorenz-96, va_kernels.hip) and generated modules
// (va_user_rhs.hip: a traced stencil in ghosted column form, codegen.ghost_form) instantiate the same code.
#pragma once
#include "va_device.h"
#include "va_epilogue.h"
#include "va_eval_flat.h"
#include "va_tile3.h"

namespace va {

// per-wave reduction strip of k_eval3: 4 rows of ST totals + ST wave totals, ST = 8 values for models with up to
// 3 parameters (EP_GP + NP <= 8), else 32 (EP_N <= 32)
VA_HD constexpr int t3_strip_stride(int np) { return EP_GP + np <= 8 ? 8 : 32; }

// ------------------------------------------------------------------ K1 (wide states): column-run
template <class RHS, int DISC, int K, int DC, int NTMAX>
__global__ __launch_bounds__(NTMAX) void k_eval3(const Dev dv)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const Dims &dm = dv.dm;
    const int nwork = dm.B * dm.ntiles;
    const int w = xcd_swizzle(blockIdx.x, nwork);
    if (w >= nwork) return;
    const int b = w / dm.ntiles, tile = w - b * dm.ntiles;

    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR, G = RHS::GHOST;
    constexpr int KP = EP_GP + RHS::NP;
    // staged double2 per lane = ceil(R / RP), RP = 2*NT/D rows per pass
    constexpr int DCs = DC > 0 ? DC : 2;
    constexpr int RPc = 2 * tile3_threads(DCs) / DCs;
    constexpr int NS = DC > 0 ? (tile3_RY(DCs) * K + HL + HR + RPc - 1) / RPc : tile3_ns_runtime(K);
    // with D fixed at compile time the whole tile geometry (and every LDS offset) is constant
    const int D = DC > 0 ? DC : dm.D;
    const int RY = DC > 0 ? tile3_RY(DC > 0 ? DC : 1) : dm.RY;
    const int T = RY * K;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int n0 = tile * T;
    const bool edge = (n0 - HL < 0) || (n0 + T + HR > dm.N);          // workgroup-uniform
    const bool evenD = (D & 1) == 0;
    const double *xg = dv.x + (size_t)b * dm.ld;

    // phase A step 1: x loads in flight before anything else is waited for
    double xr[NS][2];
    if (evenD) {
        if (edge) tile3_stage_load<DISC, K, DC, true, NS>(dm, n0, xg, tid, nt, xr);
        else tile3_stage_load<DISC, K, DC, false, NS>(dm, n0, xg, tid, nt, xr);
    }

    const SeedState &st = dv.st[b];
    const int phase = st.phase;
    if (phase != PH_START && phase != PH_LS) return;

    const int ty = tid / D, tx = tid - ty * D;
    const bool active = ty < RY;
    const int SE = tile3_stage_elems(K, D, G, RY, HL + HR);
    Tile3 t;
    t.n0 = n0; t.ty = ty; t.tx = tx; t.r0 = n0 + ty * K; t.use_d = (phase == PH_LS);
    t.l = active ? dv.pp.lmap[tx] : -1;      // the position in Lidx (which may come in any order), not a rank
    t.stp = st.stp; t.c = 2.0 * st.rf_scale * dm.cfe;
    t.xs = smem; t.ss = smem + SE;
    t.xg = xg; t.dg = dv.d + (size_t)b * dm.ld;
    t.gtg = dv.gt + (size_t)b * dm.ld;
    {   // parameters (same select-chain as tile2_params)
#pragma unroll
        for (int k = 0; k < RHS::NP; ++k) t.p[k] = dv.pp.Pfull[(size_t)b * dm.NP + k];
        for (int k = 0; k < dm.NPest; ++k) {
            double v = t.xg[dm.ND + k];
            if (t.use_d) v = trial(v, t.stp, t.dg[dm.ND + k]);
            const int dst = dv.pp.Pidx[k];
#pragma unroll
            for (int j = 0; j < RHS::NP; ++j) t.p[j] = (dst == j) ? v : t.p[j];
        }
    }
    T3Regs<K> rg;
    ThreadAcc acc;
    acc.clear();
    // observations / own d entries: early (under the staging latency) when registers allow it;
    // the 512/1024-thread variants need their groups at <= 128 VGPRs and would spill them across phase B
    constexpr bool EARLY_OBS = NTMAX <= 256;
    if (EARLY_OBS && active) tile3_obs<K>(dm, dv.pp, t, rg);
    // phase A step 2: (+ d for a line-search point) -> LDS incl. ghost columns
    if (evenD) {
        if (edge) {
            if (t.use_d) tile3_stage_store<RHS, DISC, K, DC, true, true, NS>(dm, t, tid, nt, xr);
            else tile3_stage_store<RHS, DISC, K, DC, true, false, NS>(dm, t, tid, nt, xr);
        } else {
            if (t.use_d) tile3_stage_store<RHS, DISC, K, DC, false, true, NS>(dm, t, tid, nt, xr);
            else tile3_stage_store<RHS, DISC, K, DC, false, false, NS>(dm, t, tid, nt, xr);
        }
    } else {
        if (t.use_d) tile3_stage_odd<RHS, DISC, K, DC, true, true>(dm, t, tid, nt);
        else tile3_stage_odd<RHS, DISC, K, DC, true, false>(dm, t, tid, nt);
    }
    __syncthreads();
    if (active) {
        if (edge) tile3_rows<RHS, DISC, K, true, DC>(dm, dv.pp, t, rg, acc);
        else tile3_rows<RHS, DISC, K, false, DC>(dm, dv.pp, t, rg, acc);
    }
    __syncthreads();
    if (!EARLY_OBS && active) tile3_obs<K>(dm, dv.pp, t, rg);
    if (active) {
        if (edge) tile3_grad<RHS, DISC, K, true, DC>(dm, t, rg, acc);
        else tile3_grad<RHS, DISC, K, false, DC>(dm, t, rg, acc);
    }

    // ONE row of partial sums per workgroup.  Rows of 16 lanes reduce through DPP moves (no LDS latency);
    // the four row totals of each value meet in a wave-private LDS strip and lane k finishes value k of
    // the wave; after a workgroup barrier wave 0 adds the waves' values in wave order and stores the row
    // (write-through: the tail reads it in this launch when folded).  A quarter (a sixteenth for
    // 1024-thread groups) of the rows every tail has to read back.
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nt >> 6;
    double *strips = smem + SE + tile3_s_elems(K, D, G, RY);
    constexpr int ST = t3_strip_stride(RHS::NP);
    double *strip = strips + wave * (5 * ST);             // [4 rows of 16 lanes][ST values], then the wave's ST totals
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const double r = (k == EP_GMAX) ? row16_reduce<true>(acc.v[k]) : row16_reduce<false>(acc.v[k]);
        if ((lane & 15) == 0) strip[(lane >> 4) * ST + k] = r;
    }
    wave_sync_lds();
    if (lane < KP) {
        const double r0 = strip[lane], r1 = strip[ST + lane], r2 = strip[2 * ST + lane], r3 = strip[3 * ST + lane];
        strip[4 * ST + lane] = (lane == EP_GMAX) ? fmax(fmax(r0, r1), fmax(r2, r3)) : ((r0 + r1) + r2) + r3;
    }
    __syncthreads();
    if (wave != 0) return;
    if (lane < KP) {
        double v = strips[4 * ST + lane];
        for (int w2 = 1; w2 < nw; ++w2) {
            const double o = strips[w2 * (5 * ST) + 4 * ST + lane];
            v = (lane == EP_GMAX) ? fmax(v, o) : v + o;
        }
        st_sc1(dv.evp + ((size_t)b * dm.ntiles + tile) * EP_N + lane, v);
    }
    if (dv.epi == EPI_NONE) return;
    if (arrive_last(dv.cnt_eval + (size_t)b * CNT_STRIDE, (unsigned)dm.ntiles, lane))
        eval_epilogue<true>(dv, b, lane, reinterpret_cast<SeedHot *>(strips), dv.epi);
}

inline size_t eval3_lds_bytes(const Dims &dm)
{
    const int HL = dm.disc == DISC_SH ? 2 : 1;
    const size_t elems = (size_t)tile3_stage_elems(dm.maxr, dm.D, dm.ghost, dm.RY, HL + 1) + tile3_s_elems(dm.maxr, dm.D, dm.ghost, dm.RY);
    const size_t strips = (size_t)(dm.NT / 64) * 5 * t3_strip_stride(dm.NP);      // (>= 512 B: the tail stages the seed's state there)
    return sizeof(double) * (elems + (strips < 64 ? 64 : strips));
}

// launch one instantiation (the caller has checked that dv.dm's geometry matches it)
template <class RHS, int DISC, int K, int DC, int NTMAX>
inline void launch_eval3_one(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL((k_eval3<RHS, DISC, K, DC, NTMAX>), dim3(eval_flat_grid(dv.dm)), dim3(dv.dm.NT), eval3_lds_bytes(dv.dm), s, dv);
}
template <class RHS, int DISC, int K, int DC, int NTMAX>
inline hipError_t prepare_eval3_one(const Dev &dv)
{
    if (eval3_lds_bytes(dv.dm) <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute((const void *)k_eval3<RHS, DISC, K, DC, NTMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace va
