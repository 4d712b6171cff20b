// va_eval4.h -- the wave-private column-run evaluation kernel k_eval4<RHS, DISC, K, D, W_SCALAR, SUB> and
// its launcher, as a header: compiled into libvaranneal_amd.so for the built-in Lorenz-96 and into
// every generated right-hand-side module whose model has a column form (varanneal_amd/codegen.py:
// translation-invariant stencils, or small dense systems such as the tutorial's NaKL neuron).
// Tile arithmetic: va_tile4.h; tail of an evaluation: va_epilogue.h.
#pragma once
#include "va_eval_flat.h"
#include "va_measure.h"

namespace va {

// ------------------------------------------------------------------ K1 (narrow states): wave-private runs
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// one wave copies its image: LDS piece q <- source piece tile4_src_piece(q); 1 KiB per instruction
template <int NI>
__device__ __forceinline__ void tile4_dma(const Geo4 &g, const double *src, double *dst, int lane)
{
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int sp = tile4_src_piece(g, i * 64 + lane);
        if (sp >= 0)
            __builtin_amdgcn_global_load_lds((glb_void_t *)(src + 2 * sp), (lds_void_t *)(dst + 128 * i), 16, 0, 0);
    }
}

// SUB = 1: three workgroups per CU = three waves per SIMD need <= 168 registers per lane.
// SUB > 1 (grids of a few wave-tiles per SIMD, e.g. BASELINE config 3: 3072 = 3 x 1024): ONE wave per
// SIMD works through SUB sub-tiles one after the other instead of SUB waves working side by side --
// all images are requested up front, the first sub-tile's gradient leaves while the next is computed
// (co-resident waves run in lockstep: load, then compute, then store, all of them together), and
// the wave reduces its partial sums once.
template <class RHS, int DISC, int K, int DC, int WS, int SUB>
__global__ __launch_bounds__(256, SUB > 1 ? 1 : ((K <= 7 && K * RHS::NB <= 28) ? 3 : 2)) void k_eval4(const Dev dv)      // (the neighbour values of the lane's K rows live in registers: K * NB of them)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr bool W_SCALAR = WS != 0;                        // WS: 0 weight arrays, 1 scalar weights, 2 scalar weights + merr_nskip
    const Dims &dm = dv.dm;
    const int nwork = dm.B * dm.ntiles;
    const int w = xcd_swizzle(blockIdx.x, nwork);
    if (w >= nwork) return;
    const int b = dm.ntiles > 1 ? (int)__umulhi((unsigned)w, dv.ntiles_magic) : w, tile = w - b * dm.ntiles;

    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR, NE = RHS::NE;
    constexpr int NV = EP_GP + RHS::NP;
    VA_E4_STAMP_SETUP(dv, w);      // (diagnostic builds only: va_measure.h)
    VA_E4_STAMP(0);
    // with D fixed at compile time the whole geometry (and every LDS offset) is constant
    const Geo4 g = DC > 0 ? tile4_geo<HL + HR>(DC > 0 ? DC : 4, K, NE, SUB) : dv.g4;
    constexpr int NI = DC > 0 ? (tile4_geo<HL + HR>(DC > 0 ? DC : 4, K, NE, SUB).XP + 63) / 64 : T4_NI_MAX;
    const int D = DC > 0 ? DC : g.D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RK = g.RW * K;                                  // rows of one sub-tile
    const int n0w = tile * g.T + wave * SUB * RK;
    double *xsw = smem + wave * g.WAVE, *r2w = xsw + SUB * g.XW, *strip = r2w + g.R2;
    const double *xg = dv.x + (size_t)b * dm.ld;

    // stage x: every image in flight before anything else is waited for
#pragma unroll
    for (int s = 0; s < SUB; ++s) tile4_dma<NI>(g, xg + (long)(n0w + s * RK - HL) * D, xsw + s * g.XW, lane);

    const auto *st = as_const(static_cast<const SeedHot *>(dv.st + b));
    const int phase = st->phase;
    if (phase != PH_START && phase != PH_LS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // never end a wave under its own LDS loads
        return;
    }
    const int use_d = (phase == PH_LS);
    const double stp = st->stp;
    const double *dg = dv.d + (size_t)b * dm.ld;
    if (use_d) {
#pragma unroll
        for (int s = 0; s < SUB; ++s) tile4_dma<NI>(g, dg + (long)(n0w + s * RK - HL) * D, r2w + s * g.XW, lane);
    }

    const int a = lane / D, tx = lane - a * D;
    const bool active = a < g.RW;
    Tile4 t;
    t.n0w = n0w; t.a = a; t.tx = tx; t.r0 = n0w + a * K; t.use_d = use_d;
    t.l = obs_index(dm.obsmask, tx);          // (Lidx is ascending on the device: va_problem_create sorts it)
    t.wobs = t.l >= 0 ? dm.rm : 0.0;
    t.c = 2.0 * st->rf_scale * dm.cfe;
    t.xs = xsw; t.es = r2w;
    t.gtg = dv.gt + (size_t)b * dm.ld;
    {   // parameters (estimated ones from the trial point), all uniform
#pragma unroll
        for (int k = 0; k < RHS::NP; ++k) t.p[k] = as_const(dv.pp.Pfull)[(size_t)b * dm.NP + k];
        for (int k = 0; k < dm.NPest; ++k) {
            double v = as_const(xg)[dm.ND + k];
            if (use_d) v = trial(v, stp, as_const(dg)[dm.ND + k]);
            const int dst = as_const(dv.pp.Pidx)[k];
#pragma unroll
            for (int j = 0; j < RHS::NP; ++j) t.p[j] = (dst == j) ? v : t.p[j];
        }
    }
    T4Regs<K, NE> rg[SUB];
    ThreadAcc acc;
    acc.clear();
    if constexpr (WS == 2) {
        // data at every nskip-th model row: the lane walks its rows once, counting data rows; rows without data read as 0
        // (outside the buffer) and are masked out of the measurement terms by rg.has
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            (void *)dv.pp.Y, 0, (int)(sizeof(double) * dm.N_data * dm.L), 0x00020000);
        const bool obs = active && t.l >= 0;
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            const int m0 = t.r0 + s * RK;
            int nd = m0 / dm.nskip, rem = m0 - nd * dm.nskip;
            unsigned has = 0u;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const bool h = obs && rem == 0 && nd < dm.N_data && m0 + k < dm.N;
                const v2u v = __builtin_amdgcn_raw_buffer_load_b64(yr, h ? (nd * dm.L + t.l) * 8 : 0x7ffffff0, 0, 0);
                rg[s].yv[k] = __hiloint2double((int)v.y, (int)v.x);
                has |= h ? (1u << k) : 0u;
                if (++rem == dm.nskip) { rem = 0; ++nd; }
            }
            rg[s].has = has;
        }
    } else if constexpr (W_SCALAR) {
        // observations of the lane's own rows: one address per lane, the row stride rides in an SGPR;
        // rows beyond the data and unobserved columns fall outside the buffer and read as 0
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            (void *)dv.pp.Y, 0, (int)(sizeof(double) * dm.N_data * dm.L), 0x00020000);
        const int voff = (active && t.l >= 0) ? (t.r0 * dm.L + t.l) * 8 : 0x7ffffff0;
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                const v2u v = __builtin_amdgcn_raw_buffer_load_b64(yr, voff, (s * RK + k) * dm.L * 8, 0);
                rg[s].yv[k] = __hiloint2double((int)v.y, (int)v.x);
            }
        }
    } else if (active) {
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            Tile4 ts = t;
            ts.r0 = t.r0 + s * RK;
            tile4_obs<K, NE>(dm, dv.pp, ts, rg[s]);
        }
    }
    // RF0 arrays: the lane's weights, requested with everything else (parked in the product arrays below: va_tile4.h)
    static_assert(SUB == 1 || WS != 0, "sub-tiles are compiled for the scalar-weight variants only");
    constexpr bool RFW = WS == 0 && tile4_rfw_in_lds<K, NE, HL, DC>();
    double wq[SUB][K + HL];
    if constexpr (RFW) {
        if (dv.pp.rf0_arr && active) {
#pragma unroll
            for (int s = 0; s < SUB; ++s) {
                Tile4 ts = t;
                ts.r0 = t.r0 + s * RK;
                tile4_rfw_load<K, HL>(dm, dv.pp, ts, D, wq[s]);
            }
        }
    }

    VA_E4_STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the wave's images have landed
    __builtin_amdgcn_wave_barrier();
    VA_E4_STAMP(2);
    // Workgroups that share a CU start together and their data arrives in dispatch order; the later
    // ones then compete for the vector pipe with workgroups already in their gather / reduction
    // phases and finish up to 1.7 us after the first.  The kernel is as long as its last workgroup:
    // later workgroups issue at higher priority.
    if (dv.prio == 1) {
        const int grp = (int)(blockIdx.x >> 8);
        if (grp == 1) __builtin_amdgcn_s_setprio(1);
        else if (grp >= 2) __builtin_amdgcn_s_setprio(2);
    } else if (dv.prio == 2) {
        // the seed's first and last tile run the edge variant of the rows phase (+0.5 us, profiles/r04_timeline_c3.txt) and
        // are the workgroups a seed's tail waits for: they issue ahead of the CU's other workgroups
        if (tile == 0 || tile == dm.ntiles - 1) __builtin_amdgcn_s_setprio(3);
    }
    if (use_d) {
        // trial point x + stp*d in place (the same fma as k_update's accepted iterate), then the
        // lane's own entries of d for the g.d partial; the d images are dead after that
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int q = i * 64 + lane;
                if (q < g.XP) {
                    double x0, x1, d0, d1;
                    ld2(xsw + s * g.XW + 2 * q, x0, x1); ld2(r2w + s * g.XW + 2 * q, d0, d1);
                    st2(xsw + s * g.XW + 2 * q, trial(x0, stp, d0), trial(x1, stp, d1));
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int j = k + HL;
                rg[s].dval[k] = active ? r2w[s * g.XW + a * g.PITCH + tx + (j < K ? j * D : g.PITCH + (j - K) * D)] : 0.0;
            }
        }
        wave_sync_lds();
    } else {
#pragma unroll
        for (int s = 0; s < SUB; ++s)
#pragma unroll
            for (int k = 0; k < K; ++k) rg[s].dval[k] = 0.0;
    }
    const bool lsq = dv.epi != EPI_FINALIZE;                                  // launch-uniform
    if constexpr (RFW) {
        // (the d image in the second region is dead now; every lane writes and later reads its own slots only)
        if (dv.pp.rf0_arr && active) tile4_rfw_store<K, HL>(g, t, D, wq[0]);
    }
    unsigned old = 0;
    double gvv[SUB][K];                       // the gradient stays in registers until the partial sums are out
    // The workgroup's row of partial sums: every wave reduces through the matrix pipe (max|g| through
    // DPP row moves) and lane 0 leaves the totals in the wave's LDS strip; after a workgroup barrier
    // wave 0 adds the four waves' values in a fixed order and stores ONE row (write-through).
    auto publish = [&]() {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if ((k == EP_GTD || k == EP_GN2 || k == EP_GMAX) && !lsq) continue;
            if (k == EP_GTD && !use_d) continue;
            const double r = (k == EP_GMAX) ? wave_max(acc.v[k]) : wave_sum_mfma(acc.v[k]);
            if (lane == 0) strip[k] = r;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        VA_E4_STAMP(5);
        __builtin_amdgcn_s_barrier();
        VA_E4_STAMP(6);
        if (wave == 0) {
            if (lane < NV) {
                const double r0 = strip[lane], r1 = strip[g.WAVE + lane], r2 = strip[2 * g.WAVE + lane], r3 = strip[3 * g.WAVE + lane];
                double v = (lane == EP_GMAX) ? fmax(fmax(r0, r1), fmax(r2, r3)) : ((r0 + r1) + r2) + r3;
                if ((lane == EP_GTD && !use_d) || ((lane == EP_GTD || lane == EP_GN2 || lane == EP_GMAX) && !lsq)) v = 0.0;
                st_sc1(dv.evp + ((size_t)b * dm.ntiles + tile) * EP_N + lane, v);
            }
            if (dv.epi != EPI_NONE) {
                // the wave has nothing else in flight: the row is acknowledged quickly, and the count
                // is on its way while the wave goes on (gather phase / gradient stores)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) old = __hip_atomic_fetch_add(dv.cnt_eval + (size_t)b * CNT_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
#pragma unroll
    for (int s = 0; s < SUB; ++s) {
        const int n0s = n0w + s * RK;
        const bool edge = (n0s - HL < 0) || (n0s + RK + HR > dm.N);          // wave-uniform
        t.r0 = n0s + a * K; t.xs = xsw + s * g.XW;
        if (active) {
            if (edge) tile4_rows<RHS, DISC, K, true, DC, WS>(dm, dv.pp, g, t, rg[s], acc);
            else tile4_rows<RHS, DISC, K, false, DC, WS>(dm, dv.pp, g, t, rg[s], acc);
        }
        wave_sync_lds();          // products are read by the same wave only: LDS is in order within a wave
        if (s == SUB - 1) VA_E4_STAMP(3);
        // A plain S1 evaluation needs A = me + fe and dA/dp from the sums: all known once the rows are
        // done.  The row goes out now, and the three dependent round trips of the tail (row
        // acknowledged, arrival counted, rows of the seed read back) run beside the gather phase and
        // the gradient stores instead of after them.
        if (!lsq && s == SUB - 1) publish();
#pragma unroll
        for (int k = 0; k < K; ++k) gvv[s][k] = 0.0;
        if (active) {
            if (edge) {
                if (lsq) tile4_grad<RHS, DISC, K, true, DC, WS, true>(dm, g, t, rg[s], acc, gvv[s]);
                else tile4_grad<RHS, DISC, K, true, DC, WS, false>(dm, g, t, rg[s], acc, gvv[s]);
            } else {
                if (lsq) tile4_grad<RHS, DISC, K, false, DC, WS, true>(dm, g, t, rg[s], acc, gvv[s]);
                else tile4_grad<RHS, DISC, K, false, DC, WS, false>(dm, g, t, rg[s], acc, gvv[s]);
            }
        }
        if (s + 1 < SUB) wave_sync_lds();     // the next sub-tile overwrites the product arrays
    }
    VA_E4_STAMP(4);
    // a line-search evaluation also needs g.d, g.g and max|g|: its row waits for the gradient
    if (lsq) publish();
    {
        // Gradient stores.  The wave's RW*K rows are ONE contiguous run of the path in memory: the values go
        // through the wave's (now dead) product arrays in row-major order and leave as consecutive 16-byte pieces,
        // every lane storing -- full 64-byte sectors instead of ten lanes x 16 B per row and instruction
        // (profiles/r03_ablation_c3.txt: the stores were 2.65 us of the 8.7).  Write-through (sc1) on small grids:
        // the lines leave L2 as they are written instead of being flushed, all 10 MB of them, at the end of the kernel.
        const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(dv.gt + (size_t)b * dm.ld), 0, (int)(sizeof(double) * dm.N * D), 0x00020000);
        const int npieces = RK * D / 2;                       // 16-byte pieces of one sub-tile's rows
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            wave_sync_lds();                                  // (the gather phase has read the products)
            if (active) {
#pragma unroll
                for (int k = 0; k < K; ++k) r2w[(a * K + k) * D + tx] = gvv[s][k];
            }
            wave_sync_lds();
            const int base = (n0w + s * RK) * D * 8;
#pragma unroll
            for (int i = 0; i < (K * 32 + 63) / 64; ++i) {    // (RW * D <= 64 columns, K rows: at most 32 K pieces)
                if (i * 64 >= npieces) break;
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const int q = i * 64 + lane;
                double g0, g1;
                ld2(r2w + 2 * (q < npieces ? q : 0), g0, g1);
                const v4u v = {(unsigned)__double2loint(g0), (unsigned)__double2hiint(g0), (unsigned)__double2loint(g1), (unsigned)__double2hiint(g1)};
                // rows >= N fall outside the buffer (N*D*8 bytes): the hardware drops those stores
                const int voff = q < npieces ? q * 16 : 0x7ffffff0;
                if (dv.gaux) __builtin_amdgcn_raw_buffer_store_b128(v, gr, voff, base, 16);
                else __builtin_amdgcn_raw_buffer_store_b128(v, gr, voff, base, 0);
            }
        }
    }
    // The last workgroup of the seed runs the tail (and resets the counter for the next launch).
    // (Measured at C3: running it before wave 0's own gradient stores, so that its loads do not retire
    // behind seven write-through stores, is slower -- 10.2 vs 9.6 us: those stores then end the kernel.)
    bool last = false;
    if (wave == 0 && dv.epi != EPI_NONE) {
        old = __builtin_amdgcn_readfirstlane(old);
        asm volatile("" ::: "memory");
        last = old == (unsigned)dm.ntiles - 1u;
        VA_E4_STAMP(7);
        if (last) {
            if (lane == 0) __hip_atomic_store(dv.cnt_eval + (size_t)b * CNT_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            eval_epilogue<true>(dv, b, lane, reinterpret_cast<SeedHot *>(xsw), dv.epi);
        }
    }
    VA_E4_STAMP_TAIL(last);        // (wave 0 re-uses slot 1)
}


inline size_t eval4_lds_bytes(const Dev &dv) { return sizeof(double) * (size_t)dv.g4.NW * dv.g4.WAVE; }

// launch one instantiation (the caller has checked that dv.g4 / dv.dm.maxr / dv.dm.disc match it)
template <class RHS, int DISC, int K, int DC, int WS>
inline void launch_eval4_one(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL((k_eval4<RHS, DISC, K, DC, WS, 1>), dim3(eval_flat_grid(dv.dm)), dim3(256), eval4_lds_bytes(dv), s, dv);
}
template <class RHS, int DISC, int K, int DC, int WS>
inline hipError_t prepare_eval4_one(const Dev &dv)
{
    if (eval4_lds_bytes(dv) <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute((const void *)k_eval4<RHS, DISC, K, DC, WS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace va
