// va_kernels.hip -- gfx950 kernels of the variational-annealing hot path.
//
//   k_eval3      (A, me, fe, grad A) of every live seed at x (or x + stp*d), production kernel:
//                one workgroup = T = RY*K time rows of one seed (RY lanes per state column, K rows
//                per lane); rows (+halo, +ghost columns) staged in LDS; residuals, q, direct and s
//                in registers; gradient = direct + J^T s + measurement term; DPP row reductions,
//                one partial row per wave (deterministic: no atomics).
//   k_eval2      row-strided column mapping (D <= 256), k_eval (va_eval_flat.h): flat mapping for
//                any D and any right-hand side; generic fallbacks and independent cross-checks.
//   k_ls         one wave per seed: reduce partials, More'-Thuente line-search step,
//                L-BFGS-B stopping rules, beta-ladder bookkeeping (va_core.h: ls_step).
//   k_update     x += stp*d, history pair (s, y) into its slot, g <- g_t, and all the
//                inner products the direction needs in ONE sweep over S and Y.
//   k_coeffs     one wave per seed: Gram update + compact-form (Byrd-Nocedal-Schnabel) direction
//                coefficients by two lane-parallel triangular solves.
//   k_direction  d = cg*g + sum_j cY_j*Y_j + cS_j*S_j, and g.d / d.d partials.
//
// Reference arithmetic: see va_core.h header.  All fp64.  HBM-bound streaming kernels: nothing
// here is GEMM-shaped, so no MFMA (the network action in va_nnet.hip is, and uses it).
#include "va_device.h"
#include "va_eval_flat.h"

namespace va {

// ------------------------------------------------------------------ K1 (fast path): column-mapped
template <class RHS, int DISC, int MAXR>
__global__ __launch_bounds__(256) void k_eval2(const Dev dv)
{
    extern __shared__ double smem[];
    const Dims &dm = dv.dm;
    const int nwork = dm.B * dm.ntiles;
    const int w = xcd_swizzle(blockIdx.x, nwork);
    if (w >= nwork) return;
    const int b = w / dm.ntiles, tile = w - b * dm.ntiles;
    const SeedState &st = dv.st[b];
    const int phase = st.phase;
    if (phase != PH_START && phase != PH_LS) return;

    constexpr int HL = Halo<DISC>::HL, HR = Halo<DISC>::HR;
    constexpr int K = EP_GP + RHS::NP;
    constexpr bool FUSE = RHS::CHEAP_F && DISC != DISC_SH;
    const int D = dm.D, R = dm.T + HL + HR, RD = R * D;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int ty = tid / D, tx = tid - ty * D;
    const bool active = ty < dm.RY;

    Tile2 t;
    t.n0 = tile * dm.T; t.R = R; t.RY = dm.RY; t.ty = ty; t.use_d = (phase == PH_LS);
    t.col = make_cols(tx, D);
    t.l = active ? dv.pp.lmap[tx] : -1;
    t.stp = st.stp; t.c = 2.0 * st.rf_scale * dm.cfe;
    t.xs = smem; t.qs = smem + RD; t.fs = smem + 2 * RD;
    double *red = smem + (FUSE ? 2 : 3) * RD;
    t.xg = dv.x + (size_t)b * dm.ld; t.dg = dv.d + (size_t)b * dm.ld;
    t.gtg = dv.gt + (size_t)b * dm.ld;
    tile2_params<RHS>(dm, dv.pp, b, t);

    TRegs<MAXR> rg;
    ThreadAcc acc;
    acc.clear();
    if (active) tile2_load<DISC, MAXR>(dm, dv.pp, t, rg);
    __syncthreads();
    if (!FUSE) {
        if (active) tile2_f<RHS, DISC, MAXR>(dm, t, rg);
        __syncthreads();
    }
    if (active) tile2_q<RHS, DISC, MAXR, FUSE>(dm, dv.pp, t, rg, acc);
    __syncthreads();
    if (active) tile2_g<RHS, DISC, MAXR>(dm, t, rg, acc);

    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double v = (k == EP_GMAX) ? wave_max(acc.v[k]) : wave_sum(acc.v[k]);
        if (lane == 0) red[wave * K + k] = v;
    }
    __syncthreads();
    if (tid < K) {
        double v = red[tid];
        for (int ww = 1; ww < nw; ++ww)
            v = (tid == EP_GMAX) ? fmax(v, red[ww * K + tid]) : v + red[ww * K + tid];
        dv.evp[((size_t)b * dm.ntiles + tile) * EP_N + tid] = v;
    }
}

// ------------------------------------------------------------------ K1 (production): column-run
template <class RHS, int DISC, int K, int DC, int NTMAX>
__global__ __launch_bounds__(NTMAX) void k_eval3(const Dev dv)
{
    extern __shared__ double smem[];
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

    // profiling builds only (dm.dbg & 16): per-workgroup timeline -> the update-partials table,
    // read back with va_debug_read_partials (tools/timeline.py)
    unsigned long long *tl = nullptr;
    if (dm.dbg & 16) {
        tl = reinterpret_cast<unsigned long long *>(dv.upp) + (size_t)blockIdx.x * 8;
        if (threadIdx.x == 0) {
            tl[0] = wall_clock64();
            tl[5] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_ID
            tl[6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));    // XCC_ID
        }
    }
    // phase A step 1: x loads in flight before anything else is waited for
    double xr[NS][2];
    if (evenD) {
        if (edge) tile3_stage_load<DISC, K, DC, true, NS>(dm, n0, xg, tid, nt, xr);
        else tile3_stage_load<DISC, K, DC, false, NS>(dm, n0, xg, tid, nt, xr);
    }

    const SeedState &st = dv.st[b];
    const int phase = st.phase;
    if (phase != PH_START && phase != PH_LS) return;
    if (dm.dbg & 8) return;              // ablation: launch + dispatch only

    const int ty = tid / D, tx = tid - ty * D;
    const bool active = ty < RY;
    const int SE = tile3_stage_elems(K, D, G, RY, HL + HR);
    Tile3 t;
    t.n0 = n0; t.ty = ty; t.tx = tx; t.r0 = n0 + ty * K; t.use_d = (phase == PH_LS);
    t.l = !active ? -1 : (D <= 64 ? obs_index(dm.obsmask, tx) : dv.pp.lmap[tx]);
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
    if (tl && threadIdx.x == 0) tl[1] = wall_clock64();
    if (dm.dbg & 2) {                    // ablation: copy kernel (stage -> store), no arithmetic
        if (active && !edge)
            for (int k = 0; k < K; ++k)
                t.gtg[(long)(t.r0 + k) * D + tx] = t.xs[tile3_addr(ty * K + HL + k, tx, K, D, G, HL)];
        return;
    }
    if (active) {
        if (edge) tile3_rows<RHS, DISC, K, true, DC>(dm, dv.pp, t, rg, acc);
        else tile3_rows<RHS, DISC, K, false, DC>(dm, dv.pp, t, rg, acc);
    }
    __syncthreads();
    if (tl && threadIdx.x == 0) tl[2] = wall_clock64();
    if (!EARLY_OBS && active) tile3_obs<K>(dm, dv.pp, t, rg);
    if (active) {
        if (edge) tile3_grad<RHS, DISC, K, true, DC>(dm, t, rg, acc);
        else tile3_grad<RHS, DISC, K, false, DC>(dm, t, rg, acc);
    }

    // every wave writes its own partial row (no workgroup barrier: a __syncthreads here would
    // also wait for the gradient stores to land); k_ls sums the rows in a fixed order.
    // Rows of 16 lanes reduce through DPP moves (no LDS latency); the four row totals of each
    // value meet in a wave-private LDS strip, and lane k finishes value k.
    (void)KP;
    const int lane = tid & 63, wave = tid >> 6;
    double *prow = dv.evp + (((size_t)b * dm.ntiles + tile) * (nt >> 6) + wave) * EP_N;
    double *strip = smem + SE + tile3_s_elems(K, D, G, RY) + wave * (4 * 8);      // [4 rows][8 values]
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const double r = (k == EP_GMAX) ? row16_reduce<true>(acc.v[k]) : row16_reduce<false>(acc.v[k]);
        if ((lane & 15) == 0) strip[(lane >> 4) * 8 + k] = r;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < KP) {
        const double r0 = strip[lane], r1 = strip[8 + lane], r2 = strip[16 + lane], r3 = strip[24 + lane];
        prow[lane] = (lane == EP_GMAX) ? fmax(fmax(r0, r1), fmax(r2, r3)) : ((r0 + r1) + r2) + r3;
    }
    if (tl && threadIdx.x == 0) {
        tl[3] = wall_clock64();                       // stores issued
        __builtin_amdgcn_s_waitcnt(0);
        tl[4] = wall_clock64();                       // this wave's stores acknowledged
    }
}

size_t eval_lds_bytes(const Dims &dm)
{
    const int HL = dm.disc == DISC_SH ? 2 : 1;
    const int R = dm.T + HL + 1;
    size_t elems;
    if (dm.emode == 3)
        elems = (size_t)tile3_stage_elems(dm.maxr, dm.D, 2, dm.RY, HL + 1) + tile3_s_elems(dm.maxr, dm.D, 2, dm.RY);
    else elems = (size_t)((dm.emode == 2 && dm.disc != DISC_SH) ? 2 : 3) * R * dm.D;
    return sizeof(double) * (elems + (size_t)(dm.emode == 3 ? (dm.NT / 64) * 32 : (256 / 64) * EP_N));
}

template <class RHS, int DISC, int K, int DC, int NTMAX>
static void launch_eval3_one(const Dev &dv, hipStream_t s)
{
    const dim3 grid(eval_grid(dv.dm)), block(dv.dm.NT);
    const size_t lds = eval_lds_bytes(dv.dm);
    static bool big_lds_ok = false;                       // per instantiation
    if (lds > 64 * 1024 && !big_lds_ok) {
        (void)hipFuncSetAttribute((const void *)k_eval3<RHS, DISC, K, DC, NTMAX>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        big_lds_ok = true;
    }
    hipLaunchKernelGGL((k_eval3<RHS, DISC, K, DC, NTMAX>), grid, block, lds, s, dv);
}

template <class RHS, int K, int DC, int NTMAX>
static void launch_eval3_rhs(const Dev &dv, hipStream_t s)
{
    switch (dv.dm.disc) {
    case DISC_EULER: launch_eval3_one<RHS, DISC_EULER, K, DC, NTMAX>(dv, s); break;
    case DISC_TRAPEZOID: launch_eval3_one<RHS, DISC_TRAPEZOID, K, DC, NTMAX>(dv, s); break;
    case DISC_SH: launch_eval3_one<RHS, DISC_SH, K, DC, NTMAX>(dv, s); break;
    default: launch_eval3_one<RHS, DISC_FWDMAP, K, DC, NTMAX>(dv, s); break;
    }
}

// D fixed at compile time for the state sizes of the Lorenz-96 configurations the reference
// and BASELINE.json name (D = 20: examples/Lorenz96_D20; D = 200: BASELINE config 4, 256-thread groups); any
// other D runs the same kernel with D in a register.
template <class RHS, int K>
static void launch_eval3_d(const Dev &dv, hipStream_t s)
{
    if (dv.dm.D == 20) launch_eval3_rhs<RHS, K, 20, 256>(dv, s);
    else if (dv.dm.D == 200) launch_eval3_rhs<RHS, K, 200, 256>(dv, s);
    else if (dv.dm.D <= 64) launch_eval3_rhs<RHS, K, 0, 256>(dv, s);
    else if (dv.dm.D <= 128) launch_eval3_rhs<RHS, K, 0, 512>(dv, s);
    else if (dv.dm.D <= 256) launch_eval3_rhs<RHS, K, 0, 256>(dv, s);
    else if (dv.dm.D <= 512) launch_eval3_rhs<RHS, K, 0, 512>(dv, s);
    else launch_eval3_rhs<RHS, K, 0, 1024>(dv, s);
}

int eval_grid(const Dims &dm) { return ((dm.B * dm.ntiles + 7) / 8) * 8; }

template <class RHS, int MAXR>
static void launch_eval2_rhs(const Dev &dv, hipStream_t s)
{
    const dim3 grid(eval_grid(dv.dm)), block(dv.dm.NT);
    const size_t lds = eval_lds_bytes(dv.dm);
    switch (dv.dm.disc) {
    case DISC_EULER: hipLaunchKernelGGL((k_eval2<RHS, DISC_EULER, MAXR>), grid, block, lds, s, dv); break;
    case DISC_TRAPEZOID: hipLaunchKernelGGL((k_eval2<RHS, DISC_TRAPEZOID, MAXR>), grid, block, lds, s, dv); break;
    case DISC_SH: hipLaunchKernelGGL((k_eval2<RHS, DISC_SH, MAXR>), grid, block, lds, s, dv); break;
    default: hipLaunchKernelGGL((k_eval2<RHS, DISC_FWDMAP, MAXR>), grid, block, lds, s, dv); break;
    }
}

void launch_eval(const Dev &dv, int rhs, hipStream_t s)
{
    (void)rhs;                 // VA_RHS_LORENZ96 is the only built-in RHS so far
    if (dv.dm.emode == 3) {
        if (dv.dm.maxr == 4) launch_eval3_d<RhsL96g, 4>(dv, s);
        else if (dv.dm.maxr == 5) launch_eval3_d<RhsL96g, 5>(dv, s);
        else if (dv.dm.maxr == 6) launch_eval3_d<RhsL96g, 6>(dv, s);
        else if (dv.dm.maxr == 7) launch_eval3_d<RhsL96g, 7>(dv, s);
        else launch_eval3_d<RhsL96g, 8>(dv, s);
    } else if (dv.dm.emode == 2) {
        if (dv.dm.maxr <= 8) launch_eval2_rhs<RhsL96c, 8>(dv, s);
        else launch_eval2_rhs<RhsL96c, 16>(dv, s);
    } else {
        launch_eval_rhs<RhsL96>(dv, s);
    }
}

// Sum (or max) rows r0, r0+rstep, ... of one column of a partial table.  Loads are issued
// eight at a time before any is consumed (a plain `v += p[t]` loop serialises on memory
// latency: 44 rows x ~0.4 us was most of k_ls).  Fixed order -> deterministic.
__device__ __forceinline__ double col_reduce(const double *p, int nrows, int stride, int r0, int rstep, bool is_max)
{
    double v = 0.0;
    for (int t0 = r0; t0 < nrows; t0 += 8 * rstep) {
        double tmp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + u * rstep;
            tmp[u] = t < nrows ? p[(size_t)t * stride] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) v = is_max ? fmax(v, tmp[u]) : v + tmp[u];
    }
    return v;
}

// reduce the eval partial rows of seed b with the whole wave: lane = (row group r, column k),
// 2 row groups x 32 columns (EP_N <= 32); returns the column totals broadcast into ev[].
static_assert(EP_N <= 32, "reduce_eval assumes at most 32 partial columns");
__device__ __forceinline__ void reduce_eval(const Dev &dv, int b, int lane, double *ev)
{
    const int k = lane & 31, r = lane >> 5;
    double v = 0.0;
    if (k < EP_N) v = col_reduce(dv.evp + (size_t)b * dv.dm.nprow * EP_N + k, dv.dm.nprow, EP_N, r, 2, k == EP_GMAX);
    // combine the 2 row groups (lanes k, k+32) in a fixed order
    const double v0 = __shfl(v, k, 64), v1 = __shfl(v, k + 32, 64);
    const double tot = (k == EP_GMAX) ? fmax(v0, v1) : (v0 + v1);
#pragma unroll
    for (int c = 0; c < EP_N; ++c) ev[c] = __shfl(tot, c, 64);
}

// parameter tail of grad A (sum over tiles of the per-tile parameter partials) and its
// share of the line-search sums.
__device__ __forceinline__ void eval_tail(const Dev &dv, int b, int use_d, double *ev)
{
    const Dims &dm = dv.dm;
    double *gt = dv.gt + (size_t)b * dm.ld;
    const double *d = dv.d + (size_t)b * dm.ld;
    for (int k = 0; k < dm.NPest; ++k) {
        const double g = ev[EP_GP + dv.pp.Pidx[k]];
        gt[dm.ND + k] = g;
        if (use_d) ev[EP_GTD] += g * d[dm.ND + k];
        ev[EP_GN2] += g * g;
        ev[EP_GMAX] = fmax(ev[EP_GMAX], fabs(g));
    }
}

// ------------------------------------------------------------------ K2: line search / ladder
// One wave per seed.  The seed's hot state (<= 512 B) is staged in LDS with one coalesced
// 8-byte load per lane, lane 0 runs the (branchy, scalar) state machine on the LDS copy,
// and the wave writes it back: no chain of dependent global loads on the critical path.
__global__ __launch_bounds__(64) void k_ls(const Dev dv)
{
    __shared__ SeedHot sh;
    constexpr int NW8 = sizeof(SeedHot) / 8;
    const int b = blockIdx.x, lane = threadIdx.x;
    double *gst = reinterpret_cast<double *>(static_cast<SeedHot *>(&dv.st[b]));
    double *lst = reinterpret_cast<double *>(&sh);
    if (lane < NW8) lst[lane] = gst[lane];
    __syncthreads();
    const int phase = sh.phase;
    if (phase != PH_START && phase != PH_LS) {
        if (lane == 0 && (sh.upd || sh.dir)) { dv.st[b].upd = 0; dv.st[b].dir = 0; }
        return;
    }
    const Dims &dm = dv.dm;
    double ev[EP_N], dirp[DP_N];
    reduce_eval(dv, b, lane, ev);
    {   // direction partials (g.d, d.d) left by k_direction: lane = (row group, column), 32 x 2
        const int k = lane & 1, r = lane >> 1;
        double v = col_reduce(dv.dpp + (size_t)b * dm.nchunks * DP_N + k, dm.nchunks, DP_N, r, 32, false);
#pragma unroll
        for (int o = 32; o >= 2; o >>= 1) v += __shfl_down(v, o, 64);     // lanes 0 and 1 hold the totals
#pragma unroll
        for (int c = 0; c < DP_N; ++c) dirp[c] = __shfl(v, c, 64);
    }
    if (lane == 0) {
        atomicAdd(dv.n_evals, 1ULL);
        eval_tail(dv, b, phase == PH_LS, ev);
        SeedResults r;
        r.ame = dv.ame + (size_t)b * dv.max_beta * 3;
        r.pest = nullptr;
        r.status = dv.status + (size_t)b * dv.max_beta;
        r.nit = dv.nit + (size_t)b * dv.max_beta;
        r.nfev = dv.nfev + (size_t)b * dv.max_beta;
        int dec = 0;
        ls_step(sh, ev, dirp, dv.o, dv.rf_ladder, dv.nbeta, r, &dec, dm.cme, dm.cfe);
        if (dec) atomicSub(dv.n_active, 1);
    }
    __syncthreads();
    if (lane < NW8) gst[lane] = lst[lane];
}
void launch_ls(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_ls, dim3(dv.dm.B), dim3(64), 0, s, dv);
}

// S1 epilogue: A, me, fe and the parameter tail for a plain evaluation.
__global__ __launch_bounds__(64) void k_finalize_eval(const Dev dv)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    double ev[EP_N];
    reduce_eval(dv, b, lane, ev);
    if (lane != 0) return;
    eval_tail(dv, b, 0, ev);
    const double me = ev[EP_ME] * dv.dm.cme, fe = ev[EP_FE] * dv.dm.cfe * dv.st[b].rf_scale;
    dv.outA[b] = me + fe; dv.outme[b] = me; dv.outfe[b] = fe;
}
void launch_finalize_eval(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_finalize_eval, dim3(dv.dm.B), dim3(64), 0, s, dv);
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
    s.stp_upd = 0.0; s.gd = 0.0; s.gdold = 0.0; s.gn2 = 0.0; s.dr = 0.0; s.cg = -1.0;
    s.rf_scale = rf < 0.0 ? dv.rf_ladder[0] : rf;
}
void launch_init_states(const Dev &dv, int phase, double rf, hipStream_t s)
{
    hipLaunchKernelGGL(k_init_states, dim3((dv.dm.B + 63) / 64), dim3(64), 0, s, dv, phase, rf);
}

// ------------------------------------------------------------------ K3: update + inner products
__global__ __launch_bounds__(VEC_THREADS) void k_update(const Dev dv)
{
    __shared__ double red[(VEC_THREADS / 64) * UP_N];
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
            xv.x = trial(xv.x, stp, dv2.x); xv.y = trial(xv.y, stp, dv2.y);
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
    if (!dir) return;
    const int lane = tid & 63, wave = tid >> 6;
    const int K = UP_OLD + 4 * nold;
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
    __syncthreads();
    for (int k = tid; k < K; k += VEC_THREADS) {
        double v = red[k];
#pragma unroll
        for (int ww = 1; ww < VEC_THREADS / 64; ++ww) v += red[ww * UP_N + k];
        dv.upp[((size_t)b * dm.nchunks + chunk) * dv.ups + k] = v;
    }
}
void launch_update(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_update, dim3(dv.dm.nchunks, dv.dm.B), dim3(VEC_THREADS), 0, s, dv);
}

// ------------------------------------------------------------------ K4: direction coefficients
// wave-uniform broadcast of lane i's double through SGPRs (v_readlane), i uniform
__device__ __forceinline__ double bcast_lane(double x, int i)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), i);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), i);
    return __hiloint2double(hi, lo);
}

// One wave per seed, lane j <-> j-th oldest history pair: Gram update, then the compact
// form of the two-loop recursion (va_core.h: direction_coeffs_view, which the CPU emulator
// runs serially):  p = R^-1 a,  q = (D + gamma Y'Y) p - gamma b,  u = R^-T q,
// d = -gamma g - S u + gamma Y p.  Both triangular solves are column-oriented, so a step is
// one SGPR broadcast + one fused update per lane -- no cross-lane reductions and no serial
// O(m^2) loop (the serial form took 17-40 us per cycle; this one is latency of 2m short steps).
__global__ __launch_bounds__(64) void k_coeffs(const Dev dv)
{
    __shared__ double up[UP_N];
    __shared__ double sSY[MAX_M * MAX_M], sYY[MAX_M * MAX_M];
    __shared__ double sp[MAX_M], cYs[MAX_M], cSs[MAX_M];
    const int b = blockIdx.x, lane = threadIdx.x;
    SeedState &s = dv.st[b];
    if (!s.dir) return;
    const Dims &dm = dv.dm;
    const int M = MAX_M, m = dm.m;
    const int nold = s.nold, col = s.col, sn = s.slot;
    const bool hist = (s.upd & UPD_HIST) != 0;
    const double dr = s.dr;
    double theta = s.theta;
    const int K = UP_OLD + 4 * nold;
    for (int k = lane; k < K; k += 64)
        up[k] = col_reduce(dv.upp + (size_t)b * dm.nchunks * dv.ups + k, dm.nchunks, dv.ups, 0, 1, false);
    const int myslot = lane < col ? s.order[lane] : 0;
    {   // the m x m blocks of the Gram matrices: independent loads first, LDS stores after
        double t0[4], t1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = lane + 64 * u;
            const int i = e / m, j = e - i * m;
            t0[u] = e < m * m ? s.SY[i * M + j] : 0.0;
            t1[u] = e < m * m ? s.YY[i * M + j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = lane + 64 * u;
            const int i = e / m, j = e - i * m;
            if (e < m * m) { sSY[i * M + j] = t0[u]; sYY[i * M + j] = t1[u]; }
        }
        for (int e = lane + 256; e < m * m; e += 64) {    // m > 16 only
            const int i = e / m, j = e - i * m;
            sSY[i * M + j] = s.SY[i * M + j]; sYY[i * M + j] = s.YY[i * M + j];
        }
    }
    if (lane < M) { cYs[lane] = 0.0; cSs[lane] = 0.0; }
    __syncthreads();
    if (hist) {
        // new column/row of the Gram matrices (LDS copy and the persistent one)
        if (lane < nold) {
            const double sjy = up[UP_OLD + 4 * lane + 2], yjy = up[UP_OLD + 4 * lane + 3];
            sSY[myslot * M + sn] = sjy; sYY[myslot * M + sn] = yjy; sYY[sn * M + myslot] = yjy;
            s.SY[myslot * M + sn] = sjy; s.YY[myslot * M + sn] = yjy; s.YY[sn * M + myslot] = yjy;
        }
        if (lane == 0) {
            sSY[sn * M + sn] = dr; sYY[sn * M + sn] = up[UP_YY];   // s.y as the line search saw it
            s.SY[sn * M + sn] = dr; s.YY[sn * M + sn] = up[UP_YY];
        }
        theta = up[UP_YY] / dr;
    }
    __syncthreads();
    double aj = 0.0, bj = 0.0;
    if (lane < nold) { aj = up[UP_OLD + 4 * lane + 0]; bj = up[UP_OLD + 4 * lane + 1]; }
    if (hist && lane == col - 1) { aj = up[UP_SGT]; bj = up[UP_YGT]; }
    const double gamma = 1.0 / theta;
    const double rjj = lane < col ? sSY[myslot * M + myslot] : 1.0;
    const double rinv = 1.0 / rjj;
    // p = R^-1 a  (R_ji = S_j . Y_i for j <= i)
    double pj = 0.0;
    for (int i = col - 1; i >= 0; --i) {
        const int si = __builtin_amdgcn_readlane(myslot, i);
        const double pi = bcast_lane(aj * rinv, i);
        if (lane == i) pj = pi;
        if (lane < i) aj -= sSY[myslot * M + si] * pi;
    }
    if (lane < M) sp[lane] = pj;
    __syncthreads();
    // q = (D + gamma Y'Y) p - gamma b
    double qj = 0.0;
    if (lane < col) {
        double acc = 0.0;
        for (int k = 0; k < col; ++k) acc += sYY[myslot * M + __builtin_amdgcn_readlane(myslot, k)] * sp[k];
        qj = rjj * pj + gamma * acc - gamma * bj;
    }
    // u = R^-T q  ((R^T)_ji = R_ij = S_i . Y_j for i <= j)
    double uj = 0.0;
    for (int i = 0; i < col; ++i) {
        const int si = __builtin_amdgcn_readlane(myslot, i);
        const double ui = bcast_lane(qj * rinv, i);
        if (lane == i) uj = ui;
        if (lane > i && lane < col) qj -= sSY[si * M + myslot] * ui;
    }
    if (lane < col) { cYs[myslot] = gamma * pj; cSs[myslot] = -uj; }
    __syncthreads();
    if (lane < M) { s.cY[lane] = cYs[lane]; s.cS[lane] = cSs[lane]; }
    if (lane == 0) { s.cg = -gamma; s.theta = theta; }
}
void launch_coeffs(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_coeffs, dim3(dv.dm.B), dim3(64), 0, s, dv);
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
    double gd = 0.0, dd = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (idx[e] >= dm.ld) continue;
        *reinterpret_cast<double2 *>(d + idx[e]) = acc[e];
        gd += gv[e].x * acc[e].x + gv[e].y * acc[e].y;
        dd += acc[e].x * acc[e].x + acc[e].y * acc[e].y;
    }
    const int lane = tid & 63, wave = tid >> 6;
    gd = wave_sum(gd); dd = wave_sum(dd);
    if (lane == 0) { red[wave * DP_N + DP_GD] = gd; red[wave * DP_N + DP_DD] = dd; }
    __syncthreads();
    if (tid < DP_N) {
        double v = red[tid];
#pragma unroll
        for (int ww = 1; ww < VEC_THREADS / 64; ++ww) v += red[ww * DP_N + tid];
        dv.dpp[((size_t)b * dm.nchunks + chunk) * DP_N + tid] = v;
    }
}
void launch_direction(const Dev &dv, hipStream_t s)
{
    hipLaunchKernelGGL(k_direction, dim3(dv.dm.nchunks, dv.dm.B), dim3(VEC_THREADS), 0, s, dv);
}

}  // namespace va
