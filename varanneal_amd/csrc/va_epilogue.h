// va_epilogue.h -- "last arriver finishes the seed": device-side tails that used to be kernels of
// their own (k_finalize_eval, k_ls, k_coeffs).
//
// Every wave (or workgroup) that produced a row of per-seed partial sums stores it write-through
// (sc1), drains its own stores (s_waitcnt vmcnt(0)), and adds 1 to the seed's agent-scope arrival
// counter.  The wave whose add returns `expected - 1` knows every other row is in L2 / memory: it
// reads the rows back with sc1 loads (never through its CU's L1), sums them in a FIXED order
// (bitwise reproducible, independent of which wave came last) and runs the tail:
//   eval kernels   -> A, me, fe, parameter gradient (S1: _autodiffmin.py:57-58) or one line-search /
//                     ladder step (S2/S3: va_core.h ls_step, _autodiffmin.py:72-95, va_ode.py:707-789)
//   k_update       -> Gram update + compact-form direction coefficients (the former k_coeffs)
//   k_direction    -> g.d of the new direction
// The counter is reset by its last arriver, so every launch finds it at zero.
// (Hand-off form: MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup
// visibility", sc1 stores + drained + atomic add; consumer = the adder that came last.)
#pragma once
#include "va_device.h"

namespace va {

__device__ __forceinline__ double ld_sc1(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One arrival of the calling wave (all of ITS partial-row stores issued before the call).
// Returns true, wave-uniformly, in the wave that arrived last.
__device__ __forceinline__ bool arrive_last(unsigned *cnt, unsigned expected, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own stores acknowledged
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __builtin_amdgcn_readfirstlane(old);
    asm volatile("" ::: "memory");
    if (old != expected - 1u) return false;
    if (lane == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// Sum (or max) rows r0, r0+rstep, ... of one column of a partial table.  Loads are issued
// eight at a time before any is consumed.  Fixed order -> deterministic.
template <bool SC1>
__device__ __forceinline__ double col_reduce(const double *p, int nrows, int stride, int r0, int rstep, bool is_max)
{
    double v = 0.0;
    for (int t0 = r0; t0 < nrows; t0 += 8 * rstep) {
        double tmp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = t0 + u * rstep;
            tmp[u] = 0.0;
            if (t < nrows) tmp[u] = SC1 ? ld_sc1(p + (size_t)t * stride) : p[(size_t)t * stride];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) v = is_max ? fmax(v, tmp[u]) : v + tmp[u];
    }
    return v;
}

// reduce the eval partial rows of seed b with the whole wave: lane = (row group r, column k) with
// NC = 8, 16 or 32 columns (the smallest that holds the EP_GP + NP columns in use) and 64 / NC row
// groups, so that a dozen rows come back in ONE round trip; the groups are then added in a fixed
// order.  Returns the column totals broadcast into ev[].
static_assert(EP_N <= 32, "reduce_eval assumes at most 32 partial columns");
template <bool SC1>
__device__ __forceinline__ void reduce_eval(const Dev &dv, int b, int lane, double *ev)
{
    const int nc = dv.evcols, ng = 64 / nc;
    const int k = lane & (nc - 1), r = lane / nc;
    double v = 0.0;
    if (k < EP_N)
        v = col_reduce<SC1>(dv.evp + (size_t)b * dv.dm.nprow * EP_N + k, dv.dm.nprow, EP_N, r, ng, k == EP_GMAX);
    double tot = __shfl(v, k, 64);
    for (int gi = 1; gi < ng; ++gi) {
        const double o = __shfl(v, k + gi * nc, 64);
        tot = (k == EP_GMAX) ? fmax(tot, o) : tot + o;
    }
#pragma unroll
    for (int c = 0; c < EP_N; ++c) ev[c] = c < nc ? __shfl(tot, c, 64) : 0.0;
}

// uniform, read-only-in-this-launch words through the scalar path (constant address space): such
// loads are not ordered with the vector-memory queue and cost a fraction of a vector round trip.
// (Only for memory written by EARLIER launches: the launch boundary invalidates the scalar cache.)
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T *as_const(const T *p)
{
    return (const __attribute__((address_space(4))) T *)p;
}

// The same for a whole workgroup of NW waves (tail kernels of large grids: thousands of rows per
// seed): wave w takes the rows w, w+NW, ... of every row group, leaves its column totals in LDS,
// and after the barrier every wave adds the NW wave totals in wave order.  `part`: NW*32 doubles.
__device__ __forceinline__ void reduce_eval_block(const Dev &dv, int b, int lane, int wave, int nw, double *part, double *ev)
{
    const int nc = dv.evcols, ng = 64 / nc;
    const int k = lane & (nc - 1), r = lane / nc;
    double v = 0.0;
    if (k < EP_N)
        v = col_reduce<false>(dv.evp + (size_t)b * dv.dm.nprow * EP_N + k, dv.dm.nprow, EP_N, r + wave * ng, ng * nw, k == EP_GMAX);
    double tot = __shfl(v, k, 64);
    for (int gi = 1; gi < ng; ++gi) {
        const double o = __shfl(v, k + gi * nc, 64);
        tot = (k == EP_GMAX) ? fmax(tot, o) : tot + o;
    }
    if (lane < nc) part[wave * 32 + lane] = tot;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < EP_N; ++c) {
        double t = 0.0;
        if (c < nc)
            for (int w = 0; w < nw; ++w) t = (c == EP_GMAX) ? fmax(t, part[w * 32 + c]) : t + part[w * 32 + c];
        ev[c] = t;
    }
}

// parameter tail of grad A (sum over tiles of the per-tile parameter partials) and its
// share of the line-search sums.
__device__ __forceinline__ void eval_tail(const Dev &dv, int b, int use_d, double *ev, double stp = 0.0)
{
    const Dims &dm = dv.dm;
    double *gt = dv.gt + (size_t)b * dm.ld;
    const double *d = dv.d + (size_t)b * dm.ld, *x = dv.x + (size_t)b * dm.ld;
    for (int k = 0; k < dm.NPest; ++k) {
        // (select chain, not ev[EP_GP + idx]: a run-time index would put ev[] -- and with it every wave
        // of the evaluation kernel -- on scratch memory)
        const int idx = as_const(dv.pp.Pidx)[k];
        double g = 0.0;
#pragma unroll
        for (int j = 0; j < RHS_MAX_NP; ++j) g = (idx == j) ? ev[EP_GP + j] : g;
        if (idx >= RHS_MAX_NP) {
            // (models with many parameters, flat kernel: the tiles' partials of this parameter, added in tile order;
            // written write-through by this launch and read around L1, like the rows)
            const double *pb = dv.evp_big + (size_t)b * dm.nprow * dv.npbig + (idx - RHS_MAX_NP);
            for (int t = 0; t < dm.nprow; ++t) g += ld_sc1(pb + (size_t)t * dv.npbig);
        }
        gt[dm.ND + k] = g;
        if (use_d) ev[EP_GTD] += g * as_const(d)[dm.ND + k];
        ev[EP_GN2] += g * g;
        double pg = g;
        if (dv.pp.lo) {        // bounded: the convergence test looks at the projected gradient
            double xv = as_const(x)[dm.ND + k];
            if (use_d) xv = trial_b(xv, stp, as_const(d)[dm.ND + k], dv.lb_z ? dv.lb_z + (size_t)b * dm.ld : nullptr, dv.pp, dm.ND + k);
            pg = proj_grad(xv, g, dv.pp.lo[dm.ND + k], dv.pp.hi[dm.ND + k]);
        }
        ev[EP_GMAX] = fmax(ev[EP_GMAX], fabs(pg));
    }
}

// The tail of one evaluation of seed b, run by ONE whole wave.  `sh`: 512 bytes of LDS private to
// the calling wave.  SC1: the partial rows were written in this launch (read them around L1).
// One memory round trip: the partial rows and the seed's state are requested together.
template <bool SC1>
__device__ __forceinline__ void eval_epilogue(const Dev &dv, int b, int lane, SeedHot *sh, int mode,
                                              const double *ev_ready = nullptr)
{
    const Dims &dm = dv.dm;
    constexpr int NW8 = sizeof(SeedHot) / 8;
    double *gst = reinterpret_cast<double *>(static_cast<SeedHot *>(&dv.st[b]));
    double hot = 0.0;
    if (mode != EPI_FINALIZE && lane < NW8) hot = gst[lane];       // in flight beside the row loads
    double ev[EP_N];
    if (ev_ready) {
#pragma unroll
        for (int c = 0; c < EP_N; ++c) ev[c] = ev_ready[c];
    } else reduce_eval<SC1>(dv, b, lane, ev);
    if (mode == EPI_FINALIZE) {
        if (lane != 0) return;
        eval_tail(dv, b, 0, ev);
        const double me = ev[EP_ME] * dm.cme, fe = ev[EP_FE] * dm.cfe * as_const(static_cast<const SeedHot *>(dv.st + b))->rf_scale;
        dv.outA[b] = me + fe; dv.outme[b] = me; dv.outfe[b] = fe;
        return;
    }
    // one line-search / ladder step: the seed's hot state (<= 512 B) is staged in LDS with one
    // coalesced 8-byte load per lane, lane 0 runs the (branchy, scalar) state machine on the LDS
    // copy, and the wave writes it back
    double *lst = reinterpret_cast<double *>(sh);
    if (lane < NW8) lst[lane] = hot;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
        atomicAdd(dv.n_evals, 1ULL);
        eval_tail(dv, b, sh->phase == PH_LS, ev, sh->stp);
        SeedResults r;
        r.ame = dv.ame + (size_t)b * dv.max_beta * 3;
        r.pest = nullptr;
        r.status = dv.status + (size_t)b * dv.max_beta;
        r.nit = dv.nit + (size_t)b * dv.max_beta;
        r.nfev = dv.nfev + (size_t)b * dv.max_beta;
        int dec = 0;
        double dirp[DP_N];
        dirp[DP_GD] = sh->gd_dir; dirp[DP_DD] = 0.0;
        ls_step(*sh, ev, dirp, dv.o, dv.rf_ladder, dv.nbeta, r, &dec, dm.cme, dm.cfe, dm.bounded != 0);
        if (dec) atomicSub(dv.n_active, 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < NW8) gst[lane] = lst[lane];
}

}  // namespace va
