// va_nnet_kernels.h -- action + gradient of the feed-forward-network model error
// (reference: varanneal/va_nnet.py:111-255; activation examples/nnet_twin/nnet_twin_anneal.py:20-22).
//
// In the variational formulation every layer state of every training example is an unknown,
// so the N-1 layer transitions are independent given X: there is no sequential forward pass.
// With Z_n = X_n W_n^T + b_n (examples x neurons), a = act(Z_n), r = X_{n+1} - a,
// q = 2 RF c r, delta = -q act'(Z_n):
//     dA/dX_{n+1} += q                       (direct)
//     dA/dX_n     += delta W_n               (product 2)
//     dA/dW_n      = delta^T X_n,  dA/db_n = sum_m delta        (product 3)
// Three float64 products per layer, all on v_mfma_f64_16x16x4_f64:
//   k_nnet_pack   trial point: Xw = x + stp d (line-search points only), Pw = fixed | x + stp d
//   k_nnet_fwd    Z tile -> residual, delta, q              (writes delta, q into gt, fe partial)
//   k_nnet_bwd_x  delta W + q + measurement term -> gt       (me, g.d, g.g, max|g| partials)
//   k_nnet_bwd_w  delta^T X per chunk of examples -> gpart   (no atomics: fixed-order reduce in
//   k_nnet_pred   sum over chunks, scatter to the estimated-parameter tail of gt)
//   k_nnet_rows   (large problems only) folds the per-workgroup partial rows to 32 per seed
// A workgroup owns a 64x64 output tile: 2 x 2 waves, each wave 2 x 2 MFMA blocks (32 x 32, four
// accumulators) so every LDS fragment read feeds two matrix instructions.  K is staged through
// LDS 32 at a time with the next step's global loads already in flight.  Operand tiles that
// are contiguous along K in memory are stored [row][k] (pitch 36), tiles contiguous along the
// row/column index are stored [k][row] (pitch 80): both pitches make the 16x4 MFMA fragment
// read hit 32 distinct 8-byte banks per half-wave.  MFMA blocks that lie wholly outside the
// matrix (narrow layers, few examples) are skipped wave-uniformly.
//
// A header so that generated activation modules (va_user_act.hip, codegen.activation_module_for) can
// instantiate the two kernels that apply the activation -- k_nnet_fwd and k_nnet_small -- for a traced user
// callable; they define VA_NNET_ACT_ONLY to leave the activation-independent kernels out.
#pragma once
#include "va_nnet.h"
#include "va_eval_flat.h"
#include "va_measure.h"

namespace va {


typedef double d4 __attribute__((ext_vector_type(4)));


constexpr int PRK = NN_KC + 4;     // [row][k] pitch: 36 = 4 (mod 32)
constexpr int PKR = NN_TILE + 16;  // [k][row] pitch: 80 = 16 (mod 32)
constexpr int NLD = NN_TILE * NN_KC / NN_THREADS;   // elements per thread per operand tile (8)

// Activations: a type with f(z) and d(z, a) = f'(z), where a = f(z) is at hand (the built-ins express
// the derivative through the value).  Generated modules supply `struct ActUser` (codegen.activation_header).
template <int ID> struct ActBuiltin {
    static __device__ __forceinline__ double f(double z)
    {
        if (ID == NNET_SIGMOID) return 1.0 / (1.0 + exp(-z));
        if (ID == NNET_TANH) return tanh(z);
        if (ID == NNET_RELU) return fmax(z, 0.0);
        if (ID == NNET_SOFTPLUS) return z > 30.0 ? z : log1p(exp(z));
        return z;
    }
    static __device__ __forceinline__ double d(double, double a)
    {
        if (ID == NNET_SIGMOID) return a * (1.0 - a);
        if (ID == NNET_TANH) return 1.0 - a * a;
        if (ID == NNET_RELU) return a > 0.0 ? 1.0 : 0.0;
        if (ID == NNET_SOFTPLUS) return -expm1(-a);            // 1 - exp(-softplus(z)) = sigmoid(z)
        return 1.0;
    }
};

// ---- global -> register -> LDS tile movers (256 threads, 8 elements each) -------------------
// element (r, k) at base[r*rs + k], 64 rows x 32 k; lanes run along k
__device__ __forceinline__ void load_rk(const double *base, size_t rs, int nr, int nk, int t, double v[NLD])
{
    const int k = t & 31;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int r = (t >> 5) + 8 * u;
        v[u] = (r < nr && k < nk) ? base[(size_t)r * rs + k] : 0.0;
    }
}
__device__ __forceinline__ void store_rk(double *L, int t, const double v[NLD])
{
#pragma unroll
    for (int u = 0; u < NLD; ++u) L[((t >> 5) + 8 * u) * PRK + (t & 31)] = v[u];
}
// element (k, r) at base[k*ks + r], 32 k x 64 r; lanes run along r
__device__ __forceinline__ void load_kr(const double *base, size_t ks, int nr, int nk, int t, double v[NLD])
{
    const int r = t & 63;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int k = (t >> 6) + 4 * u;
        v[u] = (r < nr && k < nk) ? base[(size_t)k * ks + r] : 0.0;
    }
}
__device__ __forceinline__ void store_kr(double *L, int t, const double v[NLD])
{
#pragma unroll
    for (int u = 0; u < NLD; ++u) L[((t >> 6) + 4 * u) * PKR + (t & 63)] = v[u];
}

// ---- one K step of 32 on the matrix cores: acc[2][2] (32x32) += A[32 x 32] B[32 x 32] -------
// fragment layout of v_mfma_f64_16x16x4_f64: lane l feeds A[row = l&15][k = l>>4] and
// B[k = l>>4][col = l&15]; result register i of lane l is C[row = (l>>4) + 4i][col = l&15].
// live: bit (2*bi + bj) set <=> block (bi, bj) of this wave intersects the matrix.
template <bool A_RK, bool B_RK>
__device__ __forceinline__ void mma_step(const double *As, const double *Bs, int wr, int wc, int lane, int live,
                                         d4 acc[2][2])
{
    const int lo = lane & 15, hi = lane >> 4;
    if (!live) return;
#pragma unroll
    for (int kk = 0; kk < NN_KC / 4; ++kk) {
        const int k = 4 * kk + hi;
        double a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wr * 32 + i * 16 + lo, c = wc * 32 + i * 16 + lo;
            a[i] = A_RK ? As[r * PRK + k] : As[k * PKR + r];
            b[i] = B_RK ? Bs[c * PRK + k] : Bs[k * PKR + c];
        }
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj)
                if (live & (1 << (2 * bi + bj)))
                    acc[bi][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], b[bj], acc[bi][bj], 0, 0, 0);
    }
}
__device__ __forceinline__ int live_mask(int wr, int wc, int nra, int nrb)
{
    int m = 0;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
            if (wr * 32 + bi * 16 < nra && wc * 32 + bj * 16 < nrb) m |= 1 << (2 * bi + bj);
    return m;
}

__device__ __forceinline__ bool seed_live(const Dev &dv, int b, int &use_d, double &stp, double &rf)
{
    const SeedState &st = dv.st[b];
    const int phase = st.phase;
    use_d = (phase == PH_LS);
    stp = use_d ? st.stp : 0.0;
    rf = st.rf_scale;
    return phase == PH_START || phase == PH_LS;
}

// workgroup reduction of 4 values (k == 3 is a max), thread 0 gets the totals
__device__ __forceinline__ void wg_reduce4(double v[4], double *red, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double w = (k == 3) ? wave_max(v[k]) : wave_sum(v[k]);
        if (lane == 0) red[wave * 4 + k] = w;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double w = red[k];
            for (int ww = 1; ww < NN_THREADS / 64; ++ww) w = (k == 3) ? fmax(w, red[ww * 4 + k]) : w + red[ww * 4 + k];
            v[k] = w;
        }
    }
}
// partial row r of seed b: straight into the table the line-search kernel reads, or into the
// raw table k_nnet_rows folds
__device__ __forceinline__ void put_row(const Dev &dv, const NnetDev &nn, int b, int r, double me, double fe,
                                        double gtd, double gn2, double gmax)
{
    double *row = nn.raw ? nn.raw + ((size_t)b * nn.nraw + r) * EP_GP
                         : dv.evp + ((size_t)b * dv.dm.nprow + r) * EP_N;
    row[EP_ME] = me; row[EP_FE] = fe; row[EP_GTD] = gtd; row[EP_GN2] = gn2; row[EP_GMAX] = gmax;
}

// Measurement term of one observed element (example m, observed index l of its layer) under a full matrix
// R (L x L, not assumed symmetric; va_nnet.py:136-139): its share of sum diff.(R.diff) and the derivative of
// that sum with respect to it.  diff_at(k) = x[observed neuron k] - data[m][k].
template <class DIFF>
__device__ __forceinline__ void nnet_meas_matrix(const double *R, int L, int l, double diff, DIFF diff_at,
                                                 double &share, double &deriv)
{
    double row = 0.0, sym = 0.0;
    for (int k = 0; k < L; ++k) {
        const double dk = diff_at(k);
        row += R[l * L + k] * dk;
        sym += (R[l * L + k] + R[k * L + l]) * dk;
    }
    share = diff * row; deriv = sym;
}

#ifndef VA_NNET_ACT_ONLY
// ------------------------------------------------------------------ K0: trial point
// NN_PACK elements per thread, strided by the workgroup: at a plain evaluation point only the workgroups that
// reach the parameter tail have work, and there are NN_PACK times fewer to dispatch and retire
__global__ __launch_bounds__(NN_THREADS) void k_nnet_pack(const Dev dv, const NnetDev nn)
{
    const int b = blockIdx.y;
    int use_d; double stp, rf;
    if (!seed_live(dv, b, use_d, stp, rf)) return;
    const int base = blockIdx.x * (NN_THREADS * NN_PACK);
    if (!use_d && base + NN_THREADS * NN_PACK <= nn.NDens) return;          // workgroup-uniform
    const size_t vo = (size_t)b * dv.dm.ld;
#pragma unroll
    for (int u = 0; u < NN_PACK; ++u) {
        const int i = base + u * NN_THREADS + threadIdx.x;
        if (i < nn.NDens) {
            // at a plain evaluation point (no line-search step) the products read x itself
            if (use_d) nn.Xw[vo + i] = trial(dv.x[vo + i], stp, dv.d[vo + i]);
        } else if (i < nn.NDens + nn.NP) {
            const int j = i - nn.NDens, k = nn.pmap[j];
            double v;
            if (k >= 0) {
                v = dv.x[vo + nn.NDens + k];
                if (use_d) v = trial(v, stp, dv.d[vo + nn.NDens + k]);
            } else v = nn.Pfix[(size_t)b * nn.NP + j];
            nn.Pw[(size_t)b * nn.NP + j] = v;
        }
    }
}

// the weights in the fragment order k_nnet_fb's products read them: for transition n, column block cb, k-step kk, lane l
//     first product  Z = X_n W_n^T:   Wf[wfoff[n]        + (cb * nkp + kk) * 64 + l] = W_n[16 cb + (l & 15)][4 kk + (l >> 4)]
//     second product G = delta W_n:   Wf[wfoff[NL-1 + n] + (cb * nkp + kk) * 64 + l] = W_n[4 kk + (l >> 4)][16 cb + (l & 15)]
// (nkp = nn_fb_steps(K) k-steps per column block, zeros outside the matrix: the products run without a bounds test)
// and the partial rows no workgroup of the fused evaluation writes (those of the separate kernels) zeroed
__global__ __launch_bounds__(NN_THREADS) void k_nnet_wfrag(const Dev dv, const NnetDev nn)
{
    __shared__ int toff[2 * NN_FB_LAYERS], ls[NN_FB_LAYERS], lw[NN_FB_LAYERS];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (blockIdx.x == 0 && b == 0 && tid == 0) nn.fb_cu[0] = 0;          // arrivals at the fused kernel (its stagger)
    int use_d; double stp, rf;
    if (!seed_live(dv, b, use_d, stp, rf)) return;
    const double *Pw = nn.Pw + (size_t)b * nn.NP;
    double *Wf = nn.Wf + (size_t)b * nn.wfsz;
    const int NT = nn.NL - 1;
    for (int i = tid; i < 2 * NT; i += NN_THREADS) toff[i] = nn.wfoff[i];
    for (int i = tid; i < nn.NL; i += NN_THREADS) { ls[i] = nn.s[i]; lw[i] = i < NT ? nn.woff[i] : 0; }
    __syncthreads();
    // a run of NN_THREADS consecutive entries lies in one table (the tables are multiples of 4096 entries long)
    for (int e0 = blockIdx.x * NN_THREADS; e0 < nn.wfsz; e0 += gridDim.x * NN_THREADS) {
        int t = 0;
        while (t + 1 < 2 * NT && e0 >= toff[t + 1]) ++t;
        const bool second = t >= NT;
        const int n = second ? t - NT : t;
        const int sn = ls[n], sn1 = ls[n + 1], nkp = nn_fb_steps(second ? sn1 : sn);
        const int f = e0 + tid - toff[t], l = f & 63, blk = f >> 6, cb = blk / nkp, kk = blk - cb * nkp;
        const int c = cb * 16 + (l & 15), k = 4 * kk + (l >> 4);
        double w = 0.0;
        if (!second) { if (c < sn1 && k < sn) w = Pw[lw[n] + (size_t)c * sn + k]; }
        else if (c < sn && k < sn1) w = Pw[lw[n] + (size_t)k * sn + c];
        Wf[e0 + tid] = w;
    }
    if (blockIdx.x == 0)
        for (int r = nn.nfb + threadIdx.x; r < nn.n1 + nn.n2; r += NN_THREADS) put_row(dv, nn, b, r, 0.0, 0.0, 0.0, 0.0, 0.0);
}

#endif  // VA_NNET_ACT_ONLY

// ------------------------------------------------------------------ K1: Z, residual, delta
template <class ACT>
__global__ __launch_bounds__(NN_THREADS) void k_nnet_fwd(const Dev dv, const NnetDev nn)
{
    __shared__ double As[NN_TILE * PRK], Bs[NN_TILE * PRK], red[16];
    const int b = blockIdx.y, tid = threadIdx.x;
    int use_d; double stp, rf;
    if (!seed_live(dv, b, use_d, stp, rf)) return;
    const NnetTile tl = nn.t1[blockIdx.x];
    const int m0 = tl.r0, i0 = tl.c0;
    const int sn = tl.sn, sn1 = tl.sn1, K = sn;
    const size_t vo = (size_t)b * dv.dm.ld;
    const double *Xs = use_d ? nn.Xw : dv.x;                 // trial point, or x itself
    const double *X = Xs + vo + (size_t)m0 * nn.NDnet + tl.offn;
    const double *W = nn.Pw + (size_t)b * nn.NP + tl.woff + (size_t)i0 * sn;
    const int nra = min(NN_TILE, nn.M - m0), nrb = min(NN_TILE, sn1 - i0);
    const int lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int live = live_mask(wr, wc, nra, nrb);

    d4 acc[2][2] = {};
    double va[NLD], vb[NLD];
    load_rk(X, nn.NDnet, nra, K, tid, va);
    load_rk(W, sn, nrb, K, tid, vb);
    for (int k0 = 0; k0 < K; k0 += NN_KC) {
        store_rk(As, tid, va); store_rk(Bs, tid, vb);
        __syncthreads();
        if (k0 + NN_KC < K) {
            load_rk(X + k0 + NN_KC, nn.NDnet, nra, K - k0 - NN_KC, tid, va);
            load_rk(W + k0 + NN_KC, sn, nrb, K - k0 - NN_KC, tid, vb);
        }
        mma_step<true, true>(As, Bs, wr, wc, lane, live, acc);
        __syncthreads();
    }
    // epilogue: lane holds Z[m][i], 16 consecutive neurons i per 16 lanes
    const double cq = 2.0 * rf * dv.dm.cfe;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
        const int i = i0 + wc * 32 + bj * 16 + (lane & 15);
        if (i >= sn1) continue;
        // (no early request of bias / x_{n+1}: 32 fewer live registers put a fourth workgroup on the CU)
        const double bias = nn.Pw[(size_t)b * nn.NP + tl.boff + i];
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + bi * 16 + (lane >> 4) + 4 * r;
                if (m >= nn.M) continue;
                const size_t idx = vo + (size_t)m * nn.NDnet + tl.offn1 + i;
                const double z = acc[bi][bj][r] + bias;
                const double a = ACT::f(z), da = ACT::d(z, a);
                const double res = Xs[idx] - a;
                const double q = cq * res;
                v[1] += res * res;
                nn.delta[idx] = -q * da;
                dv.gt[idx] = q;
            }
    }
    wg_reduce4(v, red, tid);
    if (tid == 0) put_row(dv, nn, b, blockIdx.x, 0.0, v[1], 0.0, 0.0, 0.0);
}

// one B fragment of k_nnet_fb's products (a buffer load: as an intrinsic call it stays where the source puts it --
// plain loads carried around the k loop are sunk through the loop's phi nodes by the compiler, i.e. issued where their
// values are used, and the products then wait out an L2 round trip per group of k-steps)
typedef unsigned fb_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double fb_frag(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0));
}

// ------------------------------------------------------------------ K1 + K2 in one: forward and state-gradient products
// Layers up to NN_FB_W = 128 wide (c5x: 8 x 128, M = 2048).  As separate kernels the two products move every delta and
// every q through HBM twice and re-read the states (profiles/r03_nnet_c5x_ablation.txt: 78-83 % of each kernel is
// memory time).  Here a workgroup of 4 waves owns a block of 32 EXAMPLES and marches it through the transitions (two
// workgroups per CU, 256 registers a lane each):
//   the block's states X_n sit in LDS (A operand of Z = X_n W_n^T); wave w owns the 32 output columns 32 w ... of every
//   layer (2 x 2 accumulators of 16 x 16); the B operand comes straight from L2 in fragment order (nn.Wf, written once
//   per evaluation by k_nnet_wfrag, padded so that the k loops carry no bounds test) -- 512 contiguous bytes per matrix
//   instruction, no LDS staging, requested NN_FB_PF k-steps ahead;
//   epilogue A: residual, q, delta from Z and x_{n+1} (read ONCE from HBM: it becomes the next transition's LDS operand);
//     delta goes to LDS (A operand of the second product) and to HBM (for k_nnet_bwd_w); q STAYS IN REGISTERS: the
//     accumulator layout of the next transition's state gradient is the same (example, column) map;
//   second product G_n = delta_n W_n (B operand: the second fragment table);
//   epilogue B: dA/dx_n = G_n + q_{n-1} + measurement term -> gt.
// Every state is read once and every gradient / delta entry written once: 2.79 -> 1.3 GB per evaluation at c5x.
// MEASURED (round 4, profiles/r04_nnet_fused.txt): 414 us against 270 + 258 us for the two kernels it replaces (c5x
// evaluation 709 -> 637 us).  What the first version (563 us) lost, found with the phase marks of va_measure.h and the ISA:
//   bounds tests in the k loops made every k-step a branch with a wait behind each LDS read; plain loads carried around
//   the loop were sunk to their uses by the compiler (no prefetch at all); x_{n+1} / d fetched inside the epilogues made
//   every element wait for an HBM round trip; per-layer sizes and offsets fetched from global memory put a round trip in
//   front of each phase.  The products now run at the matrix rate of two waves per SIMD; what is left are the epilogues
//   (f64 exp and division: ~4 us per layer on the vector ALUs; the stores of all workgroups of the launch come in bursts).
template <class ACT>
__global__ __launch_bounds__(NN_FB_THREADS, NN_FB_WGS) void k_nnet_fb(const Dev dv, const NnetDev nn)
{
    extern __shared__ __attribute__((aligned(16))) double fbs[];
    constexpr int R = NN_FB_R, RB = NN_FB_R / 16, PX = NN_FB_PITCH, NW = NN_FB_THREADS / 64, CW = NN_FB_W / 16 / NW;
    double *Xs = fbs, *DL = Xs + R * PX, *red = DL + R * PX;          // red: [NW * 5 + 8]
    // per-layer numbers (s, off, woff, boff, the two fragment-table offsets), read once: fetched where they are used,
    // each costs the wave a round trip to L2 per layer before the loads that depend on it can even be requested
    int *meta = reinterpret_cast<int *>(red + NW * 5 + 8);            // [6][NN_FB_LAYERS]
    const int b = blockIdx.y, tid = threadIdx.x;
    int use_d; double stp, rf;
    if (!seed_live(dv, b, use_d, stp, rf)) return;
    const int m0 = blockIdx.x * R;
    const int nra = min(R, nn.M - m0);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lo = lane & 15, hi = lane >> 4;
    const int cb0 = wave * CW;                                            // the wave's first block of 16 columns (of every layer)
    const size_t vo = (size_t)b * dv.dm.ld;
    const double *Xg = (use_d ? nn.Xw : dv.x) + vo;                      // the trial point's states
    const double *Pw = nn.Pw + (size_t)b * nn.NP;
    const double *dg = dv.d + vo;
    double *gtg = dv.gt + vo, *dlg = nn.delta + vo;
    const double cq = 2.0 * rf * dv.dm.cfe;
    const int NL = nn.NL;
    // the seed's fragment tables (k_nnet_wfrag)
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void *)(nn.Wf + (size_t)b * nn.wfsz), 0, nn.wfsz * 8, 0x00020000);
    if (nn.fb_stagger > 0) {
        // Left alone, every workgroup of the launch walks through the layers in step with every other: all of them store a
        // layer's gradients (or fetch the next layer's states) in the same few microseconds -- bursts at the HBM's full
        // rate with the matrix cores idle, then matrix phases with the memory idle.  The workgroups that start the launch
        // are put out of step: the i-th to arrive waits (i / 8 mod 8) eighths of fb_stagger (about one layer's period)
        // -- consecutive workgroups go to different XCDs, so every XCD gets every phase.
        if (tid == 0) {
            const int i = atomicAdd(nn.fb_cu, 1);
            if (i < nn.fb_slots) {
                const unsigned long long wait = (unsigned long long)nn.fb_stagger * ((i >> 3) & 7) / 8, t0 = wall_clock64();
                while (wall_clock64() - t0 < wait) __builtin_amdgcn_s_sleep(16);
            }
        }
        __syncthreads();
    }
    FB_MARK_SETUP();
    if (tid < NL) {
        meta[tid] = nn.s[tid]; meta[NN_FB_LAYERS + tid] = nn.off[tid];
        if (tid < NL - 1) {
            meta[2 * NN_FB_LAYERS + tid] = nn.woff[tid]; meta[3 * NN_FB_LAYERS + tid] = nn.boff[tid];
            meta[4 * NN_FB_LAYERS + tid] = nn.wfoff[tid]; meta[5 * NN_FB_LAYERS + tid] = nn.wfoff[NL - 1 + tid];
        }
    }
    {   // the block's input-layer states; rows / columns that do not exist are zeros (they pad K)
        const int s0 = nn.s[0];
        constexpr int NE = R * NN_FB_W / NN_FB_THREADS;
        double xin[NE];
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * NN_FB_THREADS, r = e >> 7, c = e & (NN_FB_W - 1);
            xin[i] = (r < nra && c < s0) ? Xg[(size_t)(m0 + r) * nn.NDnet + c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + i * NN_FB_THREADS, r = e >> 7, c = e & (NN_FB_W - 1);
            Xs[r * PX + c] = xin[i];
        }
    }
    __syncthreads();
    FB_MARK(0);
    // x_{n+1} of the lane's elements (epilogue A of transition n needs them) is requested a phase early -- before the
    // first transition here, after the second product of transition n - 1 later -- and lands under epilogue B and the
    // first product: asked for where it is used, every element waits out an HBM round trip (its own load's and the
    // previous element's store's)
    d4 xn[CW][RB];
    auto request_x = [&](int n1) {                                        // n1: the layer whose states are fetched
        const int s1 = meta[n1], o1 = meta[NN_FB_LAYERS + n1];
#pragma unroll
        for (int c2 = 0; c2 < CW; ++c2) {
            const int col = (cb0 + c2) * 16 + lo;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ml = rb * 16 + hi + 4 * r;
                    xn[c2][rb][r] = (col < s1 && ml < nra) ? Xg[(m0 + ml) * nn.NDnet + o1 + col] : 0.0;
                }
        }
    };
    if (NL > 1) request_x(1);
    d4 qprev[CW][RB];
#pragma unroll
    for (int c2 = 0; c2 < CW; ++c2)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) qprev[c2][rb] = d4{0.0, 0.0, 0.0, 0.0};
    double v_me = 0.0, v_fe = 0.0, v_gtd = 0.0, v_gn2 = 0.0, v_gmax = 0.0;
    {
        // the input layer's measurement term: dA/dx_0 = delta_0 W_0 + 2 cme RM (x_0 - d_in).  It rides in the accumulators the
        // second product of the first transition starts from (q of a transition before the first: zero otherwise); x_0 is in
        // LDS, the data are requested all at once while nothing else is live
        const int s0 = meta[0], Lin = nn.Lin;
        const double mc = 2.0 * dv.dm.cme * nn.rm_in;
        int lc[CW];
#pragma unroll
        for (int c2 = 0; c2 < CW; ++c2) { const int col = (cb0 + c2) * 16 + lo; lc[c2] = col < s0 ? nn.lmap_in[col] : -1; }
        d4 my[CW][RB];
#pragma unroll
        for (int c2 = 0; c2 < CW; ++c2)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ml = rb * 16 + hi + 4 * r;
                    my[c2][rb][r] = (lc[c2] >= 0 && ml < nra) ? nn.din[(size_t)(m0 + ml) * Lin + lc[c2]] : 0.0;
                }
#pragma unroll
        for (int c2 = 0; c2 < CW; ++c2) {
            const int col = (cb0 + c2) * 16 + lo;
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ml = rb * 16 + hi + 4 * r;
                    const double diff = (lc[c2] >= 0 && ml < nra) ? Xs[ml * PX + col] - my[c2][rb][r] : 0.0;
                    v_me += nn.rm_in * diff * diff;
                    qprev[c2][rb][r] = mc * diff;
                }
        }
    }
    for (int n = 0; n <= NL - 1; ++n) {
        const int sn = meta[n], offn = meta[NN_FB_LAYERS + n];
        const bool last = n == NL - 1;
        const int sn1 = last ? 0 : meta[n + 1], offn1 = last ? 0 : meta[NN_FB_LAYERS + n + 1];
        d4 q[CW][RB];
#pragma unroll
        for (int c2 = 0; c2 < CW; ++c2)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) q[c2][rb] = d4{0.0, 0.0, 0.0, 0.0};
        if (!last) {
            // ---- product 1: Z = X_n W_n^T for the wave's columns of layer n+1
            d4 acc[CW][RB];
#pragma unroll
            for (int c2 = 0; c2 < CW; ++c2)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[c2][rb] = d4{0.0, 0.0, 0.0, 0.0};
            if (cb0 * 16 < sn1) {
                // (B fragments NN_FB_PF k-steps ahead: an L2 round trip is about the matrix time of eight steps of the SIMD's two
                // waves.  The table is padded -- whole groups of k-steps and a group of zeros behind the last -- and X_n's
                // columns beyond the layer are zeros: no bounds test in the loop)
                constexpr int PF = NN_FB_PF;
                const int nkp = nn_fb_steps(sn), nkr = nkp - PF;
                const int wf = (meta[4 * NN_FB_LAYERS + n] + cb0 * nkp * 64 + lane) * 8;          // byte offset of the lane's first fragment element
                const double *xa = Xs + lo * PX + hi;
                double bq[CW][PF];
#pragma unroll
                for (int u = 0; u < PF; ++u) {                                      // (requested in the order the loop re-requests them:
#pragma unroll                                                                          //  the oldest request is the first one waited for)
                    for (int c2 = 0; c2 < CW; ++c2) bq[c2][u] = fb_frag(wr, wf + (c2 * nkp + u) * 512);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // per k-step, in this order (pinned: left alone the scheduler requests the fragments early into fresh registers
                // and copies them into the ring behind a wait, which shortens the distance to less than one group):
                // the next step's A elements from LDS, this step's matrix instructions, then the reload of the ring slot
                double a[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a[rb] = xa[rb * 16 * PX];
                for (int k0 = 0; k0 < nkr; k0 += PF) {
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        double an[RB];
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) an[rb] = xa[rb * 16 * PX + 4 * (k0 + u + 1)];   // (the last one is not used)
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                            for (int c2 = 0; c2 < CW; ++c2)
                                acc[c2][rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rb], bq[c2][u], acc[c2][rb], 0, 0, 0);
#pragma unroll
                        for (int c2 = 0; c2 < CW; ++c2) bq[c2][u] = fb_frag(wr, wf + (c2 * nkp + k0 + PF + u) * 512);
#pragma unroll
                        for (int rb = 0; rb < RB; ++rb) a[rb] = an[rb];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            FB_MARK(1);
            __syncthreads();                                   // every wave is done with X_n
            FB_MARK(2);
            // ---- epilogue A: residual, q, delta; x_{n+1} becomes the next operand (columns beyond the layer: zeros)
#pragma unroll
            for (int c2 = 0; c2 < CW; ++c2) {
                const int col = (cb0 + c2) * 16 + lo;
                const bool on = col < sn1;
                const double bias = on ? Pw[meta[3 * NN_FB_LAYERS + n] + col] : 0.0;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ml = rb * 16 + hi + 4 * r;
                        const double xv = xn[c2][rb][r];
                        double dl = 0.0;
                        if (on && ml < nra) {
                            const int e = (m0 + ml) * nn.NDnet + offn1 + col;        // (the ensemble has fewer than 2^31 states: create)
                            const double z = acc[c2][rb][r] + bias;
                            const double a = ACT::f(z), da = ACT::d(z, a);
                            const double res = xv - a, qv = cq * res;
                            v_fe += res * res;
                            dl = -qv * da;
                            q[c2][rb][r] = qv;
#if !defined(VA_FB_ABL) || VA_FB_ABL != 2
                            dlg[e] = dl;
#endif
                        }
                        DL[ml * PX + col] = dl;
                        Xs[ml * PX + col] = xv;
                    }
            }
            FB_MARK(3);
            __syncthreads();
            FB_MARK(4);
        }
        // ---- product 2 (not for the last layer: nothing leaves it) and epilogue B: dA/dx_n for the wave's columns of layer n
        // (the accumulators start from q_{n-1}: dA/dx_n = q_{n-1} + delta_n W_n, same (example, column) map)
        d4 g[CW][RB], dn[CW][RB];
#pragma unroll
        for (int c2 = 0; c2 < CW; ++c2)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) { g[c2][rb] = qprev[c2][rb]; dn[c2][rb] = d4{0.0, 0.0, 0.0, 0.0}; }
        if (use_d) {                                           // the direction's entries epilogue B multiplies with (same reason)
#pragma unroll
            for (int c2 = 0; c2 < CW; ++c2) {
                const int col = (cb0 + c2) * 16 + lo;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ml = rb * 16 + hi + 4 * r;
                        if (col < sn && ml < nra) dn[c2][rb][r] = dg[(m0 + ml) * nn.NDnet + offn + col];
                    }
            }
        }
        if (!last && cb0 * 16 < sn) {
            constexpr int PF = NN_FB_PF;
            const int nkp = nn_fb_steps(sn1), nkr = nkp - PF;
            const int wf = (meta[5 * NN_FB_LAYERS + n] + cb0 * nkp * 64 + lane) * 8;
            const double *da = DL + lo * PX + hi;
            double bq[CW][PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
#pragma unroll
                for (int c2 = 0; c2 < CW; ++c2) bq[c2][u] = fb_frag(wr, wf + (c2 * nkp + u) * 512);
                __builtin_amdgcn_sched_barrier(0);
            }
            double a[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a[rb] = da[rb * 16 * PX];
            for (int k0 = 0; k0 < nkr; k0 += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    double an[RB];
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) an[rb] = da[rb * 16 * PX + 4 * (k0 + u + 1)];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                        for (int c2 = 0; c2 < CW; ++c2)
                            g[c2][rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rb], bq[c2][u], g[c2][rb], 0, 0, 0);
#pragma unroll
                    for (int c2 = 0; c2 < CW; ++c2) bq[c2][u] = fb_frag(wr, wf + (c2 * nkp + k0 + PF + u) * 512);
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) a[rb] = an[rb];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (n + 2 <= NL - 1) request_x(n + 2);                 // (transition n + 1's; its q, delta are formed two phases from here)
        FB_MARK(5);
        const bool measured = last;                             // (the input layer's term came in through qprev, the hidden layers have none)
        // the measurement terms of the input / output layer: per column block, every operand is requested before the first is
        // used (fetched element by element they cost the two layers 16 memory round trips each: ~15 us a layer); the output
        // layer's states are still in LDS (epilogue A of the last transition put them there)
        const int *lm = nn.lmap_out;
        const double *dat = nn.dout;
        const int L = nn.Lout;
        const double mrm = measured ? nn.rm_out : 0.0, mc = 2.0 * dv.dm.cme * mrm;
#pragma unroll
        for (int c2 = 0; c2 < CW; ++c2) {
            const int col = (cb0 + c2) * 16 + lo;
            d4 mdiff[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) mdiff[rb] = d4{0.0, 0.0, 0.0, 0.0};
            if (measured) {
                const int l = col < sn ? lm[col] : -1;
                d4 mx[RB], my[RB];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ml = rb * 16 + hi + 4 * r;
                        const bool on = l >= 0 && ml < nra;
                        my[rb][r] = on ? dat[(size_t)(m0 + ml) * L + l] : 0.0;
                        mx[rb][r] = !on ? 0.0 : (last ? Xs[ml * PX + col] : Xg[(m0 + ml) * nn.NDnet + offn + col]);
                    }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) mdiff[rb] = mx[rb] - my[rb];
            }
            if (col < sn) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ml = rb * 16 + hi + 4 * r;
                        if (ml >= nra) continue;
                        const int e = (m0 + ml) * nn.NDnet + offn + col;
                        const double diff = mdiff[rb][r];                 // (0 where nothing is measured)
                        v_me += mrm * diff * diff;
                        const double gv = g[c2][rb][r] + mc * diff;
#if !defined(VA_FB_ABL) || VA_FB_ABL != 1
                        gtg[e] = gv;
#endif
                        v_gtd += gv * dn[c2][rb][r];
                        v_gn2 += gv * gv;
                        v_gmax = fmax(v_gmax, fabs(gv));
                    }
            }
        }
        FB_MARK(6);
#pragma unroll
        for (int c2 = 0; c2 < CW; ++c2)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) qprev[c2][rb] = q[c2][rb];
    }
    FB_MARK_FLUSH(dv.pz.stamps);
    // the workgroup's row of partial sums
    {
        double w;
        w = wave_sum(v_me); if (lane == 0) red[wave * 5 + 0] = w;
        w = wave_sum(v_fe); if (lane == 0) red[wave * 5 + 1] = w;
        w = wave_sum(v_gtd); if (lane == 0) red[wave * 5 + 2] = w;
        w = wave_sum(v_gn2); if (lane == 0) red[wave * 5 + 3] = w;
        w = wave_max(v_gmax); if (lane == 0) red[wave * 5 + 4] = w;
        __syncthreads();
        if (tid == 0) {
            double t[5];
            for (int k = 0; k < 5; ++k) {
                t[k] = red[k];
                for (int ww = 1; ww < NW; ++ww) t[k] = (k == 4) ? fmax(t[k], red[ww * 5 + k]) : t[k] + red[ww * 5 + k];
            }
            put_row(dv, nn, b, blockIdx.x, t[0], t[1], t[2], t[3], t[4]);
        }
    }
}
inline size_t nnet_fb_lds() { return sizeof(double) * ((size_t)2 * NN_FB_R * NN_FB_PITCH + 5 * (NN_FB_THREADS / 64) + 8) + sizeof(int) * 6 * NN_FB_LAYERS; }

#ifndef VA_NNET_ACT_ONLY
// ------------------------------------------------------------------ K2: dA/dX
// RMM: full measurement matrices (their row walk costs registers the scalar-weight instantiation does not pay)
template <bool RMM>
__global__ __launch_bounds__(NN_THREADS) void k_nnet_bwd_x(const Dev dv, const NnetDev nn)
{
    __shared__ double As[NN_TILE * PRK], Bs[NN_KC * PKR], red[16];
    const int b = blockIdx.y, tid = threadIdx.x;
    int use_d; double stp, rf;
    if (!seed_live(dv, b, use_d, stp, rf)) return;
    const NnetTile tl = nn.t2[blockIdx.x];
    const int n = tl.layer, m0 = tl.r0, j0 = tl.c0;
    const int sn = tl.sn;
    const int K = tl.sn1;                                  // 0 for the output layer: no product
    const size_t vo = (size_t)b * dv.dm.ld;
    const int nra = min(NN_TILE, nn.M - m0), nrb = min(NN_TILE, sn - j0);
    const int lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int live = live_mask(wr, wc, nra, nrb);

    d4 acc[2][2] = {};
    // q left by k_nnet_fwd in gt: requested now, consumed in the epilogue
    double q0[2][2][4];
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
        const int j = j0 + wc * 32 + bj * 16 + (lane & 15);
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + bi * 16 + (lane >> 4) + 4 * r;
                q0[bi][bj][r] = (n > 0 && j < sn && m < nn.M) ? dv.gt[vo + (size_t)m * nn.NDnet + tl.offn + j] : 0.0;
            }
    }
    if (K > 0) {
        const double *Dl = nn.delta + vo + (size_t)m0 * nn.NDnet + tl.offn1;
        const double *W = nn.Pw + (size_t)b * nn.NP + tl.woff + j0;
        double va[NLD], vb[NLD];
        load_rk(Dl, nn.NDnet, nra, K, tid, va);
        load_kr(W, sn, nrb, K, tid, vb);
        for (int k0 = 0; k0 < K; k0 += NN_KC) {
            store_rk(As, tid, va); store_kr(Bs, tid, vb);
            __syncthreads();
            if (k0 + NN_KC < K) {
                load_rk(Dl + k0 + NN_KC, nn.NDnet, nra, K - k0 - NN_KC, tid, va);
                load_kr(W + (size_t)(k0 + NN_KC) * sn, sn, nrb, K - k0 - NN_KC, tid, vb);
            }
            mma_step<true, false>(As, Bs, wr, wc, lane, live, acc);
            __syncthreads();
        }
    }
    double v[4] = {0.0, 0.0, 0.0, 0.0};          // me, g.d, g.g, max|g|
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
        const int j = j0 + wc * 32 + bj * 16 + (lane & 15);
        if (j >= sn) continue;
        int l = -1; double rm = 0.0; const double *dat = nullptr, *R = nullptr; const int *lidx = nullptr; int L = 0;
        if (n == 0) { l = nn.lmap_in[j]; rm = nn.rm_in; dat = nn.din; L = nn.Lin; R = nn.rmm_in; lidx = nn.lidx_in; }
        else if (n == nn.NL - 1) { l = nn.lmap_out[j]; rm = nn.rm_out; dat = nn.dout; L = nn.Lout; R = nn.rmm_out; lidx = nn.lidx_out; }
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + bi * 16 + (lane >> 4) + 4 * r;
                if (m >= nn.M) continue;
                const size_t idx = vo + (size_t)m * nn.NDnet + tl.offn + j;
                double g = acc[bi][bj][r] + q0[bi][bj][r];
                if (l >= 0) {
                    const double *Xt = use_d ? nn.Xw : dv.x;
                    const double diff = Xt[idx] - dat[(size_t)m * L + l];
                    const size_t rowb = vo + (size_t)m * nn.NDnet + tl.offn;
                    if (RMM) {
                        double share, deriv;
                        nnet_meas_matrix(R, L, l, diff, [&](int k) { return Xt[rowb + lidx[k]] - dat[(size_t)m * L + k]; }, share, deriv);
                        v[0] += share;
                        g += dv.dm.cme * deriv;
                    } else {
                        v[0] += rm * diff * diff;
                        g += 2.0 * dv.dm.cme * rm * diff;
                    }
                }
                dv.gt[idx] = g;
                if (use_d) v[1] += g * dv.d[idx];
                v[2] += g * g;
                v[3] = fmax(v[3], fabs(g));
            }
    }
    wg_reduce4(v, red, tid);
    if (tid == 0) put_row(dv, nn, b, nn.n1 + blockIdx.x, v[0], 0.0, v[1], v[2], v[3]);
}

// ------------------------------------------------------------------ K3: dA/dW, dA/db per chunk
__global__ __launch_bounds__(NN_THREADS) void k_nnet_bwd_w(const Dev dv, const NnetDev nn)
{
    __shared__ double As[NN_KC * PKR], Bs[NN_KC * PKR];
    const int b = blockIdx.y, tid = threadIdx.x;
    int use_d; double stp, rf;
    if (!seed_live(dv, b, use_d, stp, rf)) return;
    const NnetTile tl = nn.t3[blockIdx.x];
    const int i0 = tl.r0, j0 = tl.c0;
    const int sn = tl.sn, sn1 = tl.sn1;
    const int mb = tl.chunk * nn.mch;
    const int K = min(nn.mch, nn.M - mb);                       // examples in this chunk
    const size_t vo = (size_t)b * dv.dm.ld;
    const double *Dl = nn.delta + vo + (size_t)mb * nn.NDnet + tl.offn1 + i0;
    const double *X = (use_d ? nn.Xw : dv.x) + vo + (size_t)mb * nn.NDnet + tl.offn + j0;
    const int nra = min(NN_TILE, sn1 - i0), nrb = min(NN_TILE, sn - j0);
    const int lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int live = live_mask(wr, wc, nra, nrb);

    d4 acc[2][2] = {};
    double va[NLD], vb[NLD], bsum = 0.0;
    load_kr(Dl, nn.NDnet, nra, K, tid, va);
    load_kr(X, nn.NDnet, nrb, K, tid, vb);
    for (int k0 = 0; k0 < K; k0 += NN_KC) {
        store_kr(As, tid, va); store_kr(Bs, tid, vb);
        __syncthreads();
        if (k0 + NN_KC < K) {
            const size_t sh = (size_t)(k0 + NN_KC) * nn.NDnet;
            load_kr(Dl + sh, nn.NDnet, nra, K - k0 - NN_KC, tid, va);
            load_kr(X + sh, nn.NDnet, nrb, K - k0 - NN_KC, tid, vb);
        }
        mma_step<false, false>(As, Bs, wr, wc, lane, live, acc);
        if (j0 == 0 && tid < NN_TILE) {                         // bias gradient: column sums of delta
#pragma unroll 8
            for (int k = 0; k < NN_KC; ++k) bsum += As[k * PKR + tid];
        }
        __syncthreads();
    }
    double *gp = nn.gpart + ((size_t)b * nn.nmch + tl.chunk) * nn.NP;
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
        const int j = j0 + wc * 32 + bj * 16 + (lane & 15);
        if (j >= sn) continue;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wr * 32 + bi * 16 + (lane >> 4) + 4 * r;
                if (i < sn1) gp[tl.woff + (size_t)i * sn + j] = acc[bi][bj][r];
            }
    }
    if (j0 == 0 && tid < NN_TILE && i0 + tid < sn1) gp[tl.boff + i0 + tid] = bsum;
}

// ------------------------------------------------------------------ K4: parameter tail of grad A
__global__ __launch_bounds__(NN_THREADS) void k_nnet_pred(const Dev dv, const NnetDev nn)
{
    __shared__ double red[16];
    const int b = blockIdx.y, tid = threadIdx.x, j = blockIdx.x * NN_THREADS + tid;
    int use_d; double stp, rf;
    if (!seed_live(dv, b, use_d, stp, rf)) return;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (j < nn.NP) {
        const int k = nn.pmap[j];
        if (k >= 0) {
            const double *gp = nn.gpart + (size_t)b * nn.nmch * nn.NP + j;
            double g = 0.0;
            for (int c = 0; c < nn.nmch; ++c) g += gp[(size_t)c * nn.NP];
            const size_t i = (size_t)b * dv.dm.ld + nn.NDens + k;
            dv.gt[i] = g;
            if (use_d) v[1] = g * dv.d[i];
            v[2] = g * g;
            v[3] = fabs(g);
        }
    }
    wg_reduce4(v, red, tid);
    if (tid == 0) put_row(dv, nn, b, nn.n1 + nn.n2 + blockIdx.x, 0.0, 0.0, v[1], v[2], v[3]);
}

// ------------------------------------------------------------------ K5: fold the partial rows
// NN_RED_ROWS workgroups of one wave per seed: workgroup w sums raw rows w, w+32, ... in a
// fixed order (lane = (row phase, column)), so k_ls / k_finalize_eval see 32 rows per seed.
__global__ __launch_bounds__(64) void k_nnet_rows(const Dev dv, const NnetDev nn)
{
    const int b = blockIdx.y, w = blockIdx.x, lane = threadIdx.x;
    int use_d; double stp, rf;
    if (!seed_live(dv, b, use_d, stp, rf)) return;
    const int c = lane & 7, ph = lane >> 3;                      // 8 columns (5 used) x 8 row phases
    const double *raw = nn.raw + (size_t)b * nn.nraw * EP_GP;
    double v = 0.0;
    if (c < EP_GP)
        for (int r = w + NN_RED_ROWS * ph; r < nn.nraw; r += NN_RED_ROWS * 8) {
            const double x = raw[(size_t)r * EP_GP + c];
            v = (c == EP_GMAX) ? fmax(v, x) : v + x;
        }
#pragma unroll
    for (int o = 32; o >= 8; o >>= 1) {
        const double x = __shfl_down(v, o, 64);
        v = (c == EP_GMAX) ? fmax(v, x) : v + x;
    }
    if (lane < EP_GP) dv.evp[((size_t)b * dv.dm.nprow + w) * EP_N + lane] = v;
}

#endif  // VA_NNET_ACT_ONLY

// ------------------------------------------------------------------ small networks: one kernel
// When every layer is at most 32 wide and there are at most 32 examples (the reference's twin
// example: 20 x 10 neurons, M = 2), the six launches above are pure launch latency.  Here one
// workgroup OWNS layer n of one seed and produces everything indexed by n:
//     q_n     = 2 RF c (x_n - act(x_{n-1} W_{n-1}^T + b_{n-1}))      (transition n-1, recomputed)
//     delta_n = -q_{n+1} act'(...) from transition n                 (its residual feeds fe)
//     dA/dx_n = q_n + delta_n W_n + measurement term
//     dA/dW_n = delta_n^T x_n,  dA/db_n = sum_m delta_n  -> scattered through pmap
// so no workgroup waits for another, nothing round-trips through HBM, and the trial point is
// formed on load.  Each transition's forward product is computed twice (by the owners of its two
// ends): 4 instead of 3 tiny products per layer.  Waves = the 2 x 2 MFMA blocks of the 32x32 tile.
constexpr int PS = NN_SMALL + 4;   // LDS pitch 36
#ifndef NN_SMALL_WAVES
#define NN_SMALL_WAVES 5
#endif

// acc[16x16 block (br, bc)] = A[. x K] B[K x .], K <= 4*nk; *_T: operand stored transposed in LDS
template <bool A_T, bool B_T>
__device__ __forceinline__ d4 mma32(const double *As, const double *Bs, int br, int bc, int lane, int nk)
{
    const int lo = lane & 15, hi = lane >> 4;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < nk; ++kk) {
        const int k = 4 * kk + hi, r = br * 16 + lo, c = bc * 16 + lo;
        const double a = A_T ? As[k * PS + r] : As[r * PS + k];
        const double b = B_T ? Bs[c * PS + k] : Bs[k * PS + c];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// LDS: five operand arrays of R rows (R = 16 or 32: the largest of M and the layer widths rounded
// up to the MFMA block) x pitch 36, two bias rows, reduction scratch
inline size_t nnet_small_lds(int R) { return sizeof(double) * ((size_t)5 * R * PS + 2 * NN_SMALL + 20); }

// (5 waves per SIMD -- 1280 workgroups of the twin example = one resident round -- at 96 registers: the evaluation
// itself needs 52; the folded tail, run by ONE wave per seed,
// is what the allocator would otherwise size the whole kernel for (113), and spills a little instead.  Measured,
// complete evaluation of the twin: 12.8 us; 13.1 at 8 waves / 64 registers; 13.7 with a separate tail launch)
template <class ACT, bool RMM>
__global__ __launch_bounds__(NN_THREADS, NN_SMALL_WAVES) void k_nnet_small(const Dev dv, const NnetDev nn)
{
    extern __shared__ double sm[];
    const int R = nn.small;                       // rows staged per array
    double *Xp = sm, *Xc = sm + R * PS, *Wp = sm + 2 * R * PS, *Wc = sm + 3 * R * PS, *Dl = sm + 4 * R * PS;
    double *bp = sm + 5 * R * PS, *bc = bp + NN_SMALL, *red = bc + NN_SMALL;
    const int b = blockIdx.y, n = blockIdx.x, tid = threadIdx.x;
    int use_d; double stp, rf;
    if (!seed_live(dv, b, use_d, stp, rf)) return;
    const int NL = nn.NL, M = nn.M;
    const bool has_prev = n > 0, has_next = n < NL - 1;
    const int sn = nn.s[n], sp = has_prev ? nn.s[n - 1] : 0, sx = has_next ? nn.s[n + 1] : 0;
    const int offc = nn.off[n], offp = has_prev ? nn.off[n - 1] : 0, offx = has_next ? nn.off[n + 1] : 0;
    const int wofp = has_prev ? nn.woff[n - 1] : 0, bofp = has_prev ? nn.boff[n - 1] : 0;
    const int wofc = has_next ? nn.woff[n] : 0, bofc = has_next ? nn.boff[n] : 0;
    const size_t vo = (size_t)b * dv.dm.ld;
    const double *x = dv.x + vo, *d = dv.d + vo;
    const double *Pf = nn.Pfix + (size_t)b * nn.NP;
    const int lane = tid & 63, wave = tid >> 6, br = wave >> 1, bcol = wave & 1;

    auto xval = [&](int idx) { double v = x[idx]; if (use_d) v = trial(v, stp, d[idx]); return v; };
    auto pval = [&](int j) {
        const int k = nn.pmap[j];
        return k >= 0 ? xval(nn.NDens + k) : Pf[j];
    };
    // ---- stage the operands (zero padded to 32 x 32): 4 elements per thread per array
    for (int e = tid; e < R * 32; e += NN_THREADS) {
        const int r = e >> 5, k = e & 31;
        Xc[r * PS + k] = (r < M && k < sn) ? xval(r * nn.NDnet + offc + k) : 0.0;
        Xp[r * PS + k] = (has_prev && r < M && k < sp) ? xval(r * nn.NDnet + offp + k) : 0.0;
        Wp[r * PS + k] = (has_prev && r < sn && k < sp) ? pval(wofp + r * sp + k) : 0.0;     // W_{n-1}[i][k]
        Wc[r * PS + k] = (has_next && r < sx && k < sn) ? pval(wofc + r * sn + k) : 0.0;     // W_n[i][k]
    }
    if (tid < NN_SMALL) {
        bp[tid] = (has_prev && tid < sn) ? pval(bofp + tid) : 0.0;
        bc[tid] = (has_next && tid < sx) ? pval(bofc + tid) : 0.0;
    }
    // x_{n+1} is only needed element-wise, in the accumulator layout of this wave's block
    const int col = bcol * 16 + (lane & 15);
    double xn1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = br * 16 + (lane >> 4) + 4 * r;
        xn1[r] = (has_next && m < M && col < sx) ? xval(m * nn.NDnet + offx + col) : 0.0;
    }
    __syncthreads();

    const double cq = 2.0 * rf * dv.dm.cfe;
    double v[4] = {0.0, 0.0, 0.0, 0.0};          // me (+ fe in vfe), g.d, g.g, max|g|
    double vfe = 0.0;
    // ---- transition n-1 -> q_n in this wave's block of [m][j of layer n]
    double q[4] = {0.0, 0.0, 0.0, 0.0};
    const bool rows_m = br * 16 < M;              // this wave's block rows hold examples
    if (has_prev && rows_m && bcol * 16 < sn) {
        const d4 z = mma32<false, true>(Xp, Wp, br, bcol, lane, (sp + 3) >> 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = br * 16 + (lane >> 4) + 4 * r;
            if (m < M && col < sn) q[r] = cq * (Xc[m * PS + col] - ACT::f(z[r] + bp[col]));
        }
    }
    // ---- transition n -> delta_n in LDS, fe
    if (has_next && br * 16 < R) {
        d4 z = {0.0, 0.0, 0.0, 0.0};
        if (rows_m && bcol * 16 < sx) z = mma32<false, true>(Xc, Wc, br, bcol, lane, (sn + 3) >> 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = br * 16 + (lane >> 4) + 4 * r;
            double dl = 0.0;
            if (m < M && col < sx) {
                const double zz = z[r] + bc[col];
                const double a = ACT::f(zz);
                const double res = xn1[r] - a;
                vfe += res * res;
                dl = -cq * res * ACT::d(zz, a);
            }
            Dl[m * PS + col] = dl;
        }
    }
    __syncthreads();
    // ---- dA/dx_n = q_n + delta_n W_n + measurement term
    {
        d4 gx = {0.0, 0.0, 0.0, 0.0};
        if (has_next && rows_m && bcol * 16 < sn) gx = mma32<false, false>(Dl, Wc, br, bcol, lane, (sx + 3) >> 2);   // B[k=i][c=j] = W_n[i][j]
        int l = -1; double rm = 0.0; const double *dat = nullptr, *R = nullptr; const int *lidx = nullptr; int L = 0;
        if (col < sn) {
            if (n == 0) { l = nn.lmap_in[col]; rm = nn.rm_in; dat = nn.din; L = nn.Lin; R = nn.rmm_in; lidx = nn.lidx_in; }
            else if (n == NL - 1) { l = nn.lmap_out[col]; rm = nn.rm_out; dat = nn.dout; L = nn.Lout; R = nn.rmm_out; lidx = nn.lidx_out; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = br * 16 + (lane >> 4) + 4 * r;
            if (m >= M || col >= sn) continue;
            const int idx = m * nn.NDnet + offc + col;
            double g = gx[r] + q[r];
            if (l >= 0) {
                const double diff = Xc[m * PS + col] - dat[(size_t)m * L + l];
                if (RMM) {
                    double share, deriv;
                    nnet_meas_matrix(R, L, l, diff, [&](int k) { return Xc[m * PS + lidx[k]] - dat[(size_t)m * L + k]; }, share, deriv);
                    v[0] += share;
                    g += dv.dm.cme * deriv;
                } else {
                    v[0] += rm * diff * diff;
                    g += 2.0 * dv.dm.cme * rm * diff;
                }
            }
            dv.gt[vo + idx] = g;
            if (use_d) v[1] += g * d[idx];
            v[2] += g * g;
            v[3] = fmax(v[3], fabs(g));
        }
    }
    // ---- dA/dW_n = delta_n^T x_n (rows i of layer n+1, columns j of layer n), dA/db_n
    if (has_next && nn.NPest > 0) {
        d4 gw = {0.0, 0.0, 0.0, 0.0};
        if (br * 16 < sx && bcol * 16 < sn) gw = mma32<true, false>(Dl, Xc, br, bcol, lane, (M + 3) >> 2);   // A[r=i][k=m], B[k=m][c=j]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = br * 16 + (lane >> 4) + 4 * r;
            if (i >= sx || col >= sn) continue;
            const int k = nn.pmap[wofc + i * sn + col];
            if (k < 0) continue;
            const double g = gw[r];
            dv.gt[vo + nn.NDens + k] = g;
            if (use_d) v[1] += g * d[nn.NDens + k];
            v[2] += g * g;
            v[3] = fmax(v[3], fabs(g));
        }
        if (tid < sx) {
            const int k = nn.pmap[bofc + tid];
            if (k >= 0) {
                double g = 0.0;
                for (int m = 0; m < M; ++m) g += Dl[m * PS + tid];
                dv.gt[vo + nn.NDens + k] = g;
                if (use_d) v[1] += g * d[nn.NDens + k];
                v[2] += g * g;
                v[3] = fmax(v[3], fabs(g));
            }
        }
    }
    vfe = wave_sum(vfe);
    if (lane == 0) red[16 + wave] = vfe;
    wg_reduce4(v, red, tid);                       // (contains the barrier that publishes red[16..19])
    if (tid == 0) {
        // the layer's partial row, written through: the seed's last workgroup reads the rows in this launch
        double *row = dv.evp + ((size_t)b * dv.dm.nprow + n) * EP_N;
        st_sc1(row + EP_ME, v[0]); st_sc1(row + EP_FE, red[16] + red[17] + red[18] + red[19]);
        st_sc1(row + EP_GTD, v[1]); st_sc1(row + EP_GN2, v[2]); st_sc1(row + EP_GMAX, v[3]);
    }
    // the workgroup that completes the seed's NL rows forms A / runs the line-search step (va_epilogue.h):
    // a small network's evaluation -- and its L-BFGS cycle's first launch -- is this one kernel
#ifndef VA_NN_NOFOLD
    if (dv.epi == EPI_NONE || wave != 0) return;
    if (arrive_last(dv.cnt_eval + (size_t)b * CNT_STRIDE, (unsigned)NL, lane))
        eval_epilogue<true>(dv, b, lane, reinterpret_cast<SeedHot *>(sm), dv.epi);
#endif
}


// the two launches that depend on the activation type
template <class ACT>
inline hipError_t prepare_nnet_fb_act()
{
    return hipFuncSetAttribute((const void *)k_nnet_fb<ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// which: 0 k_nnet_fwd, 1 k_nnet_small, 2 k_nnet_fb, 3 prepare k_nnet_fb (opt in to its LDS on the current device)
template <class ACT>
inline void launch_nnet_act(const Dev &dv, const NnetDev &nn, hipStream_t s, int which)
{
    const int B = dv.dm.B;
    const bool small = which == 1;
    if (which == 3) { (void)prepare_nnet_fb_act<ACT>(); return; }
    if (which == 2) {
        hipLaunchKernelGGL(k_nnet_fb<ACT>, dim3(nn.nfb, B), dim3(NN_FB_THREADS), nnet_fb_lds(), s, dv, nn);
        return;
    }
    if (small) {
        if (nn.rmm_in) hipLaunchKernelGGL((k_nnet_small<ACT, true>), dim3(nn.NL, B), dim3(NN_THREADS), nnet_small_lds(nn.small), s, dv, nn);
        else hipLaunchKernelGGL((k_nnet_small<ACT, false>), dim3(nn.NL, B), dim3(NN_THREADS), nnet_small_lds(nn.small), s, dv, nn);
    } else hipLaunchKernelGGL(k_nnet_fwd<ACT>, dim3(nn.n1, B), dim3(NN_THREADS), 0, s, dv, nn);
}

}  // namespace va
