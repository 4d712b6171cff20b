// va_tile5.h -- geometry of the STREAMING evaluation kernel k_eval5 (va_eval5.h): wide states (D > 64,
// BASELINE config 4: D = 200) as column strips that march along the time axis.
//
// A wave owns CW state columns of one seed (a strip) over SEGL consecutive time rows (a segment) and walks
// the segment row by row; lane = column.  What the one-step discretisations (va_ode.py:341-380, 439-454)
// need from the previous row -- x, the neighbour values, f, and the adjoint weight q of the previous
// residual -- stays in registers, so every row of x is read from memory ONCE per strip and no halo row is
// recomputed (the tile kernels stage T + 2 rows for T).  Rows arrive through a ring of direct-to-LDS loads
// (two rows per 1-KiB wave instruction) that runs NSLOT - 1 slots ahead of the arithmetic; the observations
// and, at a line-search point, the direction d arrive the same way.  Nothing is shared between waves until the
// partial sums at the very end: no workgroup barrier in the data path.
//
// Columns.  The adjoint is in scatter form (va_tile4.h): lane j publishes s_j * df_j/dx_k products and gathers
// the ones addressed to its column.  With f reading columns j + nb_off and the gather reading senders
// j + g_off, a strip that OWNS columns [c0, c0 + CW) computes columns [c0 - gl, c0 + CW + gr) and stages
// x for [c0 - XL, c0 + CW + XR) (cyclic).  Lorenz-96: gl = 1, gr = 2, image 4 + CW + 4 columns.
//
// Shared with the host (geometry, LDS sizes) and with tests/cpu_emul (index arithmetic).
#pragma once
#include "va_tile4.h"

namespace va {

constexpr int T5_CW_MAX = 56;     // owned columns per strip: 56 + 8 image ghosts = 32 pieces = half a wave instruction
constexpr int T5_WPG_MAX = 4;     // strips (waves) per workgroup
constexpr int T5_MAX_STRIPS = 64;

struct Geo5 {
    int D, NS, CW;        // strips per row (tile5_c0), owned columns of the widest strip
    int NSG, WPG;         // workgroups per row of strips, waves per workgroup
    int GL;               // lane of a strip's first owned column (even: owned column pairs are lane pairs)
    int XL;               // image columns left of the first owned one (even)
    int PR;               // 16-byte pieces per staged row; a slot = 2 rows = 2*PR pieces <= 64
    int NACT;             // lanes whose products are needed
    int PW;               // doubles per product array (64 lanes + gather reach on both sides)
    int PL;               // product-array slot of lane 0
    int SEGL, NSEG;       // time rows per segment, segments per seed
    int YPMAX;            // 16-byte pieces per observation row of the widest strip
    int nslot, nslot_ls;  // ring slots: plain evaluation / launches that may hold line-search points
    int ne;               // products per element of the model's column form (LDS arrays of the non-DPP exchange)
    int xdpp;             // scatter products change lanes through DPP shifts (gather reach <= 2) instead of LDS
    int LY;               // row pitch of the observation (and RM) rows on the device: L rounded up to even
    int warr;             // the rows' weights come through the ring too: (N-1, D) RF / (N_data, L) RM arrays, data every nskip-th row
};

template <class RHS> VA_HD constexpr int t5_xl()
{
    int m = 0;
    for (int k = 0; k < RHS::NB; ++k) m = -RHS::nb_off(k) > m ? -RHS::nb_off(k) : m;
    return m;
}
template <class RHS> VA_HD constexpr int t5_xr()
{
    int m = 0;
    for (int k = 0; k < RHS::NB; ++k) m = RHS::nb_off(k) > m ? RHS::nb_off(k) : m;
    return m;
}
template <class RHS> VA_HD constexpr int t5_gl()
{
    int m = 0;
    for (int k = 0; k < RHS::NG; ++k) m = -RHS::g_off(k) > m ? -RHS::g_off(k) : m;
    return m;
}
template <class RHS> VA_HD constexpr int t5_gr()
{
    int m = 0;
    for (int k = 0; k < RHS::NG; ++k) m = RHS::g_off(k) > m ? RHS::g_off(k) : m;
    return m;
}

// first column of strip s5 (s5 = NS: D).  Strips start on multiples of 8 columns = 64 bytes, so that no 64-byte
// sector of a row of x or of the gradient is shared between two waves; widths differ by at most 8 columns
// (the last strip also takes D mod 8).
VA_HD constexpr int tile5_c0(int D, int NS, int s5)
{
    return s5 >= NS ? D : 8 * ((s5 * (D / 8)) / NS);
}
VA_HD constexpr int tile5_maxw(int D, int NS)
{
    int m = 0;
    for (int s5 = 0; s5 < NS; ++s5) {
        const int w = tile5_c0(D, NS, s5 + 1) - tile5_c0(D, NS, s5);
        m = w > m ? w : m;
    }
    return m;
}

// column geometry from the state width and the reaches of the model's column form
VA_HD constexpr Geo5 tile5_cols(int D, int xl, int xr, int gl, int gr)
{
    Geo5 g{};
    g.D = D;
    g.NS = (D + T5_CW_MAX - 1) / T5_CW_MAX;
    while (g.NS < T5_MAX_STRIPS && tile5_maxw(D, g.NS) > T5_CW_MAX) ++g.NS;
    g.CW = tile5_maxw(D, g.NS);                 // the widest strip
    g.WPG = g.NS < T5_WPG_MAX ? g.NS : T5_WPG_MAX;
    g.NSG = (g.NS + g.WPG - 1) / g.WPG;
    g.GL = (gl + 1) & ~1;
    g.XL = (g.GL + xl + 1) & ~1;
    const int xr_cols = (gr + xr + 1) & ~1;
    g.PR = (g.XL + g.CW + xr_cols) / 2;
    g.NACT = g.GL + g.CW + gr;
    g.PL = gl > 2 ? gl : 2;
    g.PW = ((g.PL + 64 + gr + 1) & ~1);
    return g;
}
template <class RHS> VA_HD constexpr Geo5 tile5_cols_rhs(int D)
{
    return tile5_cols(D, t5_xl<RHS>(), t5_xr<RHS>(), t5_gl<RHS>(), t5_gr<RHS>());
}
// can this state width run the streaming kernel with a column form of these reaches?
VA_HD constexpr bool tile5_ok(int D, int xl, int xr, int gl, int gr)
{
    if (D <= 64 || (D & 1)) return false;
    const Geo5 g = tile5_cols(D, xl, xr, gl, gr);
    return g.PR <= 32 && g.NACT <= 64 && g.NS <= T5_MAX_STRIPS && 2 * g.PR <= D;
}

VA_HD constexpr int t5_wrap(int c, int D) { return c < 0 ? c + D : (c >= D ? c - D : c); }

// doubles of LDS one wave needs: x ring (+ d ring), observation ring, product arrays (none when the products change
// lanes through DPP shifts).  The reduction strip
// and the tail's copy of the seed state re-use the rings after the walk.
VA_HD constexpr int tile5_wave_doubles(const Geo5 &g, int nslot, bool ls, bool warr = false)
{
    const int ne = g.ne;
    const int slotx = 4 * g.PR, sloty = 4 * g.YPMAX;
    const int rings = nslot * (slotx * ((ls ? 2 : 1) + (warr ? 1 : 0)) + sloty * (warr ? 2 : 1)) + (g.xdpp ? 0 : ne * g.PW) + 128;     // (+ the gradient rows of a slot on their way out)
    const int minimum = T4_STRIP + 64;        // reduction strip + SeedHot copy (512 B)
    return ((rings > minimum ? rings : minimum) + 15) & ~15;
}

}  // namespace va
