// va_persist_geo.h -- sizes and slice geometry of the persistent per-seed ladder kernel (va_persist.h), as plain
// host / device C++: shared by the kernel, by the host (va_capi.hip) and by the CPU check of the geometry
// (tests/cpu_emul/persist_check.cpp, tests/test_persist_geometry.py).
#pragma once
#include "va_core.h"

namespace va {

#ifndef PZ_THREADS_N
#define PZ_THREADS_N 1024
#endif
constexpr int PZ_THREADS = PZ_THREADS_N;
constexpr int PZ_WAVES = PZ_THREADS / 64;
constexpr int PZ_EDGE_ROWS = 3;                 // per workgroup: its last two gradient rows (right neighbour's left halo), its first row
constexpr int PZ_GDO = EP_N + UP_N;             // columns of an exchange row: [0, EP_N) evaluation partials, [EP_N, EP_N + UP_N) update
constexpr int PZ_HALO = PZ_GDO + 1;             //   partials, g.d partial, then PZ_EDGE_ROWS * D halo entries
constexpr unsigned PZ_POLL_LIMIT = 1u << 22;    // polls of one granule before the launch is abandoned (seconds)
constexpr int PZ_NSTAMP = 14;
constexpr int PZ_MREG = 10;                     // history lengths up to this keep the coefficient solve in registers

VA_HD int pz_row_granules(int D) { return PZ_HALO + PZ_EDGE_ROWS * D; }

// columns of the sums part of an exchange row that a cycle can use: evaluation partials, g.d, update partials
VA_HD int pz_sum_cols(int NP, int m) { return EP_GP + NP + 1 + UP_OLD + 4 * m; }

// LDS doubles of one workgroup (T rows per slice, history length m, G workgroups per seed)
VA_HD size_t persist_lds_doubles(int T, int D, int L, int NP, int NPest, int m, int HL, int G)
{
    const size_t RD = (size_t)(T + HL + 1) * D, nvh = RD + NPest;
    return (4 + 2 * (size_t)m) * nvh + 3 * RD + (size_t)PZ_WAVES * EP_N + 2 * (size_t)PZ_HALO + 2 * (size_t)m * m + 3 * MAX_M
           + sizeof(SeedHot) / 8 + 8 + (size_t)G * pz_sum_cols(NP, m)
           + 2 * (size_t)(T + 1) * L + RD + (size_t)(D + RHS_MAX_NP) / 2 + 2;      // the slice's observations, weight rows, column map
}

// slice geometry.  Every slice holds at least two rows; Simpson-Hermite slices start on even rows (an interval's three
// rows then reach one row past the slice: HR = 1).  want_T > 0: that many rows per slice, if admissible; 0: the
// largest slices the LDS holds, i.e. the fewest workgroups -- per cycle the vector work of a workgroup is mostly fixed
// overhead (barriers, LDS round trips), while the all-gather grows with the number of rows polled (measured:
// profiles/r04_persist_sweep.txt).
constexpr size_t PZ_LDS_BYTES = 160 * 1024;
inline bool persist_geometry_ok(int N, int D, int L, int NP, int NPest, int m, int disc, size_t lds_bytes, int max_G, int T)
{
    const bool sh = disc == DISC_SH;
    if (T < 2 || (sh && (T & 1))) return false;
    const int G = (N + T - 1) / T;
    if (G > max_G || N - (G - 1) * T < 2) return false;
    return persist_lds_doubles(T, D, L, NP, NPest, m, sh ? 2 : 1, G) * 8 <= lds_bytes;
}
inline bool persist_geometry(int N, int D, int L, int NP, int NPest, int m, int disc, size_t lds_bytes, int max_G, int want_T, int *G, int *T)
{
    if (want_T > 0) {
        if (!persist_geometry_ok(N, D, L, NP, NPest, m, disc, lds_bytes, max_G, want_T)) return false;
        *T = want_T; *G = (N + want_T - 1) / want_T;
        return true;
    }
    for (int t = N; t >= 2; --t)
        if (persist_geometry_ok(N, D, L, NP, NPest, m, disc, lds_bytes, max_G, t)) { *T = t; *G = (N + t - 1) / t; return true; }
    return false;
}

}  // namespace va
