// persist_check.cpp -- CPU check of the persistent ladder kernel's slice geometry (csrc/va_persist_geo.h: the header the
// kernel and the host include).  usage: persist_check N D L NP NPest m disc maxG wantT  ->  "OK G T lds_bytes" | "NO"
// Test infrastructure only (tests/test_persist_geometry.py).
#include <cstdio>
#include <cstdlib>

#include "va_persist_geo.h"

int main(int argc, char **argv)
{
    if (argc != 10) return 2;
    int v[9];
    for (int i = 0; i < 9; ++i) v[i] = atoi(argv[i + 1]);
    const int N = v[0], D = v[1], L = v[2], NP = v[3], NPest = v[4], m = v[5], disc = v[6], maxG = v[7], wantT = v[8];
    int G = 0, T = 0;
    if (!va::persist_geometry(N, D, L, NP, NPest, m, disc, va::PZ_LDS_BYTES, maxG, wantT, &G, &T)) { printf("NO\n"); return 0; }
    const int HL = disc == va::DISC_SH ? 2 : 1;
    printf("OK %d %d %zu\n", G, T, va::persist_lds_doubles(T, D, L, NP, NPest, m, HL, G) * 8);
    return 0;
}
