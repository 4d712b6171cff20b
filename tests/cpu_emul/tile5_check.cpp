// tile5_check.cpp -- CPU check of the streaming kernel's index arithmetic (csrc/va_tile5.h, va_eval5.h): for a state
// width D and column-form reaches, walk every strip and every lane exactly as k_eval5 computes them -- the staged
// image's source columns, the lane's own column and its stencil neighbours inside a staged row, the product-array
// slots of the gather, the packed store lanes -- and verify them against the plain definition (cyclic columns).
// Test infrastructure (tests/test_tile5_geometry.py); prints "OK <D> ..." or the first mismatch.
#include <cstdio>
#include <cstdlib>
#include <vector>
#define VA_HD inline
#include "va_tile5.h"
using namespace va;

static int fail(const char *what, int D, int s, int lane, long a, long b)
{
    std::printf("MISMATCH %s D=%d strip=%d lane=%d got=%ld want=%ld\n", what, D, s, lane, a, b);
    return 1;
}

int main(int argc, char **argv)
{
    const int xl = 2, xr = 1, gl = 1, gr = 2;                         // Lorenz-96's column form
    const int nb_off[3] = {-2, -1, 1}, g_off[3] = {1, -1, 2};
    int rc = 0;
    for (int a = 1; a < argc && !rc; ++a) {
        const int D = std::atoi(argv[a]);
        if (!tile5_ok(D, xl, xr, gl, gr)) { std::printf("NOTOK %d\n", D); continue; }
        const Geo5 g = tile5_cols(D, xl, xr, gl, gr);
        // strips tile [0, D), start on multiples of 8, are at most T5_CW_MAX wide and even
        int covered = 0;
        for (int s = 0; s < g.NS && !rc; ++s) {
            const int c0 = tile5_c0(D, g.NS, s), c1 = tile5_c0(D, g.NS, s + 1), cws = c1 - c0;
            if (c0 != covered) rc = fail("strip start", D, s, 0, c0, covered);
            if (c0 % 8) rc = fail("strip alignment", D, s, 0, c0 % 8, 0);
            if (cws < 2 || cws > g.CW || cws > T5_CW_MAX || (cws & 1)) rc = fail("strip width", D, s, 0, cws, g.CW);
            covered = c1;
            // one staged row: piece pc of the image <- source columns (wrapped), as the kernel's per-lane xoff
            std::vector<int> image(2 * g.PR, -1);
            const int need = g.XL + cws + ((gr + xr + 1) & ~1);
            for (int lane = 0; lane < 64; ++lane) {
                const int rr = lane >= g.PR ? 1 : 0, pc = lane - rr * g.PR;
                const bool on = lane < 2 * g.PR && 2 * pc < need;
                if (!on || rr) continue;
                const int scol = t5_wrap(c0 - g.XL + 2 * pc, D);
                if (scol < 0 || scol >= D) rc = fail("source column", D, s, lane, scol, 0);
                image[2 * pc] = scol; image[2 * pc + 1] = scol + 1;   // a 16-byte piece never straddles the wrap (D, c0, XL even)
                if (scol + 1 >= D) rc = fail("piece straddles the row end", D, s, lane, scol, D);
            }
            // every lane whose products are needed reads its own column and the stencil's neighbours from staged entries
            for (int lane = g.GL - gl; lane < g.GL + cws + gr && !rc; ++lane) {
                const int col = t5_wrap(c0 - g.GL + lane, D), xlane = g.XL - g.GL + lane;
                if (xlane < 0 || xlane >= 2 * g.PR || image[xlane] != col) rc = fail("own column", D, s, lane, xlane < 2 * g.PR && xlane >= 0 ? image[xlane] : -9, col);
                for (int k = 0; k < 3 && !rc; ++k) {
                    const int want = ((col + nb_off[k]) % D + D) % D, at = xlane + nb_off[k];
                    if (at < 0 || at >= 2 * g.PR || image[at] != want) rc = fail("neighbour column", D, s, lane, at >= 0 && at < 2 * g.PR ? image[at] : -9, want);
                }
            }
            if (g.GL + cws + gr > 64) rc = fail("lanes", D, s, 0, g.GL + cws + gr, 64);
            // gather: an owned lane's senders are lanes of this wave whose products were computed
            for (int lane = g.GL; lane < g.GL + cws && !rc; ++lane)
                for (int k = 0; k < 3; ++k) {
                    const int sender = lane + g_off[k];
                    if (sender < g.GL - gl || sender >= g.GL + cws + gr) rc = fail("sender lane", D, s, lane, sender, 0);
                    if (g.PL + sender < 0 || g.PL + sender >= g.PW) rc = fail("product slot", D, s, lane, g.PL + sender, g.PW);
                }
            // packed stores: lane -> (row of the pair, column pair) covers the strip's columns exactly once per row
            const int hp = cws >> 1;
            std::vector<int> hit(2 * cws, 0);
            for (int lane = 0; lane < 64; ++lane) {
                if (lane >= 2 * hp) continue;
                const int srow = lane >= hp ? 1 : 0, spc = lane - srow * hp;
                hit[srow * cws + 2 * spc]++; hit[srow * cws + 2 * spc + 1]++;
                if (g.GL + 2 * spc + 1 >= 64) rc = fail("store read lane", D, s, lane, g.GL + 2 * spc + 1, 64);
            }
            for (int e = 0; e < 2 * cws && !rc; ++e) if (hit[e] != 1) rc = fail("store coverage", D, s, e, hit[e], 1);
            if (2 * hp > 64) rc = fail("store lanes", D, s, 0, 2 * hp, 64);
        }
        if (!rc && covered != D) rc = fail("coverage", D, g.NS, 0, covered, D);
        if (!rc) std::printf("OK %d NS=%d CW=%d PR=%d NACT=%d WPG=%d NSG=%d\n", D, g.NS, g.CW, g.PR, g.NACT, g.WPG, g.NSG);
    }
    return rc;
}
