/* va_client.c -- a host program in plain C against include/varanneal_amd.h: no Python, no torch, nothing but the
 * C-ABI (the boundary a maintainer of the reference would bind: INTEGRATION.md).  Reads a problem from a binary file
 * written by tests/test_gpu_cabi.py, runs S1 (va_action_grad), S2 (va_minimize_lbfgs) and S3 (va_anneal), writes the
 * results to a binary file.  Layout of both files: little-endian int32 / float64 arrays in the order read / written below.
 *
 *   gcc -O1 -I include -o va_client tests/c_client/va_client.c -L varanneal_amd -lvaranneal_amd -Wl,-rpath,$PWD/varanneal_amd
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "varanneal_amd.h"

static void *rd(FILE *f, size_t n, size_t sz)
{
    void *p = malloc(n * sz + 8);
    if (!p || fread(p, sz, n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
    return p;
}
#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, va_last_error()); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: va_client problem.bin results.bin\n"); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t *hd = rd(f, 8, sizeof(int32_t));        /* B, D, N, L, disc, nbeta, maxiter, NPest */
    const int B = hd[0], D = hd[1], N = hd[2], L = hd[3], disc = hd[4], nbeta = hd[5], maxiter = hd[6], NPest = hd[7];
    double *sc = rd(f, 5, sizeof(double));          /* dt, rm, rf0, rf_scale (S1 / S2), alpha */
    int32_t *Lidx = rd(f, (size_t)L, sizeof(int32_t));
    double *Y = rd(f, (size_t)N * L, sizeof(double));
    double *P = rd(f, (size_t)B, sizeof(double));   /* NP = 1 */
    const size_t nv = (size_t)N * D + NPest;
    double *XP = rd(f, (size_t)B * nv, sizeof(double));
    fclose(f);

    if (va_abi_version() != VA_ABI_VERSION) { fprintf(stderr, "ABI %d != header %d\n", va_abi_version(), VA_ABI_VERSION); return 2; }
    va_problem_desc d;
    memset(&d, 0, sizeof d);
    d.struct_size = (int32_t)sizeof d;
    d.device = 0; d.batch = B; d.D = D; d.N_model = N; d.N_data = N; d.merr_nskip = 1; d.L = L; d.Lidx = Lidx; d.Y = Y;
    d.dt_model = sc[0]; d.rm = sc[1]; d.rf0 = sc[2];
    int32_t pidx[1] = {0};
    d.NP = 1; d.NPest = NPest; d.Pidx = pidx; d.P = P;
    d.disc = disc; d.rhs = VA_RHS_LORENZ96; d.lbfgs_m = 10; d.max_beta = nbeta; d.keep_paths = 0;
    va_handle h = NULL;
    CHECK(va_problem_create(&d, &h));
    int32_t ek = 0, rr = 0;
    CHECK(va_problem_eval_kernel(h, &ek, &rr));

    /* S1 */
    double *A = malloc(sizeof(double) * B * 3), *g = malloc(sizeof(double) * B * nv);
    CHECK(va_action_grad(h, XP, (int64_t)nv, VA_MEM_HOST, sc[3], A, A + B, A + 2 * B, g, (int64_t)nv));
    /* S2 on a copy */
    double *X2 = malloc(sizeof(double) * B * nv);
    memcpy(X2, XP, sizeof(double) * B * nv);
    va_lbfgs_opts o;
    memset(&o, 0, sizeof o);
    o.maxcor = 10; o.ftol = 1e-8; o.gtol = 1e-8; o.maxiter = maxiter; o.maxfun = 1000000; o.maxls = 20;
    double *A2 = malloc(sizeof(double) * B * 3);
    int32_t *st = malloc(sizeof(int32_t) * B), *nit = malloc(sizeof(int32_t) * B);
    int64_t *nfev = malloc(sizeof(int64_t) * B);
    CHECK(va_minimize_lbfgs(h, X2, (int64_t)nv, VA_MEM_HOST, sc[3], &o, A2, A2 + B, A2 + 2 * B, st, nit, nfev));
    /* S3 on another copy */
    double *X3 = malloc(sizeof(double) * B * nv), *rf = malloc(sizeof(double) * nbeta);
    memcpy(X3, XP, sizeof(double) * B * nv);
    rf[0] = 1.0;
    for (int k = 1; k < nbeta; ++k) rf[k] = rf[k - 1] * sc[4];
    double *ame = malloc(sizeof(double) * B * nbeta * 3), *pest = malloc(sizeof(double) * B * nbeta * (NPest ? NPest : 1));
    int32_t *st3 = malloc(sizeof(int32_t) * B * nbeta), *nit3 = malloc(sizeof(int32_t) * B * nbeta);
    int64_t *nf3 = malloc(sizeof(int64_t) * B * nbeta);
    CHECK(va_anneal(h, X3, (int64_t)nv, VA_MEM_HOST, rf, nbeta, &o, ame, pest, st3, nit3, nf3, NULL));
    va_problem_destroy(h);

    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fwrite(&ek, sizeof ek, 1, f);
    fwrite(A, sizeof(double), (size_t)B * 3, f);
    fwrite(g, sizeof(double), (size_t)B * nv, f);
    fwrite(A2, sizeof(double), (size_t)B * 3, f);
    fwrite(X2, sizeof(double), (size_t)B * nv, f);
    fwrite(nit, sizeof(int32_t), (size_t)B, f);
    fwrite(nfev, sizeof(int64_t), (size_t)B, f);
    fwrite(st, sizeof(int32_t), (size_t)B, f);
    fwrite(ame, sizeof(double), (size_t)B * nbeta * 3, f);
    fwrite(pest, sizeof(double), (size_t)B * nbeta * NPest, f);
    fwrite(nit3, sizeof(int32_t), (size_t)B * nbeta, f);
    fclose(f);
    printf("va_client: eval kernel %d, A[0] = %.15g, S2 nit[0] = %d, ladder A[0][last] = %.15g\n", ek, A[0], nit[0], ame[(nbeta - 1) * 3]);
    return 0;
}
