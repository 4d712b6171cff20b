// va_user_rhs.hip -- translation unit of a GENERATED right-hand-side module.
//
// varanneal_amd/codegen.py traces the user's `f(t, x, p)` (the callable of set_model,
// varanneal/va_ode.py:56-67), differentiates it symbolically and writes a header that
// defines `struct RhsUser` with f, J^T v and (df/dp)^T v.  This file instantiates the
// flat-mapped tile kernel for it; it is compiled with
//     hipcc --offload-arch=gfx950 -shared -DVA_USER_RHS_HEADER='"<header>"' va_user_rhs.hip
// and registered through va_rhs_load_module().  The reference replays an ADOL-C tape of f
// instead (_autodiffmin.py:32-58).
#include "va_eval_flat.h"
#include "va_eval3.h"
#include "va_eval4.h"
#include "va_eval5.h"
#include "va_persist.h"

#ifndef VA_USER_RHS_HEADER
#error "compile with -DVA_USER_RHS_HEADER='\"path/to/generated_header.h\"'"
#endif
#include VA_USER_RHS_HEADER

extern "C" {

// (NP, D, NSTIM, sizeof(Dev), sizeof(SeedState)) -- checked by va_rhs_load_module
void va_user_rhs_info(int *out)
{
    out[0] = va::RhsUser::NP; out[1] = va::RhsUser::D; out[2] = va::RhsUser::NSTIM;
    out[3] = (int)sizeof(va::Dev); out[4] = (int)sizeof(va::SeedState);
}

void va_user_launch_eval(const va::Dev *dv, void *stream)
{
    va::launch_eval_rhs<va::RhsUser>(*dv, (hipStream_t)stream);
}

// once per problem handle, on the handle's device: opt the kernel in to the LDS it needs
int va_user_prepare_eval(const va::Dev *dv)
{
    return (int)va::prepare_eval_rhs<va::RhsUser>(*dv);
}

// the persistent per-seed ladder kernel (va_persist.h) for this model: few seeds, short paths
int va_user_seed_kernel(const va::Dev *dv, int launch, void *stream)
{
    return (int)va::seed_kernel_op<va::RhsUser>(*dv, launch != 0, (hipStream_t)stream);
}

// Besides the flat kernel a module may carry ONE instantiation of a column-run kernel, named when the module
// was generated (va_eval_plan; -DVA_USER_EK=3|4 -DVA_USER_DISC -DVA_USER_K -DVA_USER_W):
//   EK = 4: the model's column form (struct RhsUserCol: a translation-invariant stencil, or a small dense
//           system) on the wave-private kernel k_eval4; W = 1 for scalar weights
//   EK = 5: a stencil's column form on the streaming kernel k_eval5 (wide even states, autonomous)
//   EK = 3: a stencil's ghosted form (struct RhsUserG) on the workgroup kernel k_eval3; W = threads per workgroup
// (eval kernel or 0, DISC, K, W, products per element [4, 5], ghost columns [3], reaches xl, xr, gl, gr [5], -, dense linear part)
#if defined(VA_USER_EK) && VA_USER_EK == 5 && defined(VA_USER_COL)
#define VA_USER_VARIANT 5
#elif defined(VA_USER_EK) && VA_USER_EK == 4 && defined(VA_USER_COL)
#define VA_USER_VARIANT 4
#elif defined(VA_USER_EK) && VA_USER_EK == 3 && defined(VA_USER_GHOST)
#define VA_USER_VARIANT 3
#endif
void va_user_variant_info(int *out)
{
    for (int k = 0; k < 12; ++k) out[k] = 0;
    out[10] = va::rhs_linear<va::RhsUser>::value ? 1 : 0;     // the flat kernel stages one more array (va_eval_flat.h lin_gemm)
#ifdef VA_USER_VARIANT
    out[0] = VA_USER_VARIANT; out[1] = VA_USER_DISC; out[2] = VA_USER_K; out[3] = VA_USER_W;
#if VA_USER_VARIANT == 5
    out[4] = va::RhsUserCol::NE;
    out[6] = va::t5_xl<va::RhsUserCol>(); out[7] = va::t5_xr<va::RhsUserCol>();
    out[8] = va::t5_gl<va::RhsUserCol>(); out[9] = va::t5_gr<va::RhsUserCol>();
#elif VA_USER_VARIANT == 4
    out[4] = va::RhsUserCol::NE;
#else
    out[5] = va::RhsUserG::GHOST;
#endif
#endif
}
#if defined(VA_USER_VARIANT) && VA_USER_VARIANT == 5
void va_user_launch_variant(const va::Dev *dv, void *stream)
{
    (void)va::eval5_run<va::RhsUserCol, VA_USER_DISC, va::RhsUserCol::D>(*dv, false, (hipStream_t)stream);
}
int va_user_prepare_variant(const va::Dev *dv)
{
    return (int)va::eval5_run<va::RhsUserCol, VA_USER_DISC, va::RhsUserCol::D>(*dv, true, nullptr);
}
#elif defined(VA_USER_VARIANT) && VA_USER_VARIANT == 4
void va_user_launch_variant(const va::Dev *dv, void *stream)
{
    va::launch_eval4_one<va::RhsUserCol, VA_USER_DISC, VA_USER_K, va::RhsUserCol::D, VA_USER_W != 0>(*dv, (hipStream_t)stream);
}
int va_user_prepare_variant(const va::Dev *dv)
{
    return (int)va::prepare_eval4_one<va::RhsUserCol, VA_USER_DISC, VA_USER_K, va::RhsUserCol::D, VA_USER_W != 0>(*dv);
}
#elif defined(VA_USER_VARIANT)
void va_user_launch_variant(const va::Dev *dv, void *stream)
{
    va::launch_eval3_one<va::RhsUserG, VA_USER_DISC, VA_USER_K, va::RhsUserG::D, VA_USER_W>(*dv, (hipStream_t)stream);     // (D compiled in, as the built-in's D = 200)
}
int va_user_prepare_variant(const va::Dev *dv)
{
    return (int)va::prepare_eval3_one<va::RhsUserG, VA_USER_DISC, VA_USER_K, va::RhsUserG::D, VA_USER_W>(*dv);
}
#endif

}  // extern "C"
