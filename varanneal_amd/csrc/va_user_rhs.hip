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
#include "va_eval4.h"

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

// The model's column form (struct RhsUserCol: translation-invariant stencil, or a small dense system),
// when the generator found one, on the wave-private column-run kernel -- ONE instantiation, for the
// discretisation / run length / weight kind the module was built for (-DVA_USER_DISC/K/WS):
// (has column form, DISC, K, W_SCALAR, products per element)
#if defined(VA_USER_COL) && defined(VA_USER_K)
#define VA_USER_COL_BUILT 1
#endif
void va_user_col_info(int *out)
{
#ifdef VA_USER_COL_BUILT
    out[0] = 1; out[1] = VA_USER_DISC; out[2] = VA_USER_K; out[3] = VA_USER_WS; out[4] = va::RhsUserCol::NE;
#else
    out[0] = 0; out[1] = out[2] = out[3] = out[4] = 0;
#endif
}
#ifdef VA_USER_COL_BUILT
void va_user_launch_eval4(const va::Dev *dv, void *stream)
{
    va::launch_eval4_one<va::RhsUserCol, VA_USER_DISC, VA_USER_K, va::RhsUserCol::D, VA_USER_WS != 0>(*dv, (hipStream_t)stream);
}
int va_user_prepare_eval4(const va::Dev *dv)
{
    return (int)va::prepare_eval4_one<va::RhsUserCol, VA_USER_DISC, VA_USER_K, va::RhsUserCol::D, VA_USER_WS != 0>(*dv);
}
#endif

}  // extern "C"
