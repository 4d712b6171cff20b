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

}  // extern "C"
