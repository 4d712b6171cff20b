// va_user_act.hip -- translation unit of a GENERATED activation module for the network action.
//
// The reference's network takes any callable f(x, W, b) as the layer map (varanneal/va_nnet.py:71,
// used at :260-264).  varanneal_amd/codegen.py recognises the form g(W.x + b) with an elementwise g, traces
// g on a symbol, differentiates it and writes a header that defines `struct ActUser { f(z); d(z, a); }`.
// This file instantiates the two kernels that apply the activation (va_nnet_kernels.h) for it; compiled with
//     hipcc --offload-arch=gfx950 -shared -DVA_USER_ACT_HEADER='"<header>"' va_user_act.hip
// and registered through va_act_load_module().
#define VA_NNET_ACT_ONLY 1
#include "va_nnet_kernels.h"

#ifndef VA_USER_ACT_HEADER
#error "compile with -DVA_USER_ACT_HEADER='\"path/to/generated_header.h\"'"
#endif
#include VA_USER_ACT_HEADER

extern "C" {

// (sizeof(Dev), sizeof(NnetDev), sizeof(SeedState)) -- checked by va_act_load_module
void va_user_act_info(int *out)
{
    out[0] = (int)sizeof(va::Dev); out[1] = (int)sizeof(va::NnetDev); out[2] = (int)sizeof(va::SeedState);
}

// which: 0 k_nnet_fwd, 1 k_nnet_small, 2 k_nnet_fb, 3 opt k_nnet_fb in to its LDS (va_nnet_kernels.h launch_nnet_act)
void va_user_act_launch(const va::Dev *dv, const va::NnetDev *nn, void *stream, int which)
{
    va::launch_nnet_act<va::ActUser>(*dv, *nn, (hipStream_t)stream, which);
}

}  // extern "C"
