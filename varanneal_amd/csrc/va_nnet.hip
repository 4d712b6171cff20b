// va_nnet.hip -- launch sequence of the feed-forward-network action (kernels: va_nnet_kernels.h).
#include "va_nnet_kernels.h"

namespace va {

static void launch_act(const Dev &dv, const NnetDev &nn, hipStream_t s, bool small, NnetActLaunch user)
{
    if (nn.act >= NNET_USER) { user(&dv, &nn, (void *)s, small ? 1 : 0); return; }
    switch (nn.act) {
    case NNET_SIGMOID: launch_nnet_act<ActBuiltin<NNET_SIGMOID>>(dv, nn, s, small); break;
    case NNET_TANH: launch_nnet_act<ActBuiltin<NNET_TANH>>(dv, nn, s, small); break;
    case NNET_RELU: launch_nnet_act<ActBuiltin<NNET_RELU>>(dv, nn, s, small); break;
    case NNET_SOFTPLUS: launch_nnet_act<ActBuiltin<NNET_SOFTPLUS>>(dv, nn, s, small); break;
    default: launch_nnet_act<ActBuiltin<NNET_LINEAR>>(dv, nn, s, small); break;
    }
}

void launch_nnet_eval(const Dev &dv, const NnetDev &nn, hipStream_t s, NnetActLaunch user)
{
    const int B = dv.dm.B;
    const dim3 blk(NN_THREADS);
    if (nn.small) { launch_act(dv, nn, s, true, user); return; }
    hipLaunchKernelGGL(k_nnet_pack, dim3(nn.n0, B), blk, 0, s, dv, nn);
    launch_act(dv, nn, s, false, user);
    if (nn.rmm_in) hipLaunchKernelGGL(k_nnet_bwd_x<true>, dim3(nn.n2, B), blk, 0, s, dv, nn);
    else hipLaunchKernelGGL(k_nnet_bwd_x<false>, dim3(nn.n2, B), blk, 0, s, dv, nn);
    if (nn.NPest > 0) {
        hipLaunchKernelGGL(k_nnet_bwd_w, dim3(nn.n3, B), blk, 0, s, dv, nn);
        hipLaunchKernelGGL(k_nnet_pred, dim3(nn.n4, B), blk, 0, s, dv, nn);
    }
    if (nn.raw) hipLaunchKernelGGL(k_nnet_rows, dim3(NN_RED_ROWS, B), dim3(64), 0, s, dv, nn);
}

}  // namespace va
