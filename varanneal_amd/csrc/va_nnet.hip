// va_nnet.hip -- launch sequence of the feed-forward-network action (kernels: va_nnet_kernels.h).
#include <algorithm>

#include "va_nnet_kernels.h"

namespace va {

static void launch_act(const Dev &dv, const NnetDev &nn, hipStream_t s, int which, NnetActLaunch user)
{
    if (nn.act >= NNET_USER) { user(&dv, &nn, (void *)s, which); return; }
    switch (nn.act) {
    case NNET_SIGMOID: launch_nnet_act<ActBuiltin<NNET_SIGMOID>>(dv, nn, s, which); break;
    case NNET_TANH: launch_nnet_act<ActBuiltin<NNET_TANH>>(dv, nn, s, which); break;
    case NNET_RELU: launch_nnet_act<ActBuiltin<NNET_RELU>>(dv, nn, s, which); break;
    case NNET_SOFTPLUS: launch_nnet_act<ActBuiltin<NNET_SOFTPLUS>>(dv, nn, s, which); break;
    default: launch_nnet_act<ActBuiltin<NNET_LINEAR>>(dv, nn, s, which); break;
    }
}

hipError_t prepare_nnet_fb(const NnetDev &nn, NnetActLaunch user)
{
    Dev none{};
    launch_act(none, nn, nullptr, 3, user);
    return hipGetLastError();
}

void launch_nnet_eval(const Dev &dv, const NnetDev &nn, hipStream_t s, NnetActLaunch user)
{
    const int B = dv.dm.B;
    const dim3 blk(NN_THREADS);
    if (nn.small) { launch_act(dv, nn, s, 1, user); return; }
    hipLaunchKernelGGL(k_nnet_pack, dim3(nn.n0, B), blk, 0, s, dv, nn);
    if (nn.fused) {
        hipLaunchKernelGGL(k_nnet_wfrag, dim3(std::min(nn.wfsz / NN_THREADS, 256), B), blk, 0, s, dv, nn);
        launch_act(dv, nn, s, 2, user);
    } else {
        launch_act(dv, nn, s, 0, user);
        if (nn.rmm_in) hipLaunchKernelGGL(k_nnet_bwd_x<true>, dim3(nn.n2, B), blk, 0, s, dv, nn);
        else hipLaunchKernelGGL(k_nnet_bwd_x<false>, dim3(nn.n2, B), blk, 0, s, dv, nn);
    }
    if (nn.NPest > 0) {
        hipLaunchKernelGGL(k_nnet_bwd_w, dim3(nn.n3, B), blk, 0, s, dv, nn);
        hipLaunchKernelGGL(k_nnet_pred, dim3(nn.n4, B), blk, 0, s, dv, nn);
    }
    if (nn.raw) hipLaunchKernelGGL(k_nnet_rows, dim3(NN_RED_ROWS, B), dim3(64), 0, s, dv, nn);
}

}  // namespace va
