// va_eval5.hip -- instantiations of the streaming evaluation kernel (va_eval5.h) for the built-in Lorenz-96
// (examples/Lorenz96_D20/Lorenz96_anneal.py:15-16 at the widths of BASELINE config 4).  A translation unit of
// its own so that the library's units compile side by side.
#include "va_device.h"
#include "va_eval5.h"

namespace va {

size_t eval5_lds(const Dev &dv)
{
    return eval5_lds_bytes(dv, dv.lsrun ? dv.g5.nslot_ls : dv.g5.nslot, dv.lsrun != 0, RhsL96s::NE);
}

// ring depth and "line-search points possible" are template parameters; D = 200 (BASELINE config 4) is
// compiled with the column geometry as constants
template <class RHS, int DISC, int DC>
static hipError_t eval5_slots(const Dev &dv, bool prepare, hipStream_t s)
{
    const int threads = 64 * dv.g5.WPG;
    if (prepare) {
        hipError_t err = hipSuccess;
        Dev t = dv;
        for (int ls = 0; ls < 2; ++ls) {          // both launch kinds of the handle
            t.lsrun = ls;
            if (eval5_lds(t) <= 64 * 1024) continue;
            const int ns = ls ? t.g5.nslot_ls : t.g5.nslot;
            hipError_t e = hipSuccess;
#define VA_E5_ATTR(NSL, LS) e = hipFuncSetAttribute((const void *)k_eval5<RHS, DISC, DC, NSL, LS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
            if (ls) { if (ns == 3) VA_E5_ATTR(3, true); else if (ns == 4) VA_E5_ATTR(4, true); else VA_E5_ATTR(6, true); }
            else { if (ns == 4) VA_E5_ATTR(4, false); else if (ns == 6) VA_E5_ATTR(6, false); else VA_E5_ATTR(8, false); }
#undef VA_E5_ATTR
            if (e != hipSuccess) err = e;
        }
        return err;
    }
    const size_t lds = eval5_lds(dv);
    const int ns = dv.lsrun ? dv.g5.nslot_ls : dv.g5.nslot;
#define VA_E5_GO(NSL, LS) hipLaunchKernelGGL((k_eval5<RHS, DISC, DC, NSL, LS>), dim3(eval_grid(dv.dm)), dim3(threads), lds, s, dv)
    if (dv.lsrun) { if (ns == 3) VA_E5_GO(3, true); else if (ns == 4) VA_E5_GO(4, true); else VA_E5_GO(6, true); }
    else { if (ns == 4) VA_E5_GO(4, false); else if (ns == 6) VA_E5_GO(6, false); else VA_E5_GO(8, false); }
#undef VA_E5_GO
    return hipSuccess;
}
template <class RHS, int DC>
static hipError_t eval5_disc(const Dev &dv, bool prepare, hipStream_t s)
{
    switch (dv.dm.disc) {
    case DISC_EULER: return eval5_slots<RHS, DISC_EULER, DC>(dv, prepare, s);
    case DISC_TRAPEZOID: return eval5_slots<RHS, DISC_TRAPEZOID, DC>(dv, prepare, s);
    default: return eval5_slots<RHS, DISC_FWDMAP, DC>(dv, prepare, s);
    }
}
static hipError_t eval5_d(const Dev &dv, bool prepare, hipStream_t s)
{
    if (dv.dm.D == 200) return eval5_disc<RhsL96s, 200>(dv, prepare, s);
    return eval5_disc<RhsL96s, 0>(dv, prepare, s);
}

void launch_eval5(const Dev &dv, hipStream_t s) { (void)eval5_d(dv, false, s); }
hipError_t prepare_eval5(const Dev &dv) { return eval5_d(dv, true, nullptr); }

}  // namespace va
