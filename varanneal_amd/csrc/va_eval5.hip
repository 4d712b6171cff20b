// va_eval5.hip -- instantiations of the streaming evaluation kernel (va_eval5.h) for the built-in Lorenz-96
// (examples/Lorenz96_D20/Lorenz96_anneal.py:15-16 at the widths of BASELINE config 4).  A translation unit of
// its own so that the library's units compile side by side.
#include "va_device.h"
#include "va_eval5.h"

namespace va {

size_t eval5_lds(const Dev &dv) { return eval5_lds_bytes(dv); }

// D = 200 (BASELINE config 4) is compiled with the column geometry as constants
template <class RHS, int DC>
static hipError_t eval5_disc(const Dev &dv, bool prepare, hipStream_t s)
{
    switch (dv.dm.disc) {
    case DISC_EULER: return eval5_run<RHS, DISC_EULER, DC>(dv, prepare, s);
    case DISC_TRAPEZOID: return eval5_run<RHS, DISC_TRAPEZOID, DC>(dv, prepare, s);
    case DISC_SH: return eval5_run<RHS, DISC_SH, DC>(dv, prepare, s);
    default: return eval5_run<RHS, DISC_FWDMAP, DC>(dv, prepare, s);
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
