// trace_fptr.hip -- the traversal kernels that call a device filter FUNCTION (FILT == 2, include/embree4/rtcore.h "device filter functions"), compiled on their own at -O1:
// see the MI355_FPTR_TU block of trace.hip for why.  Everything else of trace.hip (the other kernels, the host side) is left out of this translation unit.
#define MI355_FPTR_TU 1
#include "trace.hip"
