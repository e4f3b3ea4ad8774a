// fp64 instantiations of the trace / fused-spot kernels (see trace_kernel.hip)
#define OL_TRACE_TU 2
#include "trace_kernel.hip"
