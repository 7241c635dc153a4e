// rua_reduce_f64.hip — the reduction kernels instantiated for double (see rua_reduce_impl.h).
#include "rua_reduce_impl.h"
RUA_DEFINE_REDUCE_DTYPE(f64, double)
