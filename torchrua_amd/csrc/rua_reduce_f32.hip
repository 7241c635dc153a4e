// rua_reduce_f32.hip — the reduction kernels instantiated for float (see rua_reduce_impl.h).
#include "rua_reduce_impl.h"
RUA_DEFINE_REDUCE_DTYPE(f32, float)
