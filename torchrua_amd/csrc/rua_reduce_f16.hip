// rua_reduce_f16.hip — the reduction kernels instantiated for __half (see rua_reduce_impl.h).
#include "rua_reduce_impl.h"
RUA_DEFINE_REDUCE_DTYPE(f16, __half)
