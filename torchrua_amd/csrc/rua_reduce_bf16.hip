// rua_reduce_bf16.hip — the reduction kernels instantiated for __hip_bfloat16 (see rua_reduce_impl.h).
#include "rua_reduce_impl.h"
RUA_DEFINE_REDUCE_DTYPE(bf16, __hip_bfloat16)
