// apemost_model.hip -- every kernel of ONE likelihood model (pt_kernels.h), for every workgroup shape.
// Compiled once per model: -DAPEMOST_TU_MODEL=0..3 and 8..11 (the variant instantiations with a
// non-default proposal law or swap schedule, pt_device.h kVariantModel).
#include "pt_kernels.h"

#ifndef APEMOST_TU_MODEL
#error "compile with -DAPEMOST_TU_MODEL=<0..3 or 8..11>"
#endif

namespace apemost {
template hipError_t model_dispatch<APEMOST_TU_MODEL>(int, const AnyOp &);
}
