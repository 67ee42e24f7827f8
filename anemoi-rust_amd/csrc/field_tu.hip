// field_tu.hip -- compiled once per field with -DANEMOI_FIELD_ID=<0..6>: instantiates that field's
// kernels for gfx950 and exports its launcher table.
#include "anemoi_kernels.h"

#ifndef ANEMOI_FIELD_ID
#error "compile with -DANEMOI_FIELD_ID=<field id>"
#endif

#define ANEMOI_CAT2(a, b) a##b
#define ANEMOI_CAT(a, b) ANEMOI_CAT2(a, b)

namespace anemoi {
const FieldOps* ANEMOI_CAT(field_ops_, ANEMOI_FIELD_ID)() { return Launch<ANEMOI_FIELD_ID>::ops(); }
}
