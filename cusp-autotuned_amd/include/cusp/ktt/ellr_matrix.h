// cusp/ktt/ellr_matrix.h -- cusp::ktt::ellr_matrix (the fork's ELL + row_lengths container,
// reference cusp/ktt/ellr_matrix.h:17-90); defined in cusp/detail/matrices.h.
#pragma once
#include "../detail/matrices.h"
#include "../convert.h"
