// cusp/blas.h -- the reference's older include path for cusp::blas (cusp/blas.h); the routines live in cusp/blas/blas.h.
#pragma once
#include "blas/blas.h"
