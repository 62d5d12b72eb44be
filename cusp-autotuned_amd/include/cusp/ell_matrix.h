// cusp/ell_matrix.h -- cusp::ell_matrix<IndexType, ValueType, MemorySpace> (+ view); see
// cusp/detail/matrices.h for the container and cusp/convert.h for the converting constructors.
#pragma once
#include "detail/matrices.h"
#include "convert.h"
