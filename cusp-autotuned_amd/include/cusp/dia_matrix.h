// cusp/dia_matrix.h -- cusp::dia_matrix<IndexType, ValueType, MemorySpace> (+ view); see
// cusp/detail/matrices.h for the container and cusp/convert.h for the converting constructors.
#pragma once
#include "detail/matrices.h"
#include "convert.h"
