// cusp/format.h -- format tags (reference cusp/detail/format.h).
#pragma once
namespace cusp {
struct unknown_format {};
struct known_format {};
struct sparse_format : known_format {};
struct dense_format : known_format {};
struct array1d_format : dense_format {};
struct array2d_format : dense_format {};
struct coo_format : sparse_format {};
struct csr_format : sparse_format {};
struct dia_format : sparse_format {};
struct ell_format : sparse_format {};
struct hyb_format : sparse_format {};
struct row_major {};
struct column_major {};
} // namespace cusp
