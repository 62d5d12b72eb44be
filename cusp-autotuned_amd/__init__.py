"""cusp-autotuned_amd -- MI355X-native SpMV engine behind the CUSP API.

The product is the C-ABI shared library ``lib/libcusp_mi355x.so`` (hand-written gfx950 HIP kernels,
``include/cusp_mi355x.h``) and the header-only C++ layer ``include/cusp/`` that forwards
``cusp::multiply`` to it.  This Python package is plumbing for tests and ``bench.py``: a ctypes
binding of that C-ABI that takes torch tensors as device memory (``data_ptr()``) and torch's current
HIP stream.  There is NO CPU fallback: if the library is missing or was not built, importing the
binding raises.

Import name: the directory name contains a hyphen, so the repo root ships a tiny loader module
``cusp_autotuned_amd.py``; ``import cusp_autotuned_amd`` gives this package.
"""
from .binding import (  # noqa: F401
    CmiError,
    Config,
    FORMAT_CSR, FORMAT_ELL, FORMAT_DIA, FORMAT_COO, FORMAT_HYB, TABLE_COO_SORTED,
    F64, F32,
    KERNEL_AUTO, CSR_SCALAR, CSR_VECTOR, CSR_STREAM, CSR_STREAM_PIPE, CSR_BALANCED, CSR_STREAM_C16, CSR_STREAM_WAVE, CSR_STREAM_WAVEV, CSR_STREAM_WAVEX, CSR_STREAM_WAVER, CSR_STREAM_PACKED, ELL_ROW, DIA_ROW, COO_SEGMENTED, COO_LANE4, COO_TILE,
    lib, lib_path, build, version, check,
    Plan, spmv_csr_plan, spmv_coo_plan, spmv_hyb_plan, set_index_compression, get_index_compression,
    spmv_csr, spmv_csr_dot, spmv_ell_dot, spmv_dia_dot, spmv_ell, spmv_dia, spmv_coo, spmv_hyb,
    count_zeros, tuning_hyb_rule, tuning_set_hyb_rule, hyb_entries_per_row, HYB_RULE_REFERENCE, HYB_RULE_COST, HYB_RULE_COST2, tuning_hyb_light_speed, tuning_set_hyb_light_speed, tuning_waver_rule, tuning_set_waver_rule, WaverRule,
    tuning_select, tuning_set, tuning_load, tuning_save, tuning_clear,
    poisson5pt_num_entries, poisson5pt_shard_entries, poisson5pt_csr, poisson5pt_dia,
    csr_to_ell, csr_to_hyb_coo, csr_row_indices, coo_row_offsets, coo_sort_by_row, coo_is_sorted, csr_interior_rows, ell_to_csr, dia_to_csr, hyb_to_csr, ell_row_lengths,
    blas_axpy, blas_axpby, blas_copy, blas_fill, blas_dot, blas_dotd, blas_nrm2, blas_workspace,
    cg_update, cg_direction, cg_direction_x, HostScalar,
    Comm, OP_SUM, OP_MAX, OP_MIN, csr_column_span,
)
from .matrices import (  # noqa: F401
    CsrMatrix, CooMatrix, EllMatrix, DiaMatrix, HybMatrix, multiply, poisson5pt, convert,
    csr_bytes, ell_bytes, dia_bytes, coo_bytes, fill_x,
)
from . import binding, distributed, krylov  # noqa: F401,E402
