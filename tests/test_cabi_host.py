"""CPU tests (-m "not gpu"): the C-ABI library loads, exports every symbol include/*.h declares, and
its host-only entry points (status strings, argument validation, tuning table) behave.  No compute
calls: those need a GPU and live in test_spmv_gpu.py."""
import ctypes
import json
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            text = open(os.path.join(inc, f)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names |= set(re.findall(r"\b(cmi_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol(cmi):
    L = cmi.lib()
    syms = declared_symbols()
    assert len(syms) >= 50
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_version_and_status_strings(cmi):
    L = cmi.lib()
    assert cmi.version() == 400
    assert L.cmi_status_string(7) == b"CMI_ERROR_COMM"
    assert L.cmi_status_string(0) == b"CMI_SUCCESS"
    assert L.cmi_status_string(1) == b"CMI_ERROR_INVALID_VALUE"
    assert L.cmi_status_string(99) == b"CMI_ERROR_UNKNOWN"


def test_argument_validation_fails_before_touching_the_gpu(cmi):
    L = cmi.lib()
    # negative size, index overflow, pitch < rows: reported as CMI_ERROR_INVALID_VALUE with a message
    assert L.cmi_spmv_csr_f64(-1, 4, 0, None, None, None, None, None, 0, None, None) == 1
    assert b"negative" in L.cmi_last_error()
    assert L.cmi_spmv_csr_f64(2**31, 4, 0, None, None, None, None, None, 0, None, None) == 1
    assert L.cmi_spmv_csr_f64(3, 3, 2, None, None, None, None, None, 0, None, None) == 1  # null arrays
    assert L.cmi_spmv_ell_f64(10, 10, 2, 8, None, None, None, None, None, 0, None, None) == 1  # pitch < rows
    assert b"pitch" in L.cmi_last_error()
    assert L.cmi_spmv_dia_f32(10, 10, 2, 8, None, None, None, None, 0, None, None) == 1
    assert L.cmi_spmv_coo_f64(3, 3, -2, None, None, None, None, None, 0, None, None) == 1
    # empty matrix: nothing to do, success without a device
    assert L.cmi_spmv_csr_f64(0, 0, 0, None, None, None, None, None, 0, None, None) == 0
    with pytest.raises(cmi.CmiError) as e:
        cmi.check(L.cmi_blas_axpy_f64(-5, 1.0, None, None, None))
    assert e.value.status == 1


def test_communicator_entry_points_validate_before_touching_rccl_or_the_gpu(cmi):
    """cmi_comm_* / collectives (cmi_allgather_f64, cmi_allreduce_f64): null communicators, bad ranks and bad
    reduction ops come back as CMI_ERROR_INVALID_VALUE with a message; the unique id is RCCL's 128 bytes (run-time bound:
    librccl is only loaded by the first call that needs it)."""
    import ctypes
    L = cmi.lib()
    assert L.cmi_comm_create(None, 0, 1, ctypes.byref(ctypes.c_void_p())) == 1
    ident = ctypes.create_string_buffer(128)
    assert L.cmi_comm_create(ident, 2, 2, ctypes.byref(ctypes.c_void_p())) == 1 and b"rank" in L.cmi_last_error()
    assert L.cmi_comm_create(ident, 0, 0, ctypes.byref(ctypes.c_void_p())) == 1
    assert L.cmi_comm_create(ident, 0, 1, None) == 1
    assert L.cmi_allgather_f64(None, None, None, 4, None) == 1 and b"communicator" in L.cmi_last_error()
    assert L.cmi_allgatherv_f32(None, None, None, None, None, 0, None) == 1
    assert L.cmi_halo_exchange_f64(None, None, 0, None, None, None, None, None, None) == 1
    assert L.cmi_allreduce_f64(None, None, None, 1, 0, None) == 1
    assert L.cmi_comm_barrier(None, None) == 1
    assert L.cmi_comm_allgather_host(None, None, None, 8, None) == 1
    assert L.cmi_comm_rank(None, None, None) == 1
    assert L.cmi_comm_destroy(None) == 0
    assert L.cmi_comm_unique_id(None) == 1
    assert L.cmi_plan_validate(None, None, None, None, ctypes.byref(ctypes.c_int())) == 1
    lo, hi = ctypes.c_int32(5), ctypes.c_int32(5)
    assert L.cmi_csr_column_span(0, None, ctypes.byref(lo), ctypes.byref(hi), None) == 0 and (lo.value, hi.value) == (0, -1)
    assert L.cmi_csr_column_span(-1, None, ctypes.byref(lo), ctypes.byref(hi), None) == 1
    assert L.cmi_csr_rebase_offsets(3, None, 0, None, None) == 1


def test_argument_validation_of_the_newer_entry_points(cmi):
    """fused SpMV+dot, CG steps, multi-range copy, IPC, CSR->DIA, row profile: bad arguments are reported as
    CMI_ERROR_INVALID_VALUE before any HIP call (this runs without a GPU)."""
    L = cmi.lib()
    i64 = ctypes.c_int64
    assert L.cmi_spmv_csr_dot_f64(5, 5, 3, None, None, None, None, None, None, None, None, None, None) == 1
    assert b"cmi_spmv_csr_dot" in L.cmi_last_error()
    assert L.cmi_spmv_ell_dot_f64(5, 5, 2, 8, None, None, None, None, None, None, None, None, None, None) == 1
    assert L.cmi_spmv_dia_dot_f64(5, 5, 2, 8, None, None, None, None, None, None, None, None, None) == 1
    assert b"cmi_spmv_dia_dot" in L.cmi_last_error()
    assert L.cmi_cg_update_f64(-1, None, None, None, None, None, None, None, None, None, None) == 1
    assert L.cmi_cg_update_f64(4, None, None, None, None, None, None, None, None, None, None) == 1  # null scalars
    assert L.cmi_cg_direction_f64(4, None, None, None, None, None) == 1
    assert L.cmi_cg_direction_x_f64(-1, None, None, None, None, None, None, None) == 1
    assert L.cmi_cg_update_f32(4, None, None, None, None, None, None, None, None, None, None) == 1
    assert L.cmi_cg_direction_f32(-1, None, None, None, None, None) == 1 and L.cmi_cg_direction_x_f32(4, None, None, None, None, None, None, None) == 1
    assert L.cmi_blas_dotd_f32(-1, None, None, None, None, None) == 1 and L.cmi_blas_dotd_f32(4, None, None, None, None, None) == 1
    assert L.cmi_cg_direction_x_f64(4, None, None, None, None, None, None, None) == 1 and b"cmi_cg_direction_x" in L.cmi_last_error()
    n64 = ctypes.c_int64(0)
    assert L.cmi_ell_to_csr_f64(-1, 0, 0, None, None, None, None, None, 0, ctypes.byref(n64), None) == 1
    assert L.cmi_ell_to_csr_f32(4, 2, 8, None, None, None, None, None, 0, ctypes.byref(n64), None) == 1 and b"cmi_ell_to_csr" in L.cmi_last_error()
    assert L.cmi_dia_to_csr_f64(4, 4, 2, 2, None, None, None, None, None, 0, ctypes.byref(n64), None) == 1     # pitch < rows
    assert L.cmi_dia_to_csr_f32(4, 4, 2, 8, None, None, None, None, None, 0, None, None) == 1
    ok = ctypes.c_int(5)
    assert L.cmi_coo_row_offsets(-1, 0, None, None, ctypes.byref(ok), None) == 1 and L.cmi_coo_row_offsets(4, 2, None, None, ctypes.byref(ok), None) == 1
    assert L.cmi_coo_row_offsets(4, 0, None, None, None, None) == 1 and b"cmi_coo_row_offsets" in L.cmi_last_error()
    a64, b64 = ctypes.c_int64(-5), ctypes.c_int64(-5)
    assert L.cmi_csr_interior_rows(-1, None, None, 0, 4, ctypes.byref(a64), ctypes.byref(b64), None) == 1 and L.cmi_csr_interior_rows(4, None, None, 0, 4, ctypes.byref(a64), ctypes.byref(b64), None) == 1
    assert L.cmi_csr_interior_rows(0, None, None, 0, 4, ctypes.byref(a64), ctypes.byref(b64), None) == 0 and (a64.value, b64.value) == (0, 0)
    assert L.cmi_stream_wait_event(None, None) == 1 and b"cmi_stream_wait_event" in L.cmi_last_error()
    # the rest of BLAS-1 and the fused Jacobi-cg / bicgstab passes (argument checks; nothing touches a GPU)
    for suf in ("f64", "f32"):
        assert getattr(L, "cmi_blas_scal_" + suf)(-1, 1, None, None) == 1 and getattr(L, "cmi_blas_scal_" + suf)(0, 1, None, None) == 0
        assert getattr(L, "cmi_blas_xmy_" + suf)(4, None, None, None, None) == 1 and getattr(L, "cmi_blas_xmy_" + suf)(0, None, None, None, None) == 0
        assert getattr(L, "cmi_blas_asum_" + suf)(4, None, None, None, None) == 1 and getattr(L, "cmi_blas_amax_" + suf)(4, None, None, None, None, None) == 1
        assert getattr(L, "cmi_pcg_update_jacobi_" + suf)(4, None, None, None, None, None, None, None, None, None, None) == 1
        assert getattr(L, "cmi_pcg_direction_x_jacobi_" + suf)(-1, None, None, None, None, None, None, None, None) == 1
        assert getattr(L, "cmi_bicgstab_s_" + suf)(4, None, None, None, None, None, None, None, None, None) == 1
        assert getattr(L, "cmi_bicgstab_p_" + suf)(4, None, None, None, None, None, None, None, None, None) == 1
        assert getattr(L, "cmi_blas_axpy_ratio_" + suf)(4, None, None, None, None, None) == 1
        assert getattr(L, "cmi_csr_diagonal_" + suf)(-1, None, None, None, None, 1, None) == 1 and getattr(L, "cmi_csr_diagonal_" + suf)(0, None, None, None, None, 1, None) == 0
        assert getattr(L, "cmi_cr_xr_" + suf)(4, None, None, None, None, None, None, 1, None, None, None, None) == 1
        assert getattr(L, "cmi_cr_py_" + suf)(-1, None, None, None, None, None, None, None, None, None) == 1
        assert getattr(L, "cmi_blas_axpy_dot_" + suf)(4, None, None, None, None, None, None, None) == 1
    assert b"cmi_blas_axpy_dot" in L.cmi_last_error()
    # the COO container's device sort (argument checks; nothing touches a GPU)
    assert L.cmi_coo_sort_by_row_f64(-1, 4, 0, None, None, None, 0, None) == 1 and L.cmi_coo_sort_by_row_f32(4, 4, 3, None, None, None, 1, None) == 1
    assert b"cmi_coo_sort_by_row" in L.cmi_last_error()
    assert L.cmi_coo_sort_by_row_f64(4, 4, 0, None, None, None, 0, None) == 0 and L.cmi_coo_sort_by_row_f64(4, 4, 2**31, None, None, None, 0, None) == 1
    assert L.cmi_coo_is_sorted(4, 0, None, None, 1, ctypes.byref(ok), None) == 0 and ok.value == 1   # no entries: sorted
    assert L.cmi_coo_is_sorted(4, 3, None, None, 0, ctypes.byref(ok), None) == 1 and L.cmi_coo_is_sorted(4, 0, None, None, 0, None, None) == 1
    assert L.cmi_copy_ranges(17, None, None, None, None) == 1 and b"CMI_MAX_COPY_RANGES" in L.cmi_last_error()
    assert L.cmi_copy_ranges(-1, None, None, None, None) == 1
    assert L.cmi_copy_ranges(0, None, None, None, None) == 0                    # nothing to copy
    assert L.cmi_copy_ranges(2, None, None, None, None) == 1                    # null arrays
    src = (ctypes.c_void_p * 1)(None)
    dst = (ctypes.c_void_p * 1)(None)
    nbytes = (i64 * 1)(-8)
    assert L.cmi_copy_ranges(1, src, dst, nbytes, None) == 1 and b"negative" in L.cmi_last_error()
    nbytes[0] = 0
    assert L.cmi_copy_ranges(1, src, dst, nbytes, None) == 0                    # empty range: skipped
    nbytes[0] = 8
    assert L.cmi_copy_ranges(1, src, dst, nbytes, None) == 1                    # null range of 8 bytes
    assert L.cmi_ipc_get_handle(None, None) == 1 and L.cmi_ipc_open_handle(None, None) == 1
    assert L.cmi_ipc_close_handle(None) == 0
    assert L.cmi_device_can_access_peer(0, 1, None) == 1
    same = ctypes.c_int(0)
    assert L.cmi_device_can_access_peer(3, 3, ctypes.byref(same)) == 0 and same.value == 1  # a device reaches itself
    n = i64(-1)
    assert L.cmi_csr_diagonals(-1, 3, None, None, None, None, 0, ctypes.byref(n), None) == 1
    assert L.cmi_csr_diagonals(3, 3, None, None, None, None, 0, None, None) == 1
    assert L.cmi_csr_diagonals(0, 0, None, None, None, None, 0, ctypes.byref(n), None) == 0 and n.value == 0
    assert L.cmi_csr_to_dia_f64(4, 4, None, None, None, 2, 3, None, None, None, None) == 1  # pitch < rows
    assert L.cmi_csr_to_dia_f32(0, 0, None, None, None, 0, 0, None, None, None, None) == 0
    assert L.cmi_csr_max_row_length(-1, None, ctypes.byref(n), None) == 1
    assert L.cmi_csr_max_row_length(0, None, ctypes.byref(n), None) == 0 and n.value == 0
    assert L.cmi_csr_max_row_length(5, None, ctypes.byref(n), None) == 1
    assert L.cmi_malloc_host(None, 8) == 1 and L.cmi_free_host(None) == 0
    assert L.cmi_memcpy_d2h_async(None, None, 0, None) == 0 and L.cmi_memcpy_d2h_async(None, None, 8, None) == 1
    # an unknown CSR kernel id is an error, not a fallback
    cfg = cmi.Config(kernel=9)
    assert L.cmi_spmv_csr_f64(0, 0, 0, None, None, None, None, None, 0, ctypes.byref(cfg), None) == 0  # empty: nothing launched
    # two fold areas: 131072 partials + 128 chunk sums + the ticket (padded) + 16 one-line copies of the folded scalar each
    assert cmi.lib().cmi_blas_workspace_bytes() == 2 * (131072 + 128 + 2 + 16 * 16) * 8
    L.cmi_cg_update_fold_f64.argtypes = [i64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    cnt = ctypes.c_int(5)
    assert L.cmi_cg_update_fold_f64(-1, None, None, 1, None, None, None, ctypes.byref(cnt), None) == 1 and cnt.value == 0
    assert L.cmi_cg_update_fold_f64(4, None, None, 1, None, None, None, ctypes.byref(cnt), None) == 1   # null scalars
    assert L.cmi_cg_direction_x_fold_f64(4, None, None, 0, None, None, None, None, None, None, None) == 1
    assert L.cmi_spmv_csr_dot_plan_partials_f64(None, None, None, None, None, None, None, None, None, None) == 1   # null plan


def test_python_plumbing_refuses_host_tensors(cmi):
    import torch
    t = torch.zeros(4, dtype=torch.float64)
    i = torch.zeros(5, dtype=torch.int32)
    with pytest.raises(TypeError, match="device memory"):
        cmi.spmv_csr(4, 4, i, i[:0], t[:0], t, t)


def test_poisson_entry_counts(cmi):
    assert cmi.poisson5pt_num_entries(100, 100) == 49600
    assert cmi.poisson5pt_num_entries(3162, 3162) == 49978572
    assert cmi.poisson5pt_num_entries(10000, 10000) == 499960000
    assert cmi.poisson5pt_num_entries(2, 3) == 20
    assert cmi.poisson5pt_num_entries(1, 1) == 1
    assert cmi.poisson5pt_num_entries(1, 7) == 7 * 3 - 2
    # shards tile the matrix
    m, n = 37, 11
    cuts = [0, 1, 36, 37, 38, 200, 406, 407]
    assert sum(cmi.poisson5pt_shard_entries(m, n, a, b) for a, b in zip(cuts, cuts[1:])) == \
        cmi.poisson5pt_num_entries(m, n)


def test_tuning_heuristics_and_table_roundtrip(cmi, tmp_path):
    cmi.tuning_clear()
    # 5-pt Poisson CSR: short rows -> the LDS-staged stream kernel; one pass fits 204 rows of 5,
    # rounded down to whole 128-byte lines of y (192 rows); y stored with the nt hint
    c = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 9998244, 9998244, 49978572)
    assert c.kernel == cmi.CSR_STREAM and c.block_size == 256 and c.items_per_thread == 1
    assert c.rows_per_block == 192 and c.nontemporal == 2
    # long rows -> still the LDS-staged stream, summed by a power-of-two group of lanes per row
    c = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 1000, 1000, 100000)
    assert c.kernel == cmi.CSR_STREAM and c.threads_per_row == 32 and c.items_per_thread == 2
    c = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F32, 1000, 1000, 20000)
    assert c.kernel == cmi.CSR_STREAM and c.threads_per_row == 0 and c.items_per_thread == 2
    assert cmi.tuning_select(cmi.FORMAT_ELL, cmi.F64, 100, 100, 500).kernel == cmi.ELL_ROW
    assert cmi.tuning_select(cmi.FORMAT_DIA, cmi.F64, 100, 100, 500).kernel == cmi.DIA_ROW
    assert cmi.tuning_select(cmi.FORMAT_COO, cmi.F64, 100, 100, 500).kernel == cmi.COO_LANE4

    # a table entry is scaled to the matrix: rows per tile follow tuned_rows * tuned_mean / mean (the tuned fill of the LDS
    # pass), capped by what fits one pass, whole y lines kept
    cmi.tuning_set(cmi.FORMAT_CSR, cmi.F64, 5.0, cmi.Config(kernel=cmi.CSR_STREAM, block_size=256, rows_per_block=176, items_per_thread=1))
    assert cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 1000, 1000, 5000).rows_per_block == 176      # the tuned mean: as tuned
    assert cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 1000, 1000, 7000).rows_per_block == 128      # 176 * 5/7 = 126 -> 128 (nearest whole y line; fits: 896)
    assert cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 1000, 1000, 7990).rows_per_block == 112      # 110 -> 112 (fits: 895)
    assert cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 1000, 1000, 4100).rows_per_block == 208      # 215 -> 208 (fits: 853)
    # ... and its nt-load bit is dropped for a matrix whose index + value streams fit the 256 MiB Infinity Cache (x 1.25): served
    # from there when loaded plainly.  The other policy bits (2: nt stores of y, 4: lane-strided entry streams) are kept.
    cmi.tuning_set(cmi.FORMAT_CSR, cmi.F64, 5.0, cmi.Config(kernel=cmi.CSR_STREAM, block_size=256, rows_per_block=176, items_per_thread=1, nontemporal=7))
    assert cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 1000, 1000, 5000).nontemporal == 6
    assert cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 5_000_000, 5_000_000, 25_000_000).nontemporal == 6   # 300 MB of streams
    assert cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 6_000_000, 6_000_000, 30_000_000).nontemporal == 7   # 360 MB
    assert cmi.tuning_select(cmi.FORMAT_CSR, cmi.F32, 6_000_000, 6_000_000, 30_000_000).nontemporal != 7   # (another key: f32)
    cmi.tuning_set(cmi.FORMAT_CSR, cmi.F64, 5.0, cmi.Config(kernel=cmi.CSR_STREAM, block_size=256, rows_per_block=176, items_per_thread=1))
    path2 = str(tmp_path / "means.json")
    cmi.tuning_save(path2)
    assert json.load(open(path2))["entries"][0]["mean"] == 5.0
    cmi.tuning_clear()
    cmi.tuning_load(path2)                                                                          # the mean survives the file
    assert cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 1000, 1000, 7000).rows_per_block == 128
    cmi.tuning_clear()
    # persist an override, clear, reload: the selection follows the table
    cfg = cmi.Config(kernel=cmi.CSR_VECTOR, block_size=128, threads_per_row=4, nontemporal=1)
    cmi.tuning_set(cmi.FORMAT_CSR, cmi.F64, 4.99, cfg)
    path = str(tmp_path / "table.json")
    cmi.tuning_save(path)
    doc = json.load(open(path))
    assert doc["arch"] == "gfx950" and len(doc["entries"]) == 1
    assert doc["entries"][0]["format"] == "csr" and doc["entries"][0]["bucket"] == 2
    cmi.tuning_clear()
    assert cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 100, 100, 499).kernel == cmi.CSR_STREAM
    cmi.tuning_load(path)
    c = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 100, 100, 499)
    assert (c.kernel, c.block_size, c.threads_per_row, c.nontemporal) == (cmi.CSR_VECTOR, 128, 4, 1)
    # another bucket is untouched
    assert cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 100, 100, 900).kernel == cmi.CSR_STREAM
    cmi.tuning_clear()
    with pytest.raises(cmi.CmiError):
        cmi.tuning_load(str(tmp_path / "missing.json"))


def test_tuning_set_before_any_lookup_survives_the_default_table(tmp_path):
    """cmi_tuning_set as the FIRST tuning call of a process must layer on top of the shipped table -- the first
    AUTO lookup afterwards used to load the shipped file over it.  Needs a fresh process (the table is process-wide)."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import cusp_autotuned_amd as cmi\n"
        "cfg = cmi.Config(kernel=cmi.CSR_VECTOR, block_size=128, threads_per_row=8)\n"
        "cmi.tuning_set(cmi.FORMAT_CSR, cmi.F64, 5.0, cfg)            # first tuning call of the process\n"
        "c = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 1000, 1000, 5000)  # AUTO lookup: loads nothing over it\n"
        "assert (c.kernel, c.block_size, c.threads_per_row) == (cmi.CSR_VECTOR, 128, 8), c\n"
        "e = cmi.tuning_select(cmi.FORMAT_ELL, cmi.F64, 1000, 1000, 5000)   # ... and the shipped entries are there underneath\n"
        "assert e.kernel == cmi.ELL_ROW and e.block_size > 0\n"
        "d = cmi.tuning_select(cmi.FORMAT_CSR, cmi.F64, 1000, 1000, 30000)  # another bucket: the shipped CSR entry\n"
        "assert d.kernel == cmi.CSR_STREAM, d\n"
        "cmi.tuning_save(%r)\n"
    ) % (ROOT, str(tmp_path / "layered.json"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    doc = json.load(open(tmp_path / "layered.json"))
    assert len(doc["entries"]) > 10            # the shipped table was loaded before the save, not an empty one


def test_hyb_rule_and_sorted_coo_key_in_the_table(cmi, tmp_path):
    cmi.tuning_clear()
    assert cmi.tuning_hyb_rule(cmi.F64) == (cmi.HYB_RULE_REFERENCE, 3.0, 4096)   # no table: the reference's rule (csr_to_other.h:248-254)
    c = cmi.tuning_select(cmi.TABLE_COO_SORTED, cmi.F64, 1000, 1000, 5000)
    assert c.kernel == cmi.COO_TILE
    cmi.tuning_set_hyb_rule(cmi.F64, cmi.HYB_RULE_COST, 1.6, 512)
    cmi.tuning_set(cmi.TABLE_COO_SORTED, cmi.F32, 5.0, cmi.Config(kernel=cmi.COO_TILE, nontemporal=3, xcd_swizzle=64))
    with pytest.raises(cmi.CmiError):                          # the plan-less COO key must stay order-agnostic
        cmi.tuning_set(cmi.FORMAT_COO, cmi.F64, 5.0, cmi.Config(kernel=cmi.COO_TILE))
    with pytest.raises(cmi.CmiError):
        cmi.tuning_set_hyb_rule(cmi.F64, cmi.HYB_RULE_COST, 0.0, 10)
    with pytest.raises(cmi.CmiError):
        cmi.tuning_set_hyb_rule(cmi.F64, 7, 1.0, 10)
    path = str(tmp_path / "rule.json")
    cmi.tuning_save(path)
    doc = json.load(open(path))
    assert doc["hyb_rule"] == {"f64": {"kind": "cost", "relative_speed": 1.6, "threshold": 512, "light_speed": 1.6}}
    assert [e["format"] for e in doc["entries"]] == ["coo_sorted"]
    cmi.tuning_clear()
    assert cmi.tuning_hyb_rule(cmi.F64) == (cmi.HYB_RULE_REFERENCE, 3.0, 4096)
    cmi.tuning_load(path)
    assert cmi.tuning_hyb_rule(cmi.F64) == (cmi.HYB_RULE_COST, 1.6, 512)
    assert cmi.tuning_hyb_rule(cmi.F32) == (cmi.HYB_RULE_REFERENCE, 3.0, 4096)
    c = cmi.tuning_select(cmi.TABLE_COO_SORTED, cmi.F32, 1000, 1000, 5000)
    assert (c.kernel, c.nontemporal, c.xcd_swizzle) == (cmi.COO_TILE, 3, 64)
    # the two-regime kind and its fourth parameter survive the file too
    cmi.tuning_set_hyb_rule(cmi.F32, cmi.HYB_RULE_COST2, 1.1, 2_000_000)
    cmi.tuning_set_hyb_light_speed(cmi.F32, 3.0)
    cmi.tuning_save(path)
    assert json.load(open(path))["hyb_rule"]["f32"] == {"kind": "cost2", "relative_speed": 1.1, "threshold": 2000000, "light_speed": 3.0}
    cmi.tuning_clear()
    assert cmi.tuning_hyb_light_speed(cmi.F32) == 3.0                            # no table: the reference's relative speed
    cmi.tuning_load(path)
    assert cmi.tuning_hyb_rule(cmi.F32) == (cmi.HYB_RULE_COST2, 1.1, 2_000_000) and cmi.tuning_hyb_light_speed(cmi.F32) == 3.0
    with pytest.raises(cmi.CmiError):
        cmi.tuning_set_hyb_light_speed(cmi.F32, 0.0)
    cmi.tuning_clear()


def test_waver_rule_in_the_table(cmi, tmp_path):
    """csr_waver's shape and AUTO gates live in the table file ("waver_rule": tools/autotune_waver.py) -- the KTT tuner's per-kernel
    parameter space (reference cuda/ktt/csr_multiply.h:239-247) settled offline; without one, the built-in rule."""
    cmi.tuning_clear()
    r = cmi.tuning_waver_rule(cmi.F64)
    assert r.as_dict() == {"items_per_thread": 4, "cap": 0, "xcd_swizzle": 16, "min_piece": 2.2, "min_entries": 4_400_000}
    assert cmi.tuning_waver_rule(cmi.F32).min_entries == 6_400_000 and cmi.tuning_waver_rule(cmi.F32).min_piece == 1.9
    cmi.tuning_set_waver_rule(cmi.F32, items_per_thread=2, cap=3, xcd_swizzle=0, min_piece=3.0, min_entries=1234)
    for bad in (dict(items_per_thread=3), dict(cap=2), dict(xcd_swizzle=-1), dict(min_piece=0.5), dict(min_entries=-1)):
        with pytest.raises(cmi.CmiError):
            cmi.tuning_set_waver_rule(cmi.F64, **bad)
    path = str(tmp_path / "t.json")
    cmi.tuning_save(path)
    assert json.load(open(path))["waver_rule"] == {"f32": {"items_per_thread": 2, "cap": 3, "xcd_swizzle": 0, "min_piece": 3.0, "min_entries": 1234}}
    cmi.tuning_clear()
    assert cmi.tuning_waver_rule(cmi.F32).items_per_thread == 4
    cmi.tuning_load(path)
    assert cmi.tuning_waver_rule(cmi.F32).as_dict() == {"items_per_thread": 2, "cap": 3, "xcd_swizzle": 0, "min_piece": 3.0, "min_entries": 1234}
    assert cmi.tuning_waver_rule(cmi.F64).as_dict()["items_per_thread"] == 4       # f64 not in the file: the built-in rule
    # the shipped table carries the rule the sweep settled on (profiles/r04_autotune_waver.txt)
    cmi.tuning_clear()
    cmi.tuning_load(os.path.join(os.path.dirname(cmi.lib_path()), "..", "tuned", "gfx950.json"))
    doc = json.load(open(os.path.join(os.path.dirname(cmi.lib_path()), "..", "tuned", "gfx950.json")))
    for tag, code in (("f64", cmi.F64), ("f32", cmi.F32)):
        assert cmi.tuning_waver_rule(code).as_dict() == doc["waver_rule"][tag]


def test_missing_library_fails_loudly(monkeypatch):
    """No CPU fallback: without the HIP library the binding raises at first use."""
    import cusp_autotuned_amd.binding as b
    monkeypatch.setattr(b, "_lib", None)
    monkeypatch.setattr(b, "_LIB_PATH", "/nonexistent/libcusp_mi355x.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        b.lib()
