"""ctypes binding of include/cusp_mi355x.h.  Plumbing only: every function here forwards to the
C-ABI symbol of the same name with raw device pointers taken from torch tensors.  No compute
happens in Python and nothing falls back to the CPU."""
import ctypes
import os
import subprocess
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libcusp_mi355x.so")
_lib = None

FORMAT_CSR, FORMAT_ELL, FORMAT_DIA, FORMAT_COO, FORMAT_HYB = 0, 1, 2, 3, 4
TABLE_COO_SORTED = 5  # tuning-table key only: the launch shape of a COO multiply whose plan found the entries row-sorted
F64, F32 = 0, 1
KERNEL_AUTO = 0
CSR_SCALAR, CSR_VECTOR, CSR_STREAM, CSR_STREAM_PIPE, CSR_BALANCED, CSR_STREAM_C16, CSR_STREAM_WAVE, CSR_STREAM_WAVEV, CSR_STREAM_WAVEX = 1, 2, 3, 4, 5, 6, 7, 8, 9
CSR_STREAM_WAVER, CSR_STREAM_PACKED = 11, 12  # round 4: run-compressed column copy on wave tiles; ... with the values packed beside it (opt-in)
ELL_ROW, DIA_ROW, COO_SEGMENTED, COO_LANE4, COO_TILE = 10, 20, 30, 31, 32


class CmiError(RuntimeError):
    """A non-zero cmi_status (the C++ layer turns the same codes into cusp exceptions)."""

    def __init__(self, status, message):
        super().__init__(f"{message}")
        self.status = status


class Config(ctypes.Structure):
    """struct cmi_config: kernel variant + launch shape (0 = let the tuning table decide)."""
    _fields_ = [
        ("kernel", c_int32), ("block_size", c_int32), ("threads_per_row", c_int32),
        ("rows_per_block", c_int32), ("items_per_thread", c_int32), ("nontemporal", c_int32),
        ("xcd_swizzle", c_int32), ("blocks_per_cu", c_int32),
    ]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    def __repr__(self):
        return "Config(" + ", ".join(f"{k}={v}" for k, v in self.as_dict().items()) + ")"


class WaverRule(ctypes.Structure):
    """struct cmi_waver_rule: csr_waver's launch shape and AUTO gates (tuning table "waver_rule")."""
    _fields_ = [
        ("items_per_thread", c_int32), ("cap", c_int32), ("xcd_swizzle", c_int32), ("reserved", c_int32),
        ("min_piece", c_double), ("min_entries", c_int64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}

    def __repr__(self):
        return "WaverRule(" + ", ".join(f"{k}={v}" for k, v in self.as_dict().items()) + ")"


def lib_path():
    return _LIB_PATH


def build(verbose=False):
    """Compile the HIP library for gfx950 in-tree (make -C csrc).  hipcc cross-compiles without a GPU."""
    out = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
        print(out.stderr)
    if out.returncode != 0:
        raise RuntimeError("building libcusp_mi355x.so failed")
    return _LIB_PATH


_PI32 = POINTER(c_int32)


def _declare(L):
    vp, i64, i32 = c_void_p, c_int64, c_int
    cfgp = POINTER(Config)
    L.cmi_status_string.restype = c_char_p
    L.cmi_status_string.argtypes = [c_int]
    L.cmi_last_error.restype = c_char_p
    L.cmi_version.restype = c_int
    L.cmi_device_count.argtypes = [POINTER(c_int)]
    L.cmi_set_device.argtypes = [c_int]
    L.cmi_get_device.argtypes = [POINTER(c_int)]
    L.cmi_device_info.argtypes = [c_int, c_char_p, c_size_t, POINTER(c_int), POINTER(c_int64)]
    L.cmi_malloc.argtypes = [POINTER(c_void_p), c_size_t]
    L.cmi_free.argtypes = [vp]
    for n in ("cmi_memcpy_h2d", "cmi_memcpy_d2h", "cmi_memcpy_d2d"):
        getattr(L, n).argtypes = [vp, vp, c_size_t, vp]
    L.cmi_memset.argtypes = [vp, c_int, c_size_t, vp]
    L.cmi_stream_create.argtypes = [POINTER(c_void_p)]
    L.cmi_stream_destroy.argtypes = [vp]
    L.cmi_stream_synchronize.argtypes = [vp]
    L.cmi_event_create.argtypes = [POINTER(c_void_p)]
    L.cmi_event_destroy.argtypes = [vp]
    L.cmi_event_record.argtypes = [vp, vp]
    L.cmi_event_elapsed_ms.argtypes = [vp, vp, POINTER(c_float)]
    L.cmi_event_synchronize.argtypes = [vp]
    L.cmi_malloc_host.argtypes = [POINTER(c_void_p), c_size_t]
    L.cmi_free_host.argtypes = [vp]
    L.cmi_memcpy_d2h_async.argtypes = [vp, vp, c_size_t, vp]
    L.cmi_device_can_access_peer.argtypes = [c_int, c_int, POINTER(c_int)]
    L.cmi_ipc_get_handle.argtypes = [vp, vp]
    L.cmi_ipc_open_handle.argtypes = [vp, POINTER(c_void_p)]
    L.cmi_ipc_close_handle.argtypes = [vp]
    L.cmi_copy_ranges.argtypes = [c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), vp]
    L.cmi_spmv_csr_dot_f64.argtypes = [i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, cfgp, vp]
    for suf in ("f64", "f32"):
        getattr(L, f"cmi_spmv_ell_dot_{suf}").argtypes = [i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, cfgp, vp]
        getattr(L, f"cmi_spmv_dia_dot_{suf}").argtypes = [i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, cfgp, vp]
    L.cmi_tuning_load.argtypes = [c_char_p]
    L.cmi_tuning_save.argtypes = [c_char_p]
    L.cmi_tuning_set.argtypes = [c_int, c_int, c_double, cfgp]
    L.cmi_tuning_select.argtypes = [c_int, c_int, i64, i64, i64, cfgp]
    for suf in ("f64", "f32"):
        getattr(L, f"cmi_spmv_csr_{suf}").argtypes = [i64, i64, i64, vp, vp, vp, vp, vp, i32, cfgp, vp]
        getattr(L, f"cmi_spmv_coo_{suf}").argtypes = [i64, i64, i64, vp, vp, vp, vp, vp, i32, cfgp, vp]
        getattr(L, f"cmi_spmv_ell_{suf}").argtypes = [i64, i64, i64, i64, vp, vp, vp, vp, vp, i32, cfgp, vp]
        getattr(L, f"cmi_spmv_dia_{suf}").argtypes = [i64, i64, i64, i64, vp, vp, vp, vp, i32, cfgp, vp]
        getattr(L, f"cmi_spmv_hyb_{suf}").argtypes = [i64, i64, i64, i64, vp, vp, i64, vp, vp, vp, vp, vp, i32,
                                                     cfgp, cfgp, vp]
        getattr(L, f"cmi_poisson5pt_csr_{suf}").argtypes = [i64, i64, i64, i64, vp, vp, vp, vp]
        getattr(L, f"cmi_poisson5pt_dia_{suf}").argtypes = [i64, i64, i64, vp, vp, vp]
        getattr(L, f"cmi_csr_to_ell_{suf}").argtypes = [i64, vp, vp, vp, i64, i64, vp, vp, vp]
        getattr(L, f"cmi_csr_to_hyb_coo_{suf}").argtypes = [i64, vp, vp, vp, i64, vp, vp, vp, vp, vp]
    L.cmi_poisson5pt_num_entries.restype = c_int64
    L.cmi_poisson5pt_num_entries.argtypes = [i64, i64]
    L.cmi_poisson5pt_shard_entries.restype = c_int64
    L.cmi_poisson5pt_shard_entries.argtypes = [i64, i64, i64, i64]
    L.cmi_csr_row_indices.argtypes = [i64, vp, vp, vp]
    L.cmi_coo_row_offsets.argtypes = [i64, i64, vp, vp, POINTER(ctypes.c_int), vp]
    L.cmi_csr_interior_rows.argtypes = [i64, vp, vp, i64, i64, POINTER(c_int64), POINTER(c_int64), vp]
    L.cmi_stream_wait_event.argtypes = [vp, vp]
    L.cmi_coo_sort_by_row_f64.argtypes = [i64, i64, i64, vp, vp, vp, ctypes.c_int, vp]
    L.cmi_coo_sort_by_row_f32.argtypes = [i64, i64, i64, vp, vp, vp, ctypes.c_int, vp]
    L.cmi_coo_is_sorted.argtypes = [i64, i64, vp, vp, ctypes.c_int, POINTER(ctypes.c_int), vp]
    for sfx in ("f64", "f32"):
        getattr(L, "cmi_ell_to_csr_" + sfx).argtypes = [i64, i64, i64, vp, vp, vp, vp, vp, i64, POINTER(c_int64), vp]
        getattr(L, "cmi_dia_to_csr_" + sfx).argtypes = [i64, i64, i64, i64, vp, vp, vp, vp, vp, i64, POINTER(c_int64), vp]
        getattr(L, "cmi_hyb_to_csr_" + sfx).argtypes = [i64, i64, i64, vp, vp, i64, vp, vp, vp, vp, vp, vp, i64, POINTER(c_int64), vp]
    L.cmi_csr_max_row_length.argtypes = [i64, vp, POINTER(c_int64), vp]
    L.cmi_csr_diagonals.argtypes = [i64, i64, vp, vp, vp, vp, i64, POINTER(c_int64), vp]
    for suf in ("f64", "f32"):
        getattr(L, f"cmi_csr_to_dia_{suf}").argtypes = [i64, i64, vp, vp, vp, i64, i64, vp, vp, vp, vp]
    L.cmi_ell_row_lengths.argtypes = [i64, i64, i64, vp, vp, vp]
    L.cmi_blas_workspace_bytes.restype = c_size_t
    for suf, sc in (("f64", c_double), ("f32", c_float)):
        getattr(L, f"cmi_blas_axpy_{suf}").argtypes = [i64, sc, vp, vp, vp]
        getattr(L, f"cmi_blas_axpby_{suf}").argtypes = [i64, sc, vp, sc, vp, vp, vp]
        getattr(L, f"cmi_blas_copy_{suf}").argtypes = [i64, vp, vp, vp]
        getattr(L, f"cmi_blas_fill_{suf}").argtypes = [i64, sc, vp, vp]
        getattr(L, f"cmi_blas_dot_{suf}").argtypes = [i64, vp, vp, vp, vp, vp]
        getattr(L, f"cmi_blas_nrm2_{suf}").argtypes = [i64, vp, vp, vp, vp]
    for suf in ("f64", "f32"):
        getattr(L, f"cmi_spmv_csr_dot_plan_partials_{suf}").argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, POINTER(c_int), vp]
        getattr(L, f"cmi_cg_update_fold_{suf}").argtypes = [i64, vp, vp, c_int, vp, vp, vp, POINTER(c_int), vp]
        getattr(L, f"cmi_cg_direction_x_fold_{suf}").argtypes = [i64, vp, vp, c_int, vp, vp, vp, vp, vp, vp, vp]
    L.cmi_cg_update_f64.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.cmi_cg_direction_f64.argtypes = [i64, vp, vp, vp, vp, vp]
    L.cmi_cg_direction_x_f64.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp]
    L.cmi_cg_update_f32.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.cmi_cg_direction_f32.argtypes = [i64, vp, vp, vp, vp, vp]
    L.cmi_cg_direction_x_f32.argtypes = [i64, vp, vp, vp, vp, vp, vp, vp]
    L.cmi_blas_dotd_f32.argtypes = [i64, vp, vp, vp, vp, vp]
    L.cmi_count_zeros_f64.argtypes = [i64, vp, POINTER(c_int64), vp]
    L.cmi_count_zeros_f32.argtypes = [i64, vp, POINTER(c_int64), vp]
    L.cmi_tuning_hyb_rule.argtypes = [c_int, POINTER(c_int), POINTER(c_double), POINTER(c_int64)]
    L.cmi_tuning_set_hyb_rule.argtypes = [c_int, c_int, c_double, i64]
    L.cmi_tuning_waver_rule.argtypes = [c_int, POINTER(WaverRule)]
    L.cmi_tuning_set_waver_rule.argtypes = [c_int, POINTER(WaverRule)]
    L.cmi_tuning_hyb_light_speed.argtypes = [c_int, POINTER(c_double)]
    L.cmi_tuning_set_hyb_light_speed.argtypes = [c_int, c_double]
    L.cmi_hyb_entries_per_row.argtypes = [c_int, i64, vp, c_int, c_double, i64, POINTER(c_int64), vp]
    L.cmi_plan_create.argtypes = [c_int, c_int, i64, i64, i64, vp, cfgp, vp, POINTER(c_void_p)]
    L.cmi_plan_destroy.argtypes = [vp]
    L.cmi_plan_config.argtypes = [vp, cfgp]
    L.cmi_plan_validate.argtypes = [vp, vp, vp, vp, POINTER(c_int)]
    L.cmi_plan_info.argtypes = [vp, POINTER(c_int64), POINTER(c_int64), POINTER(c_int), POINTER(c_int)]
    L.cmi_plan_create_csr.argtypes = [c_int, i64, i64, i64, vp, vp, cfgp, vp, POINTER(c_void_p)]
    L.cmi_plan_create_coo.argtypes = [c_int, i64, i64, i64, vp, vp, cfgp, vp, POINTER(c_void_p)]
    L.cmi_plan_create_csr_values.argtypes = [c_int, i64, i64, i64, vp, vp, vp, cfgp, vp, POINTER(c_void_p)]
    L.cmi_plan_validate_values.argtypes = [vp, vp, vp, POINTER(c_int)]
    L.cmi_plan_device_bytes.argtypes = [vp, POINTER(c_int64)]
    L.cmi_set_index_compression.argtypes = [c_int]
    L.cmi_plan_hyb_launches.argtypes = [vp, POINTER(c_int)]
    L.cmi_plan_create_hyb.argtypes = [c_int, i64, i64, i64, i64, vp, cfgp, cfgp, vp, POINTER(c_void_p)]
    for suf in ("f64", "f32"):
        getattr(L, f"cmi_spmv_hyb_plan_{suf}").argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, vp]
        getattr(L, f"cmi_spmv_hyb_dot_plan_{suf}").argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    for suf in ("f64", "f32"):
        getattr(L, f"cmi_spmv_csr_plan_{suf}").argtypes = [vp, vp, vp, vp, vp, vp, i32, vp]
        getattr(L, f"cmi_spmv_coo_plan_{suf}").argtypes = [vp, vp, vp, vp, vp, vp, i32, vp]
        getattr(L, f"cmi_spmv_coo_dot_plan_{suf}").argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        getattr(L, f"cmi_spmv_csr_dot_plan_{suf}").argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.cmi_spmv_csr_dot_f32.argtypes = [i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, cfgp, vp]
    # communicator + collectives (RCCL behind the boundary)
    L.cmi_comm_unique_id.argtypes = [vp]
    L.cmi_comm_create.argtypes = [vp, c_int, c_int, POINTER(c_void_p)]
    L.cmi_comm_destroy.argtypes = [vp]
    L.cmi_comm_rank.argtypes = [vp, POINTER(c_int), POINTER(c_int)]
    L.cmi_comm_library_version.argtypes = [POINTER(c_int)]
    L.cmi_comm_barrier.argtypes = [vp, vp]
    L.cmi_comm_allgather_host.argtypes = [vp, vp, vp, c_size_t, vp]
    L.cmi_allreduce_f64.argtypes = [vp, vp, vp, i64, c_int, vp]
    for suf in ("f64", "f32"):
        getattr(L, f"cmi_allgather_{suf}").argtypes = [vp, vp, vp, i64, vp]
        getattr(L, f"cmi_allgatherv_{suf}").argtypes = [vp, vp, vp, POINTER(c_int64), POINTER(c_int64), c_int, vp]
        getattr(L, f"cmi_halo_exchange_{suf}").argtypes = [vp, vp, c_int, POINTER(c_int), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64),
                                                          POINTER(c_int64), vp]
    L.cmi_csr_column_span.argtypes = [i64, vp, POINTER(ctypes.c_int32), POINTER(ctypes.c_int32), vp]
    L.cmi_csr_rebase_offsets.argtypes = [i64, vp, ctypes.c_int32, vp, vp]


def lib():
    """The loaded C-ABI library.  Raises (loudly) when it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError(
                f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C cusp-autotuned_amd/csrc`.  The SpMV engine has no CPU fallback.")
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be the one this
        # process binds, so import torch BEFORE the library pulls in /opt/rocm's copy.  Loading in the
        # other order leaves two runtimes fighting over the device ("no ROCm-capable device is
        # detected" at the first launch) -- seen on MI355X when a test module loaded the library first.
        import torch  # noqa: F401
        L = ctypes.CDLL(_LIB_PATH)
        _declare(L)
        _lib = L
    return _lib


def check(status):
    if status != 0:
        L = lib()
        raise CmiError(status, f"{L.cmi_status_string(status).decode()}: {L.cmi_last_error().decode()}")


def version():
    return lib().cmi_version()


# ------------------------------------------------------------------------------------------------
# tensor plumbing
# ------------------------------------------------------------------------------------------------
def _ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def _suffix(t):
    import torch
    if t.dtype == torch.float64:
        return "f64"
    if t.dtype == torch.float32:
        return "f32"
    raise TypeError(f"value dtype must be float64 or float32, got {t.dtype}")


def _stream(stream):
    """hipStream_t of `stream` (a torch.cuda.Stream), default: torch's current stream."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream()
    return c_void_p(stream.cuda_stream)


def _need(t, name, dtype=None):
    import torch
    if t is None:
        return
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a CUDA/HIP torch tensor (device memory); got {type(t).__name__}"
                        f"{'' if not isinstance(t, torch.Tensor) else ' on ' + str(t.device)}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must have dtype {dtype}, got {t.dtype}")


def _cfg(cfg):
    return None if cfg is None else ctypes.byref(cfg)


def spmv_csr(num_rows, num_cols, Ap, Aj, Ax, x, y, accumulate=False, cfg=None, stream=None):
    import torch
    for t, n in ((Ap, "Ap"), (Aj, "Aj")):
        _need(t, n, torch.int32)
    for t, n in ((Ax, "Ax"), (x, "x"), (y, "y")):
        _need(t, n, y.dtype)
    if Ap.numel() != num_rows + 1 or x.numel() != num_cols or y.numel() != num_rows or Aj.numel() != Ax.numel():
        raise ValueError("spmv_csr: array lengths do not match the matrix shape")
    fn = getattr(lib(), "cmi_spmv_csr_" + _suffix(y))
    check(fn(num_rows, num_cols, Aj.numel(), _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(y), int(bool(accumulate)),
             _cfg(cfg), _stream(stream)))


class Plan:
    """cmi_plan: what the library learns about one matrix before its first multiply (launch shape; CSR: row-length
    profile; COO: row-sortedness).  Creating one synchronises the stream; multiplies through it never do."""

    def __init__(self, fmt, dtype, num_rows, num_cols, num_entries, index_array=None, cfg=None, stream=None):
        import torch
        self._h = c_void_p()
        if index_array is not None:
            _need(index_array, "index_array", torch.int32)
        code = F64 if dtype in (F64, torch.float64) else F32
        check(lib().cmi_plan_create(fmt, code, num_rows, num_cols, num_entries, _ptr(index_array), _cfg(cfg), _stream(stream),
                                    byref(self._h)))
        self.format, self.dtype = fmt, code
        self.num_rows, self.num_cols, self.num_entries = num_rows, num_cols, num_entries

    @classmethod
    def csr(cls, dtype, num_rows, num_cols, Ap, Aj, cfg=None, stream=None):
        """cmi_plan_create_csr: a CSR plan made from both structure arrays -- with cfg.kernel == CSR_STREAM_C16 (or after
        set_index_compression(True)) it also builds the 16-bit column copy; config().kernel tells whether it was granted."""
        import torch
        self = cls.__new__(cls)
        self._h = c_void_p()
        _need(Ap, "Ap", torch.int32)
        _need(Aj, "Aj", torch.int32)
        if Ap.numel() != num_rows + 1:
            raise ValueError("Plan.csr: row offsets must have num_rows + 1 entries")
        code = F64 if dtype in (F64, torch.float64) else F32
        check(lib().cmi_plan_create_csr(code, num_rows, num_cols, Aj.numel(), _ptr(Ap), _ptr(Aj), _cfg(cfg), _stream(stream), byref(self._h)))
        self.format, self.dtype = FORMAT_CSR, code
        self.num_rows, self.num_cols, self.num_entries = num_rows, num_cols, Aj.numel()
        return self

    @classmethod
    def coo(cls, dtype, num_rows, num_cols, Ai, Aj, cfg=None, stream=None):
        """cmi_plan_create_coo: a COO plan made from both index arrays -- sorted entries get a CSR sub-plan made with the columns (the
        run-compressed copy where the columns come in runs)."""
        import torch
        self = cls.__new__(cls)
        self._h = c_void_p()
        _need(Ai, "Ai", torch.int32)
        _need(Aj, "Aj", torch.int32)
        if Ai.numel() != Aj.numel():
            raise ValueError("Plan.coo: row and column indices must have the same length")
        code = F64 if dtype in (F64, torch.float64) else F32
        check(lib().cmi_plan_create_coo(code, num_rows, num_cols, Aj.numel(), _ptr(Ai), _ptr(Aj), _cfg(cfg), _stream(stream), byref(self._h)))
        self.format, self.dtype = FORMAT_COO, code
        self.num_rows, self.num_cols, self.num_entries = num_rows, num_cols, Aj.numel()
        return self

    @classmethod
    def csr_values(cls, num_rows, num_cols, Ap, Aj, Ax, cfg=None, stream=None):
        """cmi_plan_create_csr_values: Plan.csr plus the values -- with cfg.kernel == CSR_STREAM_PACKED the plan copies pieces AND values into
        one span per wave tile (the plan then owns a copy of the values: validate_values tells whether they have changed since)."""
        import torch
        self = cls.__new__(cls)
        self._h = c_void_p()
        _need(Ap, "Ap", torch.int32)
        _need(Aj, "Aj", torch.int32)
        if Ap.numel() != num_rows + 1 or Ax.numel() != Aj.numel():
            raise ValueError("Plan.csr_values: row offsets must have num_rows + 1 entries, values as many as columns")
        code = F64 if Ax.dtype == torch.float64 else F32
        check(lib().cmi_plan_create_csr_values(code, num_rows, num_cols, Aj.numel(), _ptr(Ap), _ptr(Aj), _ptr(Ax), _cfg(cfg), _stream(stream), byref(self._h)))
        self.format, self.dtype = FORMAT_CSR, code
        self.num_rows, self.num_cols, self.num_entries = num_rows, num_cols, Aj.numel()
        return self

    def validate_values(self, values, stream=None):
        """cmi_plan_validate_values: False when a CSR_STREAM_PACKED plan's copy of the values no longer matches `values`."""
        ok = c_int(-1)
        check(lib().cmi_plan_validate_values(self._h, _ptr(values), _stream(stream), byref(ok)))
        return bool(ok.value)

    def device_bytes(self):
        n = c_int64()
        check(lib().cmi_plan_device_bytes(self._h, byref(n)))
        return n.value

    @classmethod
    def hyb(cls, dtype, num_rows, num_cols, width, coo_row_indices, cfg_ell=None, cfg_coo=None, stream=None):
        """cmi_plan_create_hyb: both parts' launch shapes and -- COO part sorted by row -- the per-tile entry ranges of the
        one-launch HYB kernel (spmv_hyb_plan)."""
        import torch
        self = cls.__new__(cls)
        self._h = c_void_p()
        _need(coo_row_indices, "coo_row_indices", torch.int32)
        code = F64 if dtype in (F64, torch.float64) else F32
        check(lib().cmi_plan_create_hyb(code, num_rows, num_cols, width, coo_row_indices.numel(), _ptr(coo_row_indices),
                                        _cfg(cfg_ell), _cfg(cfg_coo), _stream(stream), byref(self._h)))
        self.format, self.dtype = FORMAT_HYB, code
        self.num_rows, self.num_cols, self.num_entries = num_rows, num_cols, num_rows * width
        self.width, self.coo_entries = width, coo_row_indices.numel()
        return self

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.cmi_plan_destroy(h)

    @property
    def handle(self):
        return self._h

    def validate(self, index_array, column_indices=None, stream=None):
        """cmi_plan_validate: True while the arrays the plan was made from still hold what they held then (one streaming pass
        over them + a stream synchronisation); False after an in-place edit -- make a new plan."""
        ok = c_int(-1)
        check(lib().cmi_plan_validate(self._h, _ptr(index_array), _ptr(column_indices), _stream(stream), byref(ok)))
        return bool(ok.value)

    def hyb_launches(self):
        """1 = the one-launch HYB kernel (or an empty COO part), 2 = ELL kernel + COO kernel."""
        n = c_int()
        check(lib().cmi_plan_hyb_launches(self._h, byref(n)))
        return n.value

    def config(self):
        c = Config()
        check(lib().cmi_plan_config(self._h, byref(c)))
        return c

    def info(self):
        """dict: max_row_length, entries_in_long_rows (CSR; -1 otherwise), coo_sorted (COO: True/False, else None),
        storage_order_sums (every result bit-identical to the host loop?)."""
        a, b, s, e = c_int64(), c_int64(), c_int(), c_int()
        check(lib().cmi_plan_info(self._h, byref(a), byref(b), byref(s), byref(e)))
        return {"max_row_length": a.value, "entries_in_long_rows": b.value,
                "coo_sorted": None if s.value < 0 else bool(s.value), "storage_order_sums": bool(e.value)}


def spmv_csr_plan(plan, Ap, Aj, Ax, x, y, accumulate=False, stream=None):
    """cmi_spmv_csr_plan_*: the multiply steered by a plan (no table lookup, no measurement, no sync)."""
    import torch
    for t, n in ((Ap, "Ap"), (Aj, "Aj")):
        _need(t, n, torch.int32)
    for t, n in ((Ax, "Ax"), (x, "x"), (y, "y")):
        _need(t, n, y.dtype)
    if (Ap.numel() != plan.num_rows + 1 or x.numel() != plan.num_cols or y.numel() != plan.num_rows
            or Aj.numel() != plan.num_entries or Ax.numel() != plan.num_entries):
        raise ValueError("spmv_csr_plan: array lengths do not match the plan's matrix shape")
    fn = getattr(lib(), "cmi_spmv_csr_plan_" + _suffix(y))
    check(fn(plan.handle, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(y), int(bool(accumulate)), _stream(stream)))


def spmv_coo_plan(plan, Ai, Aj, Ax, x, y, accumulate=False, stream=None):
    import torch
    for t, n in ((Ai, "Ai"), (Aj, "Aj")):
        _need(t, n, torch.int32)
    for t, n in ((Ax, "Ax"), (x, "x"), (y, "y")):
        _need(t, n, y.dtype)
    if (Ai.numel() != plan.num_entries or Aj.numel() != plan.num_entries or Ax.numel() != plan.num_entries
            or x.numel() != plan.num_cols or y.numel() != plan.num_rows):
        raise ValueError("spmv_coo_plan: array lengths do not match the plan's matrix shape")
    fn = getattr(lib(), "cmi_spmv_coo_plan_" + _suffix(y))
    check(fn(plan.handle, _ptr(Ai), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(y), int(bool(accumulate)), _stream(stream)))


def spmv_csr_dot(num_rows, num_cols, Ap, Aj, Ax, x, y, w, result, workspace, cfg=None, stream=None, plan=None):
    """y <- A x and result[0] <- <y, w> in one pass (cmi_spmv_csr_dot_{f64,f32}[_plan]); result is a float64 tensor
    for both value types (f32: products and partial sums accumulate in double)."""
    import torch
    for t, n in ((Ap, "Ap"), (Aj, "Aj")):
        _need(t, n, torch.int32)
    for t, n in ((Ax, "Ax"), (x, "x"), (y, "y"), (w, "w")):
        _need(t, n, y.dtype)
    _need(result, "result", torch.float64)
    if (Ap.numel() != num_rows + 1 or x.numel() != num_cols or y.numel() != num_rows or w.numel() != num_rows
            or Aj.numel() != Ax.numel() or result.numel() < 1):
        raise ValueError("spmv_csr_dot: array lengths do not match the matrix shape")
    suf = _suffix(y)
    if plan is not None:
        check(getattr(lib(), "cmi_spmv_csr_dot_plan_" + suf)(plan.handle, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(y), _ptr(w),
                                                             _ptr(result), _ptr(workspace), _stream(stream)))
    else:
        check(getattr(lib(), "cmi_spmv_csr_dot_" + suf)(num_rows, num_cols, Aj.numel(), _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(y),
                                                        _ptr(w), _ptr(result), _ptr(workspace), _cfg(cfg), _stream(stream)))


def spmv_ell_dot(num_rows, num_cols, width, pitch, Aj, Ax, x, y, w, result, workspace, row_lengths=None, cfg=None, stream=None):
    """ELL: y <- A x and result[0] <- <y, w> (a double) in one pass (cmi_spmv_ell_dot_{f64,f32})."""
    import torch
    for t, n in ((Ax, "Ax"), (x, "x"), (y, "y"), (w, "w")):
        _need(t, n, y.dtype)
    _need(result, "result", torch.float64)
    _need(Aj, "Aj", torch.int32)
    if x.numel() != num_cols or y.numel() != num_rows or w.numel() != num_rows or Aj.numel() < width * pitch or Ax.numel() < width * pitch:
        raise ValueError("spmv_ell_dot: array lengths do not match the matrix shape")
    check(getattr(lib(), "cmi_spmv_ell_dot_" + _suffix(y))(num_rows, num_cols, width, pitch, _ptr(Aj), _ptr(Ax), _ptr(row_lengths), _ptr(x), _ptr(y),
                                                           _ptr(w), _ptr(result), _ptr(workspace), _cfg(cfg), _stream(stream)))


def spmv_dia_dot(num_rows, num_cols, num_diagonals, pitch, offsets, values, x, y, w, result, workspace, cfg=None, stream=None):
    """DIA: y <- A x and result[0] <- <y, w> (a double) in one pass (cmi_spmv_dia_dot_{f64,f32})."""
    import torch
    for t, n in ((values, "values"), (x, "x"), (y, "y"), (w, "w")):
        _need(t, n, y.dtype)
    _need(result, "result", torch.float64)
    _need(offsets, "offsets", torch.int32)
    if x.numel() != num_cols or y.numel() != num_rows or w.numel() != num_rows or offsets.numel() != num_diagonals or values.numel() < num_diagonals * pitch:
        raise ValueError("spmv_dia_dot: array lengths do not match the matrix shape")
    check(getattr(lib(), "cmi_spmv_dia_dot_" + _suffix(y))(num_rows, num_cols, num_diagonals, pitch, _ptr(offsets), _ptr(values), _ptr(x), _ptr(y),
                                                           _ptr(w), _ptr(result), _ptr(workspace), _cfg(cfg), _stream(stream)))


def spmv_coo(num_rows, num_cols, Ai, Aj, Ax, x, y, accumulate=False, cfg=None, stream=None):
    import torch
    for t, n in ((Ai, "Ai"), (Aj, "Aj")):
        _need(t, n, torch.int32)
    for t, n in ((Ax, "Ax"), (x, "x"), (y, "y")):
        _need(t, n, y.dtype)
    if Ai.numel() != Aj.numel() or Aj.numel() != Ax.numel() or x.numel() != num_cols or y.numel() != num_rows:
        raise ValueError("spmv_coo: array lengths do not match the matrix shape")
    fn = getattr(lib(), "cmi_spmv_coo_" + _suffix(y))
    check(fn(num_rows, num_cols, Ax.numel(), _ptr(Ai), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(y), int(bool(accumulate)),
             _cfg(cfg), _stream(stream)))


def spmv_ell(num_rows, num_cols, width, pitch, Aj, Ax, x, y, row_lengths=None, accumulate=False, cfg=None,
             stream=None):
    import torch
    _need(Aj, "Aj", torch.int32)
    _need(row_lengths, "row_lengths", torch.int32)
    for t, n in ((Ax, "Ax"), (x, "x"), (y, "y")):
        _need(t, n, y.dtype)
    if Aj.numel() < width * pitch or Ax.numel() < width * pitch or x.numel() != num_cols or y.numel() != num_rows:
        raise ValueError("spmv_ell: array lengths do not match the matrix shape")
    fn = getattr(lib(), "cmi_spmv_ell_" + _suffix(y))
    check(fn(num_rows, num_cols, width, pitch, _ptr(Aj), _ptr(Ax), _ptr(row_lengths), _ptr(x), _ptr(y),
             int(bool(accumulate)), _cfg(cfg), _stream(stream)))


def spmv_dia(num_rows, num_cols, num_diagonals, pitch, offsets, values, x, y, accumulate=False, cfg=None,
             stream=None):
    import torch
    _need(offsets, "offsets", torch.int32)
    for t, n in ((values, "values"), (x, "x"), (y, "y")):
        _need(t, n, y.dtype)
    if offsets.numel() != num_diagonals or values.numel() < num_diagonals * pitch or x.numel() != num_cols \
            or y.numel() != num_rows:
        raise ValueError("spmv_dia: array lengths do not match the matrix shape")
    fn = getattr(lib(), "cmi_spmv_dia_" + _suffix(y))
    check(fn(num_rows, num_cols, num_diagonals, pitch, _ptr(offsets), _ptr(values), _ptr(x), _ptr(y),
             int(bool(accumulate)), _cfg(cfg), _stream(stream)))


def spmv_hyb(num_rows, num_cols, width, pitch, ell_Aj, ell_Ax, coo_Ai, coo_Aj, coo_Ax, x, y, accumulate=False,
             cfg_ell=None, cfg_coo=None, stream=None):
    import torch
    for t, n in ((ell_Aj, "ell_Aj"), (coo_Ai, "coo_Ai"), (coo_Aj, "coo_Aj")):
        _need(t, n, torch.int32)
    for t, n in ((ell_Ax, "ell_Ax"), (coo_Ax, "coo_Ax"), (x, "x"), (y, "y")):
        _need(t, n, y.dtype)
    fn = getattr(lib(), "cmi_spmv_hyb_" + _suffix(y))
    check(fn(num_rows, num_cols, width, pitch, _ptr(ell_Aj), _ptr(ell_Ax), coo_Ax.numel(), _ptr(coo_Ai), _ptr(coo_Aj),
             _ptr(coo_Ax), _ptr(x), _ptr(y), int(bool(accumulate)), _cfg(cfg_ell), _cfg(cfg_coo), _stream(stream)))


def hyb_plan_args(plan, pitch, ell_Aj, ell_Ax, coo_Ai, coo_Aj, coo_Ax):
    """Validate a HYB matrix's arrays against its plan ONCE and return what spmv_hyb_plan_args passes per call (a container
    that multiplies many times -- CG -- should not pay eight tensor checks per launch)."""
    import torch
    for t, n in ((ell_Aj, "ell_Aj"), (coo_Ai, "coo_Ai"), (coo_Aj, "coo_Aj")):
        _need(t, n, torch.int32)
    for t, n in ((ell_Ax, "ell_Ax"), (coo_Ax, "coo_Ax")):
        _need(t, n, ell_Ax.dtype)
    if plan.format != FORMAT_HYB:
        raise ValueError("hyb_plan_args: not a HYB plan")
    if (coo_Ai.numel() != plan.coo_entries or coo_Aj.numel() != plan.coo_entries or coo_Ax.numel() != plan.coo_entries
            or ell_Aj.numel() < plan.width * pitch or ell_Ax.numel() < plan.width * pitch or pitch < plan.num_rows):
        raise ValueError("hyb_plan_args: array lengths do not match the plan's matrix shape")
    return (getattr(lib(), "cmi_spmv_hyb_plan_" + _suffix(ell_Ax)), plan.handle, pitch, _ptr(ell_Aj), _ptr(ell_Ax), _ptr(coo_Ai), _ptr(coo_Aj),
            _ptr(coo_Ax), ell_Ax.dtype, plan.num_cols, plan.num_rows)


def spmv_hyb_plan_args(args, x, y, accumulate=False, stream=None):
    fn, h, pitch, eAj, eAx, cAi, cAj, cAx, dt, ncols, nrows = args
    if x.dtype != dt or y.dtype != dt or not x.is_cuda or not y.is_cuda or x.numel() != ncols or y.numel() != nrows \
            or not x.is_contiguous() or not y.is_contiguous():
        raise ValueError("spmv_hyb_plan: x / y must be contiguous device vectors of the matrix's value type and shape")
    check(fn(h, pitch, eAj, eAx, cAi, cAj, cAx, _ptr(x), _ptr(y), int(bool(accumulate)), _stream(stream)))


def spmv_coo_dot_plan(plan, Ai, Aj, Ax, x, y, w, result, workspace, stream=None):
    """cmi_spmv_coo_dot_plan_*: y <- A x and result[0] <- <y, w> (a double); one pass for sorted entries (the plan's row offsets)."""
    import torch
    for t, n in ((Ai, "Ai"), (Aj, "Aj")):
        _need(t, n, torch.int32)
    for t, n in ((Ax, "Ax"), (x, "x"), (y, "y"), (w, "w")):
        _need(t, n, y.dtype)
    _need(result, "result", torch.float64)
    if (Ai.numel() != plan.num_entries or Aj.numel() != plan.num_entries or Ax.numel() != plan.num_entries
            or x.numel() != plan.num_cols or y.numel() != plan.num_rows or w.numel() != plan.num_rows):
        raise ValueError("spmv_coo_dot_plan: array lengths do not match the plan's matrix shape")
    check(getattr(lib(), "cmi_spmv_coo_dot_plan_" + _suffix(y))(plan.handle, _ptr(Ai), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(y), _ptr(w), _ptr(result),
                                                                _ptr(workspace), _stream(stream)))


def spmv_hyb_dot_plan_args(args, x, y, w, result, workspace, stream=None):
    """cmi_spmv_hyb_dot_plan_*: y <- A x and result[0] <- <y, w> (a double), one pass where the plan runs one launch."""
    import torch
    fn, h, pitch, eAj, eAx, cAi, cAj, cAx, dt, ncols, nrows = args
    if x.dtype != dt or y.dtype != dt or w.dtype != dt or x.numel() != ncols or y.numel() != nrows or w.numel() != nrows:
        raise ValueError("spmv_hyb_dot_plan: x / y / w must be device vectors of the matrix's value type and shape")
    _need(result, "result", torch.float64)
    fnd = getattr(lib(), "cmi_spmv_hyb_dot_plan_" + _suffix(y))
    check(fnd(h, pitch, eAj, eAx, cAi, cAj, cAx, _ptr(x), _ptr(y), _ptr(w), _ptr(result), _ptr(workspace), _stream(stream)))


def set_index_compression(on):
    """cmi_set_index_compression: every later AUTO-kernel plan of Plan.csr / CsrMatrix tries the 16-bit column copy."""
    check(lib().cmi_set_index_compression(int(bool(on))))


def get_index_compression():
    return bool(lib().cmi_get_index_compression())


def spmv_hyb_plan(plan, pitch, ell_Aj, ell_Ax, coo_Ai, coo_Aj, coo_Ax, x, y, accumulate=False, stream=None):
    """cmi_spmv_hyb_plan_*: one launch when the plan found the COO part sorted by row (storage-order sums, y written once)."""
    import torch
    for t, n in ((ell_Aj, "ell_Aj"), (coo_Ai, "coo_Ai"), (coo_Aj, "coo_Aj")):
        _need(t, n, torch.int32)
    for t, n in ((ell_Ax, "ell_Ax"), (coo_Ax, "coo_Ax"), (x, "x"), (y, "y")):
        _need(t, n, y.dtype)
    if plan.format != FORMAT_HYB:
        raise ValueError("spmv_hyb_plan: not a HYB plan")
    if (coo_Ai.numel() != plan.coo_entries or coo_Aj.numel() != plan.coo_entries or coo_Ax.numel() != plan.coo_entries
            or ell_Aj.numel() < plan.width * pitch or ell_Ax.numel() < plan.width * pitch or pitch < plan.num_rows
            or x.numel() != plan.num_cols or y.numel() != plan.num_rows):
        raise ValueError("spmv_hyb_plan: array lengths do not match the plan's matrix shape")
    fn = getattr(lib(), "cmi_spmv_hyb_plan_" + _suffix(y))
    check(fn(plan.handle, pitch, _ptr(ell_Aj), _ptr(ell_Ax), _ptr(coo_Ai), _ptr(coo_Aj), _ptr(coo_Ax), _ptr(x), _ptr(y),
             int(bool(accumulate)), _stream(stream)))


# ------------------------------------------------------------------------------------------------
# tuning table
# ------------------------------------------------------------------------------------------------
def tuning_select(fmt, dtype, num_rows, num_cols, num_entries):
    c = Config()
    check(lib().cmi_tuning_select(fmt, dtype, num_rows, num_cols, num_entries, ctypes.byref(c)))
    return c


HYB_RULE_REFERENCE, HYB_RULE_COST, HYB_RULE_COST2 = 0, 1, 2


def count_zeros(values, stream=None):
    """explicit zeros in a device value array (cmi_count_zeros_*: the reference's ELL num_entries excludes them)"""
    c = c_int64()
    check(getattr(lib(), "cmi_count_zeros_" + _suffix(values))(values.numel(), _ptr(values), byref(c), _stream(stream)))
    return c.value


def tuning_hyb_rule(dtype):
    """(kind, relative_speed, threshold) of the HYB split rule for F64 / F32; without a table the reference's
    (HYB_RULE_REFERENCE, 3.0, 4096)."""
    k, rs, th = c_int(), c_double(), c_int64()
    check(lib().cmi_tuning_hyb_rule(dtype, byref(k), byref(rs), byref(th)))
    return k.value, rs.value, th.value


def tuning_waver_rule(dtype):
    """csr_waver's launch shape and AUTO gates for F64 / F32 (cmi_waver_rule): the table's "waver_rule", else the built-in one."""
    r = WaverRule()
    check(lib().cmi_tuning_waver_rule(dtype, byref(r)))
    return r


def tuning_set_waver_rule(dtype, items_per_thread=4, cap=0, xcd_swizzle=16, min_piece=None, min_entries=None):
    """layer a csr_waver rule on the table (tools/autotune_waver.py writes it; tuning_save persists it)"""
    r = WaverRule(int(items_per_thread), int(cap), int(xcd_swizzle), 0, float((2.2 if dtype == F64 else 1.9) if min_piece is None else min_piece),
                  int((4_400_000 if dtype == F64 else 6_400_000) if min_entries is None else min_entries))
    check(lib().cmi_tuning_set_waver_rule(dtype, byref(r)))


def tuning_hyb_light_speed(dtype):
    """HYB_RULE_COST2's fourth parameter: cost of a COO entry (ELL slots) while the COO part is light enough for one launch."""
    v = c_double()
    check(lib().cmi_tuning_hyb_light_speed(dtype, byref(v)))
    return v.value


def tuning_set_hyb_light_speed(dtype, light_speed):
    check(lib().cmi_tuning_set_hyb_light_speed(dtype, float(light_speed)))


def tuning_set_hyb_rule(dtype, kind, relative_speed, threshold):
    check(lib().cmi_tuning_set_hyb_rule(dtype, int(kind), float(relative_speed), int(threshold)))


def hyb_entries_per_row(dtype, num_rows, Ap, kind=-1, relative_speed=0.0, threshold=0, stream=None):
    """ELL width of the HYB split for the CSR matrix with row offsets Ap (device): the tuned rule (kind < 0), or an explicit
    rule -- (HYB_RULE_REFERENCE, 3.0, 4096) reproduces the reference's widths."""
    import torch
    _need(Ap, "Ap", torch.int32)
    code = F64 if dtype in (F64, torch.float64) else F32
    w = c_int64()
    check(lib().cmi_hyb_entries_per_row(code, num_rows, _ptr(Ap), int(kind), float(relative_speed), int(threshold), byref(w),
                                        _stream(stream)))
    return w.value


def tuning_set(fmt, dtype, mean_entries_per_row, cfg):
    check(lib().cmi_tuning_set(fmt, dtype, float(mean_entries_per_row), ctypes.byref(cfg)))


def tuning_load(path):
    check(lib().cmi_tuning_load(path.encode() if path else None))


def tuning_save(path):
    check(lib().cmi_tuning_save(path.encode()))


def tuning_clear():
    check(lib().cmi_tuning_clear())


# ------------------------------------------------------------------------------------------------
# on-device builders
# ------------------------------------------------------------------------------------------------
def poisson5pt_num_entries(m, n):
    return int(lib().cmi_poisson5pt_num_entries(m, n))


def poisson5pt_shard_entries(m, n, row_begin, row_end):
    return int(lib().cmi_poisson5pt_shard_entries(m, n, row_begin, row_end))


def poisson5pt_csr(m, n, Ap, Aj, Ax, row_begin=0, row_end=None, stream=None):
    import torch
    row_end = m * n if row_end is None else row_end
    _need(Ap, "Ap", torch.int32)
    _need(Aj, "Aj", torch.int32)
    _need(Ax, "Ax")
    nnz = poisson5pt_shard_entries(m, n, row_begin, row_end)
    if Ap.numel() != row_end - row_begin + 1 or Aj.numel() < nnz or Ax.numel() < nnz:
        raise ValueError("poisson5pt_csr: output arrays too small")
    fn = getattr(lib(), "cmi_poisson5pt_csr_" + _suffix(Ax))
    check(fn(m, n, row_begin, row_end, _ptr(Ap), _ptr(Aj), _ptr(Ax), _stream(stream)))


def poisson5pt_dia(m, n, pitch, offsets, values, stream=None):
    import torch
    _need(offsets, "offsets", torch.int32)
    _need(values, "values")
    if offsets.numel() != 5 or values.numel() < 5 * pitch:
        raise ValueError("poisson5pt_dia: output arrays too small")
    fn = getattr(lib(), "cmi_poisson5pt_dia_" + _suffix(values))
    check(fn(m, n, pitch, _ptr(offsets), _ptr(values), _stream(stream)))


def csr_to_ell(num_rows, Ap, Aj, Ax, width, pitch, ell_Aj, ell_Ax, stream=None):
    import torch
    for t, n in ((Ap, "Ap"), (Aj, "Aj"), (ell_Aj, "ell_Aj")):
        _need(t, n, torch.int32)
    _need(Ax, "Ax", ell_Ax.dtype)
    _need(ell_Ax, "ell_Ax")
    if ell_Aj.numel() < width * pitch or ell_Ax.numel() < width * pitch:
        raise ValueError("csr_to_ell: output arrays too small")
    fn = getattr(lib(), "cmi_csr_to_ell_" + _suffix(ell_Ax))
    check(fn(num_rows, _ptr(Ap), _ptr(Aj), _ptr(Ax), width, pitch, _ptr(ell_Aj), _ptr(ell_Ax), _stream(stream)))


def csr_to_hyb_coo(num_rows, Ap, Aj, Ax, width, coo_offsets, coo_Ai, coo_Aj, coo_Ax, stream=None):
    import torch
    for t, n in ((Ap, "Ap"), (Aj, "Aj"), (coo_offsets, "coo_offsets"), (coo_Ai, "coo_Ai"), (coo_Aj, "coo_Aj")):
        _need(t, n, torch.int32)
    _need(Ax, "Ax", coo_Ax.dtype)
    _need(coo_Ax, "coo_Ax")
    fn = getattr(lib(), "cmi_csr_to_hyb_coo_" + _suffix(coo_Ax))
    check(fn(num_rows, _ptr(Ap), _ptr(Aj), _ptr(Ax), width, _ptr(coo_offsets), _ptr(coo_Ai), _ptr(coo_Aj), _ptr(coo_Ax),
             _stream(stream)))


def csr_diagonals(num_rows, num_cols, Ap, Aj, slot_map, diag_list, stream=None):
    """Flags the occupied diagonals in slot_map, lists their offsets (unordered) in diag_list; returns the
    count (may exceed diag_list.numel(): list truncated).  Synchronises."""
    n = c_int64()
    check(lib().cmi_csr_diagonals(num_rows, num_cols, _ptr(Ap), _ptr(Aj), _ptr(slot_map), _ptr(diag_list), diag_list.numel(),
                                  byref(n), _stream(stream)))
    return n.value


def csr_to_dia(num_rows, num_cols, Ap, Aj, Ax, offsets, pitch, slot_map, values, stream=None):
    check(getattr(lib(), "cmi_csr_to_dia_" + _suffix(Ax))(num_rows, num_cols, _ptr(Ap), _ptr(Aj), _ptr(Ax), offsets.numel(), pitch,
                                                         _ptr(offsets), _ptr(slot_map), _ptr(values), _stream(stream)))


def csr_row_indices(num_rows, Ap, Ai, stream=None):
    import torch
    _need(Ap, "Ap", torch.int32)
    _need(Ai, "Ai", torch.int32)
    check(lib().cmi_csr_row_indices(num_rows, _ptr(Ap), _ptr(Ai), _stream(stream)))


def ell_row_lengths(num_rows, width, pitch, ell_Aj, row_lengths, stream=None):
    import torch
    _need(ell_Aj, "ell_Aj", torch.int32)
    _need(row_lengths, "row_lengths", torch.int32)
    check(lib().cmi_ell_row_lengths(num_rows, width, pitch, _ptr(ell_Aj), _ptr(row_lengths), _stream(stream)))


# ------------------------------------------------------------------------------------------------
# BLAS-1 (f64)
# ------------------------------------------------------------------------------------------------
def blas_workspace(device="cuda"):
    import torch
    return torch.empty(lib().cmi_blas_workspace_bytes() // 8, dtype=torch.float64, device=device)


def blas_axpy(alpha, x, y, stream=None):
    check(getattr(lib(), "cmi_blas_axpy_" + _suffix(y))(x.numel(), float(alpha), _ptr(x), _ptr(y), _stream(stream)))


def blas_axpby(alpha, x, beta, y, z, stream=None):
    check(getattr(lib(), "cmi_blas_axpby_" + _suffix(z))(x.numel(), float(alpha), _ptr(x), float(beta), _ptr(y), _ptr(z),
                                                      _stream(stream)))


def blas_copy(x, y, stream=None):
    check(getattr(lib(), "cmi_blas_copy_" + _suffix(y))(x.numel(), _ptr(x), _ptr(y), _stream(stream)))


def blas_fill(value, y, stream=None):
    check(getattr(lib(), "cmi_blas_fill_" + _suffix(y))(y.numel(), float(value), _ptr(y), _stream(stream)))


def blas_dot(x, y, result, workspace, stream=None):
    """result: 1-element device tensor of x's dtype."""
    check(getattr(lib(), "cmi_blas_dot_" + _suffix(x))(x.numel(), _ptr(x), _ptr(y), _ptr(result), _ptr(workspace),
                                                    _stream(stream)))


def blas_nrm2(x, result, workspace, stream=None):
    check(getattr(lib(), "cmi_blas_nrm2_" + _suffix(x))(x.numel(), _ptr(x), _ptr(result), _ptr(workspace), _stream(stream)))


def cg_update(rz, yp, p, y, x, r, rr_out, workspace, stream=None, mirror=None):
    """alpha = rz/yp (device scalars); x += alpha p (x None: left to cg_direction_x); r -= alpha y; rr_out = <r, r> -- one pass.
    mirror: a HostScalar that also receives <r, r> (no copy); its event is recorded behind the call."""
    check(getattr(lib(), "cmi_cg_update_" + _suffix(r))(r.numel(), _ptr(rz), _ptr(yp), _ptr(p) if x is not None else None, _ptr(y),
                                  _ptr(x) if x is not None else None, _ptr(r), _ptr(rr_out),
                                  mirror.ptr if mirror is not None else None, _ptr(workspace), _stream(stream)))
    if mirror is not None:
        mirror.record(stream)


def spmv_csr_dot_partials(plan, Ap, Aj, Ax, x, y, w, workspace, stream=None):
    """cmi_spmv_csr_dot_plan_partials_*: y <- A x, the per-tile partials of <y, w> left in the workspace for cg_update_fold.
    Returns their count (0: this plan's kernel cannot fuse the dot)."""
    n = c_int()
    check(getattr(lib(), "cmi_spmv_csr_dot_plan_partials_" + _suffix(y))(plan.handle, _ptr(Ap), _ptr(Aj), _ptr(Ax), _ptr(x), _ptr(y), _ptr(w),
                                                                         _ptr(workspace), byref(n), _stream(stream)))
    return n.value


def cg_update_fold(rz, yp_out, npartials_yp, y, r, workspace, stream=None):
    """cmi_cg_update_fold_*: yp <- fold of the workspace's <y, p> partials (also stored to yp_out); r -= (rz/yp) y; returns the
    number of <r, r> partials left in the workspace's second area."""
    n = c_int()
    check(getattr(lib(), "cmi_cg_update_fold_" + _suffix(r))(r.numel(), _ptr(rz), _ptr(yp_out), int(npartials_yp), _ptr(y), _ptr(r),
                                                             _ptr(workspace), byref(n), _stream(stream)))
    return n.value


def cg_direction_x_fold(rr_new_out, npartials_rr, rr_old, yp, r, p, x, workspace, stream=None, mirror=None):
    """cmi_cg_direction_x_fold_*: rr_new <- fold of the <r, r> partials (stored to rr_new_out and the HostScalar `mirror`);
    x += (rr_old/yp) p; p = r + (rr_new/rr_old) p."""
    check(getattr(lib(), "cmi_cg_direction_x_fold_" + _suffix(r))(r.numel(), _ptr(rr_new_out), mirror.ptr if mirror is not None else None,
                                                                  int(npartials_rr), _ptr(rr_old), _ptr(yp), _ptr(r), _ptr(p), _ptr(x),
                                                                  _ptr(workspace), _stream(stream)))
    if mirror is not None:
        mirror.record(stream)


class HostScalar:
    """8 bytes of page-locked, device-writable host memory (cmi_malloc_host) + the event that says a
    device value has landed in them: the CG convergence read without a blocking copy."""

    def __init__(self):
        L = lib()
        p, e = c_void_p(), c_void_p()
        check(L.cmi_malloc_host(byref(p), 8))
        self.ptr = p.value
        try:
            check(L.cmi_event_create(byref(e)))
        except Exception:
            L.cmi_free_host(self.ptr)
            self.ptr = None
            raise
        self.event = e.value

    def fetch(self, t, stream=None):
        """Queue device scalar -> host on the stream; returns at once."""
        check(lib().cmi_memcpy_d2h_async(self.ptr, _ptr(t), 8, _stream(stream)))
        self.record(stream)

    def record(self, stream=None):
        check(lib().cmi_event_record(self.event, _stream(stream)))

    def wait(self):
        check(lib().cmi_event_synchronize(self.event))
        return ctypes.c_double.from_address(self.ptr).value

    def close(self):
        if getattr(self, "ptr", None):
            L = lib()
            L.cmi_event_destroy(self.event)
            L.cmi_free_host(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


IPC_HANDLE_BYTES = 64
MAX_COPY_RANGES = 16


class DeviceBuffer:
    """An HBM allocation of its own (cmi_malloc, not a slice of torch's caching allocator), so that
    its IPC handle maps exactly this buffer in a peer process.  `tensor()` views it as a torch tensor."""

    def __init__(self, nbytes, device=None):
        import torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.nbytes = int(nbytes)
        p = c_void_p()
        with torch.cuda.device(self.device):
            check(lib().cmi_malloc(byref(p), max(self.nbytes, 8)))
            self.ptr = p.value
            check(lib().cmi_memset(self.ptr, 0, max(self.nbytes, 8), _stream(None)))

    def tensor(self, dtype):
        import torch
        item = torch.empty(0, dtype=dtype).element_size()
        typestr = {torch.float64: "<f8", torch.float32: "<f4", torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1"}[dtype]
        holder = _ArrayInterface(self, (self.nbytes // item,), typestr)
        t = torch.as_tensor(holder, device=self.device)
        if t.data_ptr() != self.ptr:
            raise CmiError(2, "DeviceBuffer.tensor: torch copied the buffer instead of viewing it")
        t._cmi_buffer = self  # the view keeps the allocation alive
        return t

    def ipc_handle(self):
        h = ctypes.create_string_buffer(IPC_HANDLE_BYTES)
        check(lib().cmi_ipc_get_handle(self.ptr, h))
        return h.raw

    def close(self):
        if getattr(self, "ptr", None):
            lib().cmi_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _ArrayInterface:
    def __init__(self, owner, shape, typestr):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (owner.ptr, False), "version": 2,
                                         "strides": None}


def device_can_access_peer(device, peer_device):
    flag = c_int()
    check(lib().cmi_device_can_access_peer(int(device), int(peer_device), byref(flag)))
    return bool(flag.value)


def ipc_open(handle):
    """Map a peer process's DeviceBuffer; returns the device address valid in THIS process."""
    p = c_void_p()
    check(lib().cmi_ipc_open_handle(ctypes.create_string_buffer(handle, IPC_HANDLE_BYTES), byref(p)))
    return p.value


def ipc_close(ptr):
    check(lib().cmi_ipc_close_handle(ptr))


COMM_ID_BYTES = 128
OP_SUM, OP_MAX, OP_MIN = 0, 1, 2


def comm_library_version():
    """ncclGetVersion of the RCCL the library binds at run time; raises when it cannot be bound (local call, not collective)."""
    v = c_int()
    check(lib().cmi_comm_library_version(byref(v)))
    return v.value


class Comm:
    """cmi_comm: the RCCL communicator behind the C-ABI (include/cusp_mi355x.h, csrc/comm.hip) -- what the sharded SpMV / CG use for
    their data path: all-gather of x (equal or unequal pieces), grouped send/recv halo exchange, all-reduce of CG's scalars.  Every
    call is enqueued on the given (default: torch's current) stream.  The 128-byte unique id comes from rank 0 and reaches the other
    ranks through `broadcast` (any out-of-band channel: a torch.distributed store here, a TCP socket in the C++ layer)."""

    def __init__(self, rank, world, broadcast=None):
        raw, err = None, None
        if rank == 0:
            ident = ctypes.create_string_buffer(COMM_ID_BYTES)
            try:
                check(lib().cmi_comm_unique_id(ident))
                raw = ident.raw
            except CmiError as e:  # (RCCL not loadable: the other ranks must hear about it, not wait for an id that never comes)
                err = e
        if world > 1:
            if broadcast is None:
                raise ValueError("Comm: a multi-rank communicator needs a broadcast function for the unique id")
            raw = broadcast(raw if rank == 0 else None)
        if raw is None:
            raise err if err is not None else CmiError(7, "CMI_ERROR_COMM: rank 0 could not make a communicator id")
        self._h = c_void_p()
        self.rank, self.world = rank, world
        check(lib().cmi_comm_create(ctypes.create_string_buffer(raw, COMM_ID_BYTES), rank, world, byref(self._h)))

    @classmethod
    def from_torch_distributed(cls, group=None):
        """One communicator per process group, made collectively: rank 0's id travels through torch.distributed's object broadcast
        (control plane only -- the data path is then this communicator's)."""
        import torch.distributed as dist
        key = id(group) if group is not None else 0
        if key not in _comms:
            rank, world = dist.get_rank(group), dist.get_world_size(group)

            def bcast(raw):
                box = [raw]
                dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                return box[0]
            _comms[key] = cls(rank, world, bcast)
        return _comms[key]

    @property
    def handle(self):
        return self._h

    def library_version(self):
        return comm_library_version()

    def allgather(self, send, recv, count, stream=None):
        """recv[r*count, +count) <- rank r's send[0, count); in place when send is recv's own slice."""
        check(getattr(lib(), "cmi_allgather_" + _suffix(recv))(self._h, _ptr(send), _ptr(recv), int(count), _stream(stream)))

    def allgatherv(self, send, recv, counts, displs, algorithm=1, stream=None):
        c = (c_int64 * self.world)(*[int(v) for v in counts])
        d = (c_int64 * self.world)(*[int(v) for v in displs])
        check(getattr(lib(), "cmi_allgatherv_" + _suffix(recv))(self._h, _ptr(send), _ptr(recv), c, d, int(algorithm), _stream(stream)))

    def halo_exchange(self, x_full, peers, send_lo, send_n, recv_lo, recv_n, stream=None):
        n = len(peers)
        arr = lambda v: (c_int64 * max(n, 1))(*[int(t) for t in v])  # noqa: E731
        check(getattr(lib(), "cmi_halo_exchange_" + _suffix(x_full))(self._h, _ptr(x_full), n, (c_int * max(n, 1))(*peers), arr(send_lo), arr(send_n),
                                                                     arr(recv_lo), arr(recv_n), _stream(stream)))

    def allreduce(self, t, op=OP_SUM, stream=None):
        """in-place all-reduce of a float64 device tensor (CG's scalars)"""
        import torch
        if t.dtype != torch.float64:
            raise TypeError("Comm.allreduce: float64 tensors only")
        check(lib().cmi_allreduce_f64(self._h, _ptr(t), _ptr(t), t.numel(), int(op), _stream(stream)))

    def barrier(self, stream=None):
        check(lib().cmi_comm_barrier(self._h, _stream(stream)))

    def allgather_host(self, record):
        """bytes -> list of every rank's bytes (small set-up records)"""
        out = ctypes.create_string_buffer(len(record) * self.world)
        check(lib().cmi_comm_allgather_host(self._h, record, out, len(record), _stream(None)))
        return [out.raw[i * len(record):(i + 1) * len(record)] for i in range(self.world)]

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.cmi_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_comms = {}


def csr_column_span(Aj, stream=None):
    """(smallest, largest) column index on the device; (0, -1) for no entries."""
    lo, hi = ctypes.c_int32(), ctypes.c_int32()
    check(lib().cmi_csr_column_span(Aj.numel(), _ptr(Aj), byref(lo), byref(hi), _stream(stream)))
    return lo.value, hi.value


class CopyRanges:
    """A fixed list of (src address, dst address, bytes) copied by ONE launch (cmi_copy_ranges)."""

    def __init__(self, ranges):
        ranges = [r for r in ranges if r[2] > 0]
        if len(ranges) > MAX_COPY_RANGES:
            raise ValueError(f"at most {MAX_COPY_RANGES} ranges per launch")
        n = len(ranges)
        self.n = n
        self.src = (c_void_p * max(n, 1))(*[r[0] for r in ranges])
        self.dst = (c_void_p * max(n, 1))(*[r[1] for r in ranges])
        self.bytes = (c_int64 * max(n, 1))(*[r[2] for r in ranges])

    def launch(self, stream=None):
        if self.n:
            check(lib().cmi_copy_ranges(self.n, self.src, self.dst, self.bytes, _stream(stream)))


def cg_direction(rr_new, rr_old, r, p, stream=None):
    """beta = rr_new/rr_old (device scalars); p = r + beta p."""
    check(getattr(lib(), "cmi_cg_direction_" + _suffix(p))(p.numel(), _ptr(rr_new), _ptr(rr_old), _ptr(r), _ptr(p), _stream(stream)))


def cg_direction_x(rr_new, rr_old, yp, r, p, x, stream=None):
    """alpha = rr_old/yp, beta = rr_new/rr_old (device scalars); x += alpha p (old p), then p = r + beta p -- one pass."""
    check(getattr(lib(), "cmi_cg_direction_x_" + _suffix(p))(p.numel(), _ptr(rr_new), _ptr(rr_old), _ptr(yp), _ptr(r), _ptr(p), _ptr(x),
                                                             _stream(stream)))


def blas_dotd(x, y, result, workspace, stream=None):
    """<x, y> as a DOUBLE in device memory whatever the vectors' type (the scalar type of the fused CG steps)."""
    import torch
    if x.dtype == torch.float64:
        check(lib().cmi_blas_dot_f64(x.numel(), _ptr(x), _ptr(y), _ptr(result), _ptr(workspace), _stream(stream)))
    else:
        check(lib().cmi_blas_dotd_f32(x.numel(), _ptr(x), _ptr(y), _ptr(result), _ptr(workspace), _stream(stream)))


def coo_row_offsets(num_rows, Ai, Ap, stream=None):
    """Ap[num_rows + 1] <- row offsets of row-sorted COO row indices (device); returns False (Ap unspecified) when the
    entries turn out not to be sorted by row."""
    import torch
    _need(Ai, "Ai", torch.int32)
    _need(Ap, "Ap", torch.int32)
    ok = ctypes.c_int(0)
    check(lib().cmi_coo_row_offsets(num_rows, Ai.numel(), _ptr(Ai), _ptr(Ap), byref(ok), _stream(stream)))
    return bool(ok.value)


def csr_interior_rows(num_rows, Ap, Aj, col_lo, col_hi, stream=None):
    """(first, last): rows [first, last) of a row block reference only columns in [col_lo, col_hi) -- the rows a sharded multiply can run
    while the halo is in flight (cmi_csr_interior_rows)."""
    import torch
    _need(Ap, "Ap", torch.int32)
    _need(Aj, "Aj", torch.int32)
    a, b = c_int64(0), c_int64(0)
    check(lib().cmi_csr_interior_rows(num_rows, _ptr(Ap), _ptr(Aj), col_lo, col_hi, byref(a), byref(b), _stream(stream)))
    return a.value, b.value


def coo_sort_by_row(num_rows, num_cols, Ai, Aj, Ax, and_column=False, stream=None):
    """In-place STABLE sort of COO entries by row (and_column: by (row, column)) on the device -- coo_matrix::sort_by_row
    (cusp/coo_matrix.h:208-212, cusp/sort.h:231,302).  Entries of one row keep their storage order."""
    import torch
    _need(Ai, "Ai", torch.int32)
    _need(Aj, "Aj", torch.int32)
    if not (Ai.numel() == Aj.numel() == Ax.numel()):
        raise ValueError("coo_sort_by_row: the three arrays differ in length")
    fn = getattr(lib(), "cmi_coo_sort_by_row_" + _suffix(Ax))
    check(fn(num_rows, num_cols, Ai.numel(), _ptr(Ai), _ptr(Aj), _ptr(Ax), 1 if and_column else 0, _stream(stream)))


def coo_is_sorted(num_rows, Ai, Aj=None, and_column=False, stream=None):
    """is_sorted_by_row / is_sorted_by_row_and_column (cusp/coo_matrix.h:218-224) on the device."""
    import torch
    _need(Ai, "Ai", torch.int32)
    if and_column:
        _need(Aj, "Aj", torch.int32)
    ok = ctypes.c_int(0)
    check(lib().cmi_coo_is_sorted(num_rows, Ai.numel(), _ptr(Ai), _ptr(Aj) if Aj is not None else None, 1 if and_column else 0, byref(ok), _stream(stream)))
    return bool(ok.value)


def ell_to_csr(num_rows, width, pitch, ell_Aj, ell_Ax, stream=None):
    """(Ap, Aj, Ax) of the entries of an ELL matrix with a valid column, row by row, built on the device."""
    import torch
    fn = getattr(lib(), "cmi_ell_to_csr_" + _suffix(ell_Ax))
    dev = ell_Ax.device
    Ap = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    n = c_int64(0)
    check(fn(num_rows, width, pitch, _ptr(ell_Aj), _ptr(ell_Ax), _ptr(Ap), None, None, 0, byref(n), _stream(stream)))
    Aj = torch.empty(n.value, dtype=torch.int32, device=dev)
    Ax = torch.empty(n.value, dtype=ell_Ax.dtype, device=dev)
    if n.value:
        check(fn(num_rows, width, pitch, _ptr(ell_Aj), _ptr(ell_Ax), _ptr(Ap), _ptr(Aj), _ptr(Ax), n.value, byref(n), _stream(stream)))
    return Ap, Aj, Ax


def dia_to_csr(num_rows, num_cols, num_diagonals, pitch, offsets, values, stream=None):
    """(Ap, Aj, Ax) of the non-zero entries of a DIA matrix with a valid column, row by row, built on the device."""
    import torch
    fn = getattr(lib(), "cmi_dia_to_csr_" + _suffix(values))
    dev = values.device
    Ap = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    n = c_int64(0)
    check(fn(num_rows, num_cols, num_diagonals, pitch, _ptr(offsets), _ptr(values), _ptr(Ap), None, None, 0, byref(n), _stream(stream)))
    Aj = torch.empty(n.value, dtype=torch.int32, device=dev)
    Ax = torch.empty(n.value, dtype=values.dtype, device=dev)
    if n.value:
        check(fn(num_rows, num_cols, num_diagonals, pitch, _ptr(offsets), _ptr(values), _ptr(Ap), _ptr(Aj), _ptr(Ax), n.value, byref(n),
                 _stream(stream)))
    return Ap, Aj, Ax


def hyb_to_csr(num_rows, width, pitch, ell_Aj, ell_Ax, coo_Ai, coo_Aj, coo_Ax, stream=None):
    """(Ap, Aj, Ax): each row's ELL entries, then its COO entries (row-sorted COO part), built on the device."""
    import torch
    fn = getattr(lib(), "cmi_hyb_to_csr_" + _suffix(ell_Ax))
    dev = ell_Ax.device
    Ap = torch.empty(num_rows + 1, dtype=torch.int32, device=dev)
    n = c_int64(0)
    args = (num_rows, width, pitch, _ptr(ell_Aj), _ptr(ell_Ax), coo_Ai.numel(), _ptr(coo_Ai), _ptr(coo_Aj), _ptr(coo_Ax), _ptr(Ap))
    check(fn(*args, None, None, 0, byref(n), _stream(stream)))
    Aj = torch.empty(n.value, dtype=torch.int32, device=dev)
    Ax = torch.empty(n.value, dtype=ell_Ax.dtype, device=dev)
    if n.value:
        check(fn(*args, _ptr(Aj), _ptr(Ax), n.value, byref(n), _stream(stream)))
    return Ap, Aj, Ax
