"""ctypes loaders for the two CPU checkers (numpy arrays in, numpy arrays out)."""
import ctypes
import os
import subprocess
from ctypes import c_float, c_int, c_int64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libcusp_ref.so")


def build():
    """Compile liboracle.so (gcc) and, when /root/reference is present, _ref/libcusp_ref.so."""
    out = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("building the oracle failed:\n" + out.stdout + out.stderr)


def have_reference():
    return os.path.exists(_REF)


def _p(a):
    return None if a is None else c_void_p(a.ctypes.data)


def _suf(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64"
    if dtype == np.float32:
        return "f32"
    raise TypeError(dtype)


def _i32(a):
    a = np.ascontiguousarray(a)
    assert a.dtype == np.int32, a.dtype
    return a


def fill_x(n, dtype=np.float64):
    """x[i] = ((uint32)(i*2654435761u) % 1000)/997.0 - 0.5; numpy restatement used
    for inputs, checked against orc_fill_x in the tests."""
    i = np.arange(n, dtype=np.uint64)
    h = (i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    return ((h % np.uint64(1000)).astype(np.float64) / 997.0 - 0.5).astype(dtype)


class _SpmvMixin:
    """y = A x with numpy arrays; `y0` (optional) switches to y = y0 + A x."""
    _prefix = None
    _lib = None

    def _call(self, name, dtype, *args):
        return getattr(self._lib, f"{self._prefix}_{name}_{_suf(dtype)}")(*args)

    def _y(self, rows, dtype, y0):
        if y0 is None:
            return np.full(rows, 10.0, dtype=dtype), 0  # poisoned, as testing/multiply.cu:391 does
        return np.array(y0, dtype=dtype, copy=True), 1


class Oracle(_SpmvMixin):
    """Plain-C restatement (oracle/spmv_oracle.c)."""
    _prefix = "orc"

    def __init__(self):
        if not os.path.exists(_ORACLE):
            build()
        self._lib = L = ctypes.CDLL(_ORACLE)
        L.orc_optimal_entries_per_row.restype = c_int64
        L.orc_optimal_entries_per_row.argtypes = [c_int64, c_void_p, c_float, c_int64]
        L.orc_csr_diagonals.restype = c_int64
        L.orc_csr_diagonals.argtypes = [c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64]
        L.orc_csr_row_indices.argtypes = [c_int64, c_void_p, c_void_p]
        L.orc_num_threads.restype = c_int
        L.orc_set_num_threads.argtypes = [c_int]
        # size the OpenMP team to the CPU share of this process (a one-GPU box gives 16 cores,
        # whatever the affinity mask says)
        try:
            L.orc_set_num_threads(min(len(os.sched_getaffinity(0)), 16))
        except AttributeError:
            pass
        for s in ("f64", "f32"):
            getattr(L, f"orc_spmv_csr_{s}").argtypes = [c_int64] + [c_void_p] * 5 + [c_int]
            getattr(L, f"orc_spmv_csr_omp_{s}").argtypes = [c_int64] + [c_void_p] * 5 + [c_int]
            getattr(L, f"orc_spmv_coo_{s}").argtypes = [c_int64, c_int64] + [c_void_p] * 5 + [c_int]
            getattr(L, f"orc_spmv_ell_{s}").argtypes = [c_int64, c_int64, c_int64] + [c_void_p] * 4 + [c_int]
            getattr(L, f"orc_spmv_dia_{s}").argtypes = [c_int64, c_int64, c_int64, c_int64] + [c_void_p] * 4 + [c_int]
            getattr(L, f"orc_spmv_hyb_{s}").argtypes = [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64] + \
                [c_void_p] * 5 + [c_int]
            getattr(L, f"orc_poisson5pt_dia_{s}").restype = c_int64
            getattr(L, f"orc_poisson5pt_dia_{s}").argtypes = [c_int64, c_int64, c_void_p, c_void_p]
            getattr(L, f"orc_dia_to_csr_{s}").restype = c_int64
            getattr(L, f"orc_dia_to_csr_{s}").argtypes = [c_int64, c_int64, c_int64] + [c_void_p] * 5
            getattr(L, f"orc_csr_to_ell_{s}").argtypes = [c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                                          c_void_p, c_void_p]
            getattr(L, f"orc_csr_to_hyb_coo_{s}").restype = c_int64
            getattr(L, f"orc_csr_to_hyb_coo_{s}").argtypes = [c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                                              c_void_p, c_void_p, c_void_p]
            getattr(L, f"orc_csr_to_dia_{s}").argtypes = [c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                                          c_int64, c_void_p]
            getattr(L, f"orc_fill_x_{s}").argtypes = [c_int64, c_void_p]

    def num_threads(self):
        return int(self._lib.orc_num_threads())

    def bench_csr_omp(self, num_cols, Ap, Aj, Ax, x, threads=None, seconds=4.0, min_reps=3, max_reps=500):
        """The multi-core CPU baseline (orc_bench_csr_omp_f64): OpenMP row-parallel CSR SpMV on copies first-touched in
        parallel by their users, threads pinned one per allowed CPU, timed inside C.  Returns a dict."""
        Ap, Aj = _i32(Ap), _i32(Aj)
        Ax = np.ascontiguousarray(Ax, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        rows = len(Ap) - 1
        if threads is None:
            threads = self.num_threads()
        y = np.empty(rows, dtype=np.float64)
        reps, pinned = c_int(0), c_int(0)
        cpus = (c_int * threads)(*([-1] * threads))
        fn = self._lib.orc_bench_csr_omp_f64
        fn.restype = ctypes.c_double
        fn.argtypes = [c_int64, c_int64] + [c_void_p] * 5 + [c_int, ctypes.c_double, c_int, c_int, c_void_p, c_void_p, c_void_p]
        sec = fn(rows, int(num_cols), _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y), int(threads), float(seconds), int(min_reps),
                 int(max_reps), ctypes.addressof(reps), ctypes.addressof(cpus), ctypes.addressof(pinned))
        if sec < 0:
            raise MemoryError("orc_bench_csr_omp_f64: allocation failed")
        return {"seconds_per_spmv": sec, "reps": reps.value, "threads": threads, "cpus": list(cpus),
                "pinned": bool(pinned.value), "y": y}

    def set_num_threads(self, n):
        self._lib.orc_set_num_threads(int(n))

    # ---- SpMV ----
    def spmv_csr(self, Ap, Aj, Ax, x, y0=None, omp=False):
        Ap, Aj = _i32(Ap), _i32(Aj)
        y, acc = self._y(len(Ap) - 1, Ax.dtype, y0)
        self._call("spmv_csr_omp" if omp else "spmv_csr", Ax.dtype, len(Ap) - 1, _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y), acc)
        return y

    def spmv_coo(self, rows, Ai, Aj, Ax, x, y0=None):
        Ai, Aj = _i32(Ai), _i32(Aj)
        y, acc = self._y(rows, Ax.dtype, y0)
        self._call("spmv_coo", Ax.dtype, rows, len(Ax), _p(Ai), _p(Aj), _p(Ax), _p(x), _p(y), acc)
        return y

    def spmv_ell(self, rows, width, pitch, Aj, Ax, x, y0=None):
        Aj = _i32(Aj)
        y, acc = self._y(rows, Ax.dtype, y0)
        self._call("spmv_ell", Ax.dtype, rows, width, pitch, _p(Aj), _p(Ax), _p(x), _p(y), acc)
        return y

    def spmv_dia(self, rows, cols, pitch, offsets, vals, x, y0=None):
        offsets = _i32(offsets)
        y, acc = self._y(rows, vals.dtype, y0)
        self._call("spmv_dia", vals.dtype, rows, cols, len(offsets), pitch, _p(offsets), _p(vals), _p(x), _p(y), acc)
        return y

    def spmv_hyb(self, rows, width, pitch, eAj, eAx, cAi, cAj, cAx, x, y0=None):
        y, acc = self._y(rows, eAx.dtype, y0)
        self._call("spmv_hyb", eAx.dtype, rows, width, pitch, _p(_i32(eAj)), _p(eAx), len(cAx), _p(_i32(cAi)),
                   _p(_i32(cAj)), _p(cAx), _p(x), _p(y), acc)
        return y

    # ---- generator / conversions ----
    def poisson5pt_dia(self, m, n, dtype=np.float64):
        N = m * n
        off = np.empty(5, np.int32)
        vals = np.empty(5 * N, dtype)
        nnz = self._call("poisson5pt_dia", dtype, m, n, _p(off), _p(vals))
        return off, vals, int(nnz)

    def dia_to_csr(self, rows, pitch, offsets, vals, nnz):
        Ap = np.empty(rows + 1, np.int32)
        Aj = np.empty(nnz, np.int32)
        Ax = np.empty(nnz, vals.dtype)
        got = self._call("dia_to_csr", vals.dtype, rows, len(offsets), pitch, _p(_i32(offsets)), _p(vals), _p(Ap), _p(Aj), _p(Ax))
        assert got == nnz, (got, nnz)
        return Ap, Aj, Ax

    def poisson5pt_csr(self, m, n, dtype=np.float64):
        off, vals, nnz = self.poisson5pt_dia(m, n, dtype)
        return self.dia_to_csr(m * n, m * n, off, vals, nnz)

    def csr_row_indices(self, Ap):
        Ap = _i32(Ap)
        Ai = np.empty(int(Ap[-1]), np.int32)
        self._lib.orc_csr_row_indices(len(Ap) - 1, _p(Ap), _p(Ai))
        return Ai

    def csr_to_ell(self, Ap, Aj, Ax, width, alignment=32):
        rows = len(Ap) - 1
        pitch = alignment * ((rows + alignment - 1) // alignment)
        eAj = np.empty(width * pitch, np.int32)
        eAx = np.empty(width * pitch, Ax.dtype)
        self._call("csr_to_ell", Ax.dtype, rows, _p(_i32(Ap)), _p(_i32(Aj)), _p(Ax), width, pitch, _p(eAj), _p(eAx))
        return pitch, eAj, eAx

    def csr_to_hyb(self, Ap, Aj, Ax, width, alignment=32):
        pitch, eAj, eAx = self.csr_to_ell(Ap, Aj, Ax, width, alignment)
        lens = np.diff(Ap)
        ncoo = int(np.maximum(lens - width, 0).sum())
        cAi, cAj, cAx = np.empty(ncoo, np.int32), np.empty(ncoo, np.int32), np.empty(ncoo, Ax.dtype)
        got = self._call("csr_to_hyb_coo", Ax.dtype, len(Ap) - 1, _p(_i32(Ap)), _p(_i32(Aj)), _p(Ax), width, _p(cAi), _p(cAj), _p(cAx))
        assert got == ncoo
        return pitch, eAj, eAx, cAi, cAj, cAx

    def csr_to_dia(self, rows, cols, Ap, Aj, Ax, alignment=32):
        Ap, Aj = _i32(Ap), _i32(Aj)
        nd = self._lib.orc_csr_diagonals(rows, cols, _p(Ap), _p(Aj), None, 0)
        off = np.empty(nd, np.int32)
        self._lib.orc_csr_diagonals(rows, cols, _p(Ap), _p(Aj), _p(off), nd)
        pitch = alignment * ((rows + alignment - 1) // alignment)
        vals = np.empty(nd * pitch, Ax.dtype)
        self._call("csr_to_dia", Ax.dtype, rows, _p(Ap), _p(Aj), _p(Ax), nd, _p(off), pitch, _p(vals))
        return pitch, off, vals

    def optimal_entries_per_row(self, Ap, relative_speed=3.0, breakeven_threshold=4096):
        Ap = _i32(Ap)
        return int(self._lib.orc_optimal_entries_per_row(len(Ap) - 1, _p(Ap), relative_speed, breakeven_threshold))

    def fill_x(self, n, dtype=np.float64):
        x = np.empty(n, dtype)
        self._call("fill_x", dtype, n, _p(x))
        return x


class Reference(_SpmvMixin):
    """The reference's own sequential kernels (oracle/_ref/libcusp_ref.so)."""
    _prefix = "ref"

    def __init__(self):
        if not os.path.exists(_REF):
            raise FileNotFoundError(_REF + " (built only where /root/reference exists: make -C oracle ref)")
        self._lib = L = ctypes.CDLL(_REF)
        L.ref_describe.restype = ctypes.c_char_p
        for s in ("f64", "f32"):
            getattr(L, f"ref_spmv_csr_{s}").argtypes = [c_int64] * 3 + [c_void_p] * 5 + [c_int]
            getattr(L, f"ref_spmv_coo_{s}").argtypes = [c_int64] * 3 + [c_void_p] * 5 + [c_int]
            getattr(L, f"ref_spmv_ell_{s}").argtypes = [c_int64] * 4 + [c_void_p] * 4 + [c_int]
            getattr(L, f"ref_spmv_dia_{s}").argtypes = [c_int64] * 4 + [c_void_p] * 4 + [c_int]
            getattr(L, f"ref_spmv_hyb_{s}").argtypes = [c_int64] * 4 + [c_void_p, c_void_p, c_int64] + [c_void_p] * 5 + [c_int]

    def describe(self):
        return self._lib.ref_describe().decode()

    def spmv_csr(self, cols, Ap, Aj, Ax, x, y0=None):
        Ap, Aj = _i32(Ap), _i32(Aj)
        y, acc = self._y(len(Ap) - 1, Ax.dtype, y0)
        self._call("spmv_csr", Ax.dtype, len(Ap) - 1, cols, len(Ax), _p(Ap), _p(Aj), _p(Ax), _p(x), _p(y), acc)
        return y

    def spmv_coo(self, rows, cols, Ai, Aj, Ax, x, y0=None):
        y, acc = self._y(rows, Ax.dtype, y0)
        self._call("spmv_coo", Ax.dtype, rows, cols, len(Ax), _p(_i32(Ai)), _p(_i32(Aj)), _p(Ax), _p(x), _p(y), acc)
        return y

    def spmv_ell(self, rows, cols, width, pitch, Aj, Ax, x, y0=None):
        y, acc = self._y(rows, Ax.dtype, y0)
        self._call("spmv_ell", Ax.dtype, rows, cols, width, pitch, _p(_i32(Aj)), _p(Ax), _p(x), _p(y), acc)
        return y

    def spmv_dia(self, rows, cols, pitch, offsets, vals, x, y0=None):
        y, acc = self._y(rows, vals.dtype, y0)
        self._call("spmv_dia", vals.dtype, rows, cols, len(offsets), pitch, _p(_i32(offsets)), _p(vals), _p(x), _p(y), acc)
        return y

    def spmv_hyb(self, rows, cols, width, pitch, eAj, eAx, cAi, cAj, cAx, x, y0=None):
        y, acc = self._y(rows, eAx.dtype, y0)
        self._call("spmv_hyb", eAx.dtype, rows, cols, width, pitch, _p(_i32(eAj)), _p(eAx), len(cAx), _p(_i32(cAi)),
                   _p(_i32(cAj)), _p(cAx), _p(x), _p(y), acc)
        return y
