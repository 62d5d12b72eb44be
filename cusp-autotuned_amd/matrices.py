"""Device-memory containers with the public members of the reference's
cusp::{csr,coo,ell,dia,hyb}_matrix<int, T, cusp::device_memory> and a ``multiply(A, x, y)`` that
dispatches on the container's format the way cusp::multiply does
(reference cusp/detail/multiply.inl:27-105 -> cusp/system/detail/generic/multiply.inl:173-191).

Storage is torch tensors on the HIP device (plumbing for device memory only); every multiply goes
through the C-ABI library -- see binding.py.  The C++ users' equivalent of this file is the
header-only layer in include/cusp/.
"""
from dataclasses import dataclass, field

from . import binding as B


def _round_up(n, k):
    """cusp::detail::round_up (reference cusp/detail/utils.h:24-28)."""
    return k * ((n + k - 1) // k)


class _Planned:
    """The container owns its plan (binding.Plan): made at the first multiply without an explicit config -- that call
    synchronises once -- and kept until the structure arrays are replaced (`invalidate()` after editing them in place)."""

    _plan = None
    _plan_key = None

    def _plan_for(self, fmt_code, index_array, stream, create=True):
        # (_version: torch's in-place edit counter -- a .copy_() into the same storage is a different structure)
        key = (index_array.data_ptr(), index_array._version, self.num_rows, self.num_cols, self.num_entries, self.values.dtype)
        if self._plan_key != key:
            if not create:
                return None
            self._plan = B.Plan(fmt_code, self.values.dtype, self.num_rows, self.num_cols, self.num_entries, index_array, None, stream)
            self._plan_key = key
        return self._plan

    def invalidate(self):
        self._plan_key = None
        self._plan = None


@dataclass
class CsrMatrix(_Planned):
    """reference cusp/csr_matrix.h:107-208: row_offsets, column_indices, values."""
    num_rows: int
    num_cols: int
    num_entries: int
    row_offsets: object
    column_indices: object
    values: object
    format = "csr"

    def plan(self, stream=None, create=True, compress=None):
        """compress=True asks for the 16-bit column copy (CSR_STREAM_C16; granted only if every tile qualifies -- see
        plan().config().kernel); None follows set_index_compression()."""
        if compress is not None:
            self._compress = bool(compress)  # sticky: later multiplies of this matrix keep the choice
        compress = getattr(self, "_compress", None)
        key = (self.row_offsets.data_ptr(), self.row_offsets._version, self.column_indices.data_ptr(), self.column_indices._version,
               self.num_rows, self.num_cols, self.num_entries, self.values.dtype,
               compress if compress is not None else B.get_index_compression())
        if self._plan_key != key:
            if not create:
                return None
            cfg = B.Config(kernel=B.CSR_STREAM_C16) if compress else None
            if compress is False and B.get_index_compression():
                self._plan = B.Plan(B.FORMAT_CSR, self.values.dtype, self.num_rows, self.num_cols, self.num_entries, self.row_offsets, None, stream)
            else:
                self._plan = B.Plan.csr(self.values.dtype, self.num_rows, self.num_cols, self.row_offsets, self.column_indices, cfg, stream)
            self._plan_key = key
        return self._plan


@dataclass
class CooMatrix(_Planned):
    """reference cusp/coo_matrix.h:116-224: row_indices, column_indices, values."""
    num_rows: int
    num_cols: int
    num_entries: int
    row_indices: object
    column_indices: object
    values: object
    format = "coo"

    def plan(self, stream=None, create=True):
        # made from BOTH index arrays (cmi_plan_create_coo, round 4): sorted entries get a CSR sub-plan that may hold a copy derived from
        # the columns (the run-compressed pieces), so the key carries both arrays' in-place versions
        key = (self.row_indices.data_ptr(), self.row_indices._version, self.column_indices.data_ptr(), self.column_indices._version,
               self.num_rows, self.num_cols, self.num_entries, self.values.dtype)
        if self._plan_key != key:
            if not create:
                return None
            self._plan = B.Plan.coo(self.values.dtype, self.num_rows, self.num_cols, self.row_indices, self.column_indices, None, stream)
            self._plan_key = key
        return self._plan

    # reference cusp/coo_matrix.h:208-224 (device_memory: cmi_coo_sort_by_row_*, stable; the plan is keyed on the row
    # indices' in-place version, which the sort bumps below)
    def sort_by_row(self, stream=None):
        B.coo_sort_by_row(self.num_rows, self.num_cols, self.row_indices, self.column_indices, self.values, False, stream)
        self.row_indices.add_(0)  # torch's version counter: the library wrote through the raw pointer

    def sort_by_row_and_column(self, stream=None):
        B.coo_sort_by_row(self.num_rows, self.num_cols, self.row_indices, self.column_indices, self.values, True, stream)
        self.row_indices.add_(0)

    def is_sorted_by_row(self, stream=None):
        return B.coo_is_sorted(self.num_rows, self.row_indices, None, False, stream)

    def is_sorted_by_row_and_column(self, stream=None):
        return B.coo_is_sorted(self.num_rows, self.row_indices, self.column_indices, True, stream)


@dataclass
class EllMatrix:
    """reference cusp/ell_matrix.h:119-228: column-major num_rows x num_entries_per_row arrays with
    leading dimension pitch = round_up(num_rows, alignment); padding column = invalid_index (-1).
    row_lengths (optional) is the fork's ELLR extension, cusp/ktt/ellr_matrix.h:17-90."""
    num_rows: int
    num_cols: int
    num_entries: int
    num_entries_per_row: int
    pitch: int
    column_indices: object
    values: object
    row_lengths: object = None
    invalid_index = -1
    format = "ell"


@dataclass
class DiaMatrix:
    """reference cusp/dia_matrix.h:120-226: diagonal_offsets + column-major values (pitch)."""
    num_rows: int
    num_cols: int
    num_entries: int
    pitch: int
    diagonal_offsets: object
    values: object
    format = "dia"


@dataclass
class HybMatrix:
    """reference cusp/hyb_matrix.h:142-246: ell + coo parts.  Owns a HYB plan (binding.Plan.hyb): with the COO part sorted by
    row -- what every conversion produces -- the whole multiply is one launch."""
    num_rows: int
    num_cols: int
    num_entries: int
    ell: EllMatrix
    coo: CooMatrix
    format = "hyb"
    _plan = None
    _plan_key = None
    _plan_args = None

    def plan(self, stream=None, create=True):
        e, c = self.ell, self.coo
        key = (c.row_indices.data_ptr(), c.row_indices._version, c.column_indices.data_ptr(), c.values.data_ptr(), e.column_indices.data_ptr(),
               e.values.data_ptr(), e.pitch, self.num_rows, self.num_cols, e.num_entries_per_row, c.num_entries, e.values.dtype)
        if self._plan_key != key:
            if not create:
                return None
            self._plan = B.Plan.hyb(e.values.dtype, self.num_rows, self.num_cols, e.num_entries_per_row, c.row_indices, stream=stream)
            self._plan_args = B.hyb_plan_args(self._plan, e.pitch, e.column_indices, e.values, c.row_indices, c.column_indices, c.values)
            self._plan_key = key
        return self._plan

    def invalidate(self):
        self._plan_key = None
        self._plan = None
        self._plan_args = None


def _capturing(stream):
    import torch
    try:
        return torch.cuda.is_current_stream_capturing() if stream is None else stream.is_capturing()
    except AttributeError:
        return False


def multiply(A, x, y, accumulate=False, cfg=None, stream=None):
    """y = A*x (or y += A*x).  Mirrors the 3-argument cusp::multiply (cusp/multiply.h:40).  Without an explicit config
    CSR, COO and HYB matrices multiply through their plan (made once, at the first such call; not while a stream capture is
    recording -- then the plan-less entry point runs the table's kernel)."""
    plan = None
    if isinstance(A, (CsrMatrix, CooMatrix)) and cfg is None and A.num_entries > 0:
        plan = A.plan(stream, create=not _capturing(stream))  # an existing plan is used inside a capture, none is made there
    if plan is not None and isinstance(A, CsrMatrix):
        B.spmv_csr_plan(plan, A.row_offsets, A.column_indices, A.values, x, y, accumulate, stream)
    elif plan is not None:
        B.spmv_coo_plan(plan, A.row_indices, A.column_indices, A.values, x, y, accumulate, stream)
    elif isinstance(A, CsrMatrix):
        B.spmv_csr(A.num_rows, A.num_cols, A.row_offsets, A.column_indices, A.values, x, y, accumulate, cfg, stream)
    elif isinstance(A, CooMatrix):
        B.spmv_coo(A.num_rows, A.num_cols, A.row_indices, A.column_indices, A.values, x, y, accumulate, cfg, stream)
    elif isinstance(A, EllMatrix):
        B.spmv_ell(A.num_rows, A.num_cols, A.num_entries_per_row, A.pitch, A.column_indices, A.values, x, y,
                   A.row_lengths, accumulate, cfg, stream)
    elif isinstance(A, DiaMatrix):
        B.spmv_dia(A.num_rows, A.num_cols, A.diagonal_offsets.numel(), A.pitch, A.diagonal_offsets, A.values, x, y,
                   accumulate, cfg, stream)
    elif isinstance(A, HybMatrix):
        e, c = A.ell, A.coo
        hplan = A.plan(stream, create=not _capturing(stream)) if cfg is None and c.num_entries > 0 else None
        if hplan is not None:
            # COO part sorted by row (what the conversions produce): ONE launch, y written once, the host loops' bits
            B.spmv_hyb_plan_args(A._plan_args, x, y, accumulate, stream)  # (arrays validated when the plan was made)
        else:
            B.spmv_hyb(A.num_rows, A.num_cols, e.num_entries_per_row, e.pitch, e.column_indices, e.values,
                       c.row_indices, c.column_indices, c.values, x, y, accumulate, cfg, None, stream)
    else:
        raise TypeError(f"multiply: unsupported matrix type {type(A).__name__}")
    return y


# ------------------------------------------------------------------------------------------------
# builders (setup, not the hot path)
# ------------------------------------------------------------------------------------------------
def poisson5pt(m, n, fmt="csr", dtype=None, device="cuda", row_begin=0, row_end=None, ell_alignment=32):
    """cusp::gallery::poisson5pt(A, m, n) built directly in HBM (reference
    cusp/gallery/detail/poisson.inl:29-47).  fmt in {csr, coo, ell, dia, hyb}.  row_begin/row_end
    select a row-block shard (CSR/COO/ELL only) with GLOBAL column indices."""
    import torch
    dtype = dtype or torch.float64
    N = m * n
    row_end = N if row_end is None else row_end
    if fmt == "dia":
        if row_begin != 0 or row_end != N:
            raise ValueError("dia shards are not supported")
        pitch = N  # the gallery builds DIA with pitch = num_rows (stencil.inl:174)
        off = torch.empty(5, dtype=torch.int32, device=device)
        vals = torch.empty(5 * pitch, dtype=dtype, device=device)
        B.poisson5pt_dia(m, n, pitch, off, vals)
        return DiaMatrix(N, N, B.poisson5pt_num_entries(m, n), pitch, off, vals)
    rows = row_end - row_begin
    nnz = B.poisson5pt_shard_entries(m, n, row_begin, row_end)
    Ap = torch.empty(rows + 1, dtype=torch.int32, device=device)
    Aj = torch.empty(nnz, dtype=torch.int32, device=device)
    Ax = torch.empty(nnz, dtype=dtype, device=device)
    B.poisson5pt_csr(m, n, Ap, Aj, Ax, row_begin, row_end)
    csr = CsrMatrix(rows, N, nnz, Ap, Aj, Ax)
    return csr if fmt == "csr" else convert(csr, fmt, ell_alignment=ell_alignment)


def convert(csr, fmt, num_entries_per_row=None, ell_alignment=32):
    """CSR -> {coo, ell, hyb, dia} and row-sorted COO -> CSR on the device (reference conversions/csr_to_other.h:56-306,
    coo_to_other.h).
    For ELL the width defaults to the longest row; for HYB to the tuned split rule (cmi_hyb_entries_per_row: the
    reference's compute_optimal_entries_per_row with the pair measured on MI355X, tools/autotune_hyb.py)."""
    import torch
    if isinstance(csr, CooMatrix) and fmt == "csr":
        # row-sorted COO -> CSR on the device: the offsets from the row indices in one pass (order checked on the way)
        coo = csr
        Ap = torch.empty(coo.num_rows + 1, dtype=torch.int32, device=coo.values.device)
        if not B.coo_row_offsets(coo.num_rows, coo.row_indices, Ap):
            # any order: a copy, sorted by row on the device (stable: coo_matrix::sort_by_row), then its offsets -- what the
            # reference's coo -> csr does (conversions/coo_to_other.h sorts first)
            coo = CooMatrix(coo.num_rows, coo.num_cols, coo.num_entries, coo.row_indices.clone(), coo.column_indices.clone(), coo.values.clone())
            coo.sort_by_row()  # (raises on a row index outside the matrix)
            if not B.coo_row_offsets(coo.num_rows, coo.row_indices, Ap):
                raise ValueError("convert: COO row indices outside the matrix")
        return CsrMatrix(coo.num_rows, coo.num_cols, coo.num_entries, Ap, coo.column_indices, coo.values)
    if isinstance(csr, EllMatrix) and fmt == "csr":
        Ap, Aj, Ax = B.ell_to_csr(csr.num_rows, csr.num_entries_per_row, csr.pitch, csr.column_indices, csr.values)
        return CsrMatrix(csr.num_rows, csr.num_cols, Aj.numel(), Ap, Aj, Ax)
    if isinstance(csr, DiaMatrix) and fmt == "csr":
        Ap, Aj, Ax = B.dia_to_csr(csr.num_rows, csr.num_cols, csr.diagonal_offsets.numel(), csr.pitch, csr.diagonal_offsets, csr.values)
        return CsrMatrix(csr.num_rows, csr.num_cols, Aj.numel(), Ap, Aj, Ax)
    if isinstance(csr, HybMatrix) and fmt == "csr":
        e, c = csr.ell, csr.coo
        Ap, Aj, Ax = B.hyb_to_csr(csr.num_rows, e.num_entries_per_row, e.pitch, e.column_indices, e.values, c.row_indices, c.column_indices, c.values)
        return CsrMatrix(csr.num_rows, csr.num_cols, Aj.numel(), Ap, Aj, Ax)
    if not isinstance(csr, CsrMatrix):
        raise TypeError("convert: source must be a CsrMatrix (or a Coo / Ell / Dia / Hyb matrix for fmt='csr')")
    dev = csr.values.device
    if fmt == "csr":
        return csr
    if fmt == "coo":
        Ai = torch.empty(csr.num_entries, dtype=torch.int32, device=dev)
        B.csr_row_indices(csr.num_rows, csr.row_offsets, Ai)
        return CooMatrix(csr.num_rows, csr.num_cols, csr.num_entries, Ai, csr.column_indices, csr.values)
    if fmt in ("ell", "hyb"):
        lens = (csr.row_offsets[1:] - csr.row_offsets[:-1])
        max_len = int(lens.max().item()) if csr.num_rows else 0
        if num_entries_per_row is not None:
            width = int(num_entries_per_row)
        elif fmt == "hyb" and csr.num_rows:
            width = B.hyb_entries_per_row(csr.values.dtype, csr.num_rows, csr.row_offsets)
        else:
            width = max_len
        pitch = _round_up(csr.num_rows, ell_alignment)
        eAj = torch.empty(width * pitch, dtype=torch.int32, device=dev)
        eAx = torch.empty(width * pitch, dtype=csr.values.dtype, device=dev)
        B.csr_to_ell(csr.num_rows, csr.row_offsets, csr.column_indices, csr.values, width, pitch, eAj, eAx)
        n_coo = int(torch.clamp(lens - width, min=0).sum().item())
        ell = EllMatrix(csr.num_rows, csr.num_cols, csr.num_entries - n_coo, width, pitch, eAj, eAx)
        if fmt == "ell":
            if n_coo:
                raise ValueError("convert: num_entries_per_row is smaller than the longest row (use hyb)")
            # reference csr_to_other.h:188: an ELL matrix converted from CSR reports num_entries without the explicit zeros
            ell.num_entries = csr.num_entries - (B.count_zeros(csr.values) if csr.num_entries else 0)
            return ell
        # COO part: entries at within-row index >= width, in CSR order.  Their destinations are an
        # exclusive scan of the per-row overflow counts (a function of the row offsets alone).
        over = torch.clamp(lens - width, min=0)
        offs = (torch.cumsum(over, 0) - over).to(torch.int32).contiguous()
        cAi = torch.empty(n_coo, dtype=torch.int32, device=dev)
        cAj = torch.empty(n_coo, dtype=torch.int32, device=dev)
        cAx = torch.empty(n_coo, dtype=csr.values.dtype, device=dev)
        if n_coo:
            B.csr_to_hyb_coo(csr.num_rows, csr.row_offsets, csr.column_indices, csr.values, width, offs, cAi, cAj, cAx)
        coo = CooMatrix(csr.num_rows, csr.num_cols, n_coo, cAi, cAj, cAx)
        return HybMatrix(csr.num_rows, csr.num_cols, csr.num_entries, ell, coo)
    if fmt == "dia":
        # reference csr_to_other.h:73-153 (max_fill 3.0 once the DIA array exceeds 1e6 slots)
        if csr.num_entries == 0:
            return DiaMatrix(csr.num_rows, csr.num_cols, 0, _round_up(csr.num_rows, ell_alignment), torch.empty(0, dtype=torch.int32, device=dev),
                             torch.empty(0, dtype=csr.values.dtype, device=dev))
        slot_map = torch.empty(csr.num_rows + csr.num_cols, dtype=torch.int32, device=dev)
        capacity = max(int(3.0 * csr.num_entries / max(csr.num_rows, 1)) + 1, int(1e6 // max(csr.num_rows, 1)) + 1)
        capacity = min(capacity, csr.num_rows + csr.num_cols)
        diag_list = torch.empty(capacity, dtype=torch.int32, device=dev)
        nd = B.csr_diagonals(csr.num_rows, csr.num_cols, csr.row_offsets, csr.column_indices, slot_map, diag_list)
        slots = nd * csr.num_rows
        if nd > capacity or (slots / max(1.0, csr.num_entries) > 3.0 and slots > 1e6):
            raise ValueError("convert: dia_matrix fill-in would exceed maximum tolerance")
        offsets = torch.sort(diag_list[:nd]).values.contiguous()
        pitch = _round_up(csr.num_rows, ell_alignment)
        values = torch.empty(nd * pitch, dtype=csr.values.dtype, device=dev)
        B.csr_to_dia(csr.num_rows, csr.num_cols, csr.row_offsets, csr.column_indices, csr.values, offsets, pitch, slot_map, values)
        return DiaMatrix(csr.num_rows, csr.num_cols, csr.num_entries, pitch, offsets, values)
    raise ValueError(f"convert: unknown format {fmt!r}")


def fill_x(n, dtype=None, device="cpu", start=0):
    """The deterministic, RNG-free input vector of SURVEY.md section 8(d):
    x[i] = ((uint32)(i * 2654435761u) % 1000) / 997.0 - 0.5   for i in [start, start+n).
    Computed with exact integer ops and one IEEE division on the host, then moved to `device`."""
    import torch
    dtype = dtype or torch.float64
    i = torch.arange(start, start + n, dtype=torch.int64)
    h = (i * 2654435761) & 0xFFFFFFFF
    x = (h % 1000).to(torch.float64) / 997.0 - 0.5
    return x.to(dtype).to(device)


# ------------------------------------------------------------------------------------------------
# algorithmic (compulsory) HBM bytes per SpMV -- SURVEY.md section 8(d)
# ------------------------------------------------------------------------------------------------
def csr_bytes(num_rows, num_entries, value_bytes=8):
    """Ap once + Aj once + Ax once + x once + y once (square matrix): 12*nnz + 20*N + 4 for f64."""
    return 4 * (num_rows + 1) + (4 + value_bytes) * num_entries + 2 * value_bytes * num_rows


def ell_bytes(num_rows, width, pitch, value_bytes=8):
    return width * pitch * (4 + value_bytes) + 2 * value_bytes * num_rows


def dia_bytes(num_rows, num_diagonals, pitch, value_bytes=8):
    return num_diagonals * pitch * value_bytes + 4 * num_diagonals + 2 * value_bytes * num_rows


def coo_bytes(num_rows, num_entries, value_bytes=8):
    return num_entries * (8 + value_bytes) + 2 * value_bytes * num_rows
