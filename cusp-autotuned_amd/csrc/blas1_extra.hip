// blas1_extra.hip -- the rest of the reference's BLAS-1 set on device arrays: scal, xmy, axpbypcz, asum (= nrm1), amax (+ nrmmax).
//
// Replaces (reference): cusp/blas/blas.h scal / xmy / axpbypcz / nrm1 / nrmmax / amax -> cusp/system/detail/generic/blas.h (Thrust
// transforms and reductions).  cusp::krylov::cg itself needs none of them (its five routines and their fused forms are in blas1.hip); these
// are what the OTHER callers of the multiply need around it: a Jacobi-preconditioned cg (xmy), bicgstab / cr (scal, axpbypcz), stopping tests in
// other norms.  Plain streaming kernels: one pass, nothing cached between calls.  Reductions are deterministic: a fixed grid, a fixed tree per
// workgroup, the partials combined in index order by one workgroup -- no atomics.  Accumulation in double for both value types.
#include "common.h"

#include <initializer_list>

namespace cmi {
namespace {

constexpr int kBlock = 256;
constexpr int kMaxGrid = 1024; // partials per reduction: doubles [0, kMaxGrid) and int64 [kMaxGrid, 2 kMaxGrid) of the caller's workspace

int grid_for(int64_t n)
{
    int64_t b = ceil_div(n, (int64_t)kBlock * 4);
    if (b > kMaxGrid) b = kMaxGrid;
    return b < 1 ? 1 : (int)b;
}

template <typename T> __global__ void __launch_bounds__(kBlock) scal_kernel(int64_t n, T a, T *__restrict__ x)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = a * x[i];
}
template <typename T> __global__ void __launch_bounds__(kBlock) xmy_kernel(int64_t n, const T *x, const T *y, T *z) // z may alias x or y
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) z[i] = x[i] * y[i];
}
template <typename T> __global__ void __launch_bounds__(kBlock) axpbypcz_kernel(int64_t n, T a, const T *x, T b, const T *y, T c, const T *z, T *out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // (a x + b y) + c z: the association of the reference's functor (generic/blas.h AXPBYPCZ: alpha * x + beta * y + gamma * z)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a * x[i] + b * y[i] + c * z[i];
}

// workgroup tree over LDS, thread 0 gets the result; the same shape whatever the data
__device__ __forceinline__ double block_add(double v, double *lds)
{
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) lds[threadIdx.x] += lds[threadIdx.x + s];
        __syncthreads();
    }
    return lds[0];
}
// 16 bytes of T, and the element type a streaming pass runs on: E = T (any n, any alignment) or E = wide<T>::type (n a multiple of its width and every
// pointer 16-byte aligned: 16-byte loads and stores -- the scalar form of these passes reads at 5.3-6.2 TB/s, the wide form at 6.3-6.9)
template <typename T> struct wide;
template <> struct wide<double> { typedef double __attribute__((ext_vector_type(2))) type; static constexpr int n = 2; };
template <> struct wide<float> { typedef float __attribute__((ext_vector_type(4))) type; static constexpr int n = 4; };
__device__ __forceinline__ double dot_acc(double a, double b) { return a * b; }
__device__ __forceinline__ double dot_acc(float a, float b) { return (double)a * (double)b; }
__device__ __forceinline__ double dot_acc(wide<double>::type a, wide<double>::type b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ double dot_acc(wide<float>::type a, wide<float>::type b)
{ return (double)a.x * (double)b.x + (double)a.y * (double)b.y + (double)a.z * (double)b.z + (double)a.w * (double)b.w; }
template <typename T> bool can_widen(int64_t n, std::initializer_list<const void *> ptrs)
{
    if (n % wide<T>::n != 0) return false;
    for (const void *q : ptrs) if (reinterpret_cast<uintptr_t>(q) % 16 != 0) return false;
    return true;
}

// (largest |x|, FIRST position holding it): the pair with the larger value wins, equal values: the smaller index
__device__ __forceinline__ void take_max(double &v, long long &i, double v2, long long i2)
{
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}
__device__ __forceinline__ void block_max(double &v, long long &i, double *lv, long long *li)
{
    lv[threadIdx.x] = v;
    li[threadIdx.x] = i;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            double a = lv[threadIdx.x]; long long ai = li[threadIdx.x];
            take_max(a, ai, lv[threadIdx.x + s], li[threadIdx.x + s]);
            lv[threadIdx.x] = a; li[threadIdx.x] = ai;
        }
        __syncthreads();
    }
    v = lv[0];
    i = li[0];
}

template <typename T> __global__ void __launch_bounds__(kBlock) asum_partial_kernel(int64_t n, const T *__restrict__ x, double *__restrict__ partial)
{
    __shared__ double lds[kBlock];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += fabs((double)x[i]);
    const double s = block_add(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
template <typename T> __global__ void __launch_bounds__(kBlock) asum_final_kernel(int npartial, const double *__restrict__ partial, T *__restrict__ result)
{
    __shared__ double lds[kBlock];
    double acc = 0.0;
    for (int i = threadIdx.x; i < npartial; i += kBlock) acc += partial[i];
    const double s = block_add(acc, lds);
    if (threadIdx.x == 0) *result = (T)s;
}
template <typename T>
__global__ void __launch_bounds__(kBlock) amax_partial_kernel(int64_t n, const T *__restrict__ x, double *__restrict__ pv, long long *__restrict__ pi)
{
    __shared__ double lv[kBlock];
    __shared__ long long li[kBlock];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double v = -1.0; // below every |x|: an empty or all-NaN range reports (-1 -> 0, position 0)
    long long at = 0x7fffffffffffffffLL;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) take_max(v, at, fabs((double)x[i]), (long long)i);
    block_max(v, at, lv, li);
    if (threadIdx.x == 0) { pv[blockIdx.x] = v; pi[blockIdx.x] = at; }
}
template <typename T>
__global__ void __launch_bounds__(kBlock)
amax_final_kernel(int npartial, const double *__restrict__ pv, const long long *__restrict__ pi, T *__restrict__ value, long long *__restrict__ index)
{
    __shared__ double lv[kBlock];
    __shared__ long long li[kBlock];
    double v = -1.0;
    long long at = 0x7fffffffffffffffLL;
    for (int i = threadIdx.x; i < npartial; i += kBlock) take_max(v, at, pv[i], pi[i]);
    block_max(v, at, lv, li);
    if (threadIdx.x == 0) {
        if (value) *value = (T)(v < 0.0 ? 0.0 : v);
        if (index) *index = v < 0.0 ? 0 : at;
    }
}

template <typename T> int scal_impl(int64_t n, T a, T *x, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_scal: negative n");
    if (n == 0) return CMI_SUCCESS;
    if (!x) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_scal: null array");
    hipLaunchKernelGGL((scal_kernel<T>), dim3(grid_for(n) * 4), dim3(kBlock), 0, as_stream(stream), n, a, x);
    CMI_LAUNCH_CHECK("scal");
    return CMI_SUCCESS;
}
template <typename T> int xmy_impl(int64_t n, const T *x, const T *y, T *z, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_xmy: negative n");
    if (n == 0) return CMI_SUCCESS;
    if (!x || !y || !z) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_xmy: null array");
    hipLaunchKernelGGL((xmy_kernel<T>), dim3(grid_for(n) * 4), dim3(kBlock), 0, as_stream(stream), n, x, y, z);
    CMI_LAUNCH_CHECK("xmy");
    return CMI_SUCCESS;
}
template <typename T> int axpbypcz_impl(int64_t n, T a, const T *x, T b, const T *y, T c, const T *z, T *out, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_axpbypcz: negative n");
    if (n == 0) return CMI_SUCCESS;
    if (!x || !y || !z || !out) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_axpbypcz: null array");
    hipLaunchKernelGGL((axpbypcz_kernel<T>), dim3(grid_for(n) * 4), dim3(kBlock), 0, as_stream(stream), n, a, x, b, y, c, z, out);
    CMI_LAUNCH_CHECK("axpbypcz");
    return CMI_SUCCESS;
}
template <typename T> int asum_impl(int64_t n, const T *x, T *result_dev, void *workspace, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_asum: negative n");
    if (!result_dev || !workspace) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_asum: null result or workspace");
    if (n > 0 && !x) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_asum: null array");
    const int grid = grid_for(n);
    hipLaunchKernelGGL((asum_partial_kernel<T>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n, x, (double *)workspace);
    hipLaunchKernelGGL((asum_final_kernel<T>), dim3(1), dim3(kBlock), 0, as_stream(stream), grid, (const double *)workspace, result_dev);
    CMI_LAUNCH_CHECK("asum");
    return CMI_SUCCESS;
}
template <typename T> int amax_impl(int64_t n, const T *x, T *value_dev, int64_t *index_dev, void *workspace, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_amax: negative n");
    if ((!value_dev && !index_dev) || !workspace) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_amax: null results or workspace");
    if (n > 0 && !x) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_amax: null array");
    const int grid = grid_for(n);
    double *pv = (double *)workspace;
    long long *pi = reinterpret_cast<long long *>(pv + kMaxGrid);
    hipLaunchKernelGGL((amax_partial_kernel<T>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n, x, pv, pi);
    hipLaunchKernelGGL((amax_final_kernel<T>), dim3(1), dim3(kBlock), 0, as_stream(stream), grid, (const double *)pv, (const long long *)pi, value_dev, (long long *)index_dev);
    CMI_LAUNCH_CHECK("amax");
    return CMI_SUCCESS;
}

// ---- Jacobi-preconditioned CG: the two vector passes of an iteration with z = D^-1 r never stored ---------------------------------------------
// (reference cusp/krylov/detail/cg.inl:83-103 with M = cusp::precond::diagonal: axpy, axpy, xmy, dotc, axpby = 5 passes and 2 host reads between
// two multiplies).  Here, as for the unpreconditioned solve (blas1.hip cg_update / cg_direction_x), the scalars stay in DEVICE memory:
//   update:     alpha = <r,z> / <y,p>;  r <- r - alpha y;  partials of <r, D^-1 r> and of <r, r>   -> *rz_new, *rr (+ the host's pinned mirror)
//   direction:  beta = <r,z>_new / <r,z>_old;  x <- x + alpha p;  p <- D^-1 r + beta p
// 8 + 1 vector passes (dinv is read twice) and ONE host read per iteration.  Deterministic two-stage reductions in double.
template <typename T, typename E>
__global__ void __launch_bounds__(kBlock)
pcg_update_kernel(int64_t n, const double *__restrict__ rz, const double *__restrict__ yp, const E *__restrict__ y, E *__restrict__ r, const E *__restrict__ dinv,
                  double *__restrict__ part_rz, double *__restrict__ part_rr)
{
    __shared__ double lds[kBlock];
    const T alpha = (T)(*rz / *yp);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double a_rz = 0.0, a_rr = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const E ri = r[i] - alpha * y[i];
        r[i] = ri;
        a_rz += dot_acc(ri, dinv[i] * ri);
        a_rr += dot_acc(ri, ri);
    }
    const double s1 = block_add(a_rz, lds);
    __syncthreads();
    const double s2 = block_add(a_rr, lds);
    if (threadIdx.x == 0) { part_rz[blockIdx.x] = s1; part_rr[blockIdx.x] = s2; }
}
__global__ void __launch_bounds__(kBlock)
pcg_final_kernel(int npartial, const double *__restrict__ part_rz, const double *__restrict__ part_rr, double *__restrict__ rz_new, double *__restrict__ rr,
                 double *__restrict__ rr_mirror)
{
    __shared__ double lds[kBlock];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < npartial; i += kBlock) { a += part_rz[i]; b += part_rr[i]; }
    const double s1 = block_add(a, lds);
    __syncthreads();
    const double s2 = block_add(b, lds);
    if (threadIdx.x == 0) {
        *rz_new = s1;
        *rr = s2;
        if (rr_mirror) *rr_mirror = s2;
    }
}
template <typename T, typename E>
__global__ void __launch_bounds__(kBlock)
pcg_direction_kernel(int64_t n, const double *__restrict__ rz_new, const double *__restrict__ rz_old, const double *__restrict__ yp, const E *__restrict__ r,
                     const E *__restrict__ dinv, E *__restrict__ p, E *__restrict__ x)
{
    const T alpha = (T)(*rz_old / *yp), beta = (T)(*rz_new / *rz_old);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const E pi = p[i];
        x[i] = x[i] + alpha * pi;
        p[i] = dinv[i] * r[i] + beta * pi;
    }
}

template <typename T>
int pcg_update_impl(int64_t n, const double *rz, const double *yp, const T *y, T *r, const T *dinv, double *rz_new, double *rr, double *rr_mirror, void *workspace,
                    void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_pcg_update_jacobi: negative n");
    if (!rz || !yp || !rz_new || !rr || !workspace || (n > 0 && (!y || !r || !dinv))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_pcg_update_jacobi: null argument");
    typedef typename wide<T>::type V;
    const bool w = can_widen<T>(n, {y, r, dinv});
    const int grid = grid_for(w ? n / wide<T>::n : n);
    double *part = (double *)workspace;
    if (w) hipLaunchKernelGGL((pcg_update_kernel<T, V>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n / wide<T>::n, rz, yp, (const V *)y, (V *)r, (const V *)dinv, part, part + kMaxGrid);
    else hipLaunchKernelGGL((pcg_update_kernel<T, T>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n, rz, yp, y, r, dinv, part, part + kMaxGrid);
    hipLaunchKernelGGL(pcg_final_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), grid, (const double *)part, (const double *)(part + kMaxGrid), rz_new, rr, rr_mirror);
    CMI_LAUNCH_CHECK("pcg_update_jacobi");
    return CMI_SUCCESS;
}
template <typename T>
int pcg_direction_impl(int64_t n, const double *rz_new, const double *rz_old, const double *yp, const T *r, const T *dinv, T *p, T *x, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_pcg_direction_x_jacobi: negative n");
    if (!rz_new || !rz_old || !yp || (n > 0 && (!r || !dinv || !p || !x))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_pcg_direction_x_jacobi: null argument");
    if (n == 0) return CMI_SUCCESS;
    typedef typename wide<T>::type V;
    if (can_widen<T>(n, {r, dinv, p, x}))
        hipLaunchKernelGGL((pcg_direction_kernel<T, V>), dim3(grid_for(n / wide<T>::n) * 4), dim3(kBlock), 0, as_stream(stream), n / wide<T>::n, rz_new, rz_old, yp, (const V *)r, (const V *)dinv, (V *)p, (V *)x);
    else hipLaunchKernelGGL((pcg_direction_kernel<T, T>), dim3(grid_for(n) * 4), dim3(kBlock), 0, as_stream(stream), n, rz_new, rz_old, yp, r, dinv, p, x);
    CMI_LAUNCH_CHECK("pcg_direction_x_jacobi");
    return CMI_SUCCESS;
}

// ---- BiCGstab (identity preconditioner): the three vector passes of an iteration with the scalars in DEVICE memory ----------------------------
// (reference cusp/krylov/detail/bicgstab.inl:78-125: axpby, [copy,] dotc, dotc, axpbypcz, axpby, dotc, axpbypcz + two monitor norms = ~9 passes
// and 6 host reads around its two multiplies).  Here: the multiplies carry <r*, A p> and <A s, s> (cmi_spmv_*_dot_*), and
//   s-pass:   alpha = rho / <r*, A p>;  s <- r - alpha A p;  *ss <- <s, s>                                          (+ host mirror: the early exit)
//   xr-pass:  omega = <A s, s> / <A s, A s>;  x <- x + alpha p + omega s;  r <- s - omega A s;  *rho_new <- <r*, r>;  *rr <- <r, r> (+ mirror)
//   p-pass:   beta = (rho_new / rho) (alpha / omega);  p <- r + beta (p - omega A p)
// Two host reads per iteration (||s||, ||r||), deterministic two-stage reductions in double.
template <typename T, typename E>
__global__ void __launch_bounds__(kBlock)
bicg_s_kernel(int64_t n, const double *__restrict__ rho, const double *__restrict__ d1, const E *__restrict__ r, const E *__restrict__ AMp, E *__restrict__ s,
              double *__restrict__ part)
{
    __shared__ double lds[kBlock];
    const T alpha = (T)(*rho / *d1);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const E si = r[i] - alpha * AMp[i];
        s[i] = si;
        acc += dot_acc(si, si);
    }
    const double t = block_add(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void __launch_bounds__(kBlock) sum1_final_kernel(int npartial, const double *__restrict__ part, double *__restrict__ out, double *__restrict__ mirror)
{
    __shared__ double lds[kBlock];
    double a = 0.0;
    for (int i = threadIdx.x; i < npartial; i += kBlock) a += part[i];
    const double t = block_add(a, lds);
    if (threadIdx.x == 0) { *out = t; if (mirror) *mirror = t; }
}
template <typename T, typename E>
__global__ void __launch_bounds__(kBlock)
bicg_xr_kernel(int64_t n, const double *__restrict__ rho, const double *__restrict__ d1, const double *__restrict__ d2, const double *__restrict__ d3,
               const E *__restrict__ p, const E *__restrict__ s, const E *__restrict__ AMs, const E *__restrict__ r_star, E *__restrict__ x, E *__restrict__ r,
               double *__restrict__ part_rho, double *__restrict__ part_rr)
{
    __shared__ double lds[kBlock];
    const T alpha = (T)(*rho / *d1), omega = (T)(*d2 / *d3);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double a1 = 0.0, a2 = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const E si = s[i];
        x[i] = x[i] + alpha * p[i] + omega * si;
        const E ri = si - omega * AMs[i];
        r[i] = ri;
        a1 += dot_acc(r_star[i], ri);
        a2 += dot_acc(ri, ri);
    }
    const double t1 = block_add(a1, lds);
    __syncthreads();
    const double t2 = block_add(a2, lds);
    if (threadIdx.x == 0) { part_rho[blockIdx.x] = t1; part_rr[blockIdx.x] = t2; }
}
template <typename T, typename E>
__global__ void __launch_bounds__(kBlock)
bicg_p_kernel(int64_t n, const double *__restrict__ rho_new, const double *__restrict__ rho, const double *__restrict__ d1, const double *__restrict__ d2,
              const double *__restrict__ d3, const E *__restrict__ r, const E *__restrict__ AMp, E *__restrict__ p)
{
    const double alpha = *rho / *d1, omega = *d2 / *d3;
    const T beta = (T)((*rho_new / *rho) * (alpha / omega)), bo = (T)(-(double)beta * omega);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = r[i] + beta * p[i] + bo * AMp[i];
}
template <typename T>
__global__ void __launch_bounds__(kBlock) axpy_ratio_kernel(int64_t n, const double *__restrict__ num, const double *__restrict__ den, const T *__restrict__ x, T *__restrict__ y)
{
    const T a = (T)(*num / *den);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = y[i] + a * x[i];
}

__global__ void sum1_final_kernel(int npartial, const double *__restrict__ part, double *__restrict__ out, double *__restrict__ mirror);

// ---- the main diagonal of a CSR matrix (the Jacobi preconditioner's set-up on the device) ---------------------------------------------------------
// diag[i] <- the sum of row i's entries whose column is i (0 when none is stored), or its reciprocal (reference cusp/format_utils.h:184 extract_diagonal +
// precond/detail/diagonal.inl: a Thrust transform with reciprocal_functor).  Lane per row: set-up work, one pass over the column indices.
template <typename T>
__global__ void __launch_bounds__(kBlock)
csr_diagonal_kernel(int64_t num_rows, const int *__restrict__ Ap, const int *__restrict__ Aj, const T *__restrict__ Ax, T *__restrict__ diag, int reciprocal)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < num_rows; r += stride) {
        T d = T(0);
        for (int jj = Ap[r]; jj < Ap[r + 1]; jj++)
            if ((int64_t)Aj[jj] == r) d += Ax[jj];
        diag[r] = reciprocal ? T(1) / d : d;
    }
}
template <typename T> int csr_diagonal_impl(int64_t num_rows, const int *Ap, const int *Aj, const T *Ax, T *diag, int reciprocal, void *stream)
{
    if (num_rows < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_diagonal: negative size");
    if (num_rows == 0) return CMI_SUCCESS;
    if (!Ap || !diag) return fail(CMI_ERROR_INVALID_VALUE, "cmi_csr_diagonal: null array");
    hipLaunchKernelGGL((csr_diagonal_kernel<T>), dim3(grid_for(num_rows) * 4), dim3(kBlock), 0, as_stream(stream), num_rows, Ap, Aj, Ax, diag, reciprocal);
    CMI_LAUNCH_CHECK("csr_diagonal");
    return CMI_SUCCESS;
}

// ---- conjugate residuals (identity preconditioner): the two vector passes of an iteration with the scalars in DEVICE memory ----------------
// (reference cusp/krylov/detail/cr.inl:83-124: dotc, axpy, axpy, [copy,] dotc, axpby, axpby + the monitor's norm = 7 passes and 3 host reads around
// its multiply).  rz = <r, A r>, yy = <A p, A p> are device doubles; the multiply carries <A r, r> (cmi_spmv_*_dot_*):
//   xr-pass:  alpha = rz / yy;  x <- x + alpha p;  r <- r - alpha y   (y = A p);  *rr <- <r, r>  (+ host mirror)
//             -- with update_r == 0 only x moves (every 8th iteration the caller rebuilds r = b - A x itself, cr.inl:96-107)
//   py-pass:  beta = rz_new / rz;  p <- r + beta p;  y <- A r + beta y;  *yy_new <- <y, y>
template <typename T, typename E>
__global__ void __launch_bounds__(kBlock)
cr_xr_kernel(int64_t n, const double *__restrict__ rz, const double *__restrict__ yy, const E *__restrict__ p, const E *__restrict__ y, E *__restrict__ x, E *__restrict__ r,
             int update_r, double *__restrict__ part)
{
    __shared__ double lds[kBlock];
    const T alpha = (T)(*rz / *yy);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        x[i] = x[i] + alpha * p[i];
        if (update_r) {
            const E ri = r[i] - alpha * y[i];
            r[i] = ri;
            acc += dot_acc(ri, ri);
        }
    }
    const double t = block_add(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
template <typename T, typename E>
__global__ void __launch_bounds__(kBlock)
cr_py_kernel(int64_t n, const double *__restrict__ rz_new, const double *__restrict__ rz, const E *__restrict__ r, const E *__restrict__ Ar, E *__restrict__ p, E *__restrict__ y,
             double *__restrict__ part)
{
    __shared__ double lds[kBlock];
    const T beta = (T)(*rz_new / *rz);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        p[i] = r[i] + beta * p[i];
        const E yi = Ar[i] + beta * y[i];
        y[i] = yi;
        acc += dot_acc(yi, yi);
    }
    const double t = block_add(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
template <typename T>
int cr_xr_impl(int64_t n, const double *rz, const double *yy, const T *p, const T *y, T *x, T *r, int update_r, double *rr, double *rr_mirror, void *workspace, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cr_xr: negative n");
    if (!rz || !yy || !workspace || (update_r && !rr) || (n > 0 && (!p || !y || !x || !r))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cr_xr: null argument");
    typedef typename wide<T>::type V;
    const bool w = can_widen<T>(n, {p, y, x, r});
    const int grid = grid_for(w ? n / wide<T>::n : n);
    if (w) hipLaunchKernelGGL((cr_xr_kernel<T, V>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n / wide<T>::n, rz, yy, (const V *)p, (const V *)y, (V *)x, (V *)r, update_r, (double *)workspace);
    else hipLaunchKernelGGL((cr_xr_kernel<T, T>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n, rz, yy, p, y, x, r, update_r, (double *)workspace);
    if (update_r) hipLaunchKernelGGL(sum1_final_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), grid, (const double *)workspace, rr, rr_mirror);
    CMI_LAUNCH_CHECK("cr_xr");
    return CMI_SUCCESS;
}
template <typename T>
int cr_py_impl(int64_t n, const double *rz_new, const double *rz, const T *r, const T *Ar, T *p, T *y, double *yy_new, void *workspace, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cr_py: negative n");
    if (!rz_new || !rz || !yy_new || !workspace || (n > 0 && (!r || !Ar || !p || !y))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cr_py: null argument");
    typedef typename wide<T>::type V;
    const bool w = can_widen<T>(n, {r, Ar, p, y});
    const int grid = grid_for(w ? n / wide<T>::n : n);
    if (w) hipLaunchKernelGGL((cr_py_kernel<T, V>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n / wide<T>::n, rz_new, rz, (const V *)r, (const V *)Ar, (V *)p, (V *)y, (double *)workspace);
    else hipLaunchKernelGGL((cr_py_kernel<T, T>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n, rz_new, rz, r, Ar, p, y, (double *)workspace);
    hipLaunchKernelGGL(sum1_final_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), grid, (const double *)workspace, yy_new, (double *)nullptr);
    CMI_LAUNCH_CHECK("cr_py");
    return CMI_SUCCESS;
}

// ---- GMRES's modified Gram-Schmidt as a chain of fused steps with the coefficients in device memory ---------------------------------------------
// (reference gmres.inl:145-152: per basis vector one dotc -- a host read -- and one axpy).  One step here: w <- w - (*h) v;  *out <- <w, u>  -- the
// axpy of vector k and the dot with vector k + 1 (u = V[k + 1]) in ONE pass, or the norm's square (u = w) behind the last axpy; h == NULL: the dot alone.
// vec != 0: every pointer is 16-byte aligned -- 16-byte loads and stores over the first n / W * W elements, the last few one by one
template <typename T>
__global__ void __launch_bounds__(kBlock)
axpy_dot_kernel(int64_t n, const double *__restrict__ h, const T *__restrict__ v, T *w, const T *u, double *__restrict__ part, int u_is_w, int vec)
{
    typedef typename wide<T>::type V;
    constexpr int W = wide<T>::n;
    __shared__ double lds[kBlock];
    const T a = h ? (T)(*h) : T(0);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    const int64_t nv = vec ? n / W : 0;
    for (int64_t i = t; i < nv; i += stride) {
        V wi = reinterpret_cast<const V *>(w)[i];
        if (h) {
            const V vi = reinterpret_cast<const V *>(v)[i];
#pragma unroll
            for (int k = 0; k < W; k++) wi[k] = wi[k] - a * vi[k];
            reinterpret_cast<V *>(w)[i] = wi;
        }
        if (u_is_w) {
#pragma unroll
            for (int k = 0; k < W; k++) acc += (double)wi[k] * (double)wi[k];
        } else {
            const V ui = reinterpret_cast<const V *>(u)[i];
#pragma unroll
            for (int k = 0; k < W; k++) acc += (double)wi[k] * (double)ui[k];
        }
    }
    for (int64_t i = nv * W + t; i < n; i += stride) {
        T wi = w[i];
        if (h) { wi = wi - a * v[i]; w[i] = wi; }
        acc += (double)wi * (double)(u_is_w ? wi : u[i]);
    }
    const double s = block_add(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
template <typename T> int axpy_dot_impl(int64_t n, const double *h, const T *v, T *w, const T *u, double *out, void *workspace, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_axpy_dot: negative n");
    if (!out || !workspace || (n > 0 && (!w || !u || (h && !v)))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_axpy_dot: null argument");
    const int grid = grid_for(n);
    auto aligned = [](const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    const int vec = aligned(w) && aligned(u) && (!h || aligned(v));
    hipLaunchKernelGGL((axpy_dot_kernel<T>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n, h, v, w, u, (double *)workspace, (int)(u == w), vec);
    hipLaunchKernelGGL(sum1_final_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), grid, (const double *)workspace, out, (double *)nullptr);
    CMI_LAUNCH_CHECK("axpy_dot");
    return CMI_SUCCESS;
}

template <typename T>
int bicg_s_impl(int64_t n, const double *rho, const double *d1, const T *r, const T *AMp, T *s, double *ss, double *ss_mirror, void *workspace, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_bicgstab_s: negative n");
    if (!rho || !d1 || !ss || !workspace || (n > 0 && (!r || !AMp || !s))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_bicgstab_s: null argument");
    typedef typename wide<T>::type V;
    const bool w = can_widen<T>(n, {r, AMp, s});
    const int grid = grid_for(w ? n / wide<T>::n : n);
    if (w) hipLaunchKernelGGL((bicg_s_kernel<T, V>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n / wide<T>::n, rho, d1, (const V *)r, (const V *)AMp, (V *)s, (double *)workspace);
    else hipLaunchKernelGGL((bicg_s_kernel<T, T>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n, rho, d1, r, AMp, s, (double *)workspace);
    hipLaunchKernelGGL(sum1_final_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), grid, (const double *)workspace, ss, ss_mirror);
    CMI_LAUNCH_CHECK("bicgstab_s");
    return CMI_SUCCESS;
}
template <typename T>
int bicg_xr_impl(int64_t n, const double *rho, const double *d1, const double *d2, const double *d3, const T *p, const T *s, const T *AMs, const T *r_star, T *x, T *r,
                 double *rho_new, double *rr, double *rr_mirror, void *workspace, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_bicgstab_xr: negative n");
    if (!rho || !d1 || !d2 || !d3 || !rho_new || !rr || !workspace || (n > 0 && (!p || !s || !AMs || !r_star || !x || !r)))
        return fail(CMI_ERROR_INVALID_VALUE, "cmi_bicgstab_xr: null argument");
    typedef typename wide<T>::type V;
    const bool w = can_widen<T>(n, {p, s, AMs, r_star, x, r});
    const int grid = grid_for(w ? n / wide<T>::n : n);
    double *part = (double *)workspace;
    if (w) hipLaunchKernelGGL((bicg_xr_kernel<T, V>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n / wide<T>::n, rho, d1, d2, d3, (const V *)p, (const V *)s, (const V *)AMs, (const V *)r_star, (V *)x, (V *)r, part, part + kMaxGrid);
    else hipLaunchKernelGGL((bicg_xr_kernel<T, T>), dim3(grid), dim3(kBlock), 0, as_stream(stream), n, rho, d1, d2, d3, p, s, AMs, r_star, x, r, part, part + kMaxGrid);
    hipLaunchKernelGGL(pcg_final_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), grid, (const double *)part, (const double *)(part + kMaxGrid), rho_new, rr, rr_mirror);
    CMI_LAUNCH_CHECK("bicgstab_xr");
    return CMI_SUCCESS;
}
template <typename T>
int bicg_p_impl(int64_t n, const double *rho_new, const double *rho, const double *d1, const double *d2, const double *d3, const T *r, const T *AMp, T *p, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_bicgstab_p: negative n");
    if (!rho_new || !rho || !d1 || !d2 || !d3 || (n > 0 && (!r || !AMp || !p))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_bicgstab_p: null argument");
    if (n == 0) return CMI_SUCCESS;
    typedef typename wide<T>::type V;
    if (can_widen<T>(n, {r, AMp, p}))
        hipLaunchKernelGGL((bicg_p_kernel<T, V>), dim3(grid_for(n / wide<T>::n) * 4), dim3(kBlock), 0, as_stream(stream), n / wide<T>::n, rho_new, rho, d1, d2, d3, (const V *)r, (const V *)AMp, (V *)p);
    else hipLaunchKernelGGL((bicg_p_kernel<T, T>), dim3(grid_for(n) * 4), dim3(kBlock), 0, as_stream(stream), n, rho_new, rho, d1, d2, d3, r, AMp, p);
    CMI_LAUNCH_CHECK("bicgstab_p");
    return CMI_SUCCESS;
}
template <typename T> int axpy_ratio_impl(int64_t n, const double *num, const double *den, const T *x, T *y, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_axpy_ratio: negative n");
    if (!num || !den || (n > 0 && (!x || !y))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_axpy_ratio: null argument");
    if (n == 0) return CMI_SUCCESS;
    hipLaunchKernelGGL((axpy_ratio_kernel<T>), dim3(grid_for(n) * 4), dim3(kBlock), 0, as_stream(stream), n, num, den, x, y);
    CMI_LAUNCH_CHECK("axpy_ratio");
    return CMI_SUCCESS;
}

} // namespace
} // namespace cmi

using namespace cmi;
CMI_API int cmi_bicgstab_s_f64(int64_t n, const double *rho_dev, const double *d1_dev, const double *r, const double *AMp, double *s, double *ss_dev, double *ss_host_mirror,
                               void *workspace, void *stream)
{ return bicg_s_impl<double>(n, rho_dev, d1_dev, r, AMp, s, ss_dev, ss_host_mirror, workspace, stream); }
CMI_API int cmi_bicgstab_s_f32(int64_t n, const double *rho_dev, const double *d1_dev, const float *r, const float *AMp, float *s, double *ss_dev, double *ss_host_mirror,
                               void *workspace, void *stream)
{ return bicg_s_impl<float>(n, rho_dev, d1_dev, r, AMp, s, ss_dev, ss_host_mirror, workspace, stream); }
CMI_API int cmi_bicgstab_xr_f64(int64_t n, const double *rho_dev, const double *d1_dev, const double *d2_dev, const double *d3_dev, const double *p, const double *s,
                                const double *AMs, const double *r_star, double *x, double *r, double *rho_new_dev, double *rr_dev, double *rr_host_mirror, void *workspace,
                                void *stream)
{ return bicg_xr_impl<double>(n, rho_dev, d1_dev, d2_dev, d3_dev, p, s, AMs, r_star, x, r, rho_new_dev, rr_dev, rr_host_mirror, workspace, stream); }
CMI_API int cmi_bicgstab_xr_f32(int64_t n, const double *rho_dev, const double *d1_dev, const double *d2_dev, const double *d3_dev, const float *p, const float *s,
                                const float *AMs, const float *r_star, float *x, float *r, double *rho_new_dev, double *rr_dev, double *rr_host_mirror, void *workspace,
                                void *stream)
{ return bicg_xr_impl<float>(n, rho_dev, d1_dev, d2_dev, d3_dev, p, s, AMs, r_star, x, r, rho_new_dev, rr_dev, rr_host_mirror, workspace, stream); }
CMI_API int cmi_bicgstab_p_f64(int64_t n, const double *rho_new_dev, const double *rho_dev, const double *d1_dev, const double *d2_dev, const double *d3_dev, const double *r,
                               const double *AMp, double *p, void *stream)
{ return bicg_p_impl<double>(n, rho_new_dev, rho_dev, d1_dev, d2_dev, d3_dev, r, AMp, p, stream); }
CMI_API int cmi_bicgstab_p_f32(int64_t n, const double *rho_new_dev, const double *rho_dev, const double *d1_dev, const double *d2_dev, const double *d3_dev, const float *r,
                               const float *AMp, float *p, void *stream)
{ return bicg_p_impl<float>(n, rho_new_dev, rho_dev, d1_dev, d2_dev, d3_dev, r, AMp, p, stream); }
CMI_API int cmi_csr_diagonal_f64(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const double *Ax, double *diag, int reciprocal, void *stream)
{ return csr_diagonal_impl<double>(num_rows, Ap, Aj, Ax, diag, reciprocal, stream); }
CMI_API int cmi_csr_diagonal_f32(int64_t num_rows, const int32_t *Ap, const int32_t *Aj, const float *Ax, float *diag, int reciprocal, void *stream)
{ return csr_diagonal_impl<float>(num_rows, Ap, Aj, Ax, diag, reciprocal, stream); }
CMI_API int cmi_cr_xr_f64(int64_t n, const double *rz_dev, const double *yy_dev, const double *p, const double *y, double *x, double *r, int update_r, double *rr_dev,
                          double *rr_host_mirror, void *workspace, void *stream)
{ return cr_xr_impl<double>(n, rz_dev, yy_dev, p, y, x, r, update_r, rr_dev, rr_host_mirror, workspace, stream); }
CMI_API int cmi_cr_xr_f32(int64_t n, const double *rz_dev, const double *yy_dev, const float *p, const float *y, float *x, float *r, int update_r, double *rr_dev,
                          double *rr_host_mirror, void *workspace, void *stream)
{ return cr_xr_impl<float>(n, rz_dev, yy_dev, p, y, x, r, update_r, rr_dev, rr_host_mirror, workspace, stream); }
CMI_API int cmi_cr_py_f64(int64_t n, const double *rz_new_dev, const double *rz_dev, const double *r, const double *Ar, double *p, double *y, double *yy_new_dev, void *workspace,
                          void *stream)
{ return cr_py_impl<double>(n, rz_new_dev, rz_dev, r, Ar, p, y, yy_new_dev, workspace, stream); }
CMI_API int cmi_cr_py_f32(int64_t n, const double *rz_new_dev, const double *rz_dev, const float *r, const float *Ar, float *p, float *y, double *yy_new_dev, void *workspace,
                          void *stream)
{ return cr_py_impl<float>(n, rz_new_dev, rz_dev, r, Ar, p, y, yy_new_dev, workspace, stream); }
CMI_API int cmi_blas_axpy_dot_f64(int64_t n, const double *h_dev, const double *v, double *w, const double *u, double *out_dev, void *workspace, void *stream)
{ return axpy_dot_impl<double>(n, h_dev, v, w, u, out_dev, workspace, stream); }
CMI_API int cmi_blas_axpy_dot_f32(int64_t n, const double *h_dev, const float *v, float *w, const float *u, double *out_dev, void *workspace, void *stream)
{ return axpy_dot_impl<float>(n, h_dev, v, w, u, out_dev, workspace, stream); }
CMI_API int cmi_blas_axpy_ratio_f64(int64_t n, const double *num_dev, const double *den_dev, const double *x, double *y, void *stream)
{ return axpy_ratio_impl<double>(n, num_dev, den_dev, x, y, stream); }
CMI_API int cmi_blas_axpy_ratio_f32(int64_t n, const double *num_dev, const double *den_dev, const float *x, float *y, void *stream)
{ return axpy_ratio_impl<float>(n, num_dev, den_dev, x, y, stream); }
CMI_API int cmi_pcg_update_jacobi_f64(int64_t n, const double *rz_dev, const double *yp_dev, const double *y, double *r, const double *dinv, double *rz_new_dev,
                                      double *rr_dev, double *rr_host_mirror, void *workspace, void *stream)
{ return pcg_update_impl<double>(n, rz_dev, yp_dev, y, r, dinv, rz_new_dev, rr_dev, rr_host_mirror, workspace, stream); }
CMI_API int cmi_pcg_update_jacobi_f32(int64_t n, const double *rz_dev, const double *yp_dev, const float *y, float *r, const float *dinv, double *rz_new_dev,
                                      double *rr_dev, double *rr_host_mirror, void *workspace, void *stream)
{ return pcg_update_impl<float>(n, rz_dev, yp_dev, y, r, dinv, rz_new_dev, rr_dev, rr_host_mirror, workspace, stream); }
CMI_API int cmi_pcg_direction_x_jacobi_f64(int64_t n, const double *rz_new_dev, const double *rz_old_dev, const double *yp_dev, const double *r, const double *dinv,
                                           double *p, double *x, void *stream)
{ return pcg_direction_impl<double>(n, rz_new_dev, rz_old_dev, yp_dev, r, dinv, p, x, stream); }
CMI_API int cmi_pcg_direction_x_jacobi_f32(int64_t n, const double *rz_new_dev, const double *rz_old_dev, const double *yp_dev, const float *r, const float *dinv,
                                           float *p, float *x, void *stream)
{ return pcg_direction_impl<float>(n, rz_new_dev, rz_old_dev, yp_dev, r, dinv, p, x, stream); }
CMI_API int cmi_blas_scal_f64(int64_t n, double alpha, double *x, void *stream) { return scal_impl<double>(n, alpha, x, stream); }
CMI_API int cmi_blas_scal_f32(int64_t n, float alpha, float *x, void *stream) { return scal_impl<float>(n, alpha, x, stream); }
CMI_API int cmi_blas_xmy_f64(int64_t n, const double *x, const double *y, double *z, void *stream) { return xmy_impl<double>(n, x, y, z, stream); }
CMI_API int cmi_blas_xmy_f32(int64_t n, const float *x, const float *y, float *z, void *stream) { return xmy_impl<float>(n, x, y, z, stream); }
CMI_API int cmi_blas_axpbypcz_f64(int64_t n, double alpha, const double *x, double beta, const double *y, double gamma, const double *z, double *out, void *stream)
{ return axpbypcz_impl<double>(n, alpha, x, beta, y, gamma, z, out, stream); }
CMI_API int cmi_blas_axpbypcz_f32(int64_t n, float alpha, const float *x, float beta, const float *y, float gamma, const float *z, float *out, void *stream)
{ return axpbypcz_impl<float>(n, alpha, x, beta, y, gamma, z, out, stream); }
CMI_API int cmi_blas_asum_f64(int64_t n, const double *x, double *result_dev, void *workspace, void *stream) { return asum_impl<double>(n, x, result_dev, workspace, stream); }
CMI_API int cmi_blas_asum_f32(int64_t n, const float *x, float *result_dev, void *workspace, void *stream) { return asum_impl<float>(n, x, result_dev, workspace, stream); }
CMI_API int cmi_blas_amax_f64(int64_t n, const double *x, double *value_dev, int64_t *index_dev, void *workspace, void *stream)
{ return amax_impl<double>(n, x, value_dev, index_dev, workspace, stream); }
CMI_API int cmi_blas_amax_f32(int64_t n, const float *x, float *value_dev, int64_t *index_dev, void *workspace, void *stream)
{ return amax_impl<float>(n, x, value_dev, index_dev, workspace, stream); }
