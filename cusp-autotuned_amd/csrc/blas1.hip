// blas1.hip -- the five BLAS-1 routines cusp::krylov::cg needs, on device vectors (f64).
//
// Replaces (reference tree): cusp/system/detail/generic/blas.h:175-220 (axpy/axpby/copy via
// thrust::for_each) and :283-340 (dotc / nrm2 via thrust::inner_product / transform_reduce), i.e.
// what cusp/krylov/detail/cg.inl:63-105 calls through cusp::blas::*.
//
// All streaming, HBM-bound: 16-byte-per-lane vector accesses, grid capped at 8 workgroups per CU
// and grid-strided.  dot / nrm2 are two-stage and DETERMINISTIC (fixed grid, fixed tree): stage 1
// leaves one partial per workgroup in the caller's workspace, stage 2 (one workgroup) folds them in
// index order and writes the scalar to device memory -- no host sync, no atomics, no per-call
// allocation (the reference's Thrust reductions allocate temporaries and synchronise every call).
#include "common.h"

namespace cmi {

constexpr int kBlasBlock = 256;
constexpr int kBlasMaxGrid = kCus * 8; // 2048 partials

// reductions: a fixed, capped grid (one partial per workgroup, deterministic tree)
static int blas_grid(int64_t n, int per_thread)
{
    int64_t b = ceil_div(n, (int64_t)kBlasBlock * per_thread);
    if (b > kBlasMaxGrid) b = kBlasMaxGrid;
    return b < 1 ? 1 : (int)b;
}

// element-wise kernels that STORE: one-shot grid, a workgroup per chunk (tools/membench.hip: stores
// from a capped grid-stride grid run at 4.5-5.0 TB/s, one-shot at 6.3-6.9 TB/s)
static int stream_grid(int64_t n, int per_thread)
{
    int64_t b = ceil_div(n, (int64_t)kBlasBlock * per_thread);
    const int64_t cap = (int64_t)1 << 22;
    if (b > cap) b = cap;
    return b < 1 ? 1 : (int)b;
}

__global__ void __launch_bounds__(kBlasBlock)
axpby_kernel(int64_t n, double a, const double *x, double b, const double *y, double *z, int vec) // z may alias x or y
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t n2 = n / 2;
        for (int64_t i = t; i < n2; i += stride) {
            const double2v xv = reinterpret_cast<const double2v *>(x)[i];
            const double2v yv = reinterpret_cast<const double2v *>(y)[i];
            double2v zv;
            zv.x = a * xv.x + b * yv.x;
            zv.y = a * xv.y + b * yv.y;
            reinterpret_cast<double2v *>(z)[i] = zv;
        }
        if (t == 0 && (n & 1)) z[n - 1] = a * x[n - 1] + b * y[n - 1];
    } else {
        for (int64_t i = t; i < n; i += stride) z[i] = a * x[i] + b * y[i];
    }
}

__global__ void __launch_bounds__(kBlasBlock)
fill_kernel(int64_t n, double v, double *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = v;
}

// wave64 butterfly, then one LDS slot per wave, folded by lane order: fixed summation tree
__device__ __forceinline__ double block_sum(double v, double *slots)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) slots[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x / kWave); w++) s += slots[w];
    return s; // valid in thread 0
}

__global__ void __launch_bounds__(kBlasBlock)
dot_partial_kernel(int64_t n, const double *__restrict__ x, const double *__restrict__ y, double *__restrict__ partial,
                   int vec)
{
    __shared__ double slots[kBlasBlock / kWave];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (vec) {
        const int64_t n2 = n / 2;
        for (int64_t i = t; i < n2; i += stride) {
            const double2v xv = reinterpret_cast<const double2v *>(x)[i];
            const double2v yv = reinterpret_cast<const double2v *>(y)[i];
            acc += xv.x * yv.x;
            acc += xv.y * yv.y;
        }
        if (t == 0 && (n & 1)) acc += x[n - 1] * y[n - 1];
    } else {
        for (int64_t i = t; i < n; i += stride) acc += x[i] * y[i];
    }
    const double s = block_sum(acc, slots);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(kBlasBlock)
dot_final_kernel(int npartial, const double *__restrict__ partial, double *__restrict__ result, int take_sqrt)
{
    __shared__ double slots[kBlasBlock / kWave];
    double acc = 0.0;
    for (int i = threadIdx.x; i < npartial; i += blockDim.x) acc += partial[i];
    const double s = block_sum(acc, slots);
    if (threadIdx.x == 0) *result = take_sqrt ? sqrt(s) : s;
}

static bool aligned16(const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

} // namespace cmi

using namespace cmi;

CMI_API size_t cmi_blas_workspace_bytes(void) { return (size_t)kBlasMaxGrid * sizeof(double); }

CMI_API int cmi_blas_axpby_f64(int64_t n, double alpha, const double *x, double beta, const double *y, double *z, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_axpby: negative n");
    if (n == 0) return CMI_SUCCESS;
    if (!x || !y || !z) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_axpby: null array");
    const int vec = aligned16(x) && aligned16(y) && aligned16(z);
    hipLaunchKernelGGL(axpby_kernel, dim3(stream_grid(n, 2)), dim3(kBlasBlock), 0, as_stream(stream), n, alpha, x, beta, y, z, vec);
    CMI_LAUNCH_CHECK("axpby");
    return CMI_SUCCESS;
}

// y <- alpha*x + y   (cusp::blas::axpy, cusp/blas/blas.h)
CMI_API int cmi_blas_axpy_f64(int64_t n, double alpha, const double *x, double *y, void *stream)
{
    return cmi_blas_axpby_f64(n, alpha, x, 1.0, y, y, stream);
}

CMI_API int cmi_blas_copy_f64(int64_t n, const double *x, double *y, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_copy: negative n");
    if (n == 0) return CMI_SUCCESS;
    if (!x || !y) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_copy: null array");
    CMI_HIP(hipMemcpyAsync(y, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, as_stream(stream)));
    return CMI_SUCCESS;
}

CMI_API int cmi_blas_fill_f64(int64_t n, double value, double *y, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_fill: negative n");
    if (n == 0) return CMI_SUCCESS;
    if (!y) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_fill: null array");
    hipLaunchKernelGGL(fill_kernel, dim3(stream_grid(n, 1)), dim3(kBlasBlock), 0, as_stream(stream), n, value, y);
    CMI_LAUNCH_CHECK("fill");
    return CMI_SUCCESS;
}

static int dot_impl(int64_t n, const double *x, const double *y, double *result_dev, void *workspace, void *stream, int take_sqrt)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_dot: negative n");
    if (!result_dev || !workspace) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_dot: null result or workspace");
    if (n > 0 && (!x || !y)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_dot: null array");
    const int grid = blas_grid(n, 2);
    const int vec = aligned16(x) && aligned16(y);
    hipLaunchKernelGGL(dot_partial_kernel, dim3(grid), dim3(kBlasBlock), 0, as_stream(stream), n, x, y, (double *)workspace, vec);
    hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(kBlasBlock), 0, as_stream(stream), grid, (const double *)workspace, result_dev, take_sqrt);
    CMI_LAUNCH_CHECK("dot");
    return CMI_SUCCESS;
}

CMI_API int cmi_blas_dot_f64(int64_t n, const double *x, const double *y, double *result_dev, void *workspace, void *stream)
{
    return dot_impl(n, x, y, result_dev, workspace, stream, 0);
}

CMI_API int cmi_blas_nrm2_f64(int64_t n, const double *x, double *result_dev, void *workspace, void *stream)
{
    return dot_impl(n, x, x, result_dev, workspace, stream, 1);
}
