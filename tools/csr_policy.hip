// tools/csr_policy.hip -- cache-policy sweep for the csr_stream tile body on poisson5pt 3162^2:
// matrix streams (Aj, Ax, Ap) and the y store issued as raw buffer operations so that every gfx950
// cache-policy combination (aux bits: 1 = sc0, 2 = nt, 16 = sc1) can be timed; x gathers stay plain
// (they are the only re-used data).  Results are checked against the plain variant.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/csr_policy.hip -o tools/bin/csr_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int __attribute__((ext_vector_type(4))) int4v;
typedef double __attribute__((ext_vector_type(2))) double2v;
typedef unsigned __attribute__((ext_vector_type(4))) u4;
typedef unsigned __attribute__((ext_vector_type(2))) u2;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

template <int LP, int SP, int RPB>
__global__ void __launch_bounds__(256)
stream(int64_t num_rows, int64_t num_entries, const int *__restrict__ Ap, const int *__restrict__ Aj,
       const double *__restrict__ Ax, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ double prod[1024];
    __shared__ int rowptr[260];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * RPB;
    const int nr = (int)((num_rows - r0) < RPB ? (num_rows - r0) : RPB);
    {   // row pointers through a buffer load with the stream policy
        __amdgpu_buffer_rsrc_t rp = rsrc(Ap + r0, (unsigned)((nr + 1) * 4));
        if (tid <= nr) rowptr[tid] = (int)__builtin_amdgcn_raw_buffer_load_b32(rp, tid * 4, 0, LP);
    }
    __syncthreads();
    const int nz0 = rowptr[0], nz1 = rowptr[nr];
    const int base = nz0 & ~3;
    const int span = ((nz1 + 3) & ~3) - base; // entries covered by whole vectors
    double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    if (tid * 4 < span && (int64_t)base + tid * 4 + 4 <= num_entries) {
        __amdgpu_buffer_rsrc_t rj = rsrc(Aj + base, (unsigned)(span * 4));
        __amdgpu_buffer_rsrc_t rv = rsrc(Ax + base, (unsigned)(span * 8));
        const u4 cu = __builtin_amdgcn_raw_buffer_load_b128(rj, tid * 16, 0, LP);
        const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rv, tid * 32, 0, LP);
        const u4 b = __builtin_amdgcn_raw_buffer_load_b128(rv, tid * 32 + 16, 0, LP);
        const int4v c = __builtin_bit_cast(int4v, cu);
        const double2v v01 = __builtin_bit_cast(double2v, a), v23 = __builtin_bit_cast(double2v, b);
        p0 = v01.x * x[c.x]; p1 = v01.y * x[c.y]; p2 = v23.x * x[c.z]; p3 = v23.y * x[c.w];
    }
    prod[tid * 4 + 0] = p0; prod[tid * 4 + 1] = p1; prod[tid * 4 + 2] = p2; prod[tid * 4 + 3] = p3;
    __syncthreads();
    if (tid < nr) {
        double acc = 0;
        for (int j = rowptr[tid]; j < rowptr[tid + 1]; j++) acc = acc + prod[j - base];
        __amdgpu_buffer_rsrc_t ry = rsrc(y + r0, (unsigned)(nr * 8));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, acc), ry, tid * 8, 0, SP);
    }
}

__global__ void build(int64_t m, int64_t n, int *Ap, int *Aj, double *Ax)
{
    auto prefix = [=](int64_t r) { int64_t iy = r / m, ix = r % m; int64_t c = 5 * r; c -= iy + (ix > 0); c -= iy; c -= iy > 0 ? m : ix; c -= iy >= n ? m : (iy == n - 1 ? ix : 0); return c; };
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= m * n; r += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = prefix(r); Ap[r] = (int)p; if (r == m * n) break;
        int64_t iy = r / m, ix = r % m;
        if (iy > 0) { Aj[p] = (int)(r - m); Ax[p++] = -1; }
        if (ix > 0) { Aj[p] = (int)(r - 1); Ax[p++] = -1; }
        Aj[p] = (int)r; Ax[p++] = 4;
        if (ix < m - 1) { Aj[p] = (int)(r + 1); Ax[p++] = -1; }
        if (iy < n - 1) { Aj[p] = (int)(r + m); Ax[p++] = -1; }
    }
}
__global__ void fillx(int64_t n, double *x) { for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] = (double)(((unsigned)i * 2654435761u) % 1000u) / 997.0 - 0.5; }

static std::vector<double> g_ref;
template <int LP, int SP, int RPB> void run(int64_t N, int64_t nnz, int *Ap, int *Aj, double *Ax, double *x, double *y)
{
    const int grid = (int)((N + RPB - 1) / RPB);
    auto f = [&] { hipLaunchKernelGGL((stream<LP, SP, RPB>), dim3(grid), dim3(256), 0, 0, N, nnz, Ap, Aj, Ax, x, y); };
    CK(hipMemset(y, 0xff, N * 8));
    f(); CK(hipDeviceSynchronize());
    std::vector<double> h(N);
    CK(hipMemcpy(h.data(), y, N * 8, hipMemcpyDeviceToHost));
    if (g_ref.empty()) g_ref = h;
    const bool ok = std::equal(h.begin(), h.end(), g_ref.begin());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int r = 0; r < 9; r++) {
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / 20 * 1000);
    }
    std::sort(t.begin(), t.end());
    const double bytes = 12.0 * nnz + 20.0 * N + 4;
    printf("rpb %3d  load aux %2d  store aux %2d   %7.1f us (min %7.1f)  %6.0f GB/s  %s\n", RPB, LP, SP, t[4], t[0], bytes / t[4] / 1e3, ok ? "ok" : "MISMATCH");
}

int main()
{
    const int64_t m = 3162, N = m * m, nnz = 5 * N - 4 * m;
    int *Ap, *Aj; double *Ax, *x, *y;
    CK(hipMalloc(&Ap, (N + 1) * 4)); CK(hipMalloc(&Aj, nnz * 4)); CK(hipMalloc(&Ax, nnz * 8)); CK(hipMalloc(&x, N * 8)); CK(hipMalloc(&y, N * 8));
    hipLaunchKernelGGL(build, dim3(4096), dim3(256), 0, 0, m, m, Ap, Aj, Ax);
    hipLaunchKernelGGL(fillx, dim3(4096), dim3(256), 0, 0, N, x);
    CK(hipDeviceSynchronize());
#define R(LP, SP) run<LP, SP, 192>(N, nnz, Ap, Aj, Ax, x, y)
    R(0, 0); R(0, 2); R(0, 16); R(0, 18); R(0, 17); R(0, 19); R(0, 1); R(0, 3);
    R(2, 2); R(16, 2); R(18, 2); R(17, 2); R(1, 2);
    R(16, 16); R(16, 18); R(16, 19); R(2, 18); R(18, 18);
    R(0, 2); R(0, 0);
    return 0;
}
