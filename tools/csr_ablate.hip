// tools/csr_ablate.hip -- where do the cycles of csr_stream go?  (guide section 7: ablate before
// optimising.)  Standalone timing-only variants of the csr_stream tile body on poisson5pt 3162^2:
//   bit 0: no x gather (multiply by a constant)         bit 1: no LDS pass / row sum (lane writes its own products' sum)
//   bit 2: no y store                                    bit 3: no Aj stream     bit 4: no Ax stream
//   bit 5: no row-pointer load (bounds computed as 5*row, valid only for timing)
//   bit 6: nontemporal y store      bit 7: two rows per lane, one 16-byte y store (results stay right)
// Results are WRONG by construction; only the times matter.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/csr_ablate.hip -o tools/bin/csr_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int __attribute__((ext_vector_type(4))) int4v;
typedef double __attribute__((ext_vector_type(2))) double2v;

template <int ABL>
__global__ void __launch_bounds__(256)
stream(int64_t num_rows, int64_t num_entries, const int *__restrict__ Ap, const int *__restrict__ Aj,
       const double *__restrict__ Ax, const double *__restrict__ x, double *__restrict__ y, int rpb)
{
    __shared__ double prod[1024];
    __shared__ int rowptr[260];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;
    const int nr = (int)((num_rows - r0) < rpb ? (num_rows - r0) : rpb);
    int nz0, nz1;
    if constexpr (ABL & 32) {
        nz0 = (int)(r0 * 5 < num_entries - 2048 ? r0 * 5 : num_entries - 2048); nz1 = nz0 + nr * 5;
        if (tid <= nr) rowptr[tid] = nz0 + tid * 5;
        __syncthreads();
    } else {
        if (tid <= nr) rowptr[tid] = Ap[r0 + tid];
        __syncthreads();
        nz0 = rowptr[0]; nz1 = rowptr[nr];
    }
    const int base = nz0 & ~3;
    const int e = base + tid * 4;
    double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    if (e < nz1 && e + 4 <= num_entries) {
        int4v c = {tid, tid + 1, tid + 2, tid + 3};
        if constexpr (!(ABL & 8)) c = *reinterpret_cast<const int4v *>(Aj + e);
        double2v v01 = {1.0, 2.0}, v23 = {3.0, 4.0};
        if constexpr (!(ABL & 16)) { v01 = *reinterpret_cast<const double2v *>(Ax + e); v23 = *reinterpret_cast<const double2v *>(Ax + e + 2); }
        if constexpr (ABL & 1) { p0 = v01.x * c.x; p1 = v01.y * c.y; p2 = v23.x * c.z; p3 = v23.y * c.w; }
        else { p0 = v01.x * x[c.x]; p1 = v01.y * x[c.y]; p2 = v23.x * x[c.z]; p3 = v23.y * x[c.w]; }
    }
    double acc = 0;
    if constexpr (ABL & 2) {
        acc = p0 + p1 + p2 + p3;
    } else {
        prod[tid * 4 + 0] = p0; prod[tid * 4 + 1] = p1; prod[tid * 4 + 2] = p2; prod[tid * 4 + 3] = p3;
        __syncthreads();
        if constexpr (ABL & 128) {
            const int r = 2 * tid;
            if (r < nr) {
                double a0 = 0, a1 = 0;
                const int s0 = rowptr[r], s1 = rowptr[r + 1], s2 = r + 1 < nr ? rowptr[r + 2] : s1;
                for (int j = s0; j < s1; j++) a0 = a0 + prod[j - base];
                for (int j = s1; j < s2; j++) a1 = a1 + prod[j - base];
                if (r + 1 < nr && ((r0 + r) & 1) == 0) {
                    double2v o = {a0, a1};
                    if constexpr (ABL & 64) __builtin_nontemporal_store(o, reinterpret_cast<double2v *>(y + r0 + r));
                    else *reinterpret_cast<double2v *>(y + r0 + r) = o;
                } else { y[r0 + r] = a0; if (r + 1 < nr) y[r0 + r + 1] = a1; }
            }
            return;
        }
        if (tid < nr) { for (int j = rowptr[tid]; j < rowptr[tid + 1]; j++) acc = acc + prod[j - base]; }
    }
    if (tid < nr) {
        if constexpr ((ABL & 64) != 0 && (ABL & 128) == 0) { __builtin_nontemporal_store(acc, y + r0 + tid); return; }
        if constexpr (ABL & 4) { if (acc == 123.456) y[r0 + tid] = acc; }
        else y[r0 + tid] = acc;
    }
}

__global__ void build(int64_t m, int64_t n, int *Ap, int *Aj, double *Ax)
{
    // same closed form as cmi::poisson_csr_kernel
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

template <typename F> double time_us(F f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 7; r++) {
        CK(hipEventRecord(e0)); for (int i = 0; i < 20; i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms / 20 * 1000);
    }
    std::sort(t.begin(), t.end()); return t[t.size() / 2];
}

template <int ABL> void run(const char *what, int64_t N, int64_t nnz, int *Ap, int *Aj, double *Ax, double *x, double *y, int rpb, double bytes)
{
    int grid = (int)((N + rpb - 1) / rpb);
    double us = time_us([&] { hipLaunchKernelGGL(stream<ABL>, dim3(grid), dim3(256), 0, 0, N, nnz, Ap, Aj, Ax, x, y, rpb); });
    printf("rpb %3d  abl %2d  %-44s %7.1f us   %6.0f GB/s of the bytes it still moves\n", rpb, ABL, what, us, bytes / us / 1e3);
}

int main()
{
    const int64_t m = 3162, N = m * m, nnz = 5 * N - 4 * m;
    int *Ap, *Aj; double *Ax, *x, *y;
    CK(hipMalloc(&Ap, (N + 1) * 4)); CK(hipMalloc(&Aj, nnz * 4)); CK(hipMalloc(&Ax, nnz * 8)); CK(hipMalloc(&x, N * 8)); CK(hipMalloc(&y, N * 8));
    hipLaunchKernelGGL(build, dim3(4096), dim3(256), 0, 0, m, m, Ap, Aj, Ax);
    CK(hipMemset(x, 0, N * 8)); CK(hipDeviceSynchronize());
    const double bAp = 4.0 * N, bAj = 4.0 * nnz, bAx = 8.0 * nnz, bx = 8.0 * N, by = 8.0 * N;
    for (int rpb : {204, 192, 200}) {
        run<0>("full kernel", N, nnz, Ap, Aj, Ax, x, y, rpb, bAp + bAj + bAx + bx + by);
        run<1>("no x gather", N, nnz, Ap, Aj, Ax, x, y, rpb, bAp + bAj + bAx + by);
        run<2>("no LDS pass / row sum", N, nnz, Ap, Aj, Ax, x, y, rpb, bAp + bAj + bAx + bx + by);
        run<3>("no gather, no LDS pass", N, nnz, Ap, Aj, Ax, x, y, rpb, bAp + bAj + bAx + by);
        run<4>("no y store", N, nnz, Ap, Aj, Ax, x, y, rpb, bAp + bAj + bAx + bx);
        run<7>("no gather, no LDS, no store (pure streams)", N, nnz, Ap, Aj, Ax, x, y, rpb, bAp + bAj + bAx);
        run<32>("no row-pointer load", N, nnz, Ap, Aj, Ax, x, y, rpb, bAj + bAx + bx + by);
        run<39>("pure Aj+Ax streams only", N, nnz, Ap, Aj, Ax, x, y, rpb, bAj + bAx);
        run<64>("nontemporal y store", N, nnz, Ap, Aj, Ax, x, y, rpb, bAp + bAj + bAx + bx + by);
        run<128>("two rows per lane, 16-byte y store", N, nnz, Ap, Aj, Ax, x, y, rpb, bAp + bAj + bAx + bx + by);
        run<192>("two rows per lane, 16-byte nontemporal store", N, nnz, Ap, Aj, Ax, x, y, rpb, bAp + bAj + bAx + bx + by);
        run<8>("no Aj stream", N, nnz, Ap, Aj, Ax, x, y, rpb, bAp + bAx + bx + by);
        run<16>("no Ax stream", N, nnz, Ap, Aj, Ax, x, y, rpb, bAp + bAj + bx + by);
    }
    return 0;
}
