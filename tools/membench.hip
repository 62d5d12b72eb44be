// tools/membench.hip -- what HBM rate does THIS box sustain for plain streaming kernels?
// A known-good reference for the roofline discussion (guide rule 10: never infer a ceiling from
// your own kernel): 16-byte-per-lane read-only, copy and write-only sweeps over buffers far larger
// than the 256 MiB Infinity Cache, several grid shapes, hipEvent timing.
//   hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/bin/membench && tools/bin/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float __attribute__((ext_vector_type(4))) f4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int U> __global__ void __launch_bounds__(256) k_read(const f4 *__restrict__ a, size_t n, float *out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    for (; i < n; i += stride) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *out = acc.x;
}
template <int U> __global__ void __launch_bounds__(256) k_copy(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) b[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_write(f4 *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = f4{1, 2, 3, 4};
}
// 9 reads : 1 write, like CSR SpMV's byte mix (720 MB read, 80 MB written)
__global__ void __launch_bounds__(256) k_mix(const f4 *__restrict__ a, f4 *__restrict__ b, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * 9 + 8 < n; i += stride) {
        f4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 9; u++) acc += a[i * 9 + u];
        b[i] = acc;
    }
}

template <typename F> double time_ms(F f, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; i++) f();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms / iters);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const size_t bytes = 800ull << 20; // 800 MiB per buffer
    const size_t n = bytes / 16;
    f4 *a, *b; float *out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    for (int grid : {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32, (int)(n / 256)}) {
        double r1 = time_ms([&] { hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, 0, a, n, out); }, 20);
        double r4 = time_ms([&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, a, n, out); }, 20);
        double c1 = time_ms([&] { hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, a, b, n); }, 20);
        double c4 = time_ms([&] { hipLaunchKernelGGL(k_copy<4>, dim3(grid), dim3(256), 0, 0, a, b, n); }, 20);
        double w = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n); }, 20);
        double mx = time_ms([&] { hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, 0, a, b, n); }, 20);
        printf("grid %8d  read U1 %6.0f  read U4 %6.0f  copy U1 %6.0f  copy U4 %6.0f  write %6.0f  mix9r1w %6.0f  GB/s\n", grid,
               bytes / r1 / 1e6, bytes / r4 / 1e6, 2.0 * bytes / c1 / 1e6, 2.0 * bytes / c4 / 1e6, bytes / w / 1e6,
               (bytes / 9 * 9 + bytes / 9) / mx / 1e6);
    }
    double mc = time_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 20);
    printf("hipMemcpyAsync D2D: %6.0f GB/s (read+write)\n", 2.0 * bytes / mc / 1e6);
    return 0;
}
