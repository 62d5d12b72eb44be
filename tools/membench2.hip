// tools/membench2.hip -- is ~6.6 TB/s really what a READ of a large buffer gets on this box, whatever the request pattern?
// (Every SpMV kernel here sits on that figure -- DESIGN.md 3.2 -- so a pattern that reads faster would lift all of them.)
// Variants: per-lane width (8 / 16 bytes), nt hint, how a wave's requests are laid out (one 16-byte vector per lane and a huge grid; U vectors
// per lane grid-strided; U vectors per lane CONTIGUOUS per wave: a wave owns 64 * 16 * U bytes), XCD-chunked dealing of the tiles, buffer size.
//   hipcc -O3 --offload-arch=gfx950 tools/membench2.hip -o tools/bin/membench2 && tools/bin/membench2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float __attribute__((ext_vector_type(4))) f4;
typedef float __attribute__((ext_vector_type(2))) f2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <bool NT, typename V> __device__ __forceinline__ V ldv(const V *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ float fold(f4 v) { return v.x + v.y + v.z + v.w; }
__device__ __forceinline__ float fold(f2 v) { return v.x + v.y; }

// one wave owns U consecutive wave-rows of 64 vectors: every load instruction of a wave covers 64 * sizeof(V) contiguous bytes
// SWZ: tile -> XCD chunks of SWZ consecutive tiles (workgroups b, b + 8, ... share an XCD)
template <typename V, int U, bool NT, int SWZ> __global__ void __launch_bounds__(256) k_wave_contig(const V *__restrict__ a, size_t nvec, float *out)
{
    size_t tile = blockIdx.x;
    if (SWZ > 0) {
        const size_t q = tile / 8, r = tile % 8, chunk = q / SWZ, in = q % SWZ;
        tile = (chunk * 8 + r) * SWZ + in;
    }
    const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    const size_t base = (tile * 4 + wave) * (size_t)(64 * U);
    V v[U];
    float acc = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
        const size_t i = base + (size_t)u * 64 + lane;
        v[u] = i < nvec ? ldv<NT>(a + i) : V{};
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc += fold(v[u]);
    if (acc == 123.456f) *out = acc;
}
// grid-stride with U loads in flight per lane (membench.hip's shape), V wide
template <typename V, int U, bool NT> __global__ void __launch_bounds__(256) k_grid_stride(const V *__restrict__ a, size_t nvec, float *out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0;
    for (; i + (U - 1) * stride < nvec; i += U * stride) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = ldv<NT>(a + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; u++) acc += fold(v[u]);
    }
    for (; i < nvec; i += stride) acc += fold(ldv<NT>(a + i));
    if (acc == 123.456f) *out = acc;
}

// CSR's two streams (4-byte columns + 8-byte values = 12 bytes per entry) in csr_wave's request shape -- lane-strided: K dword loads + K dwordx2 loads per
// lane, 64 * K entries per wave -- against the same bytes as 16-byte loads: int4 columns + double2 values, 256 * K entries per wave (K int4 + 2 K double2 per lane)
typedef int __attribute__((ext_vector_type(4))) i4;
typedef double __attribute__((ext_vector_type(2))) d2;
template <int K, bool NT> __global__ void __launch_bounds__(256) k_csr_lane_strided(const int *__restrict__ cols, const double *__restrict__ vals, size_t n, float *out)
{
    const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    const size_t base = ((size_t)blockIdx.x * 4 + wave) * (size_t)(64 * K);
    int c[K]; double v[K];
#pragma unroll
    for (int k = 0; k < K; k++) { const size_t i = base + (size_t)k * 64 + lane; c[k] = i < n ? ldv<NT>(cols + i) : 0; }
#pragma unroll
    for (int k = 0; k < K; k++) { const size_t i = base + (size_t)k * 64 + lane; v[k] = i < n ? ldv<NT>(vals + i) : 0.0; }
    double acc = 0;
#pragma unroll
    for (int k = 0; k < K; k++) acc += v[k] * (double)c[k];
    if (acc == 123.456) *out = (float)acc;
}
template <int K, bool NT> __global__ void __launch_bounds__(256) k_csr_vectors(const i4 *__restrict__ cols, const d2 *__restrict__ vals, size_t n4, float *out)
{
    const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    const size_t base4 = ((size_t)blockIdx.x * 4 + wave) * (size_t)(64 * K); // in units of 4 entries
    i4 c[K]; d2 v[2 * K];
#pragma unroll
    for (int k = 0; k < K; k++) { const size_t i = base4 + (size_t)k * 64 + lane; c[k] = i < n4 ? ldv<NT>(cols + i) : i4{}; }
#pragma unroll
    for (int k = 0; k < 2 * K; k++) { const size_t i = base4 * 2 + (size_t)k * 64 + lane; v[k] = i < 2 * n4 ? ldv<NT>(vals + i) : d2{}; }
    double acc = 0;
#pragma unroll
    for (int k = 0; k < K; k++) acc += (double)(c[k].x + c[k].y + c[k].z + c[k].w);
#pragma unroll
    for (int k = 0; k < 2 * K; k++) acc += v[k].x + v[k].y;
    if (acc == 123.456) *out = (float)acc;
}

// S equal streams read side by side (S separate buffers, the same tile of each by the same wave): what do CONCURRENT streams cost by themselves?
template <int S, int U, bool NT> __global__ void __launch_bounds__(256) k_streams(const f4 *__restrict__ a, size_t nvec_each, float *out)
{
    const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
    const size_t base = ((size_t)blockIdx.x * 4 + wave) * (size_t)(64 * U);
    f4 v[S * U];
#pragma unroll
    for (int s = 0; s < S; s++)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * 64 + lane;
            v[s * U + u] = i < nvec_each ? ldv<NT>(a + (size_t)s * nvec_each + i) : f4{};
        }
    float acc = 0;
#pragma unroll
    for (int k = 0; k < S * U; k++) acc += fold(v[k]);
    if (acc == 123.456f) *out = acc;
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

template <typename V, int U, bool NT, int SWZ> void run_contig(const char *name, const void *a, size_t bytes, float *out)
{
    const size_t nvec = bytes / sizeof(V);
    const size_t tiles = (nvec + (size_t)256 * U - 1) / ((size_t)256 * U);
    const size_t grid = SWZ > 0 ? (tiles + 8 * SWZ - 1) / (8 * SWZ) * (8 * SWZ) : tiles;
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_wave_contig<V, U, NT, SWZ>), dim3((unsigned)grid), dim3(256), 0, 0, (const V *)a, nvec, out); }, 20);
    printf("  %-58s %7.0f GB/s\n", name, bytes / ms / 1e6);
}
template <typename V, int U, bool NT> void run_stride(const char *name, const void *a, size_t bytes, float *out, int grid)
{
    const size_t nvec = bytes / sizeof(V);
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_grid_stride<V, U, NT>), dim3(grid), dim3(256), 0, 0, (const V *)a, nvec, out); }, 20);
    printf("  %-58s %7.0f GB/s\n", name, bytes / ms / 1e6);
}

int main()
{
    float *out;
    CK(hipMalloc(&out, 4));
    for (size_t mb : {540ull, 1200ull, 3000ull}) {
        const size_t bytes = mb << 20;
        void *a;
        CK(hipMalloc(&a, bytes));
        CK(hipMemset(a, 0, bytes));
        printf("buffer %zu MiB, replayed back to back (the 256 MiB Infinity Cache holds a part of the two smaller ones)\n", mb);
        run_contig<f4, 1, false, 0>("wave-contiguous, 16 B x 1 per lane", a, bytes, out);
        run_contig<f4, 1, true, 0>("wave-contiguous, 16 B x 1 per lane, nt", a, bytes, out);
        run_contig<f4, 4, false, 0>("wave-contiguous, 16 B x 4 per lane", a, bytes, out);
        run_contig<f4, 4, true, 0>("wave-contiguous, 16 B x 4 per lane, nt", a, bytes, out);
        run_contig<f4, 8, true, 0>("wave-contiguous, 16 B x 8 per lane, nt", a, bytes, out);
        run_contig<f4, 16, true, 0>("wave-contiguous, 16 B x 16 per lane, nt", a, bytes, out);
        run_contig<f2, 8, true, 0>("wave-contiguous, 8 B x 8 per lane, nt", a, bytes, out);
        run_contig<f2, 16, true, 0>("wave-contiguous, 8 B x 16 per lane, nt", a, bytes, out);
        run_contig<f4, 4, true, 16>("wave-contiguous, 16 B x 4, nt, XCD chunks of 16 tiles", a, bytes, out);
        run_contig<f4, 4, true, 256>("wave-contiguous, 16 B x 4, nt, XCD chunks of 256 tiles", a, bytes, out);
        run_contig<f4, 8, true, 64>("wave-contiguous, 16 B x 8, nt, XCD chunks of 64 tiles", a, bytes, out);
        run_stride<f4, 1, false>("grid-stride 16 B x 1, grid 2048", a, bytes, out, 2048);
        run_stride<f4, 1, true>("grid-stride 16 B x 1, grid 2048, nt", a, bytes, out, 2048);
        run_stride<f4, 4, true>("grid-stride 16 B x 4, grid 2048, nt", a, bytes, out, 2048);
        run_stride<f4, 4, true>("grid-stride 16 B x 4, grid 8192, nt", a, bytes, out, 8192);
        run_stride<f4, 8, true>("grid-stride 16 B x 8, grid 4096, nt", a, bytes, out, 4096);
        CK(hipFree(a));
    }
    {
        const size_t bytes = 1200ull << 20;
        void *a;
        CK(hipMalloc(&a, bytes));
        CK(hipMemset(a, 0, bytes));
        printf("1200 MiB read as S equal streams side by side (16 B x U per lane and stream, nt)\n");
        auto st = [&](auto SS, auto UU, const char *name) {
            constexpr int S = decltype(SS)::value, U = decltype(UU)::value;
            const size_t nvec_each = bytes / 16 / S, grid = (nvec_each + 256 * U - 1) / (256 * U);
            const double ms = time_ms([&] { hipLaunchKernelGGL((k_streams<S, U, true>), dim3((unsigned)grid), dim3(256), 0, 0, (const f4 *)a, nvec_each, out); }, 20);
            printf("  %-58s %7.0f GB/s\n", name, (double)(nvec_each * S * 16) / ms / 1e6);
        };
        st(std::integral_constant<int, 1>(), std::integral_constant<int, 4>(), "1 stream, U = 4");
        st(std::integral_constant<int, 2>(), std::integral_constant<int, 2>(), "2 streams, U = 2");
        st(std::integral_constant<int, 2>(), std::integral_constant<int, 4>(), "2 streams, U = 4");
        st(std::integral_constant<int, 3>(), std::integral_constant<int, 2>(), "3 streams, U = 2");
        st(std::integral_constant<int, 4>(), std::integral_constant<int, 2>(), "4 streams, U = 2");
        st(std::integral_constant<int, 8>(), std::integral_constant<int, 1>(), "8 streams, U = 1");
        CK(hipFree(a));
    }
    // the headline's streams: 50 M entries = 200 MB of columns + 400 MB of values
    {
        const size_t n = 49978572 / 4 * 4;
        int *cols; double *vals;
        CK(hipMalloc(&cols, n * 4)); CK(hipMalloc(&vals, n * 8));
        CK(hipMemset(cols, 0, n * 4)); CK(hipMemset(vals, 0, n * 8));
        const double bytes = 12.0 * n;
        printf("CSR streams of the headline matrix (%zu entries, 12 bytes each), replayed\n", n);
        auto ls = [&](auto KK, auto NN, const char *name) {
            constexpr int K = decltype(KK)::value; constexpr bool NT = decltype(NN)::value;
            const size_t grid = (n + 256 * K - 1) / (256 * K);
            const double ms = time_ms([&] { hipLaunchKernelGGL((k_csr_lane_strided<K, NT>), dim3((unsigned)grid), dim3(256), 0, 0, cols, vals, n, out); }, 20);
            printf("  %-58s %7.0f GB/s\n", name, bytes / ms / 1e6);
        };
        auto vc = [&](auto KK, auto NN, const char *name) {
            constexpr int K = decltype(KK)::value; constexpr bool NT = decltype(NN)::value;
            const size_t n4 = n / 4, grid = (n4 + 256 * K - 1) / (256 * K);
            const double ms = time_ms([&] { hipLaunchKernelGGL((k_csr_vectors<K, NT>), dim3((unsigned)grid), dim3(256), 0, 0, (const i4 *)cols, (const d2 *)vals, n4, out); }, 20);
            printf("  %-58s %7.0f GB/s\n", name, bytes / ms / 1e6);
        };
        ls(std::integral_constant<int, 5>(), std::false_type(), "lane-strided 4 B + 8 B, K = 5 (csr_wave's shape)");
        ls(std::integral_constant<int, 5>(), std::true_type(), "lane-strided 4 B + 8 B, K = 5, nt");
        ls(std::integral_constant<int, 10>(), std::true_type(), "lane-strided 4 B + 8 B, K = 10, nt");
        ls(std::integral_constant<int, 16>(), std::true_type(), "lane-strided 4 B + 8 B, K = 16, nt");
        vc(std::integral_constant<int, 1>(), std::true_type(), "int4 + 2 x double2 per lane (256 entries per wave), nt");
        vc(std::integral_constant<int, 2>(), std::true_type(), "2 int4 + 4 double2 per lane (512 entries per wave), nt");
        vc(std::integral_constant<int, 4>(), std::true_type(), "4 int4 + 8 double2 per lane (1024 entries per wave), nt");
        vc(std::integral_constant<int, 5>(), std::true_type(), "5 int4 + 10 double2 per lane (1280 entries per wave), nt");
        vc(std::integral_constant<int, 4>(), std::false_type(), "4 int4 + 8 double2 per lane (1024 entries per wave)");
    }
    return 0;
}
