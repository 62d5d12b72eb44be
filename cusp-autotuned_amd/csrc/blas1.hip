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
#include <cstdlib>

namespace cmi {

constexpr int kBlasBlock = 256;
constexpr int kCgStorePolicy = 1; // x with the nt hint; see cg_store_policy() and the note in cg_update_kernel
constexpr int kBlasMaxGrid = kCus * 8;   // reductions: 2048 partials
constexpr int kFusedMaxGrid = 1 << 16;           // fused update+reduce kernels store too: near one-shot grids (65536 partials; the workspace holds kPartialCapacity)

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

// 16 bytes per lane: 2 doubles or 4 floats
template <typename T> struct vec16;
template <> struct vec16<double> { typedef double2v type; static constexpr int n = 2; };
template <> struct vec16<float> { typedef float4v type; static constexpr int n = 4; };

template <typename T>
__global__ void __launch_bounds__(kBlasBlock)
axpby_kernel(int64_t n, T a, const T *x, T b, const T *y, T *z, int vec) // z may alias x or y
{
    typedef typename vec16<T>::type V;
    constexpr int W = vec16<T>::n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t nv = n / W;
        for (int64_t i = t; i < nv; i += stride) {
            const V xv = reinterpret_cast<const V *>(x)[i];
            const V yv = reinterpret_cast<const V *>(y)[i];
            V zv;
#pragma unroll
            for (int k = 0; k < W; k++) zv[k] = a * xv[k] + b * yv[k];
            reinterpret_cast<V *>(z)[i] = zv;
        }
        if (t < n - nv * W) { const int64_t i = nv * W + t; z[i] = a * x[i] + b * y[i]; }
    } else {
        for (int64_t i = t; i < n; i += stride) z[i] = a * x[i] + b * y[i];
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlasBlock)
fill_kernel(int64_t n, T v, T *__restrict__ y)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = v;
}

// wave64 DPP tree (common.h wave_sum_to_last: no LDS-crossbar round trips at the tail of a one-shot workgroup), then one LDS
// slot per wave, folded in wave order: fixed summation tree
__device__ __forceinline__ double block_sum(double v, double *slots)
{
    v = wave_sum_to_last(v);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == kWave - 1) slots[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x / kWave); w++) s += slots[w];
    return s; // valid in thread 0
}

// partial sums are kept in double for both value types (f32 inputs: products widened before the add)
template <typename T>
__global__ void __launch_bounds__(kBlasBlock)
dot_partial_kernel(int64_t n, const T *__restrict__ x, const T *__restrict__ y, double *__restrict__ partial, int vec)
{
    typedef typename vec16<T>::type V;
    constexpr int W = vec16<T>::n;
    __shared__ double slots[kBlasBlock / kWave];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (vec) {
        const int64_t nv = n / W;
        for (int64_t i = t; i < nv; i += stride) {
            const V xv = reinterpret_cast<const V *>(x)[i];
            const V yv = reinterpret_cast<const V *>(y)[i];
#pragma unroll
            for (int k = 0; k < W; k++) acc += (double)xv[k] * (double)yv[k];
        }
        if (t < n - nv * W) { const int64_t i = nv * W + t; acc += (double)x[i] * (double)y[i]; }
    } else {
        for (int64_t i = t; i < n; i += stride) acc += (double)x[i] * (double)y[i];
    }
    const double s = block_sum(acc, slots);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) reset_fold_state(partial);
}

// Long partial lists (one-shot fused kernels leave up to 65536 of them) are folded by several
// workgroups -- workgroup g folds partial[g*kFoldChunk, (g+1)*kFoldChunk) into folded[g] -- and the
// LAST workgroup to finish (a ticket counter) folds `folded` in index order and writes the scalar:
// one launch, and still a fixed summation tree whichever workgroup happens to be last.  The ticket
// lives in the workspace behind `folded`; the last workgroup leaves it at 0 and every stage-1 kernel
// zeroes it as well, so an uninitialised workspace is fine.
constexpr int kFoldDirect = 2048; // up to here one workgroup folds the list directly

// Hand-off of one chunk sum to whichever workgroup arrives last (thread 0 of every folding workgroup calls it; returns true in
// the last arriver).  Default: the memory model's own recipe -- payload store, agent-scope RELEASE fence (buffer_wbl2 sc1, waited
// for: the inline-asm wait is the one the compiler may not drop, MI355X_MICROARCH.md "Compiler hazard"), relaxed ticket add; the
// last arriver follows its add with an agent-scope ACQUIRE fence (buffer_inv sc1, waited for before the workgroup barrier that
// lets its other waves read).  `relaxed` != 0 ($CMI_FOLD_RELAXED=1): the fence-free form of round 2 -- write-through (sc1) payload
// store drained by s_waitcnt vmcnt(0), relaxed ticket add, sc1 loads by the last arriver -- a form the guide lists as measured-valid
// on gfx950 (inter-workgroup visibility table, first row) but which the memory model does not order; kept for measurements.
__device__ __forceinline__ bool fold_handoff(double *folded_slot, double s, unsigned int *ticket, unsigned int groups, int relaxed)
{
    __hip_atomic_store(folded_slot, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!relaxed) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == groups - 1;
    if (last && !relaxed) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    return last;
}
static int fold_relaxed()
{
    static const int env = [] { const char *e = std::getenv("CMI_FOLD_RELAXED"); return (e && e[0] == '1') ? 1 : 0; }();
    return env;
}

template <typename T>
__global__ void __launch_bounds__(kBlasBlock)
dot_fold_final_kernel(int npartial, double *__restrict__ workspace, T *__restrict__ result, double *__restrict__ mirror, int take_sqrt, int relaxed)
{
    __shared__ double slots[kBlasBlock / kWave];
    __shared__ int is_last;
    const double *partial = workspace;
    double *folded = workspace + kPartialCapacity;
    const int lo = blockIdx.x * kFoldChunk;
    const int hi = lo + kFoldChunk < npartial ? lo + kFoldChunk : npartial;
    // kFoldChunk / kBlasBlock = 4 partials per lane: all four loads in flight at once (a plain loop pays four
    // dependent round trips -- half of this kernel's 6 us), added in the loop's order
    static_assert(kFoldChunk == 4 * kBlasBlock, "the fold reads four partials per lane");
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = lo + (int)threadIdx.x + k * kBlasBlock;
        v[k] = i < hi ? partial[i] : 0.0;
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) acc += v[k];
    double s = block_sum(acc, slots);
    if (threadIdx.x == 0) is_last = fold_handoff(folded + blockIdx.x, s, ticket_of(workspace), gridDim.x, relaxed); // (at most 64 workgroups)
    __syncthreads();
    if (!is_last) return;
    acc = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x) acc += __hip_atomic_load(folded + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads(); // slots are reused
    s = block_sum(acc, slots);
    if (threadIdx.x == 0) {
        const double v = take_sqrt ? sqrt(s) : s;
        *result = (T)v;
        if (mirror) *mirror = v;
        *ticket_of(workspace) = 0;
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlasBlock)
dot_final_mirror_kernel(int npartial, const double *__restrict__ partial, T *__restrict__ result, double *__restrict__ mirror, int take_sqrt)
{
    __shared__ double slots[kBlasBlock / kWave];
    double acc = 0.0;
    for (int i = threadIdx.x; i < npartial; i += blockDim.x) acc += partial[i];
    const double s = block_sum(acc, slots);
    if (threadIdx.x == 0) {
        const double v = take_sqrt ? sqrt(s) : s;
        *result = (T)v;
        if (mirror) *mirror = v;
    }
}

// workspace layout (doubles): [0, kPartialCapacity) stage-1 partials | kFoldedMax folded | ticket
// `mirror`: optional second destination (page-locked host memory the device can write)
template <typename T>
static void reduce_partials(int npartial, double *workspace, T *result, int take_sqrt, hipStream_t s, double *mirror = nullptr)
{
    if (npartial > kFoldDirect) {
        const int groups = (npartial + kFoldChunk - 1) / kFoldChunk;
        hipLaunchKernelGGL((dot_fold_final_kernel<T>), dim3(groups), dim3(kBlasBlock), 0, s, npartial, workspace, result, mirror, take_sqrt, fold_relaxed());
    } else {
        hipLaunchKernelGGL((dot_final_mirror_kernel<T>), dim3(1), dim3(kBlasBlock), 0, s, npartial, (const double *)workspace, result, mirror, take_sqrt);
    }
}

int reduce_partials_f64(int npartial, double *workspace, double *result, hipStream_t s)
{
    if (npartial > kPartialCapacity) return fail(CMI_ERROR_INVALID_VALUE, "reduce_partials: more partials than the workspace holds");
    reduce_partials<double>(npartial, workspace, result, 0, s);
    return CMI_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// Fused CG vector updates (unpreconditioned CG: z == r), scalars read from DEVICE memory so the
// host never has to produce alpha / beta (reference cusp/krylov/detail/cg.inl:83-103 does
// dot -> host -> axpy -> axpy -> copy -> dot -> host -> axpby: seven passes and three host syncs;
// here: update (x, r, <r,r>) in ONE pass over p, y, x, r and the direction in one pass over r, p).
// ---------------------------------------------------------------------------------------------
// T = double or float; the scalars (<r,r>, <y,p>) are doubles in device memory for both (the partials of every
// reduction here are doubles), alpha / beta are rounded to T once per kernel and the vectors are updated in T.
// store with or without the nt hint by a run-time flag (uniform)
template <typename V> __device__ __forceinline__ void st_policy(V *p, V v, bool nt)
{
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}


template <typename T>
__global__ void __launch_bounds__(kBlasBlock)
cg_update_kernel(int64_t n, const double *__restrict__ rz, const double *__restrict__ yp, const T *__restrict__ p,
                 const T *__restrict__ y, T *__restrict__ x /* may be null */, T *__restrict__ r, double *__restrict__ partial, int vec, int pol)
{
    typedef typename vec16<T>::type V;
    constexpr int W = vec16<T>::n;
    __shared__ double slots[kBlasBlock / kWave];
    const T alpha = (T)(*rz / *yp); // alpha <- <r,z>/<y,p>   (cg.inl:83)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    if (vec) {
        const int64_t nv = n / W;
        for (int64_t i = t; i < nv; i += stride) {
            const V yv = reinterpret_cast<const V *>(y)[i];
            V rv = reinterpret_cast<V *>(r)[i];
            if (x) { // (uniform) x == nullptr: the x update rides with the direction kernel, which has p in registers anyway
                const V pv = reinterpret_cast<const V *>(p)[i];
                V xv = reinterpret_cast<V *>(x)[i];
#pragma unroll
                for (int k = 0; k < W; k++) xv[k] = alpha * pv[k] + xv[k];      // x <- x + alpha p   (:86)
                st_policy(reinterpret_cast<V *>(x) + i, xv, (pol & 1) != 0);
            }
#pragma unroll
            for (int k = 0; k < W; k++) rv[k] = (-alpha) * yv[k] + rv[k];       // r <- r - alpha y   (:89)
            // (r is read again by the direction kernel right behind this one: plain store.  x is not read before the matrix
            // streams of the next SpMV have flushed every cache: nt -- stored plainly it is written back DURING that SpMV, which
            // then runs ~6 us longer: archive/tools/r2_probe.hip "ctx:" lines, archive/profiles/r02_probe_ctx.txt.  p is gathered by that SpMV:
            // plain; with the nt hint on p the SpMV lost 5 us, archive/profiles/r02_cg_store_policy.txt)
            st_policy(reinterpret_cast<V *>(r) + i, rv, (pol & 4) != 0);
#pragma unroll
            for (int k = 0; k < W; k++) acc += (double)rv[k] * (double)rv[k];   // <r, r>             (:97, z == r)
        }
        if (t < n - nv * W) { // the last n % W elements: one lane each
            const int64_t i = nv * W + t;
            if (x) x[i] = alpha * p[i] + x[i];
            const T ri = (-alpha) * y[i] + r[i];
            r[i] = ri;
            acc += (double)ri * (double)ri;
        }
    } else {
        for (int64_t i = t; i < n; i += stride) {
            if (x) x[i] = alpha * p[i] + x[i];
            const T ri = (-alpha) * y[i] + r[i];
            r[i] = ri;
            acc += (double)ri * (double)ri;
        }
    }
    const double s = block_sum(acc, slots);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) reset_fold_state(partial);
}

// p <- r + beta p with beta = rr_new / rr_old read from device memory (cg.inl:100-103, z == r).  With x != null also
// x <- x + alpha p with the OLD p (alpha = rr_old / yp, as cg_update computes it) before p is overwritten: the direction
// pass reads p anyway, so moving the x update here saves one read of p per iteration (8 vector passes instead of 9);
// the expressions are the update kernel's, term for term.
template <typename T>
__global__ void __launch_bounds__(kBlasBlock)
cg_direction_kernel(int64_t n, const double *__restrict__ rr_new, const double *__restrict__ rr_old, const double *__restrict__ yp,
                    const T *__restrict__ r, T *__restrict__ p, T *__restrict__ x /* may be null */, int vec, int pol)
{
    typedef typename vec16<T>::type V;
    constexpr int W = vec16<T>::n;
    const T alpha = x ? (T)(*rr_old / *yp) : T(0);
    const T beta = (T)(*rr_new / *rr_old);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t nv = n / W;
        for (int64_t i = t; i < nv; i += stride) {
            const V rv = reinterpret_cast<const V *>(r)[i];
            V pv = reinterpret_cast<V *>(p)[i];
            if (x) {
                V xv = reinterpret_cast<V *>(x)[i];
#pragma unroll
                for (int k = 0; k < W; k++) xv[k] = alpha * pv[k] + xv[k];
                st_policy(reinterpret_cast<V *>(x) + i, xv, (pol & 1) != 0);
            }
#pragma unroll
            for (int k = 0; k < W; k++) pv[k] = T(1) * rv[k] + beta * pv[k];
            st_policy(reinterpret_cast<V *>(p) + i, pv, (pol & 2) != 0);
        }
        if (t < n - nv * W) {
            const int64_t i = nv * W + t;
            if (x) x[i] = alpha * p[i] + x[i];
            p[i] = T(1) * r[i] + beta * p[i];
        }
    } else {
        for (int64_t i = t; i < n; i += stride) {
            if (x) x[i] = alpha * p[i] + x[i];
            p[i] = T(1) * r[i] + beta * p[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fold-ahead: the CONSUMER of a partial list folds it (cmi_cg_update_fold_* / cmi_cg_direction_x_fold_*)
// ---------------------------------------------------------------------------------------------
// A fused CG iteration had two one-workgroup-deep reductions between its three big kernels: 2 x (a 4.6 us fold kernel + a
// launch gap).  Here the kernel that NEEDS the scalar folds the list itself: its first G workgroups (G = one per 1024
// partials, <= 64; they are dispatched first) fold their chunks exactly as dot_fold_final_kernel does -- same tree, same
// hand-off -- and the last of them to arrive publishes the sum in kScalarCopies slots (a single 8-byte store each: the value
// IS the flag; "pending" is a NaN payload no sum produces).  Every workgroup requests its vectors first, then one lane polls
// its slot (sc1 load, sleeping in between; the producer kernel left the slots pending) -- for all but the first ~2000
// resident workgroups the first poll succeeds, and it replaces the plain load of the scalar the old kernels did anyway.
// The spin is bounded: if the value never shows up (the folding workgroups were not resident first, or a second stream used the SAME
// workspace -- one workspace per stream) the workgroup goes on with the "pending" NaN: r / x / p are poisoned, the next <r, r> is NaN and
// so is the host mirror -- cusp::krylov::cg's fold-ahead path throws on a NaN residual instead of iterating on (no hang, no silent
// wrong answer).
constexpr int kFoldSpinLimit = 1 << 15; // x ~0.5 us of s_sleep

__device__ __forceinline__ double fold_ahead(int npartial, double *area, double *result_out, double *mirror, double *slots_lds,
                                             double *value_lds, int relaxed)
{
    const int groups = npartial <= kFoldDirect ? 1 : (npartial + kFoldChunk - 1) / kFoldChunk;
    double *slots = slots_of(area);
    if ((int)blockIdx.x < groups) { // (uniform per workgroup)
        const double *partial = area;
        double *folded = area + kPartialCapacity;
        double acc = 0.0;
        bool last = true;
        if (groups == 1) {
            for (int i = threadIdx.x; i < npartial; i += blockDim.x) acc += partial[i];
        } else {
            const int lo = blockIdx.x * kFoldChunk;
            const int hi = lo + kFoldChunk < npartial ? lo + kFoldChunk : npartial;
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int i = lo + (int)threadIdx.x + k * kBlasBlock;
                v[k] = i < hi ? partial[i] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) acc += v[k];
        }
        double s = block_sum(acc, slots_lds);
        if (groups > 1) {
            __shared__ int is_last;
            if (threadIdx.x == 0) is_last = fold_handoff(folded + blockIdx.x, s, ticket_of(area), (unsigned)groups, relaxed);
            __syncthreads();
            last = is_last != 0;
            if (last) {
                acc = 0.0;
                for (int i = threadIdx.x; i < groups; i += blockDim.x) acc += __hip_atomic_load(folded + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads(); // slots_lds is reused
                s = block_sum(acc, slots_lds);
            }
        }
        if (last) {
            if (threadIdx.x == 0) {
                *value_lds = s; // thread 0 holds the sum
                if (result_out) *result_out = s; // for the kernels behind this one (plain memory, visible at the kernel boundary)
                if (mirror) *mirror = s;
                *ticket_of(area) = 0;
            }
            __syncthreads();
            const double v = *value_lds;
            if (threadIdx.x < kScalarCopies) __hip_atomic_store(slots + threadIdx.x * kScalarStride, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return v;
        }
        __syncthreads(); // (value_lds below is written after every thread has left block_sum)
    }
    if (threadIdx.x == 0) {
        const double *slot = slots + (blockIdx.x % kScalarCopies) * kScalarStride;
        double v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; (unsigned long long)__double_as_longlong(v) == kPendingBits && spin < kFoldSpinLimit; spin++) {
            __builtin_amdgcn_s_sleep(16);
            v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *value_lds = v;
    }
    __syncthreads();
    return *value_lds;
}

// cg_update with the fold of <y, p> in front: yp = fold(area_in) (also left in *yp_out for the direction kernel)
template <typename T>
__global__ void __launch_bounds__(kBlasBlock)
cg_update_fold_kernel(int64_t n, const double *__restrict__ rz, int npartial_in, double *__restrict__ area_in, double *__restrict__ yp_out,
                      const T *__restrict__ y, T *__restrict__ r, double *__restrict__ area_out, int vec, int pol, int relaxed)
{
    typedef typename vec16<T>::type V;
    constexpr int W = vec16<T>::n;
    __shared__ double slots[kBlasBlock / kWave];
    __shared__ double value;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nv = n / W;
    // this workgroup's first vectors are requested before anybody waits for the scalar
    V y0 = {}, r0 = {};
    const bool first = vec && t < nv;
    if (first) { y0 = reinterpret_cast<const V *>(y)[t]; r0 = reinterpret_cast<V *>(r)[t]; }
    const double rzv = *rz;
    const double ypv = fold_ahead(npartial_in, area_in, yp_out, nullptr, slots, &value, relaxed);
    const T alpha = (T)(rzv / ypv);
    double acc = 0.0;
    if (vec) {
        for (int64_t i = t; i < nv; i += stride) {
            V yv, rv;
            if (i == t) { yv = y0; rv = r0; }
            else { yv = reinterpret_cast<const V *>(y)[i]; rv = reinterpret_cast<V *>(r)[i]; }
#pragma unroll
            for (int k = 0; k < W; k++) rv[k] = (-alpha) * yv[k] + rv[k];
            st_policy(reinterpret_cast<V *>(r) + i, rv, (pol & 4) != 0);
#pragma unroll
            for (int k = 0; k < W; k++) acc += (double)rv[k] * (double)rv[k];
        }
        if (t < n - nv * W) {
            const int64_t i = nv * W + t;
            const T ri = (-alpha) * y[i] + r[i];
            r[i] = ri;
            acc += (double)ri * (double)ri;
        }
    } else {
        for (int64_t i = t; i < n; i += stride) {
            const T ri = (-alpha) * y[i] + r[i];
            r[i] = ri;
            acc += (double)ri * (double)ri;
        }
    }
    __syncthreads(); // slots (block_sum) may still be read by thread 0 of a folding workgroup
    const double s = block_sum(acc, slots);
    if (threadIdx.x == 0) area_out[blockIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) reset_fold_state(area_out);
}

// cg_direction_x with the fold of <r, r> in front: rr_new = fold(area_in) (left in *rr_new_out and the host mirror)
template <typename T>
__global__ void __launch_bounds__(kBlasBlock)
cg_direction_fold_kernel(int64_t n, int npartial_in, double *__restrict__ area_in, double *__restrict__ rr_new_out, double *__restrict__ mirror,
                         const double *__restrict__ rr_old, const double *__restrict__ yp, const T *__restrict__ r, T *__restrict__ p,
                         T *__restrict__ x, int vec, int pol, int relaxed)
{
    typedef typename vec16<T>::type V;
    constexpr int W = vec16<T>::n;
    __shared__ double slots[kBlasBlock / kWave];
    __shared__ double value;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nv = n / W;
    V r0 = {}, p0 = {}, x0 = {};
    const bool first = vec && t < nv;
    if (first) { r0 = reinterpret_cast<const V *>(r)[t]; p0 = reinterpret_cast<V *>(p)[t]; x0 = reinterpret_cast<V *>(x)[t]; }
    const double rro = *rr_old, ypv = *yp;
    const double rrn = fold_ahead(npartial_in, area_in, rr_new_out, mirror, slots, &value, relaxed);
    const T alpha = (T)(rro / ypv);
    const T beta = (T)(rrn / rro);
    if (vec) {
        for (int64_t i = t; i < nv; i += stride) {
            V rv, pv, xv;
            if (i == t) { rv = r0; pv = p0; xv = x0; }
            else { rv = reinterpret_cast<const V *>(r)[i]; pv = reinterpret_cast<V *>(p)[i]; xv = reinterpret_cast<V *>(x)[i]; }
#pragma unroll
            for (int k = 0; k < W; k++) xv[k] = alpha * pv[k] + xv[k];
            st_policy(reinterpret_cast<V *>(x) + i, xv, (pol & 1) != 0);
#pragma unroll
            for (int k = 0; k < W; k++) pv[k] = T(1) * rv[k] + beta * pv[k];
            st_policy(reinterpret_cast<V *>(p) + i, pv, (pol & 2) != 0);
        }
        if (t < n - nv * W) {
            const int64_t i = nv * W + t;
            x[i] = alpha * p[i] + x[i];
            p[i] = T(1) * r[i] + beta * p[i];
        }
    } else {
        for (int64_t i = t; i < n; i += stride) {
            x[i] = alpha * p[i] + x[i];
            p[i] = T(1) * r[i] + beta * p[i];
        }
    }
}

static bool aligned16(const void *p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

} // namespace cmi

using namespace cmi;

CMI_API size_t cmi_blas_workspace_bytes(void) { return (size_t)2 * kFoldArea * sizeof(double); } // two fold areas (common.h)

namespace {

template <typename T> int axpby_impl(int64_t n, T alpha, const T *x, T beta, const T *y, T *z, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_axpby: negative n");
    if (n == 0) return CMI_SUCCESS;
    if (!x || !y || !z) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_axpby: null array");
    const int vec = aligned16(x) && aligned16(y) && aligned16(z);
    hipLaunchKernelGGL((axpby_kernel<T>), dim3(stream_grid(n, vec16<T>::n)), dim3(kBlasBlock), 0, as_stream(stream), n, alpha, x, beta, y, z, vec);
    CMI_LAUNCH_CHECK("axpby");
    return CMI_SUCCESS;
}

template <typename T> int copy_impl(int64_t n, const T *x, T *y, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_copy: negative n");
    if (n == 0) return CMI_SUCCESS;
    if (!x || !y) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_copy: null array");
    CMI_HIP(hipMemcpyAsync(y, x, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, as_stream(stream)));
    return CMI_SUCCESS;
}

template <typename T> int fill_impl(int64_t n, T value, T *y, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_fill: negative n");
    if (n == 0) return CMI_SUCCESS;
    if (!y) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_fill: null array");
    hipLaunchKernelGGL((fill_kernel<T>), dim3(stream_grid(n, 1)), dim3(kBlasBlock), 0, as_stream(stream), n, value, y);
    CMI_LAUNCH_CHECK("fill");
    return CMI_SUCCESS;
}

template <typename T> int dot_impl(int64_t n, const T *x, const T *y, T *result_dev, void *workspace, void *stream, int take_sqrt)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_dot: negative n");
    if (!result_dev || !workspace) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_dot: null result or workspace");
    if (n > 0 && (!x || !y)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_dot: null array");
    const int grid = blas_grid(n, vec16<T>::n);
    const int vec = aligned16(x) && aligned16(y);
    hipLaunchKernelGGL((dot_partial_kernel<T>), dim3(grid), dim3(kBlasBlock), 0, as_stream(stream), n, x, y, (double *)workspace, vec);
    reduce_partials<T>(grid, (double *)workspace, result_dev, take_sqrt, as_stream(stream));
    CMI_LAUNCH_CHECK("dot");
    return CMI_SUCCESS;
}

} // namespace

// y <- alpha*x + y (cusp::blas::axpy), z <- alpha*x + beta*y (axpby), copy, fill, dot, nrm2
CMI_API int cmi_blas_axpby_f64(int64_t n, double alpha, const double *x, double beta, const double *y, double *z, void *stream) { return axpby_impl<double>(n, alpha, x, beta, y, z, stream); }
CMI_API int cmi_blas_axpby_f32(int64_t n, float alpha, const float *x, float beta, const float *y, float *z, void *stream) { return axpby_impl<float>(n, alpha, x, beta, y, z, stream); }
CMI_API int cmi_blas_axpy_f64(int64_t n, double alpha, const double *x, double *y, void *stream) { return axpby_impl<double>(n, alpha, x, 1.0, y, y, stream); }
CMI_API int cmi_blas_axpy_f32(int64_t n, float alpha, const float *x, float *y, void *stream) { return axpby_impl<float>(n, alpha, x, 1.0f, y, y, stream); }
CMI_API int cmi_blas_copy_f64(int64_t n, const double *x, double *y, void *stream) { return copy_impl<double>(n, x, y, stream); }
CMI_API int cmi_blas_copy_f32(int64_t n, const float *x, float *y, void *stream) { return copy_impl<float>(n, x, y, stream); }
CMI_API int cmi_blas_fill_f64(int64_t n, double value, double *y, void *stream) { return fill_impl<double>(n, value, y, stream); }
CMI_API int cmi_blas_fill_f32(int64_t n, float value, float *y, void *stream) { return fill_impl<float>(n, value, y, stream); }
CMI_API int cmi_blas_dot_f64(int64_t n, const double *x, const double *y, double *r, void *ws, void *stream) { return dot_impl<double>(n, x, y, r, ws, stream, 0); }
CMI_API int cmi_blas_dot_f32(int64_t n, const float *x, const float *y, float *r, void *ws, void *stream) { return dot_impl<float>(n, x, y, r, ws, stream, 0); }
CMI_API int cmi_blas_nrm2_f64(int64_t n, const double *x, double *r, void *ws, void *stream) { return dot_impl<double>(n, x, x, r, ws, stream, 1); }
CMI_API int cmi_blas_nrm2_f32(int64_t n, const float *x, float *r, void *ws, void *stream) { return dot_impl<float>(n, x, x, r, ws, stream, 1); }

// ---- fused CG steps ----
static int fused_grid(int64_t n, int per_thread)
{
    int64_t b = ceil_div(n, (int64_t)kBlasBlock * per_thread);
    if (b > kFusedMaxGrid) b = kFusedMaxGrid;
    return b < 1 ? 1 : (int)b;
}

namespace {

// which of the vectors a CG step writes get the nt hint: bit 0 x, bit 1 p, bit 2 r (CMI_CG_STORE_POLICY overrides: measurements)
static int cg_store_policy()
{
    static const int pol = [] { const char *e = std::getenv("CMI_CG_STORE_POLICY"); return e ? std::atoi(e) : kCgStorePolicy; }();
    return pol;
}

template <typename T>
int cg_update_impl(int64_t n, const double *rz_dev, const double *yp_dev, const T *p, const T *y, T *x, T *r, double *rr_dev,
                   double *rr_host_mirror, void *workspace, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cg_update: negative n");
    if (!rz_dev || !yp_dev || !rr_dev || !workspace) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cg_update: null scalar or workspace");
    if (n > 0 && (!y || !r || (x && !p))) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cg_update: null array");
    const int grid = fused_grid(n, vec16<T>::n);
    const int vec = aligned16(y) && aligned16(r) && (!x || (aligned16(p) && aligned16(x)));
    hipLaunchKernelGGL((cg_update_kernel<T>), dim3(grid), dim3(kBlasBlock), 0, as_stream(stream), n, rz_dev, yp_dev, p, y, x, r, (double *)workspace, vec, cg_store_policy());
    reduce_partials<double>(grid, (double *)workspace, rr_dev, 0, as_stream(stream), rr_host_mirror);
    CMI_LAUNCH_CHECK("cg_update");
    return CMI_SUCCESS;
}

template <typename T>
int cg_direction_impl(int64_t n, const double *rr_new_dev, const double *rr_old_dev, const double *yp_dev, const T *r, T *p, T *x,
                      bool with_x, void *stream)
{
    const char *who = with_x ? "cmi_cg_direction_x" : "cmi_cg_direction";
    if (n < 0) { set_error("%s: negative n", who); return CMI_ERROR_INVALID_VALUE; }
    if (!rr_new_dev || !rr_old_dev || (with_x && !yp_dev)) { set_error("%s: null scalar", who); return CMI_ERROR_INVALID_VALUE; }
    if (n == 0) return CMI_SUCCESS;
    if (!r || !p || (with_x && !x)) { set_error("%s: null array", who); return CMI_ERROR_INVALID_VALUE; }
    const int vec = aligned16(r) && aligned16(p) && (!with_x || aligned16(x));
    hipLaunchKernelGGL((cg_direction_kernel<T>), dim3(stream_grid(n, vec16<T>::n)), dim3(kBlasBlock), 0, as_stream(stream), n, rr_new_dev,
                       rr_old_dev, yp_dev, r, p, with_x ? x : (T *)nullptr, vec, cg_store_policy());
    CMI_LAUNCH_CHECK("cg_direction");
    return CMI_SUCCESS;
}

template <typename T>
int cg_update_fold_impl(int64_t n, const double *rz_dev, double *yp_dev, int npartials_yp, const T *y, T *r, void *workspace,
                        int *npartials_rr, void *stream)
{
    if (npartials_rr) *npartials_rr = 0;
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cg_update_fold: negative n");
    if (!rz_dev || !yp_dev || !workspace || !npartials_rr) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cg_update_fold: null scalar, workspace or count");
    if (npartials_yp < 1 || npartials_yp > kPartialCapacity) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cg_update_fold: partial count out of range");
    if (n > 0 && (!y || !r)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cg_update_fold: null array");
    const int groups = npartials_yp <= kFoldDirect ? 1 : (npartials_yp + kFoldChunk - 1) / kFoldChunk;
    int grid = fused_grid(n, vec16<T>::n);
    if (grid < groups) grid = groups; // the first `groups` workgroups fold; extra ones find no elements
    const int vec = aligned16(y) && aligned16(r);
    double *area_in = (double *)workspace, *area_out = (double *)workspace + kFoldArea;
    hipLaunchKernelGGL((cg_update_fold_kernel<T>), dim3(grid), dim3(kBlasBlock), 0, as_stream(stream), n, rz_dev, npartials_yp, area_in, yp_dev,
                       y, r, area_out, vec, cg_store_policy(), fold_relaxed());
    CMI_LAUNCH_CHECK("cg_update_fold");
    *npartials_rr = grid;
    return CMI_SUCCESS;
}

template <typename T>
int cg_direction_fold_impl(int64_t n, double *rr_new_dev, double *rr_host_mirror, int npartials_rr, const double *rr_old_dev,
                           const double *yp_dev, const T *r, T *p, T *x, void *workspace, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cg_direction_x_fold: negative n");
    if (!rr_new_dev || !rr_old_dev || !yp_dev || !workspace) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cg_direction_x_fold: null scalar or workspace");
    if (npartials_rr < 1 || npartials_rr > kPartialCapacity) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cg_direction_x_fold: partial count out of range");
    if (n > 0 && (!r || !p || !x)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_cg_direction_x_fold: null array");
    const int groups = npartials_rr <= kFoldDirect ? 1 : (npartials_rr + kFoldChunk - 1) / kFoldChunk;
    int grid = stream_grid(n, vec16<T>::n);
    if (grid < groups) grid = groups;
    const int vec = aligned16(r) && aligned16(p) && aligned16(x);
    double *area_in = (double *)workspace + kFoldArea;
    hipLaunchKernelGGL((cg_direction_fold_kernel<T>), dim3(grid), dim3(kBlasBlock), 0, as_stream(stream), n, npartials_rr, area_in, rr_new_dev,
                       rr_host_mirror, rr_old_dev, yp_dev, r, p, x, vec, cg_store_policy(), fold_relaxed());
    CMI_LAUNCH_CHECK("cg_direction_x_fold");
    return CMI_SUCCESS;
}

} // namespace

// Fold-ahead forms of the two steps (see fold_ahead above): the reductions' folds ride at the front of their consumers.
//   cmi_spmv_csr_dot_plan_partials_*  leaves npartials_yp partials of <y, p> in the workspace           (spmv_csr.hip)
//   cmi_cg_update_fold_*              yp <- their sum (also stored to *yp_dev); r <- r - (rz/yp) y; leaves npartials_rr partials of <r, r>
//   cmi_cg_direction_x_fold_*         rr_new <- their sum (stored to *rr_new_dev and the host mirror); x <- x + (rr_old/yp) p; p <- r + (rr_new/rr_old) p
// Same arithmetic and the same summation trees as the plain forms: bit-identical results.
CMI_API int cmi_cg_update_fold_f64(int64_t n, const double *rz_dev, double *yp_dev, int npartials_yp, const double *y, double *r,
                                   void *workspace, int *npartials_rr, void *stream)
{
    return cg_update_fold_impl<double>(n, rz_dev, yp_dev, npartials_yp, y, r, workspace, npartials_rr, stream);
}
CMI_API int cmi_cg_update_fold_f32(int64_t n, const double *rz_dev, double *yp_dev, int npartials_yp, const float *y, float *r,
                                   void *workspace, int *npartials_rr, void *stream)
{
    return cg_update_fold_impl<float>(n, rz_dev, yp_dev, npartials_yp, y, r, workspace, npartials_rr, stream);
}
CMI_API int cmi_cg_direction_x_fold_f64(int64_t n, double *rr_new_dev, double *rr_host_mirror, int npartials_rr, const double *rr_old_dev,
                                        const double *yp_dev, const double *r, double *p, double *x, void *workspace, void *stream)
{
    return cg_direction_fold_impl<double>(n, rr_new_dev, rr_host_mirror, npartials_rr, rr_old_dev, yp_dev, r, p, x, workspace, stream);
}
CMI_API int cmi_cg_direction_x_fold_f32(int64_t n, double *rr_new_dev, double *rr_host_mirror, int npartials_rr, const double *rr_old_dev,
                                        const double *yp_dev, const float *r, float *p, float *x, void *workspace, void *stream)
{
    return cg_direction_fold_impl<float>(n, rr_new_dev, rr_host_mirror, npartials_rr, rr_old_dev, yp_dev, r, p, x, workspace, stream);
}

CMI_API int cmi_cg_update_f64(int64_t n, const double *rz_dev, const double *yp_dev, const double *p, const double *y,
                              double *x, double *r, double *rr_dev, double *rr_host_mirror, void *workspace, void *stream)
{
    return cg_update_impl<double>(n, rz_dev, yp_dev, p, y, x, r, rr_dev, rr_host_mirror, workspace, stream);
}
CMI_API int cmi_cg_update_f32(int64_t n, const double *rz_dev, const double *yp_dev, const float *p, const float *y,
                              float *x, float *r, double *rr_dev, double *rr_host_mirror, void *workspace, void *stream)
{
    return cg_update_impl<float>(n, rz_dev, yp_dev, p, y, x, r, rr_dev, rr_host_mirror, workspace, stream);
}
CMI_API int cmi_cg_direction_f64(int64_t n, const double *rr_new_dev, const double *rr_old_dev, const double *r, double *p, void *stream)
{
    return cg_direction_impl<double>(n, rr_new_dev, rr_old_dev, nullptr, r, p, nullptr, false, stream);
}
CMI_API int cmi_cg_direction_f32(int64_t n, const double *rr_new_dev, const double *rr_old_dev, const float *r, float *p, void *stream)
{
    return cg_direction_impl<float>(n, rr_new_dev, rr_old_dev, nullptr, r, p, nullptr, false, stream);
}
CMI_API int cmi_cg_direction_x_f64(int64_t n, const double *rr_new_dev, const double *rr_old_dev, const double *yp_dev, const double *r,
                                   double *p, double *x, void *stream)
{
    return cg_direction_impl<double>(n, rr_new_dev, rr_old_dev, yp_dev, r, p, x, true, stream);
}
CMI_API int cmi_cg_direction_x_f32(int64_t n, const double *rr_new_dev, const double *rr_old_dev, const double *yp_dev, const float *r,
                                   float *p, float *x, void *stream)
{
    return cg_direction_impl<float>(n, rr_new_dev, rr_old_dev, yp_dev, r, p, x, true, stream);
}

// <x, y> of float vectors as a DOUBLE in device memory (the scalar type of the fused CG steps)
CMI_API int cmi_blas_dotd_f32(int64_t n, const float *x, const float *y, double *result_dev, void *workspace, void *stream)
{
    if (n < 0) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_dotd: negative n");
    if (!result_dev || !workspace) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_dotd: null result or workspace");
    if (n > 0 && (!x || !y)) return fail(CMI_ERROR_INVALID_VALUE, "cmi_blas_dotd: null array");
    const int grid = blas_grid(n, vec16<float>::n);
    const int vec = aligned16(x) && aligned16(y);
    hipLaunchKernelGGL((dot_partial_kernel<float>), dim3(grid), dim3(kBlasBlock), 0, as_stream(stream), n, x, y, (double *)workspace, vec);
    reduce_partials<double>(grid, (double *)workspace, result_dev, 0, as_stream(stream));
    CMI_LAUNCH_CHECK("dotd");
    return CMI_SUCCESS;
}
