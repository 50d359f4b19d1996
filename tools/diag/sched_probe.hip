// What does the WORK DISTRIBUTION cost a streaming kernel on MI355X?  The same 1 read + 2 writes per 16 bytes (the block forward's traffic,
// 545 MB tensors) as: one-shot grid (a workgroup per 64 KB chunk, dispatched by the hardware as CUs free up), persistent workgroups with a
// static grid-stride walk, persistent with the XCD-ordered static walk the conv kernels use, persistent with a DYNAMIC walk (next chunk
// from an atomic counter, fetched one chunk ahead).  hipcc --offload-arch=gfx950 -O3 tools/diag/sched_probe.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int CH = 16;                                   // 16-byte pieces per thread and chunk: a chunk = 256 x 16 x 16 B = 64 KB
__device__ __forceinline__ void chunk(const f4* x, f4* y, f4* z, long c, long nchunks, int tid, bool rd) {
    const long base = c * 256 * CH + tid;
    f4 v[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) v[i] = rd ? x[base + i * 256] : f4{1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int i = 0; i < CH; ++i) { y[base + i * 256] = v[i] * 2.f; if (z) z[base + i * 256] = v[i] + 1.f; }
}
__global__ __launch_bounds__(256) void k_oneshot(const f4* x, f4* y, f4* z, long nchunks, bool rd) { chunk(x, y, z, blockIdx.x, nchunks, threadIdx.x, rd); }
__global__ __launch_bounds__(256) void k_static(const f4* x, f4* y, f4* z, long nchunks, bool rd) {
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) chunk(x, y, z, c, nchunks, threadIdx.x, rd);
}
__global__ __launch_bounds__(256) void k_xcd(const f4* x, f4* y, f4* z, long nchunks, bool rd) {
    const long per = nchunks >> 3;
    for (long v = blockIdx.x; v < nchunks; v += gridDim.x) {
        const long c = v < (per << 3) ? (v & 7) * per + (v >> 3) : v;
        chunk(x, y, z, c, nchunks, threadIdx.x, rd);
    }
}
__global__ __launch_bounds__(256) void k_dynamic(const f4* x, f4* y, f4* z, long nchunks, bool rd, unsigned* counter) {
    __shared__ unsigned nxt;
    if (threadIdx.x == 0) nxt = atomicAdd(counter, 1u);
    __syncthreads();
    unsigned c = nxt;
    while (c < nchunks) {
        __syncthreads();
        if (threadIdx.x == 0) nxt = atomicAdd(counter, 1u);      // one chunk ahead: its latency hides under this chunk's traffic
        chunk(x, y, z, c, nchunks, threadIdx.x, rd);
        __syncthreads();
        c = nxt;
    }
}
template <class F> float timeit(F f, int n = 20) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    (void)hipEventRecord(a);
    for (int i = 0; i < n; ++i) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / n;
}
int main() {
    const long bytes = 64l * 32 * 65 * 1024 * 4, nchunks = bytes / (256 * CH * 16);
    f4 *x, *y, *z; unsigned* cnt;
    (void)hipMalloc(&x, bytes); (void)hipMalloc(&y, bytes); (void)hipMalloc(&z, bytes); (void)hipMalloc(&cnt, 4);
    (void)hipMemset(x, 0, bytes);
    printf("%ld chunks of 64 KB\n", nchunks);
    for (int what = 0; what < 3; ++what) {
        const bool rd = what > 0; f4* zz = what == 2 ? z : nullptr;
        const double gb = bytes * (what == 0 ? 1.0 : what == 1 ? 2.0 : 3.0) / 1e9;
        printf("%s\n", what == 0 ? "fill (1 W)" : what == 1 ? "copy (1 R 1 W)" : "forward-like (1 R 2 W)");
        float t = timeit([&] { k_oneshot<<<nchunks, 256>>>(x, y, zz, nchunks, rd); });
        printf("  one-shot grid              %.3f ms %.2f TB/s\n", t, gb / t);
        for (int per : {2, 4, 8}) {
            const int grid = 256 * per;
            t = timeit([&] { k_static<<<grid, 256>>>(x, y, zz, nchunks, rd); });
            printf("  persistent x%d static       %.3f ms %.2f TB/s\n", per, t, gb / t);
            t = timeit([&] { k_xcd<<<grid, 256>>>(x, y, zz, nchunks, rd); });
            printf("  persistent x%d XCD-ordered  %.3f ms %.2f TB/s\n", per, t, gb / t);
            t = timeit([&] { (void)hipMemsetAsync(cnt, 0, 4); k_dynamic<<<grid, 256>>>(x, y, zz, nchunks, rd, cnt); });
            printf("  persistent x%d dynamic      %.3f ms %.2f TB/s (incl. the counter memset)\n", per, t, gb / t);
        }
    }
    return 0;
}
