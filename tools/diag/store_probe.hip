// Achievable store / copy rates on one MI355X with plain, nontemporal and sc1 16-byte stores (545 MB tensors, the block tensors' size).
// hipcc --offload-arch=gfx950 -O3 tools/diag/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE> __device__ __forceinline__ void st(f4* p, f4 v) {
    if (MODE == 0) *p = v;
    else if (MODE == 1) __builtin_nontemporal_store(v, p);
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
template <int MODE> __global__ __launch_bounds__(256) void k_fill(f4* y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) st<MODE>(y + i, f4{1.f, 2.f, 3.f, 4.f});
}
template <int MODE, int LD> __global__ __launch_bounds__(256) void k_copy(const f4* x, f4* y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        f4 v = LD ? __builtin_nontemporal_load(x + i) : x[i];
        st<MODE>(y + i, v * 2.f);
    }
}
template <int MODE> __global__ __launch_bounds__(256) void k_1r2w(const f4* x, f4* y, f4* z, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        f4 v = x[i];
        st<MODE>(y + i, v * 2.f);
        st<MODE>(z + i, v + 1.f);
    }
}
template <class F> float timeit(F f, int n = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / n;
}
int main() {
    const long bytes = 64l * 32 * 65 * 1024 * 4, n = bytes / 16;
    f4 *x, *y, *z; hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&z, bytes);
    hipMemset(x, 0, bytes);
    const char* nm[3] = {"plain", "nt", "sc1"};
    for (int grid : {2048, 8192, 65536}) {
        printf("grid %d\n", grid);
#define ROW(M)                                                                                                           \
        { float f = timeit([&] { k_fill<M><<<grid, 256>>>(y, n); });                                                     \
          float c = timeit([&] { k_copy<M, 0><<<grid, 256>>>(x, y, n); });                                               \
          float cn = timeit([&] { k_copy<M, 1><<<grid, 256>>>(x, y, n); });                                              \
          float w = timeit([&] { k_1r2w<M><<<grid, 256>>>(x, y, z, n); });                                               \
          printf("  %-5s fill %.3f ms %.2f TB/s | copy %.3f ms %.2f TB/s | copy nt-load %.3f ms %.2f TB/s | 1R2W %.3f ms %.2f TB/s\n", nm[M], f, \
                 bytes / f / 1e9, c, 2 * bytes / c / 1e9, cn, 2 * bytes / cn / 1e9, w, 3 * bytes / w / 1e9); }
        ROW(0) ROW(1) ROW(2)
    }
    return 0;
}
