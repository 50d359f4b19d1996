// The exact register numbers and instruction sequence of the k_nrb_bwd_fused build that returned a run-to-run different dW2[0][0]
// (v184 = lo lane of the last packed multiply-add before its source pair v[64:65] is overwritten), repeated under load.
//   hipcc --offload-arch=gfx950 -O3 -x hip pk_war_exact.cpp -o pk_war_exact.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k(float* out, int iters, int variant) {
    const float t = (threadIdx.x & 63) * 0.001f;
    float r184 = 0.f, r185 = 0.f;
    for (int it = 0; it < iters; ++it) {
        float lo, hi;
        asm volatile(
            "v_mov_b32 v64, %2\n v_mov_b32 v65, %3\n v_mov_b32 v60, %4\n v_mov_b32 v61, %5\n"
            "v_mov_b32 v66, 2.0\n v_mov_b32 v67, 4.0\n v_mov_b32 v70, 0.5\n v_mov_b32 v71, 1.0\n v_mov_b32 v72, -1.0\n v_mov_b32 v73, -2.0\n"
            "v_mov_b32 v184, %0\n v_mov_b32 v185, %1\n"
            "v_pk_fma_f32 v[192:193], v[64:65], v[60:61], v[192:193]\n"
            "v_pk_fma_f32 v[198:199], v[64:65], v[66:67], v[198:199] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[190:191], v[64:65], v[66:67], v[190:191] op_sel:[1,0,0]\n"
            "v_pk_fma_f32 v[196:197], v[64:65], v[70:71], v[196:197] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[188:189], v[64:65], v[70:71], v[188:189] op_sel:[1,0,0]\n"
            "v_pk_fma_f32 v[194:195], v[64:65], v[72:73], v[194:195] op_sel_hi:[0,1,1]\n"
            "v_pk_fma_f32 v[186:187], v[64:65], v[72:73], v[186:187] op_sel:[1,0,0]\n"
            "v_pk_fma_f32 v[184:185], v[64:65], v[60:61], v[184:185] op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"
            "v_pk_mov_b32 v[64:65], v[60:61], v[66:67] op_sel:[1,0]\n"
            "v_pk_mov_b32 v[66:67], v[66:67], v[70:71] op_sel:[1,0]\n"
            "v_pk_mov_b32 v[68:69], v[70:71], v[72:73] op_sel:[1,0]\n"
            "v_pk_mov_b32 v[60:61], v[72:73], v[60:61] op_sel:[1,0]\n"
            "v_mov_b32 %0, v184\n v_mov_b32 %1, v185\n"
            : "+v"(r184), "+v"(r185)
            : "v"(1.0f + t), "v"(2.0f + t), "v"(3.0f), "v"(5.0f)
            : "v60", "v61", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v184", "v185", "v186", "v187", "v188", "v189",
              "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199");
        (void)lo; (void)hi;
    }
    out[(blockIdx.x * 256 + threadIdx.x) * 2] = r184;
    out[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = r185;
}

int main() {
    const int grid = 256 * 8, iters = 8192, n = grid * 256;
    float* out; (void)hipMalloc(&out, n * 2 * 4);
    float* h = new float[n * 2];
    for (int rep = 0; rep < 3; ++rep) {
        k<<<grid, 256>>>(out, iters, 0);
        (void)hipMemcpy(h, out, n * 2 * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; ++i) {
            const float t = (i % 64) * 0.001f;
            float e0 = 0.f, e1 = 0.f;                    // lo = v64 * v61 (5.0), hi = v65 * v60 (3.0)
            for (int it = 0; it < iters; ++it) { e0 = fmaf(1.0f + t, 5.0f, e0); e1 = fmaf(2.0f + t, 3.0f, e1); }
            if (h[2 * i] != e0 || h[2 * i + 1] != e1) { if (bad < 3) printf("   lane %d: got %.3f %.3f want %.3f %.3f\n", i, h[2 * i], h[2 * i + 1], e0, e1); ++bad; }
        }
        printf("exact-register sequence, pass %d: %ld of %d lanes wrong\n", rep, bad, n);
    }
    return 0;
}
