// Issue rate of packed FP32 instructions on gfx950 (development aid): the same number of float operations as
// scalar v_fma_f32 / v_mul_f32 + v_add_f32 and as v_pk_* instructions, 8 independent chains per thread.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench/pk_rate.hip -fno-slp-vectorize -w -o tools/microbench/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {                      // 16 scalar fma
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, b);
        } else if (MODE == 1) {               // 8 packed fma (same 16 float fma)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                v2f v = {x[i], x[i + 1]};
                v = __builtin_elementwise_fma(v, (v2f){a, a}, (v2f){b, b});
                x[i] = v.x; x[i + 1] = v.y;
            }
        } else if (MODE == 2) {               // 16 scalar mul + 16 scalar add
#pragma unroll
            for (int i = 0; i < 16; i++) { x[i] = x[i] * a; x[i] = x[i] + b; }
        } else {                              // 8 packed mul + 8 packed add
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                v2f v = {x[i], x[i + 1]};
                v = v * (v2f){a, a};
                v = v + (v2f){b, b};
                x[i] = v.x; x[i + 1] = v.y;
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> static void run(const char *name, float *d, int waves_per_simd)
{
    const int iters = 20000, blocks = 256 * waves_per_simd;       // 256 CUs x (4 SIMDs x waves_per_simd waves = waves_per_simd blocks of 256)
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop_ops = 16.0 * iters * 256.0 * blocks;        // float operations (fma = 1) per launch
    printf("%-28s waves/SIMD %d  %.3f ms  %.1f G float-ops/s  (%.2f cycles per wave per 16 ops at 2.4 GHz)\n", name, waves_per_simd, ms,
           flop_ops / ms / 1e6, ms * 1e-3 * 2.4e9 / iters / waves_per_simd);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 256 * 16 * sizeof(float));
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("16 v_fma_f32", d, w);
        run<1>("8 v_pk_fma_f32", d, w);
        run<2>("16 v_mul + 16 v_add", d, w);
        run<3>("8 v_pk_mul + 8 v_pk_add", d, w);
    }
    hipFree(d);
    return 0;
}
