// Issue rate of FP64 vector instructions on gfx950 against FP32 (development aid for the VALU floors of the kernels that
// accumulate in double as OpenCV does: polynomial expansion, Sobel, the box sums of the Farneback iteration).
// 16 independent chains per thread; modes: v_fma_f32, v_fma_f64, v_mul_f64 + v_add_f64, v_cvt_f64_f32 + v_add_f64.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/microbench/f64_rate.hip -fno-slp-vectorize -w -o tools/microbench/f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b)
{
    float x[16]; double y[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { x[i] = threadIdx.x * 0.001f + i; y[i] = x[i]; }
    const double da = a, db = b;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = __builtin_fmaf(x[i], a, b);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) y[i] = __builtin_fma(y[i], da, db);
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; i++) { y[i] = y[i] * da; y[i] = y[i] + db; }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) { x[i] = x[i] + b; y[i] = y[i] + (double)x[i]; }      // f32 add, cvt, f64 add
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += x[i] + (float)y[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> static void run(const char *name, float *d, int waves_per_simd, double instr_per_iter)
{
    const int iters = 20000, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // cycles one SIMD spends per wave-level instruction, assuming 2.4 GHz (the chip clocks 2.1 - 2.4 GHz under load)
    printf("%-34s waves/SIMD %d  %.3f ms  %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / iters / waves_per_simd / instr_per_iter);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 256 * 16 * sizeof(float));
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("16 v_fma_f32", d, w, 16);
        run<1>("16 v_fma_f64", d, w, 16);
        run<2>("16 v_mul_f64 + 16 v_add_f64", d, w, 32);
        run<3>("16 x (add_f32, cvt_f64_f32, add_f64)", d, w, 48);
    }
    hipFree(d);
    return 0;
}
