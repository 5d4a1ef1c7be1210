// What does one step of a sequential double-precision chain over LDS cost on gfx950?  One workgroup of 64 threads, 25
// lanes active (the row-sum chains of k_fb_iter), 116 steps per scan, repeated; variants isolate the dependent adds, the
// LDS reads and the LDS writes.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off chain_lds.hip -o chain_lds && ./chain_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#define VS 137
#define N 116
template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, long long *cyc, int reps, int lanes)
{
    __shared__ double lds[25 * VS];
    for (int i = threadIdx.x; i < 25 * VS; i += 64) lds[i] = 1.0 + i * 1e-3;
    __syncthreads();
    double acc = 0;
    const long long t0 = clock64();
    if ((int)threadIdx.x < lanes) {
        double *row = lds + (threadIdx.x % 25) * VS;
        for (int r = 0; r < reps; r++) {
            double g = acc, sub = 0.5;
            if (MODE == 0) {                                            // dependent adds only (operands in registers)
                double d = row[r & 7];
#pragma unroll 8
                for (int i = 0; i < N; i++) g += d;
            } else if (MODE == 1) {                                     // + sub off the chain
                double d = row[r & 7], e = row[8];
#pragma unroll 8
                for (int i = 0; i < N; i++) { g += d - e; e = d; d += 1.0; }
            } else if (MODE == 2) {                                     // reads + sub + add, no writes
                double mn[8], nx[8];
                for (int k = 0; k < 8; k++) { mn[k] = row[12 + k]; nx[k] = row[k]; }
#pragma unroll 1
                for (int i0 = 0; i0 < 112; i0 += 8) {
                    double mn2[8], nx2[8];
                    for (int k = 0; k < 8; k++) { const int i = min(i0 + 8 + k, N - 1); mn2[k] = row[i + 12]; nx2[k] = row[i]; }
                    double d[8]; d[0] = mn[0] - sub;
                    for (int k = 1; k < 8; k++) d[k] = mn[k] - nx[k - 1];
                    for (int k = 0; k < 8; k++) g += d[k];
                    sub = nx[7];
                    for (int k = 0; k < 8; k++) { mn[k] = mn2[k]; nx[k] = nx2[k]; }
                }
            } else {                                                    // the kernel's loop: reads + sub + add + writes
                double mn[8], nx[8];
                for (int k = 0; k < 8; k++) { mn[k] = row[12 + k]; nx[k] = row[k]; }
#pragma unroll 1
                for (int i0 = 0; i0 < 112; i0 += 8) {
                    double mn2[8], nx2[8];
                    for (int k = 0; k < 8; k++) { const int i = min(i0 + 8 + k, N - 1); mn2[k] = row[i + 12]; nx2[k] = row[i]; }
                    double d[8]; d[0] = mn[0] - sub;
                    for (int k = 1; k < 8; k++) d[k] = mn[k] - nx[k - 1];
                    for (int k = 0; k < 8; k++) { g += d[k]; if (MODE == 3) row[i0 + k] = g; }
                    sub = nx[7];
                    for (int k = 0; k < 8; k++) { mn[k] = mn2[k]; nx[k] = nx2[k]; }
                }
            }
            acc = g * 1e-9;
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = acc;
}
int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const int reps = 2000;
    for (int lanes : {25, 64}) for (int mode = 0; mode < 4; mode++) {
        for (int it = 0; it < 2; it++) {
            if (mode == 0) k<0><<<1, 64>>>(out, cyc, reps, lanes);
            if (mode == 1) k<1><<<1, 64>>>(out, cyc, reps, lanes);
            if (mode == 2) k<2><<<1, 64>>>(out, cyc, reps, lanes);
            if (mode == 3) k<3><<<1, 64>>>(out, cyc, reps, lanes);
            hipDeviceSynchronize();
        }
        long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) k<0><<<1, 64>>>(out, cyc, reps, lanes);
        if (mode == 1) k<1><<<1, 64>>>(out, cyc, reps, lanes);
        if (mode == 2) k<2><<<1, 64>>>(out, cyc, reps, lanes);
        if (mode == 3) k<3><<<1, 64>>>(out, cyc, reps, lanes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("lanes %2d mode %d (%s): %.1f counter ticks per step, %.1f ns per step, %.2f us per 116-step scan\n", lanes, mode,
               mode == 0 ? "dependent adds" : mode == 1 ? "sub + add" : mode == 2 ? "LDS reads + sub + add" : "reads + sub + add + LDS writes",
               (double)h / reps / N, ms * 1e6 / reps / N, ms * 1e3 / reps);
    }
    return 0;
}
