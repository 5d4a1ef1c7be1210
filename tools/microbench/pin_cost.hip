// What does a pinned 1.88 GB host block cost, by the way it is made?  (round 6: the first pass of the drop-in sequence pays
// hipHostMalloc five times, 120 - 130 ms each -- four times the block's DMA.)
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/pin_cost.hip -o gpurun_out/pin_cost && gpurun_out/pin_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void touch_parallel(char *p, size_t n, int threads)
{
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++)
        th.emplace_back([=] { for (size_t i = n / threads * t; i < n / threads * (t + 1); i += 4096) p[i] = 1; });
    for (auto &t : th) t.join();
}

int main()
{
    const size_t n = (size_t)1883000000 / 4096 * 4096;
    void *d = nullptr;
    hipMalloc(&d, n);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        void *p = nullptr;
        hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
        double t1 = now();
        hipMemcpy(p, d, n, hipMemcpyDeviceToHost);
        double t2 = now();
        printf("hipHostMalloc(default)            %7.1f ms (%s)   D->H %6.1f ms\n", (t1 - t0) * 1e3, hipGetErrorString(e), (t2 - t1) * 1e3);
        hipHostFree(p);
    }
    for (unsigned flags : {(unsigned)hipHostMallocNonCoherent, (unsigned)hipHostMallocNumaUser, (unsigned)hipHostMallocPortable}) {
        double t0 = now();
        void *p = nullptr;
        hipError_t e = hipHostMalloc(&p, n, flags);
        double t1 = now();
        if (e == hipSuccess) hipMemcpy(p, d, n, hipMemcpyDeviceToHost);
        double t2 = now();
        printf("hipHostMalloc(flags 0x%x)       %7.1f ms (%s)   D->H %6.1f ms\n", flags, (t1 - t0) * 1e3, hipGetErrorString(e), (t2 - t1) * 1e3);
        if (e == hipSuccess) hipHostFree(p); else (void)hipGetLastError();
    }
    for (int huge = 0; huge < 2; huge++) for (int threads : {1, 16}) {
        double t0 = now();
        char *p = (char *)mmap(nullptr, n + (2 << 20), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        char *q = (char *)(((uintptr_t)p + (2 << 20) - 1) & ~(uintptr_t)((2 << 20) - 1));
        if (huge) madvise(q, n, MADV_HUGEPAGE);
        touch_parallel(q, n, threads);
        double t1 = now();
        hipError_t e = hipHostRegister(q, n, hipHostRegisterDefault);
        double t2 = now();
        if (e == hipSuccess) hipMemcpy(q, d, n, hipMemcpyDeviceToHost);
        double t3 = now();
        printf("mmap%s + touch (%2d thr) %7.1f ms  + hipHostRegister %7.1f ms (%s)   D->H %6.1f ms\n", huge ? " + MADV_HUGEPAGE" : "                ",
               threads, (t1 - t0) * 1e3, (t2 - t1) * 1e3, hipGetErrorString(e), (t3 - t2) * 1e3);
        if (e == hipSuccess) { double t4 = now(); hipHostUnregister(q); printf("      hipHostUnregister %7.1f ms\n", (now() - t4) * 1e3); } else (void)hipGetLastError();
        munmap(p, n + (2 << 20));
    }
    // one big arena, blocks carved out of it: the cost of a 4 x larger allocation
    {
        double t0 = now();
        void *p = nullptr;
        hipError_t e = hipHostMalloc(&p, 4 * n, hipHostMallocDefault);
        printf("hipHostMalloc(4 x)                %7.1f ms (%s)\n", (now() - t0) * 1e3, hipGetErrorString(e));
        if (e == hipSuccess) hipHostFree(p);
    }
    hipFree(d);
    return 0;
}
