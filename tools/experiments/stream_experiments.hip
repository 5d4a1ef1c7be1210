// NOT part of libtobac_flow_hip.so (round 6).  The entry points of two scheduling experiments that were measured in round 5 and
// not adopted -- the floods of finished windows on a stream confined to k CUs of every XCD, or on a low-priority stream
// (profiles/round5_scheduling_experiments.txt: both slower than a plain second stream) -- kept here for the record with the
// probe that shows which CU a mask bit selects (tools/cu_mask_probe.py builds and loads this file by itself:
//     hipcc --offload-arch=gfx950 -O2 -fPIC -shared -I tobac_flow_amd/csrc tools/experiments/stream_experiments.hip -o gpurun_out/libstream_experiments.so).
// Why they left the product ABI: a process that has called hipStreamDestroy on a CU-masked stream (hipExtStreamCreateWithCUMask)
// segfaults on this ROCm (7.2) inside a LATER large device allocation -- reproduced twice with the GPU test suite
// (profiles/round6_stream_destroy_segfault.txt): in round 5 with, in round 6 without a stream-ordered allocation in
// tf_debug_cu_histogram, so the allocation was not the cause; the stream's destruction is.  Round 5 shipped a tf_stream_destroy
// that never destroyed; the product now creates no such stream, and nothing is left to destroy.
#include "tf_common.h"
#include <stdarg.h>
void tf_set_error(const char *fmt, ...) { (void)fmt; }

// ---- streams confined to a part of the chip --------------------------------------------------------------------------
// A caller that runs the floods of finished windows BESIDE the flow (parallel.detect_stack_windows) gives them a stream
// whose kernels may only occupy some of the CUs: the flow's iteration kernel then keeps the LDS of all the others to itself
// (its faster two-part chain needs 39 KB per workgroup, four per CU -- csrc/farneback.hip), while an unrestricted flood
// stream displaces iteration workgroups on every CU.  mask_words: bit i of the mask = CU i in the runtime's numbering
// (hipExtStreamCreateWithCUMask; tf_debug_cu_histogram shows which XCD / CU a bit maps to on this part).
extern "C" int tf_stream_create_cu_mask(const uint32_t *mask_words, int n_words, void **stream_out)
{
    TF_REQUIRE(mask_words && n_words > 0 && n_words <= 32 && stream_out, "tf_stream_create_cu_mask: bad arguments");
    hipStream_t s = nullptr;
    TF_CHECK_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words));
    *stream_out = (void *)s;
    return TF_OK;
}
// a stream of the LOWEST (low != 0) or the highest priority the device offers: the flood thread of parallel.detect_stack_windows
// runs on a low-priority stream, so that the flow's workgroups are dispatched first and the floods take what is left
extern "C" int tf_stream_create_priority(int low, void **stream_out)
{
    TF_REQUIRE(stream_out, "tf_stream_create_priority: null pointer");
    int least = 0, greatest = 0;
    TF_CHECK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s = nullptr;
    TF_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, low ? least : greatest));
    *stream_out = (void *)s;
    return TF_OK;
}
// tf_stream_destroy: synchronise, then hipStreamDestroy (see the head of this file before calling it in a long-lived process).  (Round 5 shipped a version that only RETIRED the stream -- kept it alive
// until the process ended -- after the GPU suite had crashed inside a later test's 132-GiB hipMalloc; two changes had gone in
// together and that one was the wrong suspect.  The cause was tf_debug_cu_histogram's stream-ordered scratch, below: a block
// freed with hipFreeAsync stays in the device's memory pool tagged with the stream it was freed on, the test then destroyed
// that stream, and the pool trim a failing large allocation triggers walked the dead stream.  With plain hipMalloc / hipFree
// there nothing of the runtime's refers to the stream once it is idle, and destroying it is what it should be -- round 6 runs
// the suite with the real destroy: DESIGN.md section 7.)
extern "C" int tf_stream_destroy(void *stream)
{
    if (!stream) return TF_OK;
    TF_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    TF_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    return TF_OK;
}

// where do the workgroups of a stream run?  hist[xcc * 256 + (se, sh, cu) byte of HW_ID] += 1 per workgroup
__global__ void __launch_bounds__(64)
k_debug_cu_histogram(int *hist)
{
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 7u;           // HW_REG_XCC_ID, bits 3:0
        const unsigned hw = (__builtin_amdgcn_s_getreg((16 - 1) << 11 | 4)) & 0xffffu;   // HW_REG_HW_ID (id 4), 16 bits
        atomicAdd(&hist[xcc * 256 + ((hw >> 8) & 0xffu)], 1);
    }
    // (long enough for every CU of the mask to be handed workgroups)
    for (int i = 0; i < 2000; i++) __builtin_amdgcn_s_sleep(8);
}
extern "C" int tf_debug_cu_histogram(void *stream, int n_workgroups, int *hist_host_2048)
{
    TF_REQUIRE(hist_host_2048 && n_workgroups > 0, "tf_debug_cu_histogram: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    // (plain hipMalloc / hipFree: a stream-ordered allocation would leave a block of the runtime's memory pool tied to `stream`,
    // and the caller may destroy that stream -- a later out-of-memory trim of the pool then walks a dead stream: the crash
    // the first version of this function caused in a LATER test's 132-GiB allocation)
    int *d = nullptr;
    TF_CHECK_HIP(hipMalloc((void **)&d, 2048 * sizeof(int)));
    hipError_t e = hipMemsetAsync(d, 0, 2048 * sizeof(int), s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_debug_cu_histogram, dim3((unsigned)n_workgroups), dim3(64), 0, s, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hist_host_2048, d, 2048 * sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) { tf_set_error("tf_debug_cu_histogram: %s", hipGetErrorString(e)); return TF_EHIP; }
    return TF_OK;
}

