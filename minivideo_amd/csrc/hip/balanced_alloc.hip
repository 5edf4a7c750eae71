// balanced_alloc.hip -- device buffers spread over both halves of the MI355X memory system.
//
// Measured on MI355X (NPS1 / SPX; tools/placement_map.py, profiles/r02d_placement_map.log): device memory comes in regions
// of tens of GB that alternate between two halves of the memory system, and a kernel whose concurrent streams all live in
// one half sees half the bandwidth -- the 1080p Baseline launch takes 9.6-10.0 ms with planes and RGB in regions of the
// same kind and 8.2-8.4 ms with them in different kinds, whatever the offsets inside a region.  hipMalloc hands out one
// region after the other, so which case a caller gets is chance.  This file (a) tells two addresses' halves apart with a
// timing probe and (b) builds buffers out of 256-MB physical chunks (HIP virtual memory management) taken alternately
// from both halves, so that every stream of a batch is spread over the whole memory system.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <vector>

#include "minivideo_hotpath.h"

namespace {

// even workgroups stream 16-byte stores over window a, odd ones over window b
__global__ __launch_bounds__(256) void pair_write_kernel(uint4 *a, uint4 *b, size_t n16, uint32_t tag)
{
    uint4 *w = (blockIdx.x & 1) ? b : a;
    const size_t stride = (size_t)(gridDim.x >> 1) * blockDim.x;
    const uint4 v = make_uint4(tag, tag, tag, tag);
    for (size_t i = (size_t)(blockIdx.x >> 1) * blockDim.x + threadIdx.x; i < n16; i += stride) w[i] = v;
}

} // namespace

// Time `reps` passes of concurrent streaming writes over two device windows of `bytes` each (clobbers both).
MVHP_EXPORT int mvhp_probe_pair(int device, void *a, void *b, size_t bytes, int reps, float *ms)
{
    if (!a || !b || bytes < 4096 || reps <= 0 || !ms) return MVHP_FAILURE;
    if (hipSetDevice(device) != hipSuccess) return MVHP_FAILURE;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MVHP_FAILURE;
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(pair_write_kernel, dim3(2048), dim3(256), 0, 0, (uint4 *)a, (uint4 *)b, n16, 0u);   // warm
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(pair_write_kernel, dim3(2048), dim3(256), 0, 0, (uint4 *)a, (uint4 *)b, n16, (uint32_t)r);
    hipEventRecord(e1, 0);
    int rc = MVHP_SUCCESS;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(ms, e0, e1) != hipSuccess) rc = MVHP_FAILURE;
    *ms /= (float)reps;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc;
}
