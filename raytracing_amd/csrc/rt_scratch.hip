// rt_scratch.hip — the fixed device workspace behind rt_hip_render_dev_noalloc (include/rt_hip.h): callers that cannot
// call hipMalloc (HIP graph capture, tight serving loops, harnesses that forbid it) get the same kernels with their
// scratch buffers — work counter, streamed shared-origin table, continuing-ray rings, fixed-point framebuffer — carved
// out of one file-scope device array instead of the context's grow-only allocations.
//
// Every GPU the library runs on holds its own copy of the array (kNoAllocScratchBytes, until the process ends) at its
// own address.  In host code the array's name is only a stand-in for that address: it is resolved once per device with
// hipGetSymbolAddress while that device is current (rt_hip_init), and only the resolved pointer is ever passed to
// kernels, memsets or copies.
#include "rt_internal.h"

namespace rt {

static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

hipError_t noalloc_scratch_address(void** out)
{
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    *out = e == hipSuccess ? ptr : nullptr;
    return e;
}

}  // namespace rt
