// Caching device allocator behind every buffer of the library (factors, plans, kvx_dev_malloc).
// hipMalloc / hipFree cost 0.1-0.7 ms each on MI355X (mapping / unmapping); a factor owns ~40 buffers and an interior-point
// call ~60 vectors, so one `solvers.lp` call spent ~25 ms of its 80 ms releasing memory.  pool_free() keeps blocks of up to
// 1 GiB on a free list (total capped, KVX_POOL_MAX_MB, default 8192; 0 = no caching) and pool_malloc() hands them out again
// (a block up to 1/8 larger than the request may be reused).  pool_free() waits for the device first, exactly as hipFree does,
// so a block is never recycled under a kernel that still reads it.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace kvx {
hipError_t pool_malloc(void **p, size_t bytes);
hipError_t pool_free(void *p);
// Streams (non-blocking) and events are recycled too: creating and destroying the nine streams and dozen events of a factor cost
// ~5 ms + ~13 ms per factor lifetime.  A stream / event must be idle when it is put back (kvx_*_free synchronise first).
hipError_t pool_stream_get(hipStream_t *s, bool high = false);
void pool_stream_put(hipStream_t s, bool high = false);
hipError_t pool_event_get(hipEvent_t *e, bool timing);
void pool_event_put(hipEvent_t e, bool timing);
void pool_release_all();          // really free everything cached (tests, out-of-memory retry)
}  // namespace kvx
