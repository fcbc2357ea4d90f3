// lds_sync.hip.h -- flags between the waves of one workgroup, in LDS.
//
// Relaxed accesses + explicit lgkmcnt waits.  A workgroup-scope release/acquire would also order GLOBAL
// memory, i.e. put an s_waitcnt vmcnt(0) (a drain of every outstanding band / row store) on the DP's
// critical path; LDS itself is processed in order per wave, so "data writes, lgkmcnt(0), flag write" on
// one side and "flag read, then data reads" on the other is all that is needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pwa {

__device__ __forceinline__ uint32_t lds_peek(uint32_t* p) {
    const uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");   // data reads stay below the flag read
    return v;
}
__device__ __forceinline__ void lds_post(uint32_t* p, uint32_t v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS data accesses are done
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// flag write that only has to stay behind this wave's earlier LDS WRITES: one wave's LDS operations are performed in issue
// order, so no wait is needed (lds_post's lgkmcnt(0) also waits for read DATA to come back, ~100 cycles per chunk)
__device__ __forceinline__ void lds_post_after_writes(uint32_t* p, uint32_t v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace pwa
