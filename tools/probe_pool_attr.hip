// Can a pointer from the stream-ordered pool (hipMallocAsync) be told from a hipMalloc one?  Prints what
// hipPointerGetAttribute(HIP_POINTER_ATTRIBUTE_MEMPOOL_HANDLE) and hipPointerGetAttributes say for both, and the pool's release threshold.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
int main() {
    hipStream_t s;
    hipStreamCreate(&s);
    void *a = nullptr, *b = nullptr;
    printf("hipMalloc -> %d\n", (int)hipMalloc(&a, 1 << 20));
    printf("hipMallocAsync -> %d\n", (int)hipMallocAsync(&b, 1 << 20, s));
    hipStreamSynchronize(s);
    for (void* p : {a, b, (void*)((char*)b + 4096)}) {
        hipMemPool_t pool = (hipMemPool_t)(uintptr_t)0xdead;
        hipError_t e = hipPointerGetAttribute(&pool, HIP_POINTER_ATTRIBUTE_MEMPOOL_HANDLE, (hipDeviceptr_t)p);
        printf("%p: MEMPOOL_HANDLE rc=%d (%s) pool=%p", p, (int)e, hipGetErrorString(e), (void*)pool);
        hipPointerAttribute_t at;
        e = hipPointerGetAttributes(&at, p);
        printf("  attrs rc=%d type=%d device=%d alloc_flags=%u\n", (int)e, (int)at.type, at.device, at.allocationFlags);
        if (e == hipSuccess && pool && pool != (hipMemPool_t)(uintptr_t)0xdead) {
            uint64_t thr = 77;
            e = hipMemPoolGetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr);
            printf("    release threshold rc=%d %llu\n", (int)e, (unsigned long long)thr);
        }
    }
    hipMemPool_t def = nullptr;
    printf("default pool rc=%d %p\n", (int)hipDeviceGetDefaultMemPool(&def, 0), (void*)def);
    (void)hipGetLastError();
    return 0;
}
