// repro_stale.hip -- what makes a kernel read a stale value another kernel of the SAME stream wrote before it?
// (round-1 finding: with the library's scratch in hipMallocAsync memory, k_gao workgroups on some XCDs read 0 for the
// flagged-chunk counter that k_batch_recover had just incremented.)
//
// Sequence per trial, all on one stream:
//   plant   every workgroup reads word[0]            (puts the line into the L2 of every XCD)
//   clear   hipMemsetAsync(word, 0) or a kernel store
//   plant   again (now the zero is what the L2s hold)
//   bump    ONE workgroup: atomicAdd(word, 123)      (device-scope atomic, as flag_chunks does)  | or a plain store
//   read    every workgroup reads word[0] with `mode` and records (XCC id, value)
// and counts the workgroups that did not see 123, per allocation kind.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/repro_stale.hip -o tools/repro_stale
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <vector>
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
            exit(2);                                                               \
        }                                                                          \
    } while (0)

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xf;
}
__global__ void k_plant(const uint32_t* w, uint32_t* sink) {
    if (threadIdx.x == 0) sink[blockIdx.x] = w[0] + w[16];
}
__global__ void k_clear(uint32_t* w) {
    if (blockIdx.x == 0 && threadIdx.x < 32) w[threadIdx.x] = 0;
}
// stamps live in ordinary hipMalloc memory: [0] = when the bump kernel was done, [1] = when the clear kernel was done
__global__ void k_bump(uint32_t* w, int plain, unsigned long long* stamps) {
    if (threadIdx.x == 0) {
        if (plain) w[0] = 123;
        else atomicAdd(w, 123u);
        __threadfence();
        stamps[0] = __builtin_amdgcn_s_memrealtime();
    }
}
// mode 0: volatile load (compiles to global_load sc1), 1: agent-scope acquire fence (buffer_inv sc1) then plain
// global_load, 2: relaxed agent-scope atomic load (sc1), 3: atomicAdd(w, 0), 4: s_load_dword, 5: plain global_load
__global__ void k_read(uint32_t* w, uint32_t* rec, int mode, unsigned long long* when) {
    if (threadIdx.x != 0) return;
    when[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    uint32_t v;
    if (mode == 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        v = w[0];
    } else if (mode == 2) {
        v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (mode == 3) {
        v = atomicAdd(w, 0u);
    } else if (mode == 4) {
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(w) : "memory");  // what `a.counters[0]` compiles to
    } else if (mode == 5) {
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(w) : "memory");  // plain, no fence
    } else {
        v = *(volatile uint32_t*)w;
    }
    rec[2 * blockIdx.x] = xcc_id();
    rec[2 * blockIdx.x + 1] = v;
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 1000;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int blocks = 2048;
    uint32_t *rec, *sink;
    unsigned long long *stamps, *when;
    CK(hipMalloc(&rec, blocks * 8));
    CK(hipMalloc(&sink, blocks * 4));
    CK(hipMalloc(&stamps, 64));
    CK(hipMalloc(&when, blocks * 8));
    std::vector<uint32_t> h(blocks * 2);
    std::vector<unsigned long long> hw(blocks), hs(8);
    {
        // can a library tell pool memory from hipMalloc memory?
        uint32_t *a = nullptr, *b = nullptr;
        CK(hipMalloc(&a, 1 << 16));
        CK(hipMallocAsync((void**)&b, 1 << 16, s));
        CK(hipStreamSynchronize(s));
        for (uint32_t* q : {a, b}) {
            hipPointerAttribute_t at;
            memset(&at, 0, sizeof at);
            hipError_t e = hipPointerGetAttributes(&at, q);
            hipMemPool_t pool = nullptr;
            hipError_t e2 = hipPointerGetAttribute(&pool, HIP_POINTER_ATTRIBUTE_MEMPOOL_HANDLE, q);
            printf("%s: hipPointerGetAttributes -> %s type %d isManaged %d allocationFlags %u; MEMPOOL_HANDLE -> %s %p\n",
                   q == a ? "hipMalloc     " : "hipMallocAsync", hipGetErrorName(e), (int)at.type, at.isManaged, at.allocationFlags,
                   hipGetErrorName(e2), (void*)pool);
        }
        (void)hipGetLastError();
        CK(hipFree(a));
        CK(hipFreeAsync(b, s));
    }
    // Episodes: allocate, run the sequence 3 times, free, drain the stream, idle ~2 ms (the first version of this tool
    // saw its stale reads in the first sequence after such a pause).
    const char* kinds[] = {"hipMalloc", "hipMallocAsync"};
    const char* modes[] = {"global_load sc1 (volatile)", "buffer_inv sc1 + global_load", "atomic load, agent scope (sc1)", "atomicAdd(w, 0)",
                           "s_load_dword", "global_load (plain)"};
    for (int kind = 0; kind < 2; ++kind)
        for (int plain_bump = 0; plain_bump < 2; ++plain_bump)
            for (int mode = 0; mode < 6; ++mode) {
                long stale = 0, bad = 0, first = 0, early = 0;
                for (int ep = 0; ep < trials; ++ep) {
                    uint32_t* w = nullptr;
                    if (kind == 0) CK(hipMalloc(&w, 1 << 16));
                    else CK(hipMallocAsync((void**)&w, 1 << 16, s));
                    for (int t = 0; t < 3; ++t) {
                        hipLaunchKernelGGL(k_plant, dim3(blocks), dim3(64), 0, s, w, sink);
                        hipLaunchKernelGGL(k_clear, dim3(1), dim3(64), 0, s, w);
                        hipLaunchKernelGGL(k_plant, dim3(blocks), dim3(64), 0, s, w, sink);
                        hipLaunchKernelGGL(k_bump, dim3(1), dim3(64), 0, s, w, plain_bump, stamps);
                        hipLaunchKernelGGL(k_read, dim3(blocks), dim3(64), 0, s, w, rec, mode, when);
                        CK(hipMemcpyAsync(h.data(), rec, blocks * 8, hipMemcpyDeviceToHost, s));
                        CK(hipMemcpyAsync(hw.data(), when, blocks * 8, hipMemcpyDeviceToHost, s));
                        CK(hipMemcpyAsync(hs.data(), stamps, 64, hipMemcpyDeviceToHost, s));
                        CK(hipStreamSynchronize(s));
                        long st = 0;
                        for (int b = 0; b < blocks; ++b) {
                            early += hw[b] < hs[0];
                            st += h[2 * b + 1] != 123;
                        }
                        bad += st != 0;
                        first += st != 0 && t == 0;
                        stale += st;
                    }
                    if (kind == 0) CK(hipFree(w));
                    else CK(hipFreeAsync(w, s));
                    CK(hipStreamSynchronize(s));
                    usleep(2000);
                }
                printf("%-15s bump=%-6s read=%-32s: %7ld stale reads in %3ld of %d sequences (%ld of them the first after the pause); "
                       "readers that started before the writer finished: %ld\n", kinds[kind], plain_bump ? "store" : "atomic", modes[mode], stale, bad,
                       3 * trials, first, early);
                fflush(stdout);
            }
    return 0;
}
