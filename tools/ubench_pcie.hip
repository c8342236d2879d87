// ubench_pcie.hip -- what the host boundary of this box can deliver: pageable vs pinned hipMemcpy in both
// directions, both directions at once, and multi-threaded CPU memcpy between pinned and pageable memory.
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/ubench_pcie tools/ubench_pcie.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <atomic>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_copy(char* dst, const char* src, size_t bytes, int T) {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { size_t a = bytes * t / T, b = bytes * (t + 1) / T; memcpy(dst + a, src + a, b - a); });
    for (auto& x : th) x.join();
}
int main() {
    const size_t N = 512u << 20;
    void *d0, *d1, *pin0, *pin1;
    CK(hipMalloc(&d0, N)); CK(hipMalloc(&d1, N));
    CK(hipHostMalloc(&pin0, N, hipHostMallocDefault)); CK(hipHostMalloc(&pin1, N, hipHostMallocDefault));
    char* pg = (char*)malloc(N); memset(pg, 1, N);
    memset(pin0, 2, N); memset(pin1, 3, N);
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    printf("hardware threads: %u\n", std::thread::hardware_concurrency());
    for (int r = 0; r < 2; ++r) {
        double t = now(); CK(hipMemcpy(d0, pg, N, hipMemcpyHostToDevice)); printf("pageable H2D  %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now(); CK(hipMemcpy(pg, d0, N, hipMemcpyDeviceToHost)); printf("pageable D2H  %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now(); CK(hipMemcpyAsync(d0, pin0, N, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); printf("pinned   H2D  %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now(); CK(hipMemcpyAsync(pin1, d1, N, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); printf("pinned   D2H  %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now(); CK(hipMemcpyAsync(d0, pin0, N, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(pin1, d1, N, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); printf("pinned   both %.1f GB/s (sum)\n", 2 * N / (now() - t) / 1e9);
    }
    {   // fresh pageable destination: page faults included
        double t = now(); char* fresh = (char*)malloc(N); CK(hipMemcpy(fresh, d0, N, hipMemcpyDeviceToHost)); printf("pageable D2H into fresh malloc %.1f GB/s\n", N / (now() - t) / 1e9); free(fresh);
    }
    {   // can the library make a caller's FRESH output buffer cheaper to fill?  (the reference returns fresh Vecs)
        FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
        char line[128] = "?";
        if (f) { if (!fgets(line, sizeof line, f)) line[0] = 0; fclose(f); }
        printf("transparent_hugepage/enabled: %s", line);
        for (int mode = 0; mode < 5; ++mode) {
            char* fresh = (char*)aligned_alloc(1 << 21, N);
            double t = now();
            const char* what = "";
            if (mode == 0) what = "plain (2 MiB-aligned allocation)";
            if (mode == 1 || mode == 3 || mode == 4) { int rc = madvise(fresh, N, MADV_HUGEPAGE); what = rc ? "MADV_HUGEPAGE failed" : "MADV_HUGEPAGE"; }
            if (mode == 2 || mode == 3 || mode == 4) {
                const int T = mode == 4 ? 16 : 8;
                std::vector<std::thread> th;
                std::atomic<int> bad{0};
                for (int i = 0; i < T; ++i)
                    th.emplace_back([&, i] {
                        const size_t per = ((N / T) >> 21) << 21;
                        const size_t lo = i * per, hi = i == T - 1 ? N : lo + per;
                        if (madvise(fresh + lo, hi - lo, 23 /* MADV_POPULATE_WRITE */)) bad++;
                    });
                for (auto& x : th) x.join();
                what = mode == 2 ? (bad ? "MADV_POPULATE_WRITE x8 FAILED" : "MADV_POPULATE_WRITE x8 threads")
                                 : mode == 3 ? (bad ? "HUGEPAGE + POPULATE_WRITE x8 FAILED" : "MADV_HUGEPAGE + MADV_POPULATE_WRITE x8 threads")
                                             : (bad ? "HUGEPAGE + POPULATE_WRITE x16 FAILED" : "MADV_HUGEPAGE + MADV_POPULATE_WRITE x16 threads");
            }
            const double prep = now() - t;
            CK(hipMemcpy(fresh, d0, N, hipMemcpyDeviceToHost));
            const double all = now() - t;
            printf("pageable D2H into fresh memory, %-48s: prepare %.1f ms, total %.1f ms = %.1f GB/s\n", what, prep * 1e3, all * 1e3, N / all / 1e9);
            free(fresh);
        }
    }
    for (int T : {1, 2, 4, 8, 16}) {
        double t = now(); par_copy((char*)pin0, pg, N, T); double a = now() - t;
        t = now(); par_copy(pg, (char*)pin1, N, T); double b = now() - t;
        char* fresh = (char*)malloc(N);
        t = now(); par_copy(fresh, (char*)pin1, N, T); double c = now() - t; free(fresh);
        printf("cpu memcpy %2d threads: pageable->pinned %.1f GB/s, pinned->pageable %.1f GB/s, pinned->fresh pageable %.1f GB/s\n", T, N / a / 1e9, N / b / 1e9, N / c / 1e9);
    }
    {   // hipHostRegister cost
        char* reg = (char*)malloc(N); memset(reg, 1, N);
        double t = now(); CK(hipHostRegister(reg, N, hipHostRegisterDefault)); double a = now() - t;
        t = now(); CK(hipMemcpyAsync(d0, reg, N, hipMemcpyHostToDevice, s0)); CK(hipStreamSynchronize(s0)); double b = now() - t;
        t = now(); CK(hipHostUnregister(reg)); double c = now() - t;
        printf("hipHostRegister 512 MiB: %.1f ms (%.1f GB/s), copy from it %.1f GB/s, unregister %.1f ms\n", a * 1e3, N / a / 1e9, N / b / 1e9, c * 1e3);
        free(reg);
    }
    return 0;
}
