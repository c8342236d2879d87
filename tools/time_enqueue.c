/* Host-side cost of enqueueing device calls (no Python in the way):
 *   gcc -O2 -Iinclude tools/time_enqueue.c -o tools/time_enqueue -Lmpc-protocols_amd -lhbmpc_hip -Wl,-rpath,'$ORIGIN/../mpc-protocols_amd' */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "hbmpc_hip.h"
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(void) {
    hbmpc_ctx* ctx; if (hbmpc_create(0, Bls12_381Fr, &ctx)) return 1;
    enum { N = 16, T = 5, D = 5, G = 64, REPS = 5000 };
    void* buf; if (hbmpc_dev_alloc(ctx, 1 << 22, &buf)) return 1;
    U256 *x = (U256*)buf, *y = x + G * (D + 1), *co = y + N * G; uint8_t* st = (uint8_t*)(co + G * (D + 1));
    hbmpc_recover_summary* sm = (hbmpc_recover_summary*)(st + 4096);
    U256 h[G * (D + 1)]; memset(h, 0, sizeof h); for (int i = 0; i < G * (D + 1); ++i) h[i].data[0] = 1000 + i;
    size_t ids[N]; for (int i = 0; i < N; ++i) ids[i] = i;
    void* s; hbmpc_stream_create(ctx, &s);
    hbmpc_memcpy_h2d(ctx, x, h, sizeof h, s);
    hbmpc_dev_vandermonde_apply(ctx, x, G, N, D, y, s);
    hbmpc_dev_batch_recover(ctx, ids, N, y, G, N, D, T, co, NULL, st, sm, s);
    hbmpc_stream_sync(ctx, s);
    double t0 = now();
    for (int i = 0; i < REPS; ++i) hbmpc_dev_vandermonde_apply(ctx, x, G, N, D, y, s);
    double t1 = now(); hbmpc_stream_sync(ctx, s); double t1s = now();
    for (int i = 0; i < REPS; ++i) hbmpc_dev_batch_recover(ctx, ids, N, y, G, N, D, T, co, NULL, st, sm, s);
    double t2 = now(); hbmpc_stream_sync(ctx, s); double t2s = now();
    for (int i = 0; i < REPS; ++i) hbmpc_dev_triple_finalize(ctx, x, y, G, co, s);
    double t3 = now(); hbmpc_stream_sync(ctx, s); double t3s = now();
    printf("vandermonde_apply (1 launch): enqueue %.2f us/call, with drain %.2f\n", (t1 - t0) / REPS * 1e6, (t1s - t0) / REPS * 1e6);
    printf("batch_recover (2 launches):   enqueue %.2f us/call, with drain %.2f\n", (t2 - t1s) / REPS * 1e6, (t2s - t1s) / REPS * 1e6);
    printf("triple_finalize (1 launch):   enqueue %.2f us/call, with drain %.2f\n", (t3 - t2s) / REPS * 1e6, (t3s - t2s) / REPS * 1e6);
    return 0;
}
