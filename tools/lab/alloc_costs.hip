// What the pieces of a workspace cost to create (HIP runtime, MI355X): streams, events, pinned and device allocations.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    (void)hipFree(0);
    void *warm; hipMalloc(&warm, 1 << 20); hipFree(warm);
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        hipStream_t s[4];
        hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking);
        hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking);
        double t1 = now();
        int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipStreamCreateWithPriority(&s[2], hipStreamNonBlocking, hi);
        hipStreamCreateWithPriority(&s[3], hipStreamNonBlocking, hi);
        double t2 = now();
        std::vector<hipEvent_t> ev(400);
        for (auto &e : ev) hipEventCreate(&e);
        double t3 = now();
        std::vector<hipEvent_t> ev2(12);
        for (auto &e : ev2) hipEventCreateWithFlags(&e, hipEventDisableTiming);
        double t4 = now();
        void *h[5]; size_t hs[5] = {64, 2 * 8192 * 8, 8192 * 8, 8192 * 8, 64};
        for (int i = 0; i < 5; i++) hipHostMalloc(&h[i], hs[i], hipHostMallocDefault);
        double t5 = now();
        void *d[5]; size_t ds[5] = {64, 8 * 8192 * 8, 8192 * 8, 8192 * 8, 8192 * 8};
        for (int i = 0; i < 5; i++) hipMalloc(&d[i], ds[i]);
        double t6 = now();
        void *big; hipMalloc(&big, (size_t)8192 * 8320 * 8);
        double t7 = now();
        // first use of the streams
        for (int i = 0; i < 4; i++) hipMemsetAsync(d[1], 0, 64, s[i]);
        for (int i = 0; i < 4; i++) hipStreamSynchronize(s[i]);
        double t8 = now();
        printf("rep %d: 2 streams %.2f ms | 2 priority streams %.2f | 400 events %.2f | 12 no-timing events %.2f | 5 pinned allocs %.2f | 5 small device allocs %.2f | 545 MB device alloc %.2f | first use of the 4 streams %.2f\n",
               rep, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7);
        double u0 = now();
        hipFree(big);
        double u1 = now();
        for (int i = 0; i < 5; i++) hipFree(d[i]);
        for (int i = 0; i < 5; i++) hipHostFree(h[i]);
        double u2 = now();
        for (auto &e : ev) hipEventDestroy(e);
        for (auto &e : ev2) hipEventDestroy(e);
        for (int i = 0; i < 4; i++) hipStreamDestroy(s[i]);
        double u3 = now();
        printf("       free 545 MB %.2f ms | small + pinned frees %.2f | events + streams destroyed %.2f\n", u1 - u0, u2 - u1, u3 - u2);
    }
    return 0;
}
