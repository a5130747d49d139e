// Does hipStreamCreate scale over host threads?  48 streams (a 12-workspace handle) created by 1, 4, 8, 16 threads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    (void)hipFree(0);
    hipStream_t w;
    (void)hipStreamCreateWithFlags(&w, hipStreamNonBlocking);
    int lo, hi;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int nt : {1, 4, 8, 16, 1}) {
        std::vector<hipStream_t> s(48);
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                (void)hipSetDevice(0);
                for (int i = t; i < 48; i += nt) {
                    if (i & 2) (void)hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, hi);
                    else (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
                }
            });
        for (auto &x : th) x.join();
        const double t1 = now();
        for (auto x : s) (void)hipStreamDestroy(x);
        const double t2 = now();
        printf("%2d threads: 48 streams created in %.2f ms, destroyed (one thread) in %.2f ms\n", nt, t1 - t0, t2 - t1);
    }
    return 0;
}
