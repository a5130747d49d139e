// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on this stack (gfx950)?  Kernel A spins ~100 us and then
// raises a flag; kernel B, launched behind it in the same stream with the flag, reports whether it started before A was done.
//   hipcc --offload-arch=gfx950 -O2 tools/lab/anyorder_probe.hip -o /tmp/anyorder_probe && /tmp/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void k_a(int *flag, long long spin_ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_b(int *flag, int *out) {
    out[0] = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);  // 0: started while A was still spinning
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0 && spins < 4000000) {
        __builtin_amdgcn_s_sleep(8);
        spins++;
    }
    out[1] = spins;
}

int main() {
    int *flag, *out, h[2];
    hipMalloc(&flag, 4);
    hipMalloc(&out, 8);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const long long ticks = 100 * 100;  // wall_clock64 runs at 100 MHz: 100 us
    for (int mode = 0; mode < 2; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            hipMemsetAsync(flag, 0, 4, s);
            hipMemsetAsync(out, 0xff, 8, s);
            hipStreamSynchronize(s);
            hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, s, flag, ticks);
            if (mode == 0) hipLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s, flag, out);
            else hipExtLaunchKernelGGL(k_b, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, flag, out);
            hipError_t e = hipStreamSynchronize(s);
            hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
            std::printf("%s launch: B saw flag %d at its start, polled %d times (%s)\n", mode ? "any-order" : "in-order ", h[0], h[1],
                        hipGetErrorString(e));
        }
    }
    return 0;
}
