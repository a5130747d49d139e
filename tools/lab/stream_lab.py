#!/usr/bin/env python3
"""Where do the stream kernel's K loops lose their 7 %?  Builds k_gemm_stream in variants that are WRONG on purpose (a textual
edit of a temporary copy of egobox_amd/csrc/kernels_chol.hip, never of the product) -- no workgroup barrier in the loop, no wait
for the LDS-DMA pieces, no LDS reads in the loop, no LDS-DMA issues -- with the per-tile stamps of the trace build, runs a
launch of the long update's shape (n = 16384, rows from 9216, 1024 columns, K = 8192, four matrices) on random data and prints
the MFMA issue rate inside the K loop (nch x 8192 cycles / measured cycles) and the shader clock per variant.
    python tools/lab/stream_lab.py            (on the GPU box; ~1 min of hipcc per variant)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = open(os.path.join(ROOT, "egobox_amd", "csrc", "kernels_chol.hip")).read()
BAR = '''                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of chunk g + 1 (the only ones in flight)
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // (the MFMAs stay outside the branches'''
assert SRC.count(BAR) == 1


def variant(name):
    s = SRC
    if name in ("nobar", "nobar_nowait"):
        s = s.replace(BAR, BAR.replace("                __builtin_amdgcn_s_barrier();\n", ""))
    if name in ("nowait", "nobar_nowait"):
        s = s.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of chunk g + 1 (the only ones in flight)\n                __builtin_amdgcn_s_barrier();\n            }\n            __builtin_amdgcn_sched_barrier(0);\n            {   // (the MFMAs stay',
                      '__builtin_amdgcn_s_barrier();\n            }\n            __builtin_amdgcn_sched_barrier(0);\n            {   // (the MFMAs stay', 1) if name == "nowait" else \
            s.replace('                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own pieces of chunk g + 1 (the only ones in flight)\n            }\n            __builtin_amdgcn_sched_barrier(0);\n            {   // (the MFMAs stay',
                      '            }\n            __builtin_amdgcn_sched_barrier(0);\n            {   // (the MFMAs stay', 1)
    if name == "noldsread":
        a = "            read_half(stage, 1, a1, b1);\n            mma_half(a0, b0);"
        assert s.count(a) == 1
        s = s.replace(a, "            for (int i_ = 0; i_ < 4; i_++) a1[i_] = a0[i_], b1[i_] = b0[i_];\n            asm volatile(\"\" : \"+v\"(a1[0]), \"+v\"(b1[0]));\n            mma_half(a0, b0);")
        b = "            if (next) read_half(st1, 0, a0, b0);"
        assert s.count(b) == 1
        s = s.replace(b, "")
    if name == "asmloads":  # the LDS-DMA pieces as inline asm in the SADDR form: SGPR base + 32-bit VGPR byte offset, M0 set by hand
        a = """                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(ga + offA[i]), (lds_ptr_t)(dst + i * 1024), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(gb + offB[i - 2]), (lds_ptr_t)(dst + 2048 + (i - 2) * 1024), 16,
                                                 0, 0);"""
        assert s.count(a) == 1
        s = s.replace(a, r"""                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"((unsigned)(size_t)(lds_ptr_t)(dst + i * 1024)), "v"(offA[i]), "s"(ga) : "memory");
            else
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"((unsigned)(size_t)(lds_ptr_t)(dst + 2048 + (i - 2) * 1024)), "v"(offB[i - 2]), "s"(gb) : "memory");""")
        for ab, ldx in (("A", "lda"), ("B", "ldb")):
            b = f"off{ab}[i] = (unsigned)((8 * (wave + 8 * i) + lrow) * (int){ldx}) + lpart;"
            assert s.count(b) == 1, b
            s = s.replace(b, f"off{ab}[i] = 8u * ((unsigned)((8 * (wave + 8 * i) + lrow) * (int){ldx}) + lpart);")
    if name == "noloads":
        c = "if (ld) issue_pieces(ist, q, q + 1);"
        assert s.count(c) == 1
        s = s.replace(c, "")
    return s


MAIN = r'''
#include "KC_VARIANT"
#include "../egobox_amd/csrc/kernels_pipe.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
namespace egx {
void set_error(const std::string &m) { fprintf(stderr, "error: %s\n", m.c_str()); }
hipError_t dev_malloc_bytes(void **p, size_t bytes) { return hipMalloc(p, bytes); }
}
using namespace egx;
extern "C" long long egx_dev_stream_trace(int mode, long long *out, long long cap);
__global__ void k_fill(double *p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned long long x = i * 6364136223846793005ull + 1442695040888963407ull; x ^= x >> 29; p[i] = 1e-3 * ((double)(x & 0xffffff) / 16777216.0 - 0.5); }
}
int main() {
    const int n = 16384, g0 = 9216, K = 8192, N = 1024, nz = 4;
    const int64_t ld = n + 128, sM = (int64_t)(n + 128) * ld;
    double *M; if (hipMalloc(&M, sizeof(double) * sM * nz) != hipSuccess) return 1;
    k_fill<<<4096, 256>>>(M, (size_t)sM * nz);
    hipDeviceSynchronize();
    GemmBatch gb; gb.count = nz; gb.sC = gb.sA = gb.sB = sM; gb.sInfo = 0;
    for (int rep = 0; rep < 3; rep++) {
        if (rep == 2) egx_dev_stream_trace(1, nullptr, 20000);
        if (launch_gemm_nt_sub(0, M + (int64_t)g0 * ld + g0, ld, M + (int64_t)g0 * ld, ld, M + (int64_t)g0 * ld, ld, n + 128 - g0, N, K, 1, 0, nullptr, nullptr, &gb, 1)) return 2;
        hipDeviceSynchronize();
    }
    std::vector<long long> r(20000 * 10);
    const long long got = egx_dev_stream_trace(0, r.data(), 20000);
    std::vector<double> busy, ghz, us;
    for (long long i = 0; i < got; i++) {
        const long long *q = &r[i * 10];
        if (!q[8]) continue;
        const double nch = (double)((q[1] >> 32) & 0xfffff), cyc = (double)(q[7] - q[5]), wall = (double)(q[6] - q[4]) * 10.0;  // ns
        busy.push_back(nch * 8192.0 / cyc), ghz.push_back(cyc / wall), us.push_back((double)(q[8] - q[3]) * 0.01);
    }
    std::vector<double> crow((size_t)1024);
    double chk = 0.0;
    for (int rr = 0; rr < 3; rr++) {  // three rows of the updated region of the last matrix: a checksum against the base variant
        hipMemcpy(crow.data(), M + (nz - 1) * sM + (int64_t)(g0 + 1500 + 1777 * rr) * ld + g0, sizeof(double) * 1024, hipMemcpyDeviceToHost);
        for (double v : crow) chk += v;
    }
    std::sort(busy.begin(), busy.end()), std::sort(ghz.begin(), ghz.end()), std::sort(us.begin(), us.end());
    const size_t m = busy.size() / 2;
    printf("VARIANT_NAME: %zu tiles, MFMA issue in the K loop median %.3f (5 %% %.3f, 95 %% %.3f), clock %.3f GHz, tile %.0f us, checksum %.17g\n", busy.size(), busy[m],
           busy[busy.size() / 20], busy[busy.size() * 19 / 20], ghz[m], us[m], chk);
    return 0;
}
'''

out = []
for name in sys.argv[1:] or ["base", "nobar", "nowait", "nobar_nowait", "noldsread", "noloads"]:
    kc = f"/tmp/kc_{name}.hip"
    open(kc, "w").write(variant(name))
    mainf = f"/tmp/stream_lab_{name}.hip"
    open(mainf, "w").write(MAIN.replace("KC_VARIANT", kc).replace("../egobox_amd/csrc/kernels_pipe.hip", os.path.join(ROOT, "egobox_amd", "csrc", "kernels_pipe.hip")).replace("VARIANT_NAME", name))
    exe = f"/tmp/stream_lab_{name}"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-DEGX_STREAM_TRACE", "-DEGX_TEST_HOOKS", "-I", os.path.join(ROOT, "egobox_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), "-Wno-unused-result", "-Wno-unused-value", mainf, "-o", exe], check=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print((p.stdout + p.stderr).strip(), flush=True)
