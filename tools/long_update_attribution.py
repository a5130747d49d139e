#!/usr/bin/env python3
"""Where the left-looking LONG update loses against the FP64 MFMA peak IN SITU -- under the product schedule of bench.py's timed
region (n = 16384, d = 32, lock-step groups of eight, two groups in flight), not alone.

Needs the profiling build (tools/dev_build.sh trace -> egobox_amd/lib/_dev/libegx_gp_hip_trace.so): every tile of the tagged
k_gemm_stream launches (TAG 1 = long update, TAG 2 = short update) records the compute unit it ran on, the 100-MHz wall clock at
its start / around its K loop / after its stores, and the shader-clock counter (s_memtime) around its K loop.  From these:

    achieved / peak  =  useful-flop share  x  tile-slot occupancy  x  K-loop share of a tile's residency
                        x  MFMA issue inside the K loop  x  shader clock / 2.4 GHz

    python tools/long_update_attribution.py [candidates = 32] > profiles/r06_long_update_in_situ_attribution.txt"""
import ctypes as C
import os
import sys
import time

os.environ["EGX_TEST_LIBRARY"] = "trace"
import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402
from egobox_amd import _lib as L  # noqa: E402

lib = L.load()
lib.egx_dev_stream_trace.restype = C.c_longlong
lib.egx_dev_stream_trace.argtypes = [C.c_int, C.POINTER(C.c_longlong), C.c_longlong]
W = 10
PEAK, F_NOM = 78.6, 2.4  # TFLOP/s at 2.4 GHz: 256 CUs x 4 SIMDs x 2048 flop per 64-cycle v_mfma_f64_16x16x4_f64
ncand = int(sys.argv[1]) if len(sys.argv) > 1 else 32
in_flight = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n, d = 16384, 32
x, y = egx.workload.make_training_set(n, d, 42)
base = egx.workload.default_theta(d)
rng = np.random.default_rng(3)
h = egx.GpHandle(x, y, corr=0, n_workspaces=in_flight)
h.set_lockstep(8)
print(f"# {sys.argv[0]}: n = {n}, d = {d}, {in_flight} workspaces, lock-step {h.set_lockstep(8)}, schedule {h.schedule()}")
cands = base * 10.0 ** rng.uniform(-0.15, 0.15, size=(max(ncand, in_flight), d))
h.likelihood_batch(cands[:in_flight])  # warm-up (allocations, first launches)
t0 = time.perf_counter()
h.likelihood_batch(cands[:ncand])
t_plain = time.perf_counter() - t0
cap = 400000
assert lib.egx_dev_stream_trace(1, None, cap) == 0
t0 = time.perf_counter()
lk, st = h.likelihood_batch(cands[:ncand])
t_traced = time.perf_counter() - t0
buf = np.zeros(cap * W, dtype=np.int64)
got = lib.egx_dev_stream_trace(0, buf.ctypes.data_as(C.POINTER(C.c_longlong)), cap)
h.close()
assert np.all(st == 0), st
r = buf[: got * W].reshape(got, W)
r = r[r[:, 8] != 0]  # (complete records only)
print(f"# {ncand} candidates: {ncand / t_plain:.2f} fits/s untraced, {ncand / t_traced:.2f} fits/s traced; {got} tile records "
      f"({len(r)} complete), times from the 100-MHz counter (s_memrealtime), shader cycles from s_memtime")
key, word, hw = r[:, 0], r[:, 1], r[:, 2]
tag = (word >> 60) & 0xF
z = (word >> 52) & 0xFF
nch = (word >> 32) & 0xFFFFF
tile = word & 0xFFFFFFFF
bx, by = r[:, 9] >> 32, r[:, 9] & 0xFFFFFFFF
cu = ((hw >> 32) & 0xF) * 256 + ((hw >> 8) & 0xFF)  # XCC_ID, then SE | SH | CU of HW_ID
w_begin, w_l0, c_l0, w_l1, c_l1, w_end = (r[:, i].astype(np.float64) for i in (3, 4, 5, 6, 7, 8))
T0 = w_begin.min()
us = lambda ticks: ticks * 0.01  # noqa: E731
res_us, loop_us = us(w_end - w_begin), us(w_l1 - w_l0)
ghz = (c_l1 - c_l0) / (loop_us * 1e3)
busy = nch * 8192.0 / (c_l1 - c_l0)  # per SIMD and 16-deep chunk: 2 waves x 64 MFMAs x 64 cycles
print(f"# compute units seen: {len(np.unique(cu))}")


def tile_flops(k):  # executed by one 128 x 256 tile
    return 2.0 * 128 * 256 * k


for tg, name in ((1, "LONG update (K = every column before the previous group; bench.py's `roofline` launch)"),
                 (2, "SHORT update (K = the previous group, 1024)"),
                 (4, "IN-GROUP updates that take the stream kernel (rectangles below the next diagonal block, K = 256)")):
    m = tag == tg
    if not m.any():
        continue
    print(f"\n== {name}: {m.sum()} tiles")
    # a launch = the tiles of one (first matrix' C, K); a slot's next group reuses both, so split where no tile of the key started
    # for 20 ms (the longest tile runs 4 ms)
    by_key = {}
    for i in np.nonzero(m)[0]:
        by_key.setdefault((int(key[i]), int(nch[i])), []).append(i)
    launches = []
    for (kk, nc), idx in by_key.items():
        idx = np.array(idx)
        idx = idx[np.argsort(w_begin[idx])]
        cuts = np.nonzero(np.diff(w_begin[idx]) > 2.0e6)[0] + 1
        for part in np.split(idx, cuts):
            launches.append((nc, part))
    rows = []
    for nc, idx in sorted(launches, key=lambda kv: w_begin[kv[1]].min()):
        K = nc * 16
        span = us(w_end[idx].max() - w_begin[idx].min())
        nmat = len(np.unique(z[idx]))
        # useful flops of the launch: the lower triangle's share of the tiles that cross the diagonal (rows bx*128.., cols by*256..)
        gw = (by[idx].max() + 1) * 256
        rows_m = (bx[idx].max() + 1) * 128
        useful = nmat * 2.0 * K * (rows_m * gw - (0.5 * gw * (gw - 1.0) if tg != 4 else 0.0))
        executed = len(idx) * tile_flops(K)
        occ = res_us[idx].sum() / (256.0 * span)
        loop_share = loop_us[idx].sum() / res_us[idx].sum()
        b = np.average(busy[idx], weights=loop_us[idx])
        f = np.average(ghz[idx], weights=loop_us[idx])
        ach = useful / span / 1e6
        rows.append((us(w_begin[idx].min() - T0), K, nmat, len(idx), span, ach, useful / executed, occ, loop_share, b, f,
                     np.median(res_us[idx]), executed, useful, res_us[idx].sum(), loop_us[idx].sum()))
    print("   start us      K  mats  tiles   span us  TFLOP/s  of peak = useful x occupancy x loop share x MFMA issue x clock/2.4   (clock GHz, median tile us)")
    for (ts, K, nmat, nt, span, ach, u, occ, ls, b, f, med, *_rest) in (rows if tg != 4 else rows[:12]):  # (the in-group launches: the first few)
        print(f"  {ts:9.0f}  {K:5d}  {nmat:4d}  {nt:5d}  {span:8.0f}  {ach:7.2f}  {ach / PEAK:.3f} = {u:.3f} x {occ:.3f} x {ls:.3f} x {b:.3f} x {f / F_NOM:.3f}"
              f"   ({f:.3f}, {med:.0f})   product {u * occ * ls * b * f / F_NOM:.3f}")
    A = np.array([[q[4], q[12], q[13], q[14], q[15]] for q in rows])
    idx = np.nonzero(m)[0]
    b_all, f_all = np.average(busy[idx], weights=loop_us[idx]), np.average(ghz[idx], weights=loop_us[idx])
    u_all, occ_all, ls_all = A[:, 2].sum() / A[:, 1].sum(), A[:, 3].sum() / (256.0 * A[:, 0].sum()), A[:, 4].sum() / A[:, 3].sum()
    ach_all = A[:, 2].sum() / A[:, 0].sum() / 1e6
    print(f"  all {len(rows)} launches (flop-weighted; spans overlap when two groups are in flight): {ach_all:.2f} TFLOP/s over the SUM of the spans = "
          f"{ach_all / PEAK:.3f} = useful {u_all:.3f} x occupancy {occ_all:.3f} x loop share {ls_all:.3f} x MFMA issue {b_all:.3f} x clock "
          f"{f_all:.3f} GHz / 2.4 = {f_all / F_NOM:.3f}   (product {u_all * occ_all * ls_all * b_all * f_all / F_NOM:.3f})")
    print(f"  shader clock inside the K loops: min {ghz[idx].min():.3f}, 5 % {np.percentile(ghz[idx], 5):.3f}, median "
          f"{np.median(ghz[idx]):.3f}, 95 % {np.percentile(ghz[idx], 95):.3f}, max {ghz[idx].max():.3f} GHz")
    print(f"  MFMA issue inside the K loops: 5 % {np.percentile(busy[idx], 5):.3f}, median {np.median(busy[idx]):.3f}, 95 % "
          f"{np.percentile(busy[idx], 95):.3f}; prologue (C tile in, first chunks) median {np.median(us(w_l0[idx] - w_begin[idx])):.1f} us, "
          f"epilogue (stores drained) median {np.median(us(w_end[idx] - w_l1[idx])):.1f} us")

# ---- the compute units' time over the whole traced batch
win0, win1 = w_begin.min(), w_end.max()
window = us(win1 - win0)
tot = {1: res_us[tag == 1].sum(), 2: res_us[tag == 2].sum(), 4: res_us[(tag != 1) & (tag != 2)].sum()}
print(f"\n== compute-unit time over the traced batch ({window / 1e3:.1f} ms from the first to the last traced tile, {len(np.unique(cu))} CUs)")
print(f"  in LONG-update tiles {tot[1] / (256 * window):.3f}, in SHORT-update tiles {tot[2] / (256 * window):.3f}, in the in-group updates' "
      f"stream tiles {tot[4] / (256 * window):.3f}, in none of them {1 - (tot[1] + tot[2] + tot[4]) / (256 * window):.3f} (diagonal blocks, panel "
      f"solves, the in-group updates on the 64 x 64 kernel, the correlation builds, and idle)")
gaps = []
for c in np.unique(cu):
    i = np.nonzero(cu == c)[0]
    o = i[np.argsort(w_begin[i])]
    g = us(w_begin[o][1:] - w_end[o][:-1])
    gaps.append(g[g > 0])
gaps = np.concatenate(gaps)
print(f"  between two traced tiles on the same CU: {len(gaps)} gaps, median {np.median(gaps):.1f} us, mean {gaps.mean():.1f} us, 90 % "
      f"{np.percentile(gaps, 90):.0f} us, 99 % {np.percentile(gaps, 99):.0f} us; gaps under 20 us (a workgroup handing the CU to the next) "
      f"{(gaps < 20).mean():.2f} of them, holding {gaps[gaps < 20].sum() / gaps.sum():.3f} of the gap time")
# the flops of the whole batch against the window
fl = ncand * n ** 3 / 3.0
print(f"  whole batch: {ncand} x n^3/3 = {fl:.3e} flop in {t_traced * 1e3:.0f} ms (host clock) = {fl / t_traced / 1e12:.2f} TFLOP/s = "
      f"{fl / t_traced / 1e12 / PEAK:.3f} of peak")
