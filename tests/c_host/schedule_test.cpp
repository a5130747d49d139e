// Host check of the schedule table (egobox_amd/csrc/schedule.h): the documented rows, the knobs' meaning, and that nothing
// but (padded size, lock-step width, workspaces) enters.  Built and run by tests/test_tile_tables_cpu.py (g++, no GPU).
#include <cstdio>
#include <initializer_list>

#include "../../egobox_amd/csrc/schedule.h"

static int fails = 0;
static void expect(int n_pad, int lockstep, int nws, const egx::ScheduleKnobs &k, int left, int w_left, int pipe, int whole, int gp) {
    const egx::PotrfSchedule s = egx::schedule_table(n_pad, lockstep, nws, k);
    if (s.left != left || s.w_left != w_left || s.pipe != pipe || s.whole != whole || s.group_panels != gp) {
        std::printf("n_pad %d lockstep %d workspaces %d: got {%d %d %d %d %d}, table says {%d %d %d %d %d}\n", n_pad, lockstep, nws,
                    s.left, s.w_left, s.pipe, s.whole, s.group_panels, left, w_left, pipe, whole, gp);
        fails++;
    }
}

int main() {
    const egx::ScheduleKnobs d;  // the defaults
    // the table's rows
    expect(4096, 1, 1, d, 0, 0, 1, 1, 2);     // a lone fit at config 2's size: one launch
    expect(4096, 2, 2, d, 0, 0, 1, 1, 2);     // 2 x 16 panels = 32 diagonal blocks: still one launch
    expect(4096, 12, 12, d, 0, 0, 1, 0, 2);   // a tuned fit's twelve starts: chain launches per group
    expect(2176, 6, 6, d, 0, 0, 1, 0, 2);     // 6 x 9 > 32
    expect(1024, 8, 8, d, 0, 0, 1, 1, 2);     // 8 x 4 = 32
    expect(4352, 1, 1, d, 0, 0, 1, 1, 2);        // a lone matrix up to 7168 columns: still one launch
    expect(7168, 1, 1, d, 0, 0, 1, 1, 2);
    expect(4352, 2, 2, d, 0, 0, 0, 0, 2);        // beyond 4096 columns with more diagonal blocks than 32: separate launches
    expect(7424, 1, 1, d, 0, 0, 0, 0, 2);
    expect(8192, 1, 1, d, 0, 0, 0, 0, 2);
    expect(8192, 2, 2, d, 0, 0, 0, 0, 2);
    expect(8192, 12, 12, d, 0, 0, 0, 0, 2);
    expect(14336, 1, 1, d, 0, 0, 0, 0, 4);
    expect(16384, 1, 1, d, 0, 0, 0, 0, 4);       // a lone fit at the metric's size
    expect(16384, 4, 4, d, 0, 1, 0, 0, 4);
    expect(16384, 7, 16, d, 0, 1, 0, 0, 4);
    expect(16384, 8, 16, d, 1, 1, 0, 0, 4);
    expect(16384, 8, 8, d, 1, 1, 0, 0, 4);
    expect(14400, 8, 8, d, 1, 0, 0, 0, 4);       // (the rider's left-looking form needs 256-column panels throughout)
    // round 6: a lone one-workspace handle between 5376 and 14080 columns factors as ONE flow launch (the other bits stay what
    // such a handle falls back to); never with several workspaces, a lock-step width, odd panels, or EGX_PIPE != 1
    auto flow = [&](int n_pad, int w, int nws, const egx::ScheduleKnobs &kk) { return egx::schedule_table(n_pad, w, nws, kk).flow; };
    for (int n_pad : {5376, 6144, 7168, 8192, 12288, 14080})
        if (!flow(n_pad, 1, 1, d)) std::printf("n_pad %d: a lone handle should take the flow launch\n", n_pad), fails++;
    for (int n_pad : {1024, 4096, 5120, 14336, 16384})
        if (flow(n_pad, 1, 1, d)) std::printf("n_pad %d: no flow launch at this size\n", n_pad), fails++;
    if (flow(8192, 1, 2, d) || flow(8192, 2, 2, d) || flow(8320, 1, 1, d)) std::printf("flow needs one workspace, width 1, 256-column panels\n"), fails++;
    // ... and from 14336 columns on its LAST 6144 .. 7167 columns, behind a switch on a four-panel group boundary
    for (int n_pad = 14336; n_pad <= 24576; n_pad += 256) {
        const egx::PotrfSchedule a = egx::schedule_table(n_pad, 1, 1, d);
        if (a.flow || a.flow_tail < 6144 || a.flow_tail >= 7168 || (n_pad - a.flow_tail) % 1024 || a.flow_tail % 256)
            std::printf("n_pad %d: flow tail %d\n", n_pad, a.flow_tail), fails++;
        if (egx::schedule_table(n_pad, 1, 2, d).flow_tail || egx::schedule_table(n_pad, 8, 16, d).flow_tail) fails++;
    }
    if (egx::schedule_table(8192, 1, 1, d).flow_tail || egx::schedule_table(16384 + 128, 1, 1, d).flow_tail) fails++;
    // the width enters through the lock-step thresholds only, the workspaces through the whole-launch bound only
    for (int n_pad = 128; n_pad <= 20480; n_pad += 128)
        for (int w = 1; w <= 16; w++)
            for (int nws = w; nws <= 24; nws += 5) {
                const egx::PotrfSchedule a = egx::schedule_table(n_pad, w, nws, d), b = egx::schedule_table(n_pad, w, nws + 100, d);
                if (a.left != b.left || a.w_left != b.w_left || a.group_panels != b.group_panels || (n_pad <= 4096 && a.pipe != b.pipe)) fails++;
                const egx::PotrfSchedule c = egx::schedule_table(n_pad, w + 100, nws, d);
                if (a.whole != c.whole || a.group_panels != c.group_panels) fails++;
                if (a.whole && !(n_pad <= 7168 && nws * ((n_pad + 255) / 256) <= 32)) fails++;
                if (a.pipe != (n_pad <= 4096 || a.whole)) fails++;
                if (a.left && !a.w_left && n_pad % 256 == 0) fails++;  // left-looking handles' riders are left-looking too
            }
    // the knobs
    egx::ScheduleKnobs k = d;
    k.pipe = 0;
    expect(4096, 1, 1, k, 0, 0, 0, 0, 2);
    if (flow(8192, 1, 1, k)) std::printf("EGX_PIPE=0 must switch the flow launch off\n"), fails++;
    k = d, k.pipe = 2;                           // chain launches per group of panels only
    if (flow(8192, 1, 1, k)) std::printf("EGX_PIPE=2 must switch the flow launch off\n"), fails++;
    expect(4096, 1, 1, k, 0, 0, 1, 0, 2);
    expect(8192, 1, 1, k, 0, 0, 0, 0, 2);
    k = d, k.potrf_left = 0;
    expect(16384, 8, 16, k, 0, 0, 0, 0, 4);
    k = d, k.potrf_left = 2;
    expect(2048, 1, 1, k, 1, 1, 1, 1, 2);
    k = d, k.potrf_group = 3;
    expect(16384, 8, 16, k, 1, 1, 0, 0, 3);
    if (fails) {
        std::printf("%d schedule checks failed\n", fails);
        return 1;
    }
    std::printf("schedule table ok\n");
    return 0;
}
