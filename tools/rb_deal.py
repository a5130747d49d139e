#!/usr/bin/env python3
"""Tile -> (update wave, slot) table of the 16-wave register-resident diagonal-block kernel (rb_rc15 in
egobox_amd/csrc/potf2_blocks.h).  135 tiles of 16x16 (lower triangle of a 256x256 block without tile (0,0)), 15 update
waves x 9 slots.  Constraints: the pair {(r, r-1), (r, r)} on one wave in consecutive slots; update waves 3, 7, 11 (they
share the chain wave's SIMD) own tiles of columns 0..4 only, at most two per column; everything else dealt column-major to
the least loaded wave, different waves within a column where possible.  Prints the per-strip maxima (TRSM tiles, trailing
update tiles, trailing-update tiles on the three partner waves) and the C++ initialiser."""
NU, NS, partners = 15, 9, [3, 7, 11]
nonp = [w for w in range(NU) if w not in partners]


def deal(off, step, pcols):
    lists = {w: [] for w in range(NU)}
    for r in range(1, 16):
        w = nonp[(off + step * (r - 1)) % len(nonp)]
        lists[w] += [(r, r - 1), (r, r)]
    need, taken = {p: NS for p in partners}, set()
    for c, cnt in pcols:
        rows = list(range(15, c + 1, -1))
        for j in range(cnt):
            p = partners[j % 3]
            if need[p] == 0:
                continue
            lists[p].append((rows[j], c))
            taken.add((rows[j], c))
            need[p] -= 1
    for (r, c) in [(r, c) for c in range(16) for r in range(c + 2, 16) if (r, c) not in taken]:
        cand = sorted(nonp, key=lambda w: (sum(1 for (rr, cc) in lists[w] if cc == c and rr > c), len(lists[w]), w))
        lists[[w for w in cand if len(lists[w]) < NS][0]].append((r, c))
    return lists


def score(lists):
    tot, rows = 0, []
    for k in range(15):
        b = [sum(1 for (r, c) in lists[w] if c >= k + 1 and not (r == k + 1 and c == k + 1)) for w in range(NU)]
        a = [sum(1 for (r, c) in lists[w] if c == k and r > k) + (1 if (k + 1, k) in lists[w] else 0) for w in range(NU)]
        rows.append((k, max(a), max(b), max(b[p] for p in partners)))
        tot += max(b) + max(a)
    return tot, rows


best = None
for off in range(12):
    for step in range(1, 12):
        l = deal(off, step, [(0, 6), (1, 6), (2, 6), (3, 6), (4, 3)])
        if any(len(v) != NS for v in l.values()):
            continue
        t, rows = score(l)
        if best is None or t < best[0]:
            best = (t, off, step, l, rows)
print("score", best[0], "pair offset / stride", best[1], best[2])
for r in best[4]:
    print("strip %2d: max TRSM tiles %d, max update tiles %d, on the partner waves %d" % r)
print("{" + ",\n ".join("{" + ", ".join("0x%02x" % (r * 16 + c) for (r, c) in best[3][w]) + "}" for w in range(NU)) + "}")
