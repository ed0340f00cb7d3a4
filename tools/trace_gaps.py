"""Developer tool: from a rocprofv3 --kernel-trace CSV (one bench.py run) print, for the LAST forward in the trace, the GPU's busy
time (union of kernel intervals over all streams), the idle gaps between kernels by size class, and the longest gaps with the kernels
around them.  usage: python tools/trace_gaps.py <kernel_trace.csv> [n_forwards_in_trace]"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the last forward = from the last stem kernel of the x frames backwards: use the final `conv_last_taps` as the end marker
ends = [i for i, e in enumerate(ev) if "conv_last_taps" in e[2]]
if len(ends) < 2:
    sys.exit("need at least two forwards in the trace")
lo, hi = ends[-2] + 1, ends[-1]
seg = ev[lo:hi + 1]
t0, t1 = seg[0][0], max(e[1] for e in seg)
busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
gaps = []
last_name = seg[0][2]
for s, e, n in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    last_name = n
busy += cur_e - cur_s
wall = t1 - t0
print(f"forward: {len(seg)} kernels, wall {wall / 1e6:.3f} ms, GPU busy (union) {busy / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
for lim in (2e3, 5e3, 20e3, 100e3, 1e9):
    sel = [g for g in gaps if g[0] < lim]
    print(f"  gaps < {lim / 1e3:.0f} us: {len(sel)} totalling {sum(g[0] for g in sel) / 1e6:.3f} ms")
for g in sorted(gaps, reverse=True)[:15]:
    print(f"  {g[0] / 1e3:8.1f} us  after {g[1][:60]}  before {g[2][:60]}")
