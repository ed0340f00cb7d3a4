import sys, os, csv, glob
# usage: gap_probe.py <rocprof dir>: per stream, time covered by kernels vs idle gaps inside the busy span of the LAST forward
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# take the last 40% of events (steady state)
n = len(ev)
ev = ev[int(n * 0.6):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
# union of busy intervals
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]
gaps = []
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = t1 - t0
print(f"kernels {len(ev)}  span {span/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms  idle {100*(span-busy)/span:.1f} %  gaps: n={len(gaps)} mean {sum(gaps)/max(1,len(gaps))/1e3:.2f} us  sum {sum(gaps)/1e6:.2f} ms")
gaps.sort()
print("gap percentiles us:", [round(gaps[int(len(gaps)*q)]/1e3, 2) for q in (0.1, 0.5, 0.9, 0.99)])
