"""Developer tool: mean per launch of every counter in rocprofv3 counter_collection.csv files, per kernel symbol.
usage: pmc_counters.py <dir-or-csv> [...] [--match substring]"""
import sys, os, csv, glob, collections


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""
    if match in args:
        args.remove(match)
    files = []
    for a in args:
        files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(a) else [a]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        per_dispatch = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per_dispatch[(r["Kernel_Name"], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, c, _), v in per_dispatch.items():
            a = acc[(k, c)]
            a[0] += v
            a[1] += 1
    for (k, c), (s, n) in sorted(acc.items()):
        if match in k:
            print(f"{k[:90]:90s} {c:28s} {s / n:14.5g}  ({n} launches)")


if __name__ == "__main__":
    main()
