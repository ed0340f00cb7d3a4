"""Developer tool: summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) per kernel symbol.
    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.txt> [--json <out.json> <kernel-substring> <family> <precision>]
                                                                  [--families <out.json> <precision> <commit>]
--families writes profiles/traffic_rNN.json in the form bench.py replays: HBM bytes per launch for every kernel family of its
event profiler (bench.MFMA16 / bench.HBM_FAMILIES map a family to kernel-symbol substrings), with the commit of the run.
FETCH_SIZE reads 1/2 of the bytes of a 16-B/lane coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM): HBM bytes = 2 * FETCH + WRITE."""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(d, name):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    F, Wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    rows = []
    for k in set(F) | set(Wr):
        f = sum(F.get(k, [0])) / max(1, len(F.get(k, [])))
        w = sum(Wr.get(k, [0])) / max(1, len(Wr.get(k, [])))
        n = max(len(F.get(k, [])), len(Wr.get(k, [])))
        rows.append(((2 * f + w) * n, k, n, f, w))
    rows.sort(reverse=True)
    with open(out, "w") as fh:
        fh.write("# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, no trace domains): per kernel symbol the launches,\n"
                 "# mean KiB per launch as reported, HBM bytes per launch = (2 * FETCH + WRITE) * 1024 (gfx950: FETCH_SIZE reads half of a\n"
                 "# 16-B/lane stream, MI355X_MICROARCH.md), and the total over the run in GiB.\n")
        fh.write("launches  FETCH_KiB  WRITE_KiB  HBM_MB_per_launch  total_GiB  kernel\n")
        for tot, k, n, f, w in rows[:40]:
            fh.write(f"{n:6d} {f:12.1f} {w:12.1f} {(2 * f + w) * 1024 / 1e6:12.1f} {tot * 1024 / 2**30:10.2f}  {k[:150]}\n")
    if "--dcn" in sys.argv:      # the DCN forward line of bench.py: main kernel + everything else the operator call launches
        i = sys.argv.index("--dcn")
        jpath, commit = sys.argv[i + 1:i + 3]
        def tot(match):
            f = sum(sum(v) / len(v) for k, v in F.items() if match(k))
            w = sum(sum(v) / len(v) for k, v in Wr.items() if match(k))
            return f, w
        main_sym = next((m for m in ("dcn_win_kernel", "dcn_fast_kernel") if any(m in k for k in F)), "dcn_fwd_kernel")
        fm, wm = tot(lambda k: main_sym in k)
        fp, wp = tot(lambda k: main_sym not in k and ("dcn" in k.lower()))
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        json.dump({"kernel": main_sym, "workload": "C=Co=64 dg=16 3x3 272x480 B=8", "commit": commit, "source_sha256": bench.source_hashes("dcn"),
                   "fetch_kib_raw": fm, "write_kib_raw": wm,
                   "fetch_correction": 2.0, "hbm_bytes_per_launch": round((2.0 * fm + wm) * 1024.0),
                   "prepass_hbm_bytes_per_launch": round((2.0 * fp + wp) * 1024.0),
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/bench_dcn.py, tools/pmc_summary.py --dcn; "
                             "prepass = the call's other kernels (weight maximum / packing, the idle re-run kernel)"}, open(jpath, "w"), indent=1)
    if "--families" in sys.argv:
        i = sys.argv.index("--families")
        families(F, Wr, *sys.argv[i + 1:i + 4])
    if "--json" in sys.argv:
        i = sys.argv.index("--json")
        jpath, kern, family, prec = sys.argv[i + 1:i + 5]
        fk = [v for k, vs in F.items() if kern in k for v in vs]
        wk = [v for k, vs in Wr.items() if kern in k for v in vs]
        f, w = sum(fk) / len(fk), sum(wk) / len(wk)
        json.dump({"kernel": family, "precision": prec, "kernel_symbol_contains": kern, "launches_averaged": [len(fk), len(wk)],
                   "fetch_kib_raw": f, "write_kib_raw": w, "fetch_correction": 2.0, "hbm_bytes_per_launch": round((2.0 * f + w) * 1024.0),
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/pmc_summary.py"}, open(jpath, "w"), indent=1)


def families(F, Wr, jpath, prec, commit):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    fmap = {k: [v[0]] for k, v in bench.MFMA16.items()}
    fmap.update(bench.HBM_FAMILIES)
    out = {}
    for fam, subs in fmap.items():
        fk = [v for k, vs in F.items() if any(sb in k for sb in subs) for v in vs]
        wk = [v for k, vs in Wr.items() if any(sb in k for sb in subs) for v in vs]
        if not fk and not wk:
            continue
        n = max(len(fk), len(wk))
        f, w = sum(fk) / max(1, len(fk)), sum(wk) / max(1, len(wk))
        out[fam] = {"kernel_symbol_contains": subs, "launches_averaged": [len(fk), len(wk)], "fetch_kib_raw": f, "write_kib_raw": w,
                    "hbm_bytes_per_launch": round((2.0 * f + w) * 1024.0), "launches": n, "source_sha256": bench.source_hashes(fam)}
    json.dump({"precision": prec, "commit": commit, "fetch_correction": 2.0, "families": out,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py, tools/pmc_summary.py --families; HBM bytes "
                         "= (2 * FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts half of a 16-B/lane stream, MI355X_MICROARCH.md)"},
              open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
