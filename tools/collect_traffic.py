"""Developer tool: turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/traffic_r01.json.
Usage: python tools/collect_traffic.py <fetch_dir> <write_dir> <kernel-substring> <family-name> <precision>
FETCH_SIZE reads 1/2 of the bytes of a 16-B/lane stream on gfx950 (MI355X_MICROARCH.md, HBM): doubled here."""
import csv, glob, json, os, sys


def mean_counter(d, name, kern):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and kern in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


fetch_dir, write_dir, kern, family, prec = sys.argv[1:6]
f, nf = mean_counter(fetch_dir, "FETCH_SIZE", kern)
w, nw = mean_counter(write_dir, "WRITE_SIZE", kern)
out = {"kernel": family, "precision": prec, "kernel_symbol_contains": kern, "launches_averaged": [nf, nw],
       "fetch_kib_raw": f, "write_kib_raw": w, "fetch_correction": 2.0,
       "hbm_bytes_per_launch": round((2.0 * f + w) * 1024.0), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)"}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "profiles", "traffic_r01.json"), "w"), indent=1)
print(out)
