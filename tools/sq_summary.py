"""Per-kernel sums of the counter passes of tools/sq_passes.sh:  python tools/sq_summary.py <outdir> [kernel-name substring]"""
import csv, glob, os, sys, collections, json
out = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "gru"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(os.path.join(out, "p*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if sub not in k: continue
        k = k.split("(")[0].replace("void ggnn::", "")[:90]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
for k in acc:
    per = {c: acc[k][c] / max(n[k][c], 1) for c in acc[k]}
    print(k); print(json.dumps({c: round(v, 1) for c, v in sorted(per.items())}, indent=0).replace("\n", " "))
