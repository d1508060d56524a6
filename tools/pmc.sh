#!/bin/bash
# usage: tools/pmc.sh <outdir-under-gpurun_out> "<counter list>" [bench args]   (run on the GPU box)
# Collects PMC counters for the dysf kernels of a short bench run (no forks under --pmc: --bank-workers 1).
set -e
out=gpurun_out/$1; shift
ctrs="$1"; shift
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --bank-workers 1 "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dysf::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("dysf::")[1].split("<")[0]
            a = agg[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print(f"{k:24s} {c:28s} per-launch {v / n:16.1f}  (launches {n})")
PY
