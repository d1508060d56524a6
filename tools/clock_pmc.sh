#!/bin/bash
# Runs ON THE GPU BOX: GRBM_GUI_ACTIVE (shader-clock cycles summed over the 8 XCDs) per kernel for the CNN and the
# reference_149 presets -> effective shader clock of every kernel = cycles / 8 / duration.
export TMPDIR=/tmp
out=gpurun_out/clock; mkdir -p $out
for cfg in "cnn --head cnn" "reference_149 --preset reference_149" "bench"; do
  set -- $cfg; name=$1; shift
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/$name -- python3 bench.py --steps 5 --warmup 20 --no-cpu-baseline --no-host-rate --no-reference-leg --bank-workers 1 --lanes 1 "$@" > $out/$name.log 2>&1 || { tail -3 $out/$name.log; continue; }
  python3 - "$out/$name" "$name" <<'PY'
import csv, glob, re, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
ctr = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
for f in ctr:
    for r in csv.DictReader(open(f)):
        m = re.search(r"dysf_[a-z0-9_]+_kernel", r["Kernel_Name"])
        if m and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a = agg[m.group(0)]; a[0] += float(r["Counter_Value"]); a[2] += 1
            if "Start_Timestamp" in r: a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, (v, ns, n) in sorted(agg.items()):
    print(f"{sys.argv[2]}\t{k}\tcycles/launch {v / n / 8:.0f}\tns/launch {ns / n:.0f}\tclock {v / 8 / ns if ns else float('nan'):.2f} GHz\t({n} launches)")
PY
  rm -rf $out/$name
done
