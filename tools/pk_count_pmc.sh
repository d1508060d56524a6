#!/bin/bash
# Runs ON THE GPU BOX: how SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU count v_pk_fma_f32 -- the packed-fma microbenchmark kernels
# issue a known number of them (first kernel of the list: 1024 waves x 16 x 2000 = 32.8 M, plus 10-iteration warm-ups).
export TMPDIR=/tmp
out=gpurun_out/pkpmc; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate 2>/dev/null || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/run -- /tmp/valu_rate x > $out/run.log 2>&1
python3 - "$out/run" <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:40], r["Counter_Name"], float(r["Counter_Value"])))
by = collections.defaultdict(dict)
for d, k, c, v in rows: by[(d, k)][c] = v
for (d, k), c in sorted(by.items())[:14]:
    print(d, k, {n: int(v) for n, v in c.items()})
PY
rm -rf $out/run
