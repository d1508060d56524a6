#!/bin/bash
# Runs ON THE GPU BOX: vector-pipe counters of the CNN kernel with conv2's product compiled out (DYSF_CNN_EXP=2: conv1 alone)
# (EXP=1: no conv1 arithmetic; EXP=3: half of conv1's channels, no conv2 product) and of the normal build.
export TMPDIR=/tmp
out=gpurun_out/cnnpmc; mkdir -p $out
cd recognizing-speech-dysfluencies-in-stuttering_amd/csrc
for e in 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDYSF_CNN_EXP=$e -shared dysf_kernels.hip dysf_capi.hip dysf_forest.hip dysf_svc.hip dysf_mlp.hip dysf_resample.hip dysf_cnn.hip dysf_hfratio.hip dysf_clean.hip -o /tmp/libdysf_exp$e.so 2>/dev/null || exit 1
done
cd ../..
for name in normal exp1 exp2 exp3; do
  if [ $name != normal ]; then export DYSF_LIB=/tmp/libdysf_$name.so; fi
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/$name -- python3 tools/cnn_time.py > $out/$name.log 2>&1
  python3 - "$out/$name" "$name" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv2" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
c = {k: v[0] / v[1] for k, v in agg.items()}
cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
print(sys.argv[2], {k: round(v) for k, v in c.items()})
if cyc: print("   kernel cycles %.0f; VALU instructions per SIMD %.0f; VALU busy %.2f; waves: issuing %.2f, stalled %.2f, waiting %.2f" % (
    cyc, c["SQ_INSTS_VALU"] / 1024, c["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024), c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
  rm -rf $out/$name
done
