#!/bin/bash
# Runs ON THE GPU BOX: the bench lines committed under profiles/r03_bench*.json
out=gpurun_out/r03b; mkdir -p $out
run() { name=$1; shift; python3 bench.py "$@" > $out/$name.json 2> $out/$name.err || { tail -3 $out/$name.err; return 1; }; tail -c 300 $out/$name.json | head -c 10 > /dev/null; echo "$name done"; }
run r03_bench_driver_w5 --steps 20 --warmup 5 &&
run r03_bench &&
run r03_bench_cnn --head cnn &&
run r03_bench_config4 --config 4 &&
run r03_bench_config5 --config 5 &&
run r03_bench_reference_149 --preset reference_149
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03b/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"].get("frac"), d["roofline"].get("kernel_ms"))
    except Exception as e:
        print(f, "unreadable", e)
PY
