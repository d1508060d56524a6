#!/bin/bash
# GPU diagnostic: CNN-head kernel time against the conv2 tile width (DYSF_CNN_CW) -- LDS per workgroup, workgroups per CU
mkdir -p gpurun_out/ex
for cw in 320 96 80 64 48 32; do
  echo "DYSF_CNN_CW=$cw"
  DYSF_CNN_CW=$cw timeout -k 10 200 python bench.py --head cnn --no-cpu-baseline --no-host-rate --steps 30 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  clips/s', d['value'], 'ms/step', d['ms_per_step'], 'cnn kernel ms', d['roofline']['kernel_ms'])"
done
