#!/bin/bash
# usage: tools/envsweep.sh "VAR1=a VAR2=b" "VAR1=c" ...  -- per-kernel HIP-event times of the bench batch for several env settings
for rep in 1 2; do
for setting in "$@"; do
  env $setting python3 tools/kdrive.py 60 | sed "s|^|$setting  |"
done; done
