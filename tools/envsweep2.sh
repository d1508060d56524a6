#!/bin/bash
# usage: tools/envsweep2.sh  -- kernel A2 geometry sweep (workgroup size, register budget, quads per wave)
for rep in 1 2; do
for cfg in "4 4" "4 5" "8 4" "8 6" "6 6"; do set -- $cfg
for q in 4 5 7 10 13 19; do
  DYSF_A2_NW=$1 DYSF_A2_OCC=$2 DYSF_QUADS=$q python3 tools/kdrive.py 40 2>/dev/null | sed "s|^|NW=$1 OCC=$2 QUADS=$q  |"
done; done; done
