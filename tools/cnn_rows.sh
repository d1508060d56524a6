#!/bin/bash
# GPU diagnostic: CNN-head time against conv2's row-tile height (kRowTile, compile-time) and column-tile width (DYSF_CNN_CW).
cd recognizing-speech-dysfluencies-in-stuttering_amd/csrc
for r in 4 6 8; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDYSF_CNN_ROWS=$r -shared dysf_kernels.hip dysf_capi.hip dysf_forest.hip dysf_svc.hip dysf_mlp.hip dysf_resample.hip dysf_cnn.hip dysf_hfratio.hip dysf_clean.hip -o /tmp/libdysf_rows$r.so 2>/dev/null || exit 1
done
cd ../..
for r in 4 6 8; do for cw in 0 160 96 80; do
  if [ $cw = 0 ]; then unset DYSF_CNN_CW; else export DYSF_CNN_CW=$cw; fi
  echo -n "rows $r cw ${cw}: "; DYSF_LIB=/tmp/libdysf_rows$r.so python tools/cnn_time.py 2>/dev/null | tail -1
done; done
