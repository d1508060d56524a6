#!/bin/bash
# GPU diagnostic: same-box A/B of the CNN head: the built library against a build with -DDYSF_CNN_OLD_POOL (earlier patch loads).
cd recognizing-speech-dysfluencies-in-stuttering_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDYSF_CNN_OLD_POOL -shared dysf_kernels.hip dysf_capi.hip dysf_forest.hip dysf_svc.hip dysf_mlp.hip dysf_resample.hip dysf_cnn.hip dysf_hfratio.hip dysf_clean.hip -o /tmp/libdysf_exp4.so 2>/dev/null || exit 1
cd ../..
for rep in 1 2 3; do
  echo -n "as built:       "; python tools/cnn_time.py 2>/dev/null | tail -1
  echo -n "fmaxf pooling:  "; DYSF_LIB=/tmp/libdysf_exp4.so python tools/cnn_time.py 2>/dev/null | tail -1
done
