#!/bin/bash
# GPU diagnostic: CNN-head kernel time with conv1's arithmetic (EXP=1), conv2's product (EXP=2), or conv2's product and half
# of conv1's channels (EXP=3) compiled out.
# Builds throw-away libraries under /tmp on the GPU box; results are wrong by construction, only the times mean something.
cd recognizing-speech-dysfluencies-in-stuttering_amd/csrc
for e in 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDYSF_CNN_EXP=$e -shared dysf_kernels.hip dysf_capi.hip dysf_forest.hip dysf_svc.hip dysf_mlp.hip dysf_resample.hip dysf_cnn.hip dysf_hfratio.hip dysf_clean.hip -o /tmp/libdysf_exp$e.so 2>/dev/null || exit 1
done
cd ../..
for lib in recognizing-speech-dysfluencies-in-stuttering_amd/libdysf_hip.so /tmp/libdysf_exp1.so /tmp/libdysf_exp2.so /tmp/libdysf_exp3.so; do
  echo "$lib"
  DYSF_LIB=$lib python tools/cnn_time.py 2>/dev/null | tail -3
done
