#!/bin/bash
# usage: tools/sweep.sh VAR v1 v2 ...   -- kernel time of the bench batch for several values of one env knob
var=$1; shift
for v in "$@"; do env $var=$v tools/ab.sh recognizing-speech-dysfluencies-in-stuttering_amd/libdysf_hip.so 2>&1 | grep -m1 median | sed "s/^.*median/$var=$v median/"; done
