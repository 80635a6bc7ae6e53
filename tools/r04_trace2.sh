#!/bin/bash
for v in "$@"; do
cp tools/ab/$v.so orbslam2_amd/liborbfe.so
bash tools/step_trace.sh $v
done
