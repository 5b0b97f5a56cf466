#!/bin/bash
# conv3x3y against the direct form on one box (clock-warmed single-op timings)
for shape in "32 128 128 128" "32 128 256 128" "32 64 256 256" "32 32 256 256" "32 128 128 128 gn" "32 64 256 256 gn"; do
  echo "shape $shape: direct $(PDAE_W1=0 python tools/y_one.py $shape 2>/dev/null | grep ms)   W1 $(python tools/y_one.py $shape 2>/dev/null | grep ms)"
done
