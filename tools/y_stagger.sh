#!/bin/bash
# conv3x3y phase-stagger sweep (PDAE_Y_STAGGER x 1024 cycles between the four phase groups) on one box, against the direct form (PDAE_W1=0)
for shape in "32 128 128 128" "32 128 256 128" "32 64 256 256" "32 128 128 128 gn"; do
  echo "shape $shape: direct $(PDAE_W1=0 python tools/y_one.py $shape 2>/dev/null | grep ms)"
  for st in 0 2 4 6 8 12 16; do
    echo "   stagger $st: $(PDAE_Y_STAGGER=$st python tools/y_one.py $shape 2>/dev/null | grep ms)"
  done
done
