#!/bin/bash
for a in 0 15; do
  echo "== ablate=$a"
  UNET_IGEMM=1 UNET_CLOCK=1 UNET_CLOCK_PRINT=1 UNET_ABLATE=$a timeout -k 10 120 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | grep -E "in-kernel clock|tiles/sec" | tail -5 | cut -c1-200
done
