#!/bin/bash
for a in 0 1 2 3 4 7 8 15; do
  UNET_IGEMM=1 UNET_ABLATE=$a timeout -k 10 120 python bench.py --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$a', 'igemm ms/step %.2f' % d['kernels']['igemm_f32']['ms_per_step'], 'TF %.1f' % d['kernels']['igemm_f32']['tflops'])"
done
