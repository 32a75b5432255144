#!/bin/bash
for n in 256 384 512 768 1024; do
for fb in 4096 2048 1536 1024; do
  echo -n "N=$n FUSED_BLOCKS=$fb: "; N=$n P3D_FUSED_BLOCKS=$fb python tools/dev/fused_time.py 2>&1 | tail -1 | sed -e 's/.*k_fused/k_fused/'
done; done
