#!/bin/bash
# dev: sweep launch geometry knobs on the 512^3 bench (whole call, pipelined)
run() { echo -n "$* : "; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])"; }
for cb in 128 256 384; do run P3D_COMPACT_BLOCKS=$cb; done
for xt in 8 10 12 16; do for td in 4 8 16; do for xtt in 2 4; do run P3D_COMPACT_BLOCKS=256 P3D_FUSED_XT=$xt P3D_FUSED_TAIL_DIV=$td P3D_FUSED_XT_TAIL=$xtt; done; done; done
