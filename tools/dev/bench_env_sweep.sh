#!/bin/bash
# dev: launch-shape knobs of the streaming kernel swept INSIDE bench.py's back-to-back call stream (sustained load
# shifts some optima that isolated-call timings show as ties)
run() { python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])"; }
for rep in 1 2; do
  echo -n "default: "; run
  for xt in 8 10 12 14 16; do echo -n "XT=$xt: "; P3D_FUSED_XT=$xt run; done
  for tl in 2 3 4 6; do echo -n "XT_TAIL=$tl: "; P3D_FUSED_XT_TAIL=$tl run; done
  for dv in 2 3 6 8; do echo -n "TAIL_DIV=$dv: "; P3D_FUSED_TAIL_DIV=$dv run; done
  for cb in 128 512; do echo -n "COMPACT_BLOCKS=$cb: "; P3D_COMPACT_BLOCKS=$cb run; done
  for ce in 1 2 5; do echo -n "COMPACT_EARLY=$ce: "; P3D_COMPACT_EARLY=$ce run; done
done
