#!/bin/bash
# per-phase cycles of the conv kernel with and without the fused pointwise epilogue (-DNUHTC_DEV -DNUHTC_CONV_STAMPS build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/fuse
B="python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 5 --warmup 2 --no-settle --in-flight 0"
for at in 161 165 169 172; do
  for fz in 1 0; do
    NUHTC_CONV_FUSE=$fz NUHTC_STAMP_AT=$at timeout 300 $B > /dev/null 2> gpurun_out/fuse/st_$at.err
    echo "== launch $at fuse $fz"; python tools/dev/conv_stamps.py /tmp/conv_stamps.txt
  done
done
