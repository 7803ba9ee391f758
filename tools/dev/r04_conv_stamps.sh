#!/bin/bash
# per-phase cycles of the conv kernel for the 16 conv launches of one bench step: -DNUHTC_DEV -DNUHTC_CONV_STAMPS build on the box
cd "$GRAFT_REPO_ROOT" || exit 1
NUHTC_EXTRA_CFLAGS="-DNUHTC_DEV -DNUHTC_CONV_STAMPS" python -m nuhtc_amd.build --force > /dev/null 2>&1
B="python bench.py --no-cpu-baseline --no-fp32-pipe --no-roi-load --steps 5 --warmup 2 --no-settle --in-flight 0"
for at in $(seq 161 176); do
  NUHTC_STAMP_AT=$at timeout 300 $B > /dev/null 2>&1
  echo "== launch $at"; python tools/dev/conv_stamps.py /tmp/conv_stamps.txt | head -3
done
