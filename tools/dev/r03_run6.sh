#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3f
NUHTC_PLANES=0 timeout 200 python tools/dev/dump_stage.py /tmp/a.npz > gpurun_out/r3f/dump.log 2>&1
NUHTC_PLANES=1 NUHTC_HID_PLANES=0 timeout 200 python tools/dev/dump_stage.py /tmp/b.npz >> gpurun_out/r3f/dump.log 2>&1
NUHTC_PLANES=1 NUHTC_HID_PLANES=1 timeout 200 python tools/dev/dump_stage.py /tmp/c.npz >> gpurun_out/r3f/dump.log 2>&1
tail -3 gpurun_out/r3f/dump.log
python tools/dev/cmp_stage.py /tmp/a.npz /tmp/b.npz | grep -c "identical True"
python tools/dev/cmp_stage.py /tmp/a.npz /tmp/c.npz | tee gpurun_out/r3f/cmp.txt | grep -c "identical True"
bash tools/dev/r03_ab.sh NUHTC_PLANES=0 NUHTC_PLANES=1,NUHTC_HID_PLANES=0 NUHTC_PLANES=1,NUHTC_HID_PLANES=1 NUHTC_PLANES=0
