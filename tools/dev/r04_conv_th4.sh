# dev: (apply tools/dev/probe/r04_conv_two_per_cu.patch first) builds the dev library and runs r04_conv_th4.py on the GPU box (results -> gpurun_out/conv_th4.txt)
set -x
export NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV
python -m nuhtc_amd.build --force > /dev/null || exit 1
mkdir -p gpurun_out
timeout 600 python tools/dev/r04_conv_th4.py ${1:-8} > gpurun_out/conv_th4.txt 2>&1
tail -30 gpurun_out/conv_th4.txt
