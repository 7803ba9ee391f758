# dev: what each kernel group costs the step with four batches in flight, final round-6 tree (tools/dev/r04_ablate.py, -DNUHTC_DEV build)
mkdir -p gpurun_out; O=gpurun_out/r06_ablation_in_flight.txt
NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV python -m nuhtc_amd.build --force > /dev/null || exit 1
timeout 600 python tools/dev/r04_ablate.py > $O 2>/dev/null
python -m nuhtc_amd.build --force > /dev/null
cat $O
