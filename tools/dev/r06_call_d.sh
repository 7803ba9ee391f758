# dev (round 6, one GPU call): FC1 reading a 128-row-blocked A image (timing probe), bench line of the tree
O=gpurun_out/r06d; mkdir -p $O
python bench.py --steps 20 --warmup 5 --gemm-shapes > $O/bench_driver_like.json 2> $O/bench_driver_like.err; tail -c 300 $O/bench_driver_like.json; echo
NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV python -m nuhtc_amd.build --force > /dev/null || exit 1
for rs in 12,40 40,100; do
timeout 600 python tools/dev/r06_knob_tags.py FC1_BLK_PROBE 0 1 --tag 'gemm_kernel<2>|N256|K3136' --roi-size $rs 2>&1 | grep -v amdgpu >> $O/fc1_blk_probe.txt
done
cat $O/fc1_blk_probe.txt
python -m nuhtc_amd.build --force > /dev/null
