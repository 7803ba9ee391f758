# dev (round 6, one GPU call): parity of the one-launch RPN head, slide-level bench with documents, A/B of the RPN launch form, attention presplit probe
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_full.py tests/test_hip_edges.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -5 > $O/parity.log; tail -2 $O/parity.log
timeout 600 python tools/bench_wsi.py > $O/bench_wsi.json 2> $O/bench_wsi.err; tail -c 900 $O/bench_wsi.json; echo
cp nuhtc_amd/libnuhtc_hip.so /tmp/keep_default.so
NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV python -m nuhtc_amd.build --force > /dev/null || exit 1
timeout 600 python tools/dev/knob_ab.py RPN_ONE_LAUNCH 0 1 --rounds 10 --steps 10 --tags gemm > $O/rpn_one_launch_ab.txt 2>&1; grep -v amdgpu $O/rpn_one_launch_ab.txt | tail -6
python -m nuhtc_amd.build --force > /dev/null
bash tools/dev/r06_attn_presplit.sh > $O/attn_presplit.log 2>&1; cp gpurun_out/r06_attn_presplit.txt $O/; tail -9 $O/r06_attn_presplit.txt
