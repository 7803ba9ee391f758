# dev (round 6, one GPU call): the new two-rank .svs slide bench test; RPN head in one launch against one launch per level with four batches in flight
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_api.py -m gpu -x -q -k "svs_file_on_every_rank or eight_ranks" 2>&1 | grep -v amdgpu.ids | tail -4 > $O/tests.log; tail -2 $O/tests.log
NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV python -m nuhtc_amd.build --force > /dev/null || exit 1
: > $O/rpn_one_launch_in_flight.txt
for r in 1 2 3; do for v in 0 1; do
  NUHTC_RPN_ONE_LAUNCH=$v python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RPN_ONE_LAUNCH=$v', 'value', round(d['value'],1), 'steady', round(d['value_steady'],1), 'sequential', round(d['sequential']['value'],1), 'dense_convs_fcs', d['kernel_groups']['dense_convs_fcs']['frac_of_fp32_mfma'], 'clock', d['roofline']['shader_clock_ghz_under_step'])" >> $O/rpn_one_launch_in_flight.txt
done; done
cat $O/rpn_one_launch_in_flight.txt
python -m nuhtc_amd.build --force > /dev/null
