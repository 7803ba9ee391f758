# dev: the fast / slow state of the dense launches by process: N back-to-back bench processes with the 10 Hz power / clock log (bench.py),
# optionally without address-space randomisation (ASLR=0: setarch -R), printing gemm_kernel<3> per step and power / sclk / fclk / socclk / mclk / temperatures
cd "$GRAFT_REPO_ROOT"
N=${1:-8}
PRE=""
if [ "$ASLR" = "0" ]; then PRE="setarch $(uname -m) -R"; fi
for i in $(seq $N); do $PRE python bench.py --no-cpu-baseline --no-roi-load --no-fp32-pipe --steps 150 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; ph=d['power']['phases']
print('run $i aslr ${ASLR:-1} value', round(d['value'],1), 'seq', round(d['sequential']['value'],1), 'gemm3', k['gemm_kernel<3>'], {n:(round(v['power_w_mean']), round(v['sclk_mhz_mean']), v['fclk_mhz_mean'], round(v['socclk_mhz_mean'] or 0), v['mclk_mhz_mean'], v['temp_junction_c_max'], v['temp_mem_c_max']) for n,v in ph.items() if n in ('timed_in_flight','sequential')})"; done
