# dev: the two PMC passes (FETCH_SIZE / WRITE_SIZE) of round5_all.sh alone -> gpurun_out/round5/r05_traffic.json + r05_hbm_per_kernel.txt
R=r05; OUT=$GRAFT_REPO_ROOT/gpurun_out/round5; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/hf /tmp/hw
P="--steps 3 --warmup 1 --no-cpu-baseline --no-roi-load --no-settle --no-fp32-pipe --in-flight 0 --no-force-collective"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/hf -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/hw -- python3 $GRAFT_REPO_ROOT/bench.py $P > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_traffic.py '/tmp/h[fw]/**/*counter_collection.csv' 'gemm_split_kernel<[12], 3, [02]>' $GRAFT_REPO_ROOT/nuhtc_amd/csrc/gemm.hip > $OUT/${R}_traffic.json; cat $OUT/${R}_traffic.json
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_hbm_all.py '/tmp/h[fw]/**/*counter_collection.csv' 30 > $OUT/${R}_hbm_per_kernel.txt
NS=$(python3 - <<'P'
import csv, glob
n = 0
for f in glob.glob('/tmp/hf/**/*counter_collection.csv', recursive=True):
    n += sum(1 for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith('patch_embed') and r['Counter_Name'] == 'FETCH_SIZE')
print(n)
P
)
echo "steps in the PMC pass: $NS" >> $OUT/${R}_hbm_per_kernel.txt
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_swin_traffic.py '/tmp/h[fw]/**/*counter_collection.csv' $NS 16 >> $OUT/${R}_hbm_per_kernel.txt; tail -16 $OUT/${R}_hbm_per_kernel.txt
