# dev (round 6): the shared FC1 launch (gemm_kernel<2>|N256|K3136) with and without its output stores, nothing else changed (fixed load)
export NUHTC_DEV=1
O=gpurun_out/r06_fc1_nostore.txt; : > $O
for v in tree nostore; do
  if [ $v = tree ]; then unset NUHTC_EXTRA_CFLAGS_GEMM; else export NUHTC_EXTRA_CFLAGS_GEMM="-DNUHTC_GEMM_NOSTORE -DNUHTC_GEMM_NOSTORE_K=3136"; fi
  python -m nuhtc_amd.build --force > /dev/null || exit 1
  for r in 1 2; do timeout 200 python tools/dev/r06_tags_fixed.py "K3136" "N256|K256" 2>/dev/null | sed "s/^/$v: /" >> $O; done
done
unset NUHTC_EXTRA_CFLAGS_GEMM
python -m nuhtc_amd.build --force > /dev/null
cat $O
