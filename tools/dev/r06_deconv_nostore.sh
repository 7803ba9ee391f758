# dev (round 6): what fusing the mask head's deconv with the 64 -> 1 logits layer could give at most: the deconv GEMM (gemm_kernel<4>|N256|K64, fp32 MFMA kernel,
# ST_DECONV2 scatter of 184 MB per step) timed without its stores (-DNUHTC_GEMM_NOSTORE, wrong results), beside the tree; the logits kernel (conv1x1_n1, 31 us) would go too
export NUHTC_DEV=1
O=gpurun_out/r06_deconv_nostore.txt; : > $O
for v in tree nostore; do
  if [ $v = tree ]; then unset NUHTC_EXTRA_CFLAGS_GEMM; else export NUHTC_EXTRA_CFLAGS_GEMM=-DNUHTC_GEMM_NOSTORE; fi
  python -m nuhtc_amd.build --force > /dev/null || exit 1
  for r in 1 2; do timeout 200 python tools/dev/r06_tags_fixed.py "N256|K64" conv1x1 paste tile_post 2>/dev/null | sed "s/^/$v: /" >> $O; done
done
unset NUHTC_EXTRA_CFLAGS_GEMM
python -m nuhtc_amd.build --force > /dev/null
cat $O
