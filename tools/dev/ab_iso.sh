for v in old new old new; do
  cp tmp_ab/$v.so nuhtc_amd/libnuhtc_hip.so; echo "== $v"
  python tools/dev/gemm_iso.py 262144x384x96 262144x288x96 65536x768x192 16384x1536x384 16384x3072x3072 2>&1 | grep TF
done
bash tools/dev/ab.sh
