# dev: column-tile width / row-tile count sweep of the split GEMM on the path's shapes, in isolation
# (the NUHTC_* switches are read by -DNUHTC_DEV builds only: build one first, the default build ignores the environment)
NUHTC_EXTRA_CFLAGS=-DNUHTC_DEV python -m nuhtc_amd.build --force > /dev/null || exit 1
S="16384x1152x384 16384x384x384 16384x1536x384 16384x384x1536 4096x768x1536 4096x2304x768 4096x768x768 4096x3072x768 4096x768x3072 16384x256x3136 65536x576x192 65536x192x192 65536x768x192 65536x192x768 16384x384x768 262144x288x96 262144x96x96 262144x384x96 262144x96x384 65536x192x384"
for cfg in "0 0" "1 1" "2 1" "3 1" "3 2" "4 1"; do set -- $cfg
  echo "== NT=$1 MT=$2"; NUHTC_SPLIT_NT=$1 NUHTC_SPLIT_MT=$2 python tools/dev/split_iso.py $S 2>/dev/null | tr '|' '\n'
done
