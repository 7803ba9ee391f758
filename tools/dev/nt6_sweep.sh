# dev: 192-column wave tiles (NT=6) against the defaults on the shapes with N % 192 == 0
S="16384x1152x384 16384x384x384 16384x1536x384 16384x384x1536 4096x768x1536 4096x2304x768 4096x768x768 4096x3072x768 4096x768x3072 65536x576x192 65536x192x192 65536x768x192 65536x192x768 16384x384x768 262144x384x96 65536x192x384 16384x3072x3072"
for cfg in "0 0" "6 1" "0 0" "6 1"; do set -- $cfg
  echo "== NT=$1 MT=$2"; NUHTC_SPLIT_NT=$1 NUHTC_SPLIT_MT=$2 python tools/dev/split_iso.py $S 2>/dev/null | tr '|' '\n'
done
