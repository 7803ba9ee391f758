# dev: the split GEMM on the step's main shapes with and without its output stores (-DNUHTC_GEMM_NOSTORE: results lost) -- the most that hiding the
export NUHTC_DEV=1   # the probe builds below give wrong results by design: nuhtc_create refuses them without this
# store phase of every tile behind other tiles' k-loops could buy, per shape, in isolation (back-to-back launches of one shape)
cd "$GRAFT_REPO_ROOT"
S="16384x1536x384 16384x384x1536 16384x1152x384 16384x384x384 65536x768x192 65536x192x768 65536x576x192 65536x192x192 4096x3072x768 4096x768x3072 4096x2304x768 4096x768x768"
for v in "" "-DNUHTC_GEMM_NOSTORE"; do
  NUHTC_EXTRA_CFLAGS="$v" python -m nuhtc_amd.build --force > /dev/null 2>&1
  for r in 1 2; do echo "== build '$v' round $r"; ISO_N=20 python tools/dev/split_iso.py $S 2>/dev/null | tr '|' '\n'; done
done
