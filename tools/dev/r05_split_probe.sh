# dev (round 5, verdict item 5): what a producer-side operand split could give the consumer GEMM at most, per shape of the step.
# The split GEMM with the A operand's three-way split compiled OUT of its k-loop (-DNUHTC_GEMM_PROBE_NOSPLIT: wrong results; the planes are
# bit casts) against the kernel as it is, on isolated shapes (back-to-back launches of one shape, events inside the library), with 128-row
# (latency schedule) and 256-row (throughput schedule: NUHTC_SPLIT_MT=2) block tiles.  A real producer-side split would ALSO read 1.5x the
# A bytes (three bf16 planes instead of one fp32), which this probe does not charge: it is an upper bound.
export NUHTC_DEV=1   # the probe build gives wrong results by design: nuhtc_create refuses it without this
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out/r05_split_probe.txt; : > $O
S="65536x576x192 65536x192x192 65536x768x192 65536x192x768 16384x1152x384 16384x384x384 16384x1536x384 16384x384x1536 4096x2304x768 4096x768x768 4096x3072x768 4096x768x3072 17024x256x3136"
for v in "" "-DNUHTC_GEMM_PROBE_NOSPLIT"; do
  NUHTC_EXTRA_CFLAGS="-DNUHTC_DEV $v" python -m nuhtc_amd.build --force > /dev/null 2>&1 || exit 1
  for mt in 1 2; do for r in 1 2; do
    echo "== build '$v' SPLIT_MT=$mt round $r" >> $O
    NUHTC_SPLIT_MT=$mt ISO_N=20 python tools/dev/split_iso.py $S 2>/dev/null | tr '|' '\n' >> $O
  done; done
done
python -m nuhtc_amd.build --force > /dev/null 2>&1
cat $O
