# dev: per-wave stamps of the split GEMM on isolated shapes, dumped from the 60th back-to-back launch (sustained clocks)
cp nuhtc_amd/libnuhtc_hip.so /tmp/keep.so
cp tmp_ab/probe_STAMPS.so nuhtc_amd/libnuhtc_hip.so
for s in "$@"; do echo "=== $s"; ISO_N=80 NUHTC_STAMP_AT=60 python tools/dev/split_iso.py $s 2>/dev/null; STAMPS_RT=1 python tools/dev/stamps.py | head -13; done
cp /tmp/keep.so nuhtc_amd/libnuhtc_hip.so
