cp nuhtc_amd/libnuhtc_hip.so /tmp/keep.so
cp tmp_ab/probe_STAMPS.so nuhtc_amd/libnuhtc_hip.so
for s in "$@"; do echo "=== $s"; python tools/dev/split_iso.py $s 2>/dev/null; STAMPS_RT=1 python tools/dev/stamps.py | head -12; done
cp /tmp/keep.so nuhtc_amd/libnuhtc_hip.so
