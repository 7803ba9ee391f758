# dev (round 6): what a producer-split QKV image could give window attention at most: the kernel with its K / Q / V operand splits replaced
# by bit moves (-DNUHTC_ATTN_PROBE=16, wrong results) and with half as many K / Q bytes again loaded (32: three bf16 planes are 6 B per
# element against 4) -- `window_attn` ms per step, one batch at a time, builds alternating on one box
export NUHTC_DEV=1
mkdir -p gpurun_out tmp_ab; O=gpurun_out/r06_attn_presplit.txt; : > $O
cp nuhtc_amd/libnuhtc_hip.so /tmp/keep_attn.so
for m in 0 16 48; do
  if [ $m = 0 ]; then unset NUHTC_EXTRA_CFLAGS_SWIN; else export NUHTC_EXTRA_CFLAGS_SWIN=-DNUHTC_ATTN_PROBE=$m; fi
  python -m nuhtc_amd.build --force > /dev/null || exit 1
  cp nuhtc_amd/libnuhtc_hip.so tmp_ab/attn$m.so
done
unset NUHTC_EXTRA_CFLAGS_SWIN
for r in 1 2 3; do for m in 0 16 48; do cp tmp_ab/attn$m.so nuhtc_amd/libnuhtc_hip.so
  timeout 200 python - >> $O 2>/dev/null <<PY
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256), bind_host=True)
torch.cuda.set_stream(eng.stream)
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
for _ in range(40): eng.infer_async(tiles, hip.CH_SWAP)
torch.cuda.synchronize(); hip.profile_enable(True)
for _ in range(8): eng.infer_async(tiles, hip.CH_SWAP)
p = hip.profile_read()
print('probe mask $m: window_attn', round(sum(x['ms'] for k, x in p.items() if k.startswith('window_attn')) / 8, 4), 'ms per step', {k.split('|')[1]: round(x['ms'] / 8, 4) for k, x in p.items() if k.startswith('window_attn')})
PY
done; done
cp /tmp/keep_attn.so nuhtc_amd/libnuhtc_hip.so
python -m nuhtc_amd.build --force > /dev/null
cat $O
