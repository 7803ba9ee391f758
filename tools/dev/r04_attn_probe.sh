# dev: what the window-attention launches wait for: the kernel with its bias (1), V (2) or K / Q (4) loads or its output stores (8) left out (-DNUHTC_ATTN_PROBE=mask, wrong
export NUHTC_DEV=1   # the probe builds below give wrong results by design: nuhtc_create refuses them without this
# results; the attention launches' own work does not depend on the data) -- `window_attn` ms per step, one batch at a time
mkdir -p gpurun_out tmp_ab; O=gpurun_out/attn_probe.txt; : > $O
for m in 0 8 15; do
  if [ $m = 0 ]; then unset NUHTC_EXTRA_CFLAGS_SWIN; else export NUHTC_EXTRA_CFLAGS_SWIN=-DNUHTC_ATTN_PROBE=$m; fi
  python -m nuhtc_amd.build --force > /dev/null || exit 1
  cp nuhtc_amd/libnuhtc_hip.so tmp_ab/attn$m.so
done
unset NUHTC_EXTRA_CFLAGS_SWIN
for r in 1 2; do for m in 0 8 15; do cp tmp_ab/attn$m.so nuhtc_amd/libnuhtc_hip.so
  timeout 200 python - >> $O 2>/dev/null <<PY
import torch
from nuhtc_amd import hip, synth, weights
from nuhtc_amd.engine import Engine
eng = Engine(weights.bench_state_dict(), device=0, max_batch=16, tile=(256, 256))
torch.cuda.set_stream(eng.stream)
tiles = eng.to_device(synth.nuclei_tiles(16, 256))
for _ in range(30): eng.infer_async(tiles, hip.CH_SWAP)
torch.cuda.synchronize(); hip.profile_enable(True)
for _ in range(5): eng.infer_async(tiles, hip.CH_SWAP)
p = hip.profile_read()
print('probe mask $m: window_attn', round(sum(x['ms'] for k, x in p.items() if k.startswith('window_attn')) / 5, 4), 'ms per step')
PY
done; done
cp tmp_ab/attn0.so nuhtc_amd/libnuhtc_hip.so
cat $O
