# dev: what the fused conv epilogue spends its cycles on: -DNUHTC_CONV_PROBE_EPI=mask (1 no stores, 2 no residual loads, 4 no exchange barrier,
export NUHTC_DEV=1   # the probe builds below give wrong results by design: nuhtc_create refuses them without this
# 8 no second product; wrong results) -- the fixed-size fused conv tags in ms per step, one batch at a time
mkdir -p gpurun_out tmp_ab; O=gpurun_out/conv_probe_epi.txt; : > $O
for m in 0 1 2 4 8 15; do
  if [ $m = 0 ]; then unset NUHTC_EXTRA_CFLAGS_CONV; else export NUHTC_EXTRA_CFLAGS_CONV=-DNUHTC_CONV_PROBE_EPI=$m; fi
  python -m nuhtc_amd.build --force > /dev/null || exit 1
  cp nuhtc_amd/libnuhtc_hip.so tmp_ab/cv$m.so
done
unset NUHTC_EXTRA_CFLAGS_CONV
for r in 1 2; do for m in 0 1 2 4 8 15; do cp tmp_ab/cv$m.so nuhtc_amd/libnuhtc_hip.so
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
g = {k.split('|', 3)[-1]: round(x['ms'] / 5, 4) for k, x in sorted(p.items()) if 'conv3halo+' in k}
print('probe mask $m:', g)
PY
done; done
cp tmp_ab/cv0.so nuhtc_amd/libnuhtc_hip.so
cat $O
