# dev: is the conv main loop bound by its LDS reads?  The kernel with a quarter of them left out (-DNUHTC_CONV_PROBE_READS, wrong results)
export NUHTC_DEV=1   # the probe builds below give wrong results by design: nuhtc_create refuses them without this
# against the real one: conv tags per step, alternating processes (results -> gpurun_out/conv_probe_reads.txt)
mkdir -p gpurun_out tmp_ab; O=gpurun_out/conv_probe_reads.txt; : > $O
python -m nuhtc_amd.build --force > /dev/null && cp nuhtc_amd/libnuhtc_hip.so tmp_ab/real.so
NUHTC_EXTRA_CFLAGS_CONV=-DNUHTC_CONV_PROBE_READS python -m nuhtc_amd.build --force > /dev/null && cp nuhtc_amd/libnuhtc_hip.so tmp_ab/probe.so
for r in 1 2 3; do for v in real probe; do cp tmp_ab/$v.so nuhtc_amd/libnuhtc_hip.so
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
g = {k.split('|', 3)[-1]: round(x['ms'] / 5, 4) for k, x in sorted(p.items()) if 'conv3' in k}
print('$v', g, 'sum', round(sum(x['ms'] for k, x in p.items() if 'conv3' in k) / 5, 4))
PY
done; done
cp tmp_ab/real.so nuhtc_amd/libnuhtc_hip.so
cat $O
