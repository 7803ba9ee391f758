"""Dev helper: print per-shape GEMM TFLOP/s and the step breakdown of one bench-like run."""
import json, subprocess, sys, os
env = dict(os.environ)
out = subprocess.run([sys.executable, 'bench.py', '--steps', '8', '--warmup', '2', '--no-cpu-baseline', '--gemm-shapes'], capture_output=True, text=True, env=env).stdout
d = json.loads(out.strip().splitlines()[-1])
print('tiles/s %.1f  ms/step %.2f  frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
print({k: v for k, v in d['kernel_ms_per_step'].items() if v > 0.3})
for k, v in sorted(d['gemm_shapes'].items()):
    if v['ms_per_step'] > 0.2:
        print('  %-36s %6.3f ms  %6.1f TF' % (k, v['ms_per_step'], v['tflops']))
