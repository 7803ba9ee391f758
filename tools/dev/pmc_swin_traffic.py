"""HBM bytes of the Swin stage per tile from the FETCH_SIZE / WRITE_SIZE passes (same corrections as pmc_hbm_all.py):
    pmc_swin_traffic.py '<glob>' <steps in the profiled run> <tiles per step>"""
import csv, glob, re, sys, collections
steps, tiles = int(sys.argv[2]), int(sys.argv[3])
SWIN = re.compile(r'gemm_split_kernel<[12], 3, [02]>|swin_mlp_kernel|swin_lnqkv_kernel|qkv_pad_rows|layernorm|ln_stats|window_attn|merge_ln|patch_embed')
rd = collections.defaultdict(float); wr = collections.defaultdict(float)
for f in glob.glob(sys.argv[1], recursive=True):
    for row in csv.DictReader(open(f)):
        if SWIN.search(row['Kernel_Name']):
            k = row['Kernel_Name'].split('(')[0][:48]
            if row['Counter_Name'] == 'FETCH_SIZE': rd[k] += 2 * float(row['Counter_Value']) * 1024
            elif row['Counter_Name'] == 'WRITE_SIZE': wr[k] += float(row['Counter_Value']) * 1024
tot = 0.0
for k in sorted(set(rd) | set(wr), key=lambda k: -(rd[k] + wr[k])):
    b = (rd[k] + wr[k]) / steps / tiles
    tot += b
    print(f'{k:50s} {b / 1e6:8.1f} MB per tile  (read {rd[k] / steps / tiles / 1e6:.1f}, write {wr[k] / steps / tiles / 1e6:.1f})')
print(f'Swin stage (96-column GEMMs, fused stage-1 kernels, LayerNorm, attention, patch embed / merge): {tot / 1e6:.1f} MB per tile '
      f'(SURVEY 8d fused ideal 119.5 MB per tile; round 2: ~980)')
