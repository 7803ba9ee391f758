"""dev: run a command as a child process and log the GPU's power / clocks / temperatures at ~10 Hz beside it (sysfs hwmon of the first
amdgpu card; no GPU call is made here, so the child may be anything -- also a rocprofv3 line).

    python tools/dev/power_log.py out.csv -- python bench.py ...

CSV: t_s, power_w, sclk_mhz, mclk_mhz, temp_edge_c, temp_junction_c, temp_mem_c, gpu_busy_pct (whatever the box exposes; missing = empty).
The child's stdout / stderr pass through; exit code = the child's."""
import glob
import os
import subprocess
import sys
import threading
import time


def _read(path, scale=1.0):
    try:
        with open(path) as f:
            return float(f.read().split()[0]) * scale
    except Exception:
        return None


def find_cards():
    out = []
    for card in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
        hm = glob.glob(os.path.join(card, 'hwmon', 'hwmon*'))
        if hm and (os.path.exists(os.path.join(hm[0], 'power1_average')) or os.path.exists(os.path.join(hm[0], 'power1_input'))):
            out.append((card, hm[0]))
    return out


def main():
    out, cmd = sys.argv[1], sys.argv[sys.argv.index('--') + 1:]
    cards = find_cards()           # a box shows every GPU of its host in sysfs; the one the child used is the one that got busy
    stop = threading.Event()
    rows = {c: [] for c, _ in cards}
    labels = {}
    for card, hm in cards:
        for p in glob.glob(os.path.join(hm, 'temp*_label')):
            try:
                labels[(card, open(p).read().strip())] = p.replace('_label', '_input')
            except Exception:
                pass

    def loop():
        t0 = time.time()
        while not stop.is_set():
            for card, hm in cards:
                pw = _read(os.path.join(hm, 'power1_average'), 1e-6)
                if pw is None:
                    pw = _read(os.path.join(hm, 'power1_input'), 1e-6)
                t = lambda l: _read(labels[(card, l)], 1e-3) if (card, l) in labels else None
                rows[card].append((time.time() - t0, pw, _read(os.path.join(hm, 'freq1_input'), 1e-6), _read(os.path.join(hm, 'freq2_input'), 1e-6),
                                   t('edge'), t('junction'), t('mem'), _read(os.path.join(card, 'gpu_busy_percent'))))
            time.sleep(0.1)
    th = None
    if cards:
        th = threading.Thread(target=loop, daemon=True)
        th.start()
    rc = subprocess.call(cmd)
    stop.set()
    if th:
        th.join()
    best = max(rows, key=lambda c: (sum((r[7] or 0) for r in rows[c]), max([(r[1] or 0) for r in rows[c]] or [0]))) if rows else None
    with open(out, 'w') as f:
        f.write('t_s,power_w,sclk_mhz,mclk_mhz,temp_edge_c,temp_junction_c,temp_mem_c,gpu_busy_pct\n')
        if best is None:
            f.write('# no amdgpu hwmon readable on this box\n')
        else:
            f.write(f'# {os.path.realpath(best)} (the busiest of {len(cards)} cards visible in sysfs); power cap '
                    f'{_read(os.path.join(dict(cards)[best], "power1_cap"), 1e-6)} W\n')
            for r in rows[best]:
                f.write(','.join('' if v is None else f'{v:.3f}' for v in r) + '\n')
    sys.exit(rc)


if __name__ == '__main__':
    main()
