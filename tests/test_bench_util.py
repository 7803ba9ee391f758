"""CPU tests of bench.py's host-side helpers (no GPU): the 10 Hz power / clock sampler against a fake sysfs tree, and the argument parser."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_card(root, idx, bdf, power_uw, sclk_hz, busy):
    dev = os.path.join(root, 'devices', bdf)
    hm = os.path.join(dev, 'hwmon', 'hwmon3')
    os.makedirs(hm)
    os.makedirs(os.path.join(root, 'class/drm'), exist_ok=True)
    os.makedirs(os.path.join(root, 'class/drm', f'card{idx}'))
    os.symlink(dev, os.path.join(root, 'class/drm', f'card{idx}', 'device'))
    w = lambda p, v: open(p, 'w').write(v)
    w(os.path.join(hm, 'power1_input'), f'{power_uw}\n')
    w(os.path.join(hm, 'power1_cap'), '1400000000\n')
    w(os.path.join(hm, 'freq1_input'), f'{sclk_hz}\n')
    w(os.path.join(hm, 'temp2_label'), 'junction\n'); w(os.path.join(hm, 'temp2_input'), '55000\n')
    w(os.path.join(hm, 'temp3_label'), 'mem\n'); w(os.path.join(hm, 'temp3_input'), '48000\n')
    w(os.path.join(dev, 'gpu_busy_percent'), f'{busy}\n')
    w(os.path.join(dev, 'pp_dpm_fclk'), '0: 1250Mhz *\n')
    w(os.path.join(dev, 'pp_dpm_mclk'), '0: 900Mhz\n1: 2000Mhz *\n')
    return dev


def test_power_log_phases_and_card_choice(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    root = str(tmp_path / 'sys')
    _fake_card(root, 0, '0000:05:00.0', 300_000_000, 1_400_000_000, 100)       # somebody else's GPU: busier
    _fake_card(root, 8, '0000:0a:00.0', 1_300_000_000, 2_000_000_000, 90)       # ours
    pl = bench.PowerLog(sysfs=root)
    assert len(pl.cards) == 2
    pl.mark('_setup')
    time.sleep(0.25)
    pl.mark('timed_in_flight')
    time.sleep(0.45)
    pl.mark('_after')
    time.sleep(0.15)
    csv = str(tmp_path / 'p.csv')
    out = pl.summary(csv, pci_bdf='0000:0a:00.0')
    assert out['available'] and out['device'].endswith('0000:0a:00.0') and out['device_chosen_by'].startswith('pci bus id') and out['power_cap_w'] == 1400.0
    assert list(out['phases']) == ['timed_in_flight']                          # phases whose name starts with '_' are not reported
    ph = out['phases']['timed_in_flight']
    assert 2 <= ph["samples"] <= 7 and ph['power_w_mean'] == 1300.0 and ph['sclk_mhz_mean'] == 2000.0
    assert ph['temp_junction_c_max'] == 55.0 and ph['temp_mem_c_max'] == 48.0 and ph['fclk_mhz_mean'] == 1250.0 and ph['mclk_mhz_mean'] == 2000.0
    lines = open(csv).read().splitlines()
    assert lines[0].startswith('t_s,power_w,sclk_mhz') and lines[-1].endswith('_after') and any(l.endswith('timed_in_flight') for l in lines)
    # without a bus id (or without a match) the busiest card is taken, and the line says so
    pl2 = bench.PowerLog(sysfs=root)
    pl2.mark('x')
    time.sleep(0.25)
    out2 = pl2.summary(None, pci_bdf='0000:ff:00.0')
    assert out2['device'].endswith('0000:05:00.0') and out2['device_chosen_by'] == 'busiest card in sysfs'
    # a box without readable hwmon files
    assert bench.PowerLog(sysfs=str(tmp_path / 'none')).summary() == {'available': False, 'note': 'no amdgpu hwmon files readable on this box'}


def test_bench_help_lists_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--help'], capture_output=True, text=True, check=True).stdout
    for flag in ('--gpus', '--steps', '--warmup', '--in-flight', '--cpu-full', '--power-csv', '--fixed-load'):
        assert flag in out
