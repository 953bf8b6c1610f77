"""Samples the card's shader clock and socket power (rocm-smi) while a command runs -- the evidence behind DESIGN section 6's statement that
the MFMA kernels read lower inside the sustained 360 ms step than in isolation.  Usage (on the GPU box):
    python scripts/clock_trace.py OUT.json -- python bench.py --steps 25 --warmup 2 --no-cpu-baseline
The sampler never touches the HIP runtime (rocm-smi reads sysfs); the command runs as a child process."""
import json
import re
import subprocess
import sys
import time


def sample():
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    card = None
    for line in out.splitlines():
        if line.startswith('{'):
            try:
                d = json.loads(line)
            except ValueError:
                continue
            card = d.get('card0') or next(iter(d.values()))
    if card is None:
        return None
    rec = {}
    for k, v in card.items():
        kl = k.lower()
        m = re.search(r'(\d+)\s*mhz', str(v).lower())
        if kl.startswith('sclk clock speed') and m:
            rec['sclk_mhz'] = int(m.group(1))
        elif kl.startswith('mclk clock speed') and m:
            rec['mclk_mhz'] = int(m.group(1))
        elif 'power (w)' in kl:
            try:
                rec['power_w'] = float(v)
            except ValueError:
                pass
    return rec or None


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index('--') + 1:]
    t0 = time.time()
    idle = sample()
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE)
    samples = []
    while child.poll() is None:
        s = sample()
        if s is not None:
            s['t'] = round(time.time() - t0, 2)
            samples.append(s)
        time.sleep(0.1)
    line = child.stdout.read().decode().strip().splitlines()
    res = {'command': ' '.join(cmd), 'idle': idle, 'samples': samples, 'stdout_last_line': line[-1] if line else None, 'rc': child.returncode}
    both = [(s['sclk_mhz'], s['power_w']) for s in samples if 'sclk_mhz' in s and 'power_w' in s]
    if both:
        pmax = max(p for _, p in both)
        load = [(c, p) for c, p in both if p > 0.6 * pmax]
        res['summary'] = {'samples': len(both), 'samples_under_load': len(load), 'power_max_w': pmax,
                          'power_median_under_load_w': median([p for _, p in load]),
                          'sclk_max_mhz': max(c for c, _ in both), 'sclk_median_under_load_mhz': median([c for c, _ in load]),
                          'sclk_min_under_load_mhz': min(c for c, _ in load), 'sclk_mean_under_load_mhz': round(sum(c for c, _ in load) / len(load), 1)}
    json.dump(res, open(out_path, 'w'), indent=1)
    print(json.dumps(res.get('summary')))
    return child.returncode


if __name__ == '__main__':
    sys.exit(main())
