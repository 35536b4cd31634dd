#!/usr/bin/env python3
"""What clock and power does the part hold under the field kernels?  Samples the amdgpu hwmon files (sclk, mclk, socket power) at
~20 Hz and `amd-smi metric` (per-XCD gfx clocks, throttle status) at ~1 Hz while `bench.py` runs one workload per operand policy,
and prints the distribution of the samples taken inside each bench run.

    python tools/clock_probe.py [steps=6]      (on the GPU box, from the repository root)"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "6"


def hwmon_files():
    out = {}
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input"):
            p = os.path.join(d, name)
            if os.path.exists(p):
                out.setdefault(name, p)
    return out


samples, smi, stop = [], [], threading.Event()


def poll_hwmon(files):
    while not stop.is_set():
        row = {"t": time.time()}
        for k, p in files.items():
            try:
                row[k] = int(open(p).read().strip())
            except Exception:
                pass
        samples.append(row)
        time.sleep(0.05)


def poll_smi():
    while not stop.is_set():
        try:
            o = subprocess.run(["amd-smi", "metric", "-g", "0", "--clock", "--power", "--json"], capture_output=True, text=True, timeout=20).stdout
            smi.append({"t": time.time(), "raw": json.loads(o)})
        except Exception as e:
            smi.append({"t": time.time(), "err": repr(e)})
        time.sleep(0.5)


files = hwmon_files()
print("hwmon files:", files)
threads = [threading.Thread(target=poll_hwmon, args=(files,), daemon=True), threading.Thread(target=poll_smi, daemon=True)]
for t in threads:
    t.start()
windows = {}
for dt in ("f32", "f16_split", "bf16"):
    env = dict(os.environ, NEDDF_BENCH_PMC="0")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dtype", dt, "--steps", steps, "--warmup", "2", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True)
    t1 = time.time()
    line = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else {"error": p.stderr[-500:]}
    # the timed region is the last steps * ms_per_step before the supplementary passes: take the samples of the middle of the run
    windows[dt] = (t0, t1, line)
stop.set()
time.sleep(0.2)


def stats(v):
    v = sorted(v)
    n = len(v)
    return {"n": n, "min": v[0], "p10": v[n // 10], "median": v[n // 2], "p90": v[(9 * n) // 10], "max": v[-1], "mean": sum(v) / n} if n else {"n": 0}


report = {}
for dt, (t0, t1, line) in windows.items():
    ms, st = line.get("ms_per_step", 0), line.get("steps", 0)
    busy = [s for s in samples if t0 <= s["t"] <= t1]
    # samples under load = those whose power is in the upper half of the window's range
    pk = "power1_average" if any("power1_average" in s for s in busy) else "power1_input"
    pw = [s[pk] for s in busy if pk in s]
    thr = (max(pw) + min(pw)) / 2 if pw else 0
    hot = [s for s in busy if s.get(pk, 0) >= thr]
    r = {"rays_per_s": line.get("value"), "ms_per_step": ms, "roofline_frac": (line.get("roofline") or {}).get("frac"),
         "avg_launch_ms": (line.get("roofline") or {}).get("avg_launch_ms"),
         "sclk_MHz_under_load": stats([s["freq1_input"] / 1e6 for s in hot if "freq1_input" in s]),
         "mclk_MHz_under_load": stats([s["freq2_input"] / 1e6 for s in hot if "freq2_input" in s]),
         "power_W_under_load": stats([s[pk] / 1e6 for s in hot if pk in s]),
         "power_W_all": stats([s[pk] / 1e6 for s in busy if pk in s])}
    ss = [x for x in smi if t0 <= x["t"] <= t1 and "raw" in x]
    r["amd_smi_samples"] = [x["raw"] for x in ss[len(ss) // 2:len(ss) // 2 + 3]]
    report[dt] = r
print(json.dumps(report, indent=1))
