#!/usr/bin/env python
"""What differs between a fast and a slow box (DESIGN.md "timing modes": the same binary, C3 at full width, 39 ms on one box and 46 ms on another,
while 2048 / 1024 / 512 chains run alike on both)?  Clocks and power sampled from sysfs WHILE three loads run -- a streaming copy, the random-line
probe, the C3 tracked slice -- plus their rates.      python tools/mode_fingerprint.py      (one JSON line)"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def dev_dir():
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
            return d
    return None


DEV = dev_dir()


def cur_level(name):
    try:
        for line in open(os.path.join(DEV, name)):
            if "*" in line:
                return line.split(":")[1].replace("*", "").strip()
    except Exception:
        return None
    return None


def power_w():
    for p in glob.glob(os.path.join(DEV or "", "hwmon/hwmon*/power1_average")) + glob.glob(os.path.join(DEV or "", "hwmon/hwmon*/power1_input")):
        try:
            return int(open(p).read()) / 1e6
        except Exception:
            pass
    return None


class Sampler:
    def __init__(self):
        self.rows, self.stop = [], False
        self.t = threading.Thread(target=self.run, daemon=True)

    def run(self):
        while not self.stop:
            self.rows.append((cur_level("pp_dpm_sclk"), cur_level("pp_dpm_mclk"), cur_level("pp_dpm_fclk"), cur_level("pp_dpm_socclk"), power_w()))
            time.sleep(0.05)

    def __enter__(self):
        self.t.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.t.join()

    def summary(self):
        def col(k):
            v = [r[k] for r in self.rows if r[k] is not None]
            if not v:
                return None
            if isinstance(v[0], str):
                u = sorted(set(v))
                return u if len(u) > 1 else u[0]
            return {"min": min(v), "max": max(v), "mean": sum(v) / len(v)}
        return {"sclk": col(0), "mclk": col(1), "fclk": col(2), "socclk": col(3), "power_W": col(4), "samples": len(self.rows)}


out = {"sysfs": DEV}
try:
    smi = subprocess.run(["rocm-smi", "--showuniqueid", "--showperflevel", "--showmaxpower"], capture_output=True, text=True, timeout=60).stdout
    out["smi"] = [l.split(":", 1)[1].strip() for l in smi.splitlines() if "Unique ID" in l or "Performance Level" in l or "Max Graphics" in l]
except Exception as e:  # noqa: BLE001
    out["smi"] = repr(e)

import torch  # noqa: E402

a = torch.empty(1 << 30, dtype=torch.float32, device="cuda")  # 4 GiB
b = torch.empty_like(a)
a.fill_(1.0)
torch.cuda.synchronize()
with Sampler() as s:
    t0 = time.time()
    for _ in range(40):
        b.copy_(a)
    torch.cuda.synchronize()
    dt = time.time() - t0
out["stream_copy"] = {"GBps_read_plus_write": 40 * 2 * a.numel() * 4 / dt / 1e9, **s.summary()}
with Sampler() as s:
    t0 = time.time()
    for _ in range(40):
        x = a.sum()
    torch.cuda.synchronize()
    dt = time.time() - t0
out["stream_read"] = {"GBps": 40 * a.numel() * 4 / dt / 1e9, **s.summary()}
del a, b

pkg = load_package()
with Sampler() as s:
    r = {}
    for write, name in ((0, "read"), (1, "read+write")):
        ms = min(pkg._lib.sector_probe(4096, 16384, 1000, write) for _ in range(3))
        n = 4096 * 64 * 4 * 1000 * (1 if write == 0 else 2)
        r[name] = {"lines_per_s": n / (ms * 1e-3), "TBps_of_128B_lines": n * 128 / (ms * 1e-3) / 1e12}
out["random_lines"] = {**r, **s.summary()}

G = pkg.problems.gmrf_precision(128)
d = G.shape[0]
c = pkg.problems.column_norms(G)
for nch in (4096, 2048):
    ens = pkg.Ensemble(nch, d, trace_capacity=2 * d + 1024)
    ens.set_flow(pkg.ZigZag(G, np.zeros(d)))
    ens.set_target(pkg.GaussianTarget(G))
    ens.set_gradient_tracking(True)
    ens.set_state_synthetic(0.0, c, 0x5EED0000)
    ms = []
    with Sampler() as s:
        for k in range(12):
            ens.run(float(k + 1), pkg._lib.RUN_STOP_BEFORE, sync=True)
            ms.append(ens.last_run_ms())
            ens.trace_reset()
    out["C3_tracked_%d" % nch] = {"ms_per_step": float(np.mean(ms[2:])), "ms_all": [round(m, 2) for m in ms], **s.summary()}
    ens.close()
print(json.dumps(out))
