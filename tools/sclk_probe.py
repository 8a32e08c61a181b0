#!/usr/bin/env python3
"""Which files tell the shader clock on this box, and what they read while the CVA kernel runs back to back (bench.py samples one of
them from a side thread during its exclusive launches: roofline.issue_frac_at_measured_clock)."""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for pat in ("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input", "/sys/class/drm/card*/device/pp_dpm_sclk"):
    for f in sorted(glob.glob(pat)):
        try:
            print(f, "->", open(f).read().strip().replace("\n", " | ")[:200])
        except OSError as e:
            print(f, "-> unreadable:", e)
import bench
import montecarlocuda_amd as mc
from montecarlocuda_amd import sclk
PCI = mc.pci_bus_id(0)
print("HIP device 0 is", PCI, "-> sclk source:", sclk.source(0, PCI))
eng = mc.Engine(0)
samples = []
stop = threading.Event()


def sampler():
    while not stop.is_set():
        samples.append((time.perf_counter(), sclk.read_mhz(0, PCI)))
        time.sleep(0.002)


for label, fn in (("idle", lambda: time.sleep(0.3)),
                  ("cva 1e7 f64 x 120", lambda: [eng.cva(bench.CVA, 10 ** 7, mc.MC_DEFAULT_SEED, 0, "f64") for _ in range(120)]),
                  ("vanilla 1e8 f32 x 12000", lambda: [eng.vanilla(bench.VAN, 10 ** 8, mc.MC_DEFAULT_SEED, 0, "f32") for _ in range(12000)])):
    samples.clear()
    stop.clear()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    v = [m for _, m in samples if m]
    half = v[len(v) // 2:]
    print(f"{label:24s} {dt*1e3:8.1f} ms, {len(v)} samples: first {v[:3]} ... second half mean {sum(half)/max(1,len(half)):.0f} MHz, min {min(v)}, max {max(v)}")
