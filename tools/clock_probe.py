#!/usr/bin/env python3
"""Long single launches (>= 10 ms) of each kernel, to be run under `rocprofv3 --pmc GRBM_GUI_ACTIVE ...`:
effective shader clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md 'DVFS give-back').

    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
        --output-format csv -d gpurun_out/clock -- python3 tools/clock_probe.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import montecarlocuda_amd as mc  # noqa: E402

which = sys.argv[1:] or ["vanilla_f32", "vanilla_f64", "vanilla_f64_n32", "basket4_f32", "basket16_f32", "basket16_f64", "basket16_f64_n32",
                         "cva256_f64", "cva256_f64_n32", "cva256_f32"]
SIZES = {"vanilla_f32": 2 * 10 ** 10, "vanilla_f64": 4 * 10 ** 9, "vanilla_f64_n32": 8 * 10 ** 9, "basket4_f32": 5 * 10 ** 9,
         "basket16_f64": 3 * 10 ** 8, "basket16_f64_n32": 5 * 10 ** 8, "basket16_f32": 10 ** 9,
         "cva256_f64": 10 ** 7, "cva256_f64_n32": 14 * 10 ** 6, "cva256_f32": 4 * 10 ** 7}
W = bench.workloads(mc)
for name in which:
    prod, X, inputs, _, _, _ = W[name]
    if callable(inputs):
        inputs = inputs()
    eng = mc.Engine(0)
    if "normals" in bench.workload_settings(name):
        eng.set_normals(bench.workload_settings(name)["normals"])
    for rep in range(3):
        e = getattr(eng, prod)(inputs, SIZES[name], mc.MC_DEFAULT_SEED, 0, X)
    eng.close()
    print(f"{name:14s} n={SIZES[name]:.3g} kernel_ms={e.kernel_ms:.3f} rate={SIZES[name]/e.kernel_ms*1e3:.4g}/s value={e.expected:.6f}")
