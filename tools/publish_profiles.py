#!/usr/bin/env python3
"""gpurun_out/final/ (tools/collect_profiles.sh on the GPU box) -> profiles/rNN_* (tracked).
usage: python tools/publish_profiles.py [round, default 02]"""
import csv
import glob
import json
import os
import shutil
import sys

R = "r" + (sys.argv[1] if len(sys.argv) > 1 else "02")
F, P = "gpurun_out/final", "profiles"
FB = "gpurun_out/finalb" if os.path.isdir("gpurun_out/finalb") else F      # round 4: the collection runs as two gpurun calls
                                                                         # (tools/collect_profiles_a.sh / _b.sh)
cp = lambda a, b: shutil.copy(a, os.path.join(P, R + "_" + b))
clean = lambda path: "".join(l for l in open(path) if "amdgpu.ids" not in l and "run_backward" not in l)
cp(F + "/bench_C3.json", "bench_C3.json")
cp(F + "/bench_C3_kernel_summary.md", "bench_C3_kernel_summary.md")
cp(glob.glob(F + "/prof/*/*kernel_stats.csv")[0], "bench_C3_kernel_stats.csv")
cp(F + "/bench_C3_profiled_run.json", "bench_C3_profiled_run.json")
cp(F + "/single_stream_kernel_summary.md", "single_stream_kernel_summary.md")
cp(F + "/single_stream_step_timeline.txt", "single_stream_step_timeline.txt")
cp(F + "/C5_step_timeline.txt", "C5_step_timeline.txt")
cp(F + "/gemm_variants_kernel_summary.md", "gemm_variants_isolated_rocprof_summary.md")
cp(glob.glob(F + "/rk/*/*kernel_stats.csv")[0], "gemm_variants_isolated_kernel_stats.csv")
cp(FB + "/pmc_gemm.json", "pmc_gemm.json")
for name in ("instep_clock.md", "tile_sweep.txt", "C2_step_timeline.txt", "C1_step_timeline.txt", "cond_iteration_timeline.txt"):
    if os.path.exists(F + "/" + name):
        cp(F + "/" + name, name)
if os.path.exists(F + "/gemm_stamps.txt"):
    open(os.path.join(P, R + "_gemm_stamps.txt"), "w").write(clean(F + "/gemm_stamps.txt"))
if os.path.exists(F + "/gemm_instep_vs_isolated.txt"):
    open(os.path.join(P, R + "_gemm_instep_vs_isolated.txt"), "w").write(clean(F + "/gemm_instep_vs_isolated.txt"))
with open(os.path.join(P, R + "_gemm_variants.txt"), "w") as o:
    for name, title in (("gemm_variants.txt", "M = 512, N' = 16384 (C3 top layer: the 2048 rows of the top fidelity x 8 samples)"),
                        ("gemm_variants_M1024.txt", "M = 1024, N' = 16384 (C5 top layer)"),
                        ("gemm_variants_layer0.txt", "M = 512, N' = 8192 (C3 layer 0)"),
                        ("gemm_variants_reflayout.txt", "M = 512, N' = 65536 (C3 top layer in the reference's layout: every row)")):
        o.write("# tools/gemm_variants.py -- %s; HIP events, 20 launches per timing, three interleaved rounds after 0.3 s of load\n" % title)
        o.write(clean(F + "/" + name) + "\n")
with open(os.path.join(P, R + "_other_measurements.txt"), "w") as o:
    o.write("# bench.py on the other configurations (same harness; steps/s, whole job)\n" + open(FB + "/other_configs.txt").read() + "\n")
    o.write("# tools/size_sweep.py -- one surrogate, M = N, S = 1, two fidelities, rows shuffled (general branch)\n" + clean(FB + "/size_sweep.txt") + "\n")
    o.write("# tools/acq_bench.py 50\n" + clean(FB + "/acq_bench.txt") + "\n")
    o.write("# tools/cond_bench.py 400\n" + clean(FB + "/cond_bench.txt") + "\n")
    o.write("# examples/example_acquisition_mfdgp_forrester.py (the reference's walk-through at its own schedule)\n" + clean(FB + "/forrester_walkthrough.txt") + "\n")
    if os.path.exists(F + "/mfma_peak.txt"):
        o.write("# tools/mfma_peak (pure instruction streams, no memory traffic)\n" + open(F + "/mfma_peak.txt").read())
# SQ counters of the dominant kernel
acc = {}
for r in csv.DictReader(open(glob.glob(FB + "/pmc_sq/*/*counter_collection.csv")[0])):
    if "gemm_f64" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
     for r in csv.DictReader(open(glob.glob(FB + "/pmc_sq/*/*kernel_trace.csv")[0])) if "gemm_f64" in r["Kernel_Name"]]
dur = sum(d) / len(d)
mf = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc["SQ_VALU_MFMA_BUSY_CYCLES"])
clk = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8 / dur / 1e3
ms = json.load(open(F + "/bench_C3.json"))["roofline"]["kernel_ms"]
shape = json.load(open(FB + "/pmc_gemm.json"))["shape"]
alg = shape[0] * shape[0] * shape[1] / 1e9
open(os.path.join(P, R + "_pmc_sq_counters.md"), "w").write(("# SQ counters of the dominant kernel (A = L^-1 K_mn with the "
    "column-statistics epilogue, %d x %d x %d)\n" % tuple(shape)) + """

command: `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 tools/pmc_gemm.py` (5 cold launches in a fresh process)

| quantity | value |
|---|---|
| duration under the profiler (cold clock) | %.1f us |
| SQ_VALU_MFMA_BUSY_CYCLES, summed over the 1024 SIMDs | %.3e  (= %.2e per SIMD; x 32 flops = %.2f GFLOP executed: M^2 N' = %.2f algorithmic + the dense part of the diagonal blocks) |
| effective clock (GRBM_GUI_ACTIVE / 8 / duration) | %.2f GHz |
| MFMA pipe busy / launch duration at that clock | %.2f |
| the same busy cycles over the settled-clock duration (%.3f ms, bench) at 2.4 GHz | %.2f |
""" % (dur, mf, mf / 1024, mf * 32 / 1e9, alg, clk, mf / 1024 / (dur * 1e-6 * clk * 1e9), ms, mf / 1024 / (ms * 1e-3 * 2.4e9)))
print("published to", P, "as", R + "_*")
