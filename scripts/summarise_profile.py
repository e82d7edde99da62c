"""Condense gpurun_out/prof_<tag> (scripts/profile_gpu.sh) into the files kept under profiles/:
  <out>_kernel_stats.csv  rocprofv3 --kernel-trace --stats summary, verbatim
  <out>_pmc.json          per-kernel means of every PMC pass + per-frame HBM traffic
  <out>_bench.json        the bench line printed under the stats pass
usage: python scripts/summarise_profile.py <tag> <out-prefix> [frames-per-run]
(frames per run: taken from the bench line's "frames_issued" when not given)"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, out = sys.argv[1], sys.argv[2]
root = f"gpurun_out/prof_{tag}"
frames = int(sys.argv[3]) if len(sys.argv) > 3 else int(json.load(open(f"{root}/stats_bench.json"))["frames_issued"])
# gpurun merges every call's output into the same directories: of each pass, the newest file counts
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)
shutil.copy(newest(f"{root}/stats/*/*kernel_stats.csv"), f"{out}_kernel_stats.csv")
shutil.copy(f"{root}/stats_bench.json", f"{out}_bench.json")
per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(f"{root}/pmc_*/")):
    f = newest(f"{d}*/*_counter_collection.csv")
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        per_kernel[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summary = {}
fetch_kb = write_kb = 0.0
sq = collections.defaultdict(float)  # SQ / TCC / GRBM counters summed over the render kernels' launches
for k, cs in sorted(per_kernel.items()):
    summary[k] = {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v), "sum": sum(v)} for c, v in sorted(cs.items())}
    if k.startswith(("void k_wf", "void k_render", "k_wf", "k_render")):
        fetch_kb += sum(cs.get("FETCH_SIZE", []))
        write_kb += sum(cs.get("WRITE_SIZE", []))
        for c, v in cs.items():
            if c.startswith(("SQ_", "TCC_", "TCP_", "GRBM_")):
                sq[c] += sum(v)
doc = {
    "command": "rocprofv3 --pmc <one pass per group> -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline",
    "frames_per_run": frames,
    "units": {"FETCH_SIZE": "KB as reported; gfx950 tallies 128-B requests at 64 B, so bytes = 2 * 1024 * value",
              "WRITE_SIZE": "KB, exact"},
    "render_kernels_per_frame": {
        "fetch_bytes_corrected": 2 * 1024 * fetch_kb / frames,
        "write_bytes": 1024 * write_kb / frames,
        "hbm_bytes": (2 * 1024 * fetch_kb + 1024 * write_kb) / frames,
        # wave-level instruction counts; SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_ACTIVE_INST_* count
        # quad-cycles (x4 = shader cycles); GRBM_GUI_ACTIVE is summed over the 8 XCDs
        "counters": {c: v / frames for c, v in sorted(sq.items())},
    },
    "kernels": summary,
}
json.dump(doc, open(f"{out}_pmc.json", "w"), indent=1)
print(json.dumps(doc["render_kernels_per_frame"]))
