"""Every benchmark configuration through bench.py (one GPU): writes the JSON lines to
profiles/<tag>_all_configs.jsonl and prints a markdown table.  usage: results_table.py <tag>"""
import json, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
runs = [  # (label, bench.py arguments[, environment]); the tree is the device-built one unless --bvh says otherwise
    ("cover400", ["--workload", "cover400"]),
    ("pawn1080", ["--workload", "pawn1080"]),
    ("pawn1080 host SAH tree", ["--workload", "pawn1080", "--bvh", "sah", "--no-cpu-baseline"]),
    ("pawn1080 shadow pass", ["--workload", "pawn1080", "--no-cpu-baseline"], {"RT_SHADOW_PASS": "1"}),
    ("pawn1080 f32 search", ["--workload", "pawn1080", "--mode", "f32_fast", "--no-cpu-baseline"]),
    ("teaset1080", ["--workload", "teaset1080"]),
    ("teaset1080 megakernel", ["--workload", "teaset1080", "--pipeline", "megakernel", "--no-cpu-baseline"]),
    ("pawn8k", ["--workload", "pawn8k", "--steps", "10", "--no-cpu-baseline"]),
    ("synthetic4k", ["--workload", "synthetic4k", "--steps", "10"]),
    ("synthetic4k host SAH tree", ["--workload", "synthetic4k", "--steps", "10", "--bvh", "sah", "--no-cpu-baseline"]),
]
out = open(os.path.join(REPO, "profiles", f"{tag}_all_configs.jsonl"), "w")
print("| config | ms/frame x4 | Mrays/s x4 | ms one frame | Mrays/s one frame | roof | frac | CPU Mrays/s (threads) |")
print("|---|---|---|---|---|---|---|---|")
for run in runs:
    name, extra = run[0], run[1]
    env = dict(os.environ, **(run[2] if len(run) > 2 else {}))
    r = subprocess.run([sys.executable, "-u", os.path.join(REPO, "bench.py"), "--sustain-s", "2"] + extra, capture_output=True,
                       text=True, timeout=600, env=env)
    if r.returncode != 0:
        print(f"| {name} | failed: {r.stderr[-200:]} |")
        continue
    d = json.loads(r.stdout.strip().splitlines()[-1])
    d["_run"] = name
    out.write(json.dumps(d) + "\n")
    out.flush()
    cb = d.get("cpu_baseline")
    cpu = f"{cb['threads_all']['value']:.3f} ({cb['threads_all']['threads']}), {cb['threads_6']['value']:.3f} (6)" if cb else "-"
    rf = d["roofline"]
    frac = "-" if rf["frac"] is None else f"{rf['frac']:.3f}"
    print(f"| {name} | {d['ms_per_step']:.3f} | {d['value']:.0f} | {d['single_frame']['ms']:.3f} | {d['single_frame']['Mrays/s']:.0f} | "
          f"{rf['bound']} | {frac} | {cpu} |", flush=True)
