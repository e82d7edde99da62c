"""Where a wave of k_wf_level spends its cycles (needs a build with -DRT_SECTION_TIMES:
bash scripts/build_variant.sh sect -DRT_SECTION_TIMES; RT_LIB=variants/sect.so python scripts/section_times.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
name, hs, vs = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("pawn", 1920, 1080)
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 5  # 1: the primary level alone
w = rt.World.example(name, hs, vs).commit()
w.render(depth=depth); w.render(depth=depth)
L = w.L
out = (C.c_ulonglong * 8)()
L.rt_debug_section_cycles(None, 1)
w.render(depth=depth)
torch.cuda.synchronize()
L.rt_debug_section_cycles(out, 0)
names = ["task load / ray generation", "closest-hit traversal",
         "containers, hit preparation, pattern, optics, child rays, light terms", "shadow traversal",
         "predecessor traversal", "surface sum, node write"]
tot = sum(out[:6])
for n, v in zip(names, out):
    print(f"{n:40s} {100.0 * v / tot:5.1f} %  {v / 1e6:10.1f} Mcycles")
print(f"shadow traversal in waves with fewer than 16 / 32 hits of 64: {100.0 * out[6] / tot:.1f} % / {100.0 * out[7] / tot:.1f} % of all cycles")
