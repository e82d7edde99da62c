import sys, os
sys.path.insert(0, os.getcwd())
import raytracer_amd as rt
w = rt.World.example("pawn", 1920, 1080).commit()
_, _, st = w.render(pipeline=rt.PIPELINE_WAVEFRONT, want_stats=True)
print(st.as_dict())
