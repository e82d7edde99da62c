import sys, os
sys.path.insert(0, os.getcwd())
import raytracer_amd as rt
w = rt.World.example("pawn", 64, 36).commit()
w.render(pipeline=rt.PIPELINE_WAVEFRONT)
import torch; torch.cuda.synchronize()
