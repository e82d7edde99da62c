"""One rank's share of the pawn frame (tile shard 0 of N), a few frames, for rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = rt.World.example("pawn", 1920, 1080).commit()
per = rt.shard_tiles(1920, 1080, 0, n)
buf = torch.zeros(max(per * rt.TILE * rt.TILE * 3, 1920 * 1080 * 3), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for _ in range(6):
    w.render_device(d_u8=buf.data_ptr(), stream=stream, tile_first=0, tile_step=n)
torch.cuda.synchronize()
