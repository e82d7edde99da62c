"""bench.py — Mrays/s of the per-pixel render path on N MI355X.

A step is one full frame of the workload (default: the configuration the metric is
quoted on, the glass pawn at 1920x1080, depth 5, f64 parity mode) with the committed
scene already resident in HBM.  With N > 1 the frame's 32x32-pixel tiles are dealt
round-robin to the ranks (one process per GPU), every rank renders its tiles, and one RCCL
gather + a de-tile kernel on rank 0 assemble the image inside the timed step ("strong"
scaling: the frame is fixed, the GPUs share it).

Prints one JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

WORKLOADS = {
    # name: (example, hsize, vsize, synthetic_tris, description)
    "cover400": ("cover", 400, 400, 0, "cover 400x400 (19 primitives, 2 lights)"),
    "pawn1080": ("pawn", 1920, 1080, 0, "pawn 1920x1080 (31776 smooth triangles + plane, glass)"),
    "teaset1080": ("tea set", 1920, 1080, 0, "tea set 1920x1080 (23076 smooth triangles in 3 instances + plane)"),
    "pawn8k": ("pawn", 7680, 4320, 0, "pawn 7680x4320"),
    "synthetic4k": ("synthetic", 3840, 2160, 1000000, "synthetic 1M flat triangles + 64 reflective spheres 3840x2160"),
}

# MI355X peaks: HBM3E 8 TB/s; FP64 vector 78.6 TFLOP/s (half the 157.3 TFLOP/s FP32 vector peak),
# which counts a fused multiply-add as two flops.  The parity arithmetic may not fuse (every
# product and sum is rounded like the reference's), so one flop per lane and cycle, 39.3
# TFLOP/s, is the most this code can reach.
PEAK_HBM_GBS = 8000.0
PEAK_F64_TFLOPS = 78.6
PEAK_F64_NO_FMA_TFLOPS = 39.3


def algorithmic_work(st, node_bytes):
    """Per-frame algorithmic bytes and flops from the traversal counters:
    bytes = node_bytes*nodes + 72*tri_tests + 72*shaded mesh hits (normals);
    flops = 18*children*nodes (one 18-flop box test per child of a node: 16 bytes of a node
            hold one child) + 46*tri_tests + 56*instance entries + 30*primitive tests
            + 200*shade calls."""
    b = node_bytes * st["nodes_visited"] + 72 * st["tri_tests"] + 72 * st["shade_calls"]
    f = (18 * (node_bytes // 16) * st["nodes_visited"] + 46 * st["tri_tests"] + 56 * st["inst_entries"] +
         30 * st["prim_tests"] + 200 * st["shade_calls"])
    return b, f


def host_cpu_info():
    """(physical cores of the host from /proc/cpuinfo, threads this process may use).  The GPU
    boxes of the pool give a one-GPU job a share of the host's cores (exported as
    OMP_NUM_THREADS); os.cpu_count() shows the whole machine."""
    phys = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    physical = len(phys) or (os.cpu_count() or 1)
    share = physical
    try:
        share = min(share, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for var in ("RT_CPU_THREADS", "OMP_NUM_THREADS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            share = min(share, int(v)) if var == "OMP_NUM_THREADS" else int(v)
            break
    return physical, max(1, share)


def cpu_baseline(example, hs, vs, ntris, pixels, target_s=7.0):
    """The CPU oracle (brute force, as the reference: every ray against every triangle; the
    sample's pixels are dealt to std::threads in runs of 16) on a strided sample of the frame's
    pixels, timed twice: on every core this job may use (count stated), and on 6 threads, the
    reference's default (main.rs:24).  A short calibration run sizes each sample so that its
    timed run takes about `target_s` seconds."""
    from oracle import oracle as O
    physical, share = host_cpu_info()
    sc = O.Scene.example(example, hs, vs, synthetic_tris=ntris)
    if ntris:
        sc.use_bvh()  # brute force over 1M triangles is not feasible; say so in `sample`
    n = hs * vs

    def run(count, threads):
        idx = (np.arange(count, dtype=np.int64) * n // count + (n // count) // 2) % n
        xs, ys = (idx % hs).astype(np.int32), (idx // hs).astype(np.int32)
        t0 = time.time()
        _, _, cnt = sc.render_pixels(xs, ys, threads=threads)
        dt = time.time() - t0
        return cnt["primary"] + cnt["shadow"] + cnt["reflect"] + cnt["refract"], dt

    def timed(threads):
        px = max(64, pixels * threads // 16)
        rays, dt = run(px, threads)
        if dt < target_s / 2:
            px = int(min(n, px * min(64.0, target_s / max(dt, 1e-3))))
            rays, dt = run(px, threads)
        return {"value": rays / dt / 1e6, "threads": threads, "pixels": px, "rays": rays, "seconds": dt}

    a, six = timed(share), timed(min(6, share))
    how = ("oracle culling mode (brute force infeasible)" if ntris else
           "brute force over all triangles like the reference")
    return {"value": a["value"], "unit": "Mrays/s", "cores": share, "kind": "port",
            "host_physical_cores": physical,
            "threads_all": a, "threads_6": six,
            "sample": f"{a['pixels']} pixels on a uniform stride over the {hs}x{vs} frame, {a['rays']} rays, "
                      f"{a['seconds']:.2f} s wall on {share} threads (all the cores this job may use; the host "
                      f"has {physical} physical cores); {six['pixels']} pixels, {six['rays']} rays, "
                      f"{six['seconds']:.2f} s on {six['threads']} threads (the reference's default); " + how}


def measured_traffic(workload, pipeline):
    """HBM bytes per frame of the render kernels from the committed rocprofv3 PMC passes of
    this same command (FETCH_SIZE doubled as gfx950 requires, plus WRITE_SIZE), the per-frame
    sums of the other counters of those passes (or {}), and the file's name; (None, {}, None)
    without a committed profile."""
    for rnd in ("r03", "r02", "r01"):
        path = os.path.join(REPO, "profiles", f"{rnd}_{pipeline}_{workload}_pmc.json")
        try:
            with open(path) as f:
                per_frame = json.load(f)["render_kernels_per_frame"]
            return float(per_frame["hbm_bytes"]), per_frame.get("counters", {}), os.path.basename(path)
        except (OSError, KeyError, ValueError):
            continue
    return None, {}, None


SIMDS = 1024          # 256 CUs x 4
PEAK_CLOCK_HZ = 2.4e9  # MI355X peak engine clock


def pmc_derived(c, kernel_ms):
    """Utilisation figures from the committed PMC passes (per-frame counter sums `c`) and the
    frame time measured live.  SQ_INSTS_VALU counts wave instructions, each of which holds a SIMD's
    vector unit for 4 cycles: valu_issue_util = SQ_INSTS_VALU x 4 / (1024 SIMDs x clock x time).
    SQ_WAVE_CYCLES / SQ_WAIT_ANY count quad-cycles per resident wave.  The clock is the peak
    engine clock (the chip runs below it under load, so the true utilisation is a little higher)."""
    if not c:
        return None
    slots = SIMDS * PEAK_CLOCK_HZ * kernel_ms * 1e-3  # SIMD cycles in one frame's time
    out = {"source": "committed PMC passes (profiles/), per frame; time = this run's kernel_ms"}
    if "SQ_INSTS_VALU" in c:
        out["valu_insts_per_frame"] = c["SQ_INSTS_VALU"]
        out["valu_issue_util"] = c["SQ_INSTS_VALU"] * 4 / slots
    if "SQ_INSTS_SALU" in c:
        out["salu_insts_per_frame"] = c["SQ_INSTS_SALU"]
    if "SQ_WAVE_CYCLES" in c:
        out["waves_per_simd"] = c["SQ_WAVE_CYCLES"] * 4 / slots
        if "SQ_WAIT_ANY" in c:
            out["wave_wait_frac"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        out["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pawn1080", choices=sorted(WORKLOADS))
    ap.add_argument("--pipeline", default="wavefront", choices=["megakernel", "wavefront", "auto"])
    ap.add_argument("--mode", default="f64_parity", choices=["f64_parity", "f32_fast"],
                    help="f32_fast: triangle search in f32, everything else f64; reported with its pixel-match rate")
    ap.add_argument("--inflight", type=int, default=0,
                    help="frames in flight: step k is issued on stream k %% inflight with its own buffers, so the "
                         "tail of one frame's recursion level runs under the next frame (1 = strictly one at a time; "
                         "0 = 4 on one GPU, 8 per rank on several: a rank's share of the frame fills its GPU only "
                         "together with its neighbours)")
    ap.add_argument("--hw-queues", type=int, default=-1,
                    help="GPU_MAX_HW_QUEUES for this process (HIP's default is 4: more frames in flight than "
                         "hardware queues share queues and serialise); 0 leaves the environment as it is; "
                         "-1 = 8 on one GPU, 12 per rank on several (the exchange has a stream of its own)")
    ap.add_argument("--bvh", default="lbvh", choices=["sah", "lbvh", "auto"],
                    help="device-built tree (the default everywhere), host binned-SAH build, or by mesh size "
                         "(device from 100k triangles on)")
    ap.add_argument("--sustain-s", type=float, default=6.0,
                    help="after the timed steps, the same loop again for at least this long (0 = skip); long "
                         "enough by default for a coarse GPU-activity sampler to land inside it")
    ap.add_argument("--cpu-pixels", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.inflight <= 0:
        args.inflight = 4 if args.gpus == 1 else 8
    if args.hw_queues < 0:
        args.hw_queues = 8 if args.gpus == 1 else 12
    if args.hw_queues > 0:  # read by the HIP runtime when it starts: set before anything touches the GPU
        os.environ["GPU_MAX_HW_QUEUES"] = str(min(args.hw_queues, 32))

    import torch
    import raytracer_amd as rt
    from raytracer_amd import parallel

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dist = None
    # rehearsal of the N-rank code path on a one-GPU box: every rank on cuda:0, gloo with host
    # staging instead of RCCL (numbers from it mean nothing)
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL_ONE_GPU") == "1"
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world_size > 1 else 0)

    def all_reduce(t, op=None):
        kw = {} if op is None else {"op": op}
        if rehearsal:
            h = t.cpu()
            dist.all_reduce(h, **kw)
            t.copy_(h)
        else:
            dist.all_reduce(t, **kw)
        return t

    example, hs, vs, ntris, desc = WORKLOADS[args.workload]
    pipeline = {"megakernel": rt.PIPELINE_MEGAKERNEL, "wavefront": rt.PIPELINE_WAVEFRONT,
                "auto": rt.PIPELINE_AUTO}[args.pipeline]
    mode = rt.MODE_F32_FAST if args.mode == "f32_fast" else rt.MODE_F64_PARITY
    w = rt.World.example(example, hs, vs, synthetic_tris=ntris).commit(device=dev.index, bvh_builder={"sah": 0, "lbvh": 1, "auto": 2}[args.bvh])
    info = w.info()
    # a process's first commit also loads the library's code objects; the same scene committed again
    # costs the build and the upload alone
    w_again = rt.World.example(example, hs, vs, synthetic_tris=ntris).commit(
        device=dev.index, bvh_builder={"sah": 0, "lbvh": 1, "auto": 2}[args.bvh])
    commit_ms_warm = w_again.info()["commit_ms"]
    del w_again
    F = max(1, args.inflight)
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)]

    n = world_size
    per = max(rt.shard_tiles(hs, vs, s, n) for s in range(n))
    images = [torch.zeros((vs, hs, 3), dtype=torch.uint8, device=dev) for _ in range(F)]
    tiles = [torch.zeros(per * rt.TILE * rt.TILE * 3, dtype=torch.uint8, device=dev) for _ in range(F)] if n > 1 else None
    # rank 0 receives every shard straight into its row of one contiguous buffer
    all_tiles = [torch.zeros((n, per * rt.TILE * rt.TILE * 3), dtype=torch.uint8, device=dev) for _ in range(F)] \
        if (n > 1 and rank == 0) else None
    image = images[0]

    def exchange(slot):
        """The one exchange step: shards -> rank 0 (RCCL gather), de-tile kernel there."""
        stream = torch.cuda.current_stream().cuda_stream
        if rehearsal:
            h = parallel.gather_tiles(tiles[slot].cpu(), rank, n, dist)
            if rank == 0:
                all_tiles[slot].copy_(torch.stack(h))
        else:
            parallel.gather_tiles(tiles[slot], rank, n, dist,
                                  out=list(all_tiles[slot].unbind(0)) if rank == 0 else None)
        if rank == 0:
            rt.detile_u8(all_tiles[slot].data_ptr(), n, per, hs, vs, images[slot].data_ptr(), stream)

    frames_issued = [0]

    def step(k=0, want_stats=False):
        """Frame k: rendered (and, for n > 1, exchanged) on stream k % F with that slot's buffers."""
        frames_issued[0] += 1
        slot = k % F
        with torch.cuda.stream(streams[slot]):
            stream = streams[slot].cuda_stream
            if n == 1:
                w.render_device(d_u8=images[slot].data_ptr(), stream=stream, pipeline=pipeline, mode=mode,
                                want_stats=want_stats)
            else:
                w.render_device(d_u8=tiles[slot].data_ptr(), stream=stream, pipeline=pipeline, mode=mode,
                                tile_first=rank, tile_step=n, want_stats=want_stats)
                exchange(slot)

    # counters of one frame (outside the timed region; deterministic per scene)
    step(0, want_stats=True)
    torch.cuda.synchronize()
    st = w.stats().as_dict()
    keys = ["rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "depth0_skipped", "pred_queries",
            "nodes_visited", "tri_tests", "prim_tests", "inst_entries", "shade_calls", "node_wave_steps",
            "tri_wave_steps"]
    if n > 1:
        t = all_reduce(torch.tensor([st[k] for k in keys], dtype=torch.int64, device=dev))
        st.update({k: int(v) for k, v in zip(keys, t.tolist())})
    rays_per_frame = st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"] + st["rays_refract"]

    for k in range(max(args.warmup, F)):  # at least one frame per slot, so every slot's queues are sized
        step(k)
    torch.cuda.synchronize()
    # one frame at a time (not part of the timed region): mean over a few frames issued and
    # waited for one by one — what a host that wants ONE image gets
    single_n = 8
    for _ in range(6):  # after four frames on one stream the library splits frames; the first split frame is planned
        step(0)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(single_n):
        step(0)
        torch.cuda.synchronize()
    frame_latency_ms = (time.perf_counter() - t0) * 1e3 / single_n
    if n > 1:
        dist.barrier()
    torch.cuda.synchronize()

    def timed(steps):
        """`steps` frames, F in flight; returns (wall seconds, HIP-event ms per frame).  HIP events
        on the render streams bracket the frames (the frames of neighbouring slots overlap, so
        the span is taken over all streams and divided by the frame count)."""
        ev_start = torch.cuda.Event(enable_timing=True)
        ev_end = [torch.cuda.Event(enable_timing=True) for _ in range(F)]
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev_start.record(streams[0])
        for k in range(steps):
            step(k)
        for slot in range(F):
            ev_end[slot].record(streams[slot])
        torch.cuda.synchronize()
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt, max(ev_start.elapsed_time(e) for e in ev_end) / steps

    elapsed, kernel_ms = timed(args.steps)
    if n > 1:
        t = all_reduce(torch.tensor([elapsed, kernel_ms, frame_latency_ms], dtype=torch.float64, device=dev),
                       dist.ReduceOp.MAX)
        elapsed, kernel_ms, frame_latency_ms = t.tolist()
    # the same loop sustained for some seconds (the K timed steps above last ~50 ms).  Its length comes
    # from the time just measured — the maximum over the ranks, so that every rank issues the same
    # number of frames (each frame is one exchange: ranks that disagreed would wait for one another
    # for ever)
    sustained_steps = max(args.steps, int(math.ceil(args.sustain_s / max(elapsed / args.steps, 1e-6))))
    sustained_s, sustained_kernel_ms = timed(sustained_steps) if args.sustain_s > 0 else (elapsed, kernel_ms)
    if n > 1:
        t = all_reduce(torch.tensor([sustained_s, sustained_kernel_ms], dtype=torch.float64, device=dev), dist.ReduceOp.MAX)
        sustained_s, sustained_kernel_ms = t.tolist()

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = rays_per_frame / (ms_per_step * 1e3)
        import zlib
        frame_crc = zlib.crc32(image.cpu().numpy().tobytes())  # same for every N: the frame is fixed
        abytes, aflops = algorithmic_work(st, info["node_bytes"])
        # per launch: this rank's share of the frame (1/n of the counters for n > 1)
        ach_gbs = abytes / n / (kernel_ms * 1e-3) / 1e9
        ach_tf = aflops / n / (kernel_ms * 1e-3) / 1e12
        # the committed PMC passes are of the default build and settings: a run with another tree, mode
        # or tuning knob gets no counters from them
        tuned = [k for k in os.environ if k.startswith("RT_") and k not in ("RT_CPU_THREADS", "RT_BENCH_REHEARSAL_ONE_GPU")]
        as_profiled = n == 1 and args.mode == "f64_parity" and args.bvh == "lbvh" and not tuned
        traffic, pmc, traffic_src = measured_traffic(args.workload, args.pipeline) if as_profiled else (None, {}, None)
        # what the frame has to move whatever the kernels do: the scene once, the image, every
        # queued ray written and read once (64 B), every shaded call's result node written and
        # read once (96 B) and its flag byte
        npx = hs * vs
        traffic_needed = (info["device_bytes"] + 3 * npx + 128 * (st["rays_reflect"] + st["rays_refract"]) +
                          194 * st["shade_calls"]) / n
        lanes = {"node_lanes_per_step": st["nodes_visited"] / max(1, st["node_wave_steps"]),
                 "tri_lanes_per_step": st["tri_tests"] / max(1, st["tri_wave_steps"]),
                 "note": "of 64; from the STATS counters of one frame (visits / wave iterations)"}
        util = pmc_derived(pmc, kernel_ms)
        kernel = ("k_render_mega" if args.pipeline == "megakernel" else
                  "k_wf_level (one launch per recursion level; the frame's launches taken together)")
        # Which roof binds: the example scenes (a few MB) are served from L2 / Infinity Cache, so
        # their roof is the FP64 vector ALU; the 1M-triangle scene is the HBM case, and its
        # achieved bandwidth is the MEASURED traffic of the committed PMC passes over the frame's
        # time (algorithmic bytes, mostly cache hits, say nothing about HBM).
        valu = {"bound": "valu_f64", "achieved": ach_tf, "peak": PEAK_F64_TFLOPS, "unit": "TFLOP/s",
                "frac": ach_tf / PEAK_F64_TFLOPS,
                "peak_no_fma": PEAK_F64_NO_FMA_TFLOPS, "frac_no_fma": ach_tf / PEAK_F64_NO_FMA_TFLOPS,
                "traffic": traffic, "traffic_source": traffic_src, "traffic_needed": traffic_needed,
                "traffic_ratio": None if traffic is None else traffic / traffic_needed,
                "utilisation": util, "lanes": lanes,
                "kernel": kernel, "kernel_ms": kernel_ms,
                "algorithmic_flops_per_launch": aflops / n,
                "single_frame_frac": (aflops / n / (frame_latency_ms * 1e-3) / 1e12) / PEAK_F64_TFLOPS,
                "note": "flops = 18 per child box test (4 per node) + 46 per triangle test + 56 per ray-into-group "
                        "transform + 30 per closed-form shape test + 200 per shade call, from the STATS counters of "
                        "one frame; time = HIP-event span per frame over the timed steps (frames in flight overlap); "
                        "peak counts an FMA as 2 flops, the parity arithmetic may not fuse: peak_no_fma is its ceiling"}
        hbm_measured = None if traffic is None else traffic / (kernel_ms * 1e-3) / 1e9
        hbm = {"bound": "hbm", "achieved": hbm_measured, "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": None if hbm_measured is None else hbm_measured / PEAK_HBM_GBS,
               "traffic": traffic, "traffic_source": traffic_src, "traffic_needed": traffic_needed,
               "traffic_ratio": None if traffic is None else traffic / traffic_needed,
               "utilisation": util, "lanes": lanes,
               "kernel": kernel, "kernel_ms": kernel_ms,
               "algorithmic_bytes_per_launch": abytes / n,
               "algorithmic_GBps": ach_gbs,
               "note": "achieved = measured HBM bytes per frame (rocprofv3 FETCH_SIZE x2 as gfx950 requires + WRITE_SIZE, "
                       "profiles/) / frame time; algorithmic bytes = node_bytes*nodes_visited + 72*tri_tests + "
                       "72*shade_calls are mostly cache hits and are given for reference only"}
        hbm_bound = args.workload == "synthetic4k"
        single_ms = frame_latency_ms
        out = {
            "metric": "Mrays/s (primary+shadow+reflection+refraction), pawn @1080p" if args.workload == "pawn1080"
                      else f"Mrays/s (primary+shadow+reflection+refraction), {args.workload}",
            "value": value, "unit": "Mrays/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if args.mode == "f64_parity" else "f32 triangle search + f64 shading",
            "data": ("synthetic (seeded generator, 1M triangles)" if ntris else
                     "the reference's own example scene (examples.rs; OBJ assets of the reference repository under "
                     "assets/obj), camera size as the benchmark config states; no external dataset"),
            "config": {"workload": desc, "hsize": hs, "vsize": vs, "depth": 5, "mode": args.mode,
                       "pipeline": args.pipeline, "frames_in_flight": F, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "frame_latency_ms": frame_latency_ms, "bvh": args.bvh, "commit_ms": info["commit_ms"],
                       "commit_ms_build_and_upload": commit_ms_warm,
                       "commit_ms_code_load": max(0.0, info["commit_ms"] - commit_ms_warm), "rays_per_frame": rays_per_frame,
                       "parallelism": f"tiles{rt.TILE}x{rt.TILE} round-robin over {n} GPU(s)" + (", RCCL gather to rank 0" if n > 1 else ""),
                       "triangles": info["n_triangles"], "bvh_nodes": info["n_bvh_nodes"],
                       "scene_bytes": info["device_bytes"], "frame_crc32": frame_crc},
            "single_frame": {"ms": single_ms, "Mrays/s": rays_per_frame / (single_ms * 1e3),
                             "note": "one frame issued and waited for at a time (mean of 8), same entry point"},
            "sustained": {"steps": sustained_steps, "seconds": sustained_s,
                          "ms_per_step": sustained_s * 1e3 / sustained_steps,
                          "value": rays_per_frame / (sustained_s * 1e3 / sustained_steps * 1e3),
                          "kernel_ms": sustained_kernel_ms},
            "roofline": hbm if hbm_bound else valu,
            ("roofline_valu_f64" if hbm_bound else "roofline_hbm"): valu if hbm_bound else hbm,
            "counters_per_frame": {k: st[k] for k in keys},
            "frames_issued": frames_issued[0],  # by this process, all told (a profile of the run divides by it)
        }
        if args.mode == "f32_fast" and n == 1:
            # the fast mode is reported with its pixel-match rate against the parity mode's frame
            fast = image.cpu().numpy().astype(np.int16)
            w.render_device(d_u8=image.data_ptr(), stream=streams[0].cuda_stream, pipeline=pipeline, mode=rt.MODE_F64_PARITY)
            torch.cuda.synchronize()
            d = np.abs(fast - image.cpu().numpy().astype(np.int16)).max(axis=-1)
            out["pixel_match_vs_f64_parity"] = {"exact": float((d == 0).mean()), "within_1": float((d <= 1).mean()),
                                                "worst": int(d.max()), "pixels": int(d.size)}
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(example, hs, vs, ntris, args.cpu_pixels)
        print(json.dumps(out), flush=True)
    if n > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
