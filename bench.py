#!/usr/bin/env python3
"""Headline benchmark: frames/s (+ raster-kernel HBM GB/s) on BASELINE.json configs[2]:
1 M splats (lilly_boquet-class synthetic PLY), SH degree 3, 1920x1080, 8-view orbit.

  python bench.py --gpus 1 --steps 64 --warmup 8
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one frame of the renderer: clip data -> depth sort of the splats -> tile list -> tile sort -> tile ranges ->
raster with the SH colours evaluated on demand (the work of src/Demo.cpp:394-479; same image bytes as the reference's stage
order, tests/test_gpu_parity.py), inputs resident in HBM, plus -- for N > 1 -- the framebuffer gather.  With N GPUs
the screen is split into N bands of tile rows (splat buffers replicated); every rank renders its band and
rank 0 receives the bands over RCCL directly into its framebuffer.  Total work is fixed -> "strong" scaling.

Rank 0 prints ONE JSON line (see README/DESIGN.md for the field meanings).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(n, p, pc, tiles, nonempty_tiles, w, h, sh_degree, sort_passes, colours_on_demand=False):
    """bytes each stage must move per frame, independent of implementation.
    With colours on demand there is no SH pass; a consumed list entry then costs the raster its position (16 B) and
    coefficient row (12 (L+1)^2 B) instead of the 16-byte colour: 44 - 16 + 16 + 12 (L+1)^2 bytes per consumed pair."""
    sh_row = 12 * (sh_degree + 1) ** 2
    sh = n * (16 + sh_row + 16) if sh_degree and not colours_on_demand else 0
    per_pair = 44 + (sh_row if colours_on_demand else 0)
    return {
        "sh": sh,
        "clip": 76 * n,
        "tile_list": 28 * n + 12 * p,
        "sort": p * (8 + 24 * sort_passes),
        "ranges": 8 * p + 8 * nonempty_tiles,
        "raster": per_pair * pc + 8 * tiles + 4 * w * h,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--splats", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--opacity-mean", type=float, default=1.0, help="raw (pre-sigmoid) opacity mean of the synthetic scene")
    ap.add_argument("--cpu-frames", type=int, default=4, help="frames of the CPU oracle timed as cpu_baseline (0 = skip)")
    ap.add_argument("--ppl", type=int, default=0, help="raster pixels per lane override (1,2,4)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel individually instead of replaying a captured frame")
    ap.add_argument("--even-bands", action="store_true", help="multi-GPU: equal-height bands instead of cost-balanced ones")
    ap.add_argument("--pipeline", type=int, default=1, help="1 two-level sort (default, fastest at C3), 2 binned tile lists (no pair sort)")
    ap.add_argument("--band-cull", type=int, default=0, help="0 auto (compact the visible splats on strict bands only), 1 always, 2 never")
    ap.add_argument("--epanechnikov", action="store_true", help="the reference's EPANECHNIKOV_KERNEL option (footprint sqrt(7) sigma, density 1 - q/7)")
    ap.add_argument("--rehearse", action="store_true", help="multi-rank control flow on ONE GPU: every rank on cuda:0, gloo instead of RCCL (not a measurement)")
    ap.add_argument("--inflight", type=int, default=0, help="frames in flight per GPU (renderers on separate HIP streams); 0 = 3 on one GPU, 4 per rank when the screen is split (measured optima)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    import cudagaussianrenderer_amd as gs
    from cudagaussianrenderer_amd import multigpu, scenes

    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    lib = gs.lib()  # fails loudly if libgsplat_hip.so is missing
    lib.gs_set_device(local_rank)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    stream = torch.cuda.current_stream(device)
    lib.gs_set_stream(stream.cuda_stream)
    if args.ppl:
        lib.gs_set_raster_pixels_per_lane(args.ppl)
    if args.epanechnikov:
        gs.set_footprint(gs.FOOTPRINT_EPANECHNIKOV)

    W, H, N = args.width, args.height, args.splats

    # ---- scene: synthesise a .ply FILE and ingest it through the product's parsePly ----------------
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, f"scene_rank{rank}.ply")
        names, table = scenes.synthetic_vertex_table(N, sh_degree=args.sh_degree, seed=args.seed)
        if args.opacity_mean != 1.0:
            col = names.index("opacity")
            table[:, col] += np.float32(args.opacity_mean - 1.0)
        scenes.write_ply(path, names, table)
        del table
        ply = gs.parse_ply(path)
    t_ingest = time.time() - t0
    assert ply["vertexCount"] == N

    tiles_x, tiles_y = gs.tile_counts(W, H)
    cameras = [gs.camera_from_pose(p, q, W, H) for p, q in scenes.orbit_camera_poses(ply["boundsMin"], ply["boundsMax"], args.views)]
    bands = scenes.tile_row_bands(tiles_y, world)
    if world > 1 and not args.even_bands:
        # Cost-balanced contiguous bands: every rank renders the orbit once full-frame (a calibration pass outside the
        # timed region) and sums a per-tile-row cost -- pairs in the row, plus the entries a raster tile can consume
        # before it saturates -- then cuts the rows so the summed cost per band is even. Same scene, same numbers on
        # every rank, so all ranks derive the same bands without talking to each other.
        cal = gs.Renderer(W, H)
        cal.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
        row_cost = np.zeros(tiles_y, np.float64)
        for cam in cameras:
            cal.render(cam)
            rng = cal.download(cal.buffers().tileRange, 2 * tiles_x * tiles_y, np.int32).reshape(tiles_y, tiles_x, 2)
            counts = (rng[..., 1] - rng[..., 0]).astype(np.float64)
            row_cost += counts.sum(1) + 6.0 * np.minimum(counts, 256.0).sum(1)
        cal.close()
        bands = scenes.balanced_tile_row_bands(row_cost, world)
    band = bands[rank]
    # D frames in flight: D renderers, each with its own stream, intermediates and framebuffer (the scene is uploaded
    # to each; 0.6 GB apiece), so the latency-bound stages of one frame overlap the others' kernels.
    depth = args.inflight if args.inflight > 0 else (3 if world == 1 else 4)
    streams = [torch.cuda.Stream(device) for _ in range(depth)]
    renderers, framebuffers = [], []
    for d in range(depth):
        r = gs.Renderer(W, H, tile_rows=band if world > 1 else None, band_cull=args.band_cull, pipeline=args.pipeline)
        r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
        fb = torch.zeros((H, W, 4), dtype=torch.uint8, device=device)
        r.set_back_buffer(fb.data_ptr())
        r.set_stream(streams[d].cuda_stream)
        r.set_graph(not args.no_graph)
        renderers.append(r)
        framebuffers.append(fb)
    renderer, framebuffer = renderers[0], framebuffers[0]
    torch.cuda.synchronize(device)

    def pixel_rows(b):
        return multigpu.pixel_rows(b, H)

    # One host thread per frame in flight: each drives its renderer's whole frame (it blocks on its own pair-count
    # readback only), so kernel launches of different frames are issued in parallel; the main thread collects the
    # frames in order and enqueues the band gather (RCCL calls stay on one thread, in the same order on every rank).
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor

    def init_worker():
        torch.cuda.set_device(local_rank)
        lib.gs_set_device(local_rank)

    pools = [ThreadPoolExecutor(1, initializer=init_worker) for _ in range(depth)]

    def render_frame(d, i, timed):
        renderers[d].render(cameras[i % len(cameras)], timed=timed)

    def collect(j, future):
        """Frame j is enqueued: its bands -> rank 0 (one grouped RCCL send/recv set, receives land directly in the
        framebuffer rows), ordered on the frame's own stream so the next use of that renderer waits for it."""
        future.result()
        d = j % depth
        with torch.cuda.stream(streams[d]):
            multigpu.gather_bands(framebuffers[d], bands, rank, world)

    def run_frames(count, first=0, timed=0):
        inflight = deque()
        for i in range(first, first + count):
            if len(inflight) == depth:
                collect(*inflight.popleft())
            inflight.append((i, pools[i % depth].submit(render_frame, i % depth, i, timed)))
        while inflight:
            collect(*inflight.popleft())

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # ---- set-up outside the measured steps: one frame per renderer, one at a time (frame capture, scratch sizing,
    # RCCL channel set-up for the gather), so the pipelined loop below starts from a steady state -------------------
    for d in range(depth):
        pools[d].submit(render_frame, d, d, 0).result()
        with torch.cuda.stream(streams[d]):
            multigpu.gather_bands(framebuffers[d], bands, rank, world)
        torch.cuda.synchronize(device)

    # ---- warmup, then EXACTLY --steps timed frames ----------------------------------------------------
    run_frames(args.warmup)
    sync_all()
    t_start = time.perf_counter()
    run_frames(args.steps, first=args.warmup)
    sync_all()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=("cpu" if args.rehearse else device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1: one more frame through the same path, compared on rank 0 with a full-frame render of its own --------
    gathered_ok = None
    if world > 1:
        check_view = 1 % len(cameras)
        with torch.cuda.stream(streams[0]):
            renderers[0].render(cameras[check_view])
            multigpu.gather_bands(framebuffers[0], bands, rank, world)
        torch.cuda.synchronize(device)
        if rank == 0:
            whole = gs.Renderer(W, H)
            whole.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
            whole.render(cameras[check_view])
            gathered_ok = bool(np.array_equal(whole.read_image(), framebuffers[0].cpu().numpy()))
            whole.close()
        dist.barrier()

    # ---- per-stage GPU times: the same frames again with one non-blocking event pair per stage --------
    # (one renderer, frames back to back, so the stage intervals do not overlap another frame's kernels)
    with torch.cuda.stream(streams[0]):
        for i in range(args.steps):
            renderer.render(cameras[i % len(cameras)], timed=2)
            multigpu.gather_bands(framebuffer, bands, rank, world)
        frames, sums = renderer.collect_stage_times()
    stage_ms = {
        "sh": sums.msSphericalHarmonics / frames, "clip": sums.msClipData / frames, "tile_list": sums.msBuildTileList / frames,
        "sort": sums.msSortTileList / frames, "ranges": sums.msTileRanges / frames, "raster": sums.msRasterize / frames,
    }
    # list sizes per view (deterministic), P_c from the raster kernel's own counter
    lib.gs_set_collect_raster_stats(1)
    p_sum = pc_sum = 0
    with torch.cuda.stream(streams[0]):
        for i in range(len(cameras)):
            st = renderer.render(cameras[i], stats=True)
            p_sum += st.tileListSize
            pc_sum += st.consumedPairs
    lib.gs_set_collect_raster_stats(0)
    views = len(cameras)
    p_mean, pc_mean = p_sum / views, pc_sum / views
    band_tiles = (band[1] - band[0]) * tiles_x

    result = None
    if rank == 0:
        end_bit = gs.sort_end_bit(W, H)
        sort_passes = (end_bit + 7) // 8
        band_h = pixel_rows(band)[1] - pixel_rows(band)[0]
        on_demand = renderer.colours_on_demand()
        if on_demand:
            stage_ms["sh"] = 0.0  # no SH pass: the colours are evaluated inside the raster kernel (the timer pair is empty)
        ab = algorithmic_bytes(N, p_mean, pc_mean, band_tiles, band_tiles, W, band_h, args.sh_degree, sort_passes, on_demand)
        stages = {k: {"ms": round(stage_ms[k], 4), "algorithmic_MB": round(ab[k] / 1e6, 2),
                      "GBps": round(ab[k] / 1e9 / (stage_ms[k] / 1e3), 1) if stage_ms[k] > 0 else None} for k in stage_ms}
        dominant = max(stage_ms, key=lambda k: stage_ms[k])
        # The metric names the raster kernel; report it, and name the dominant stage beside it.
        achieved = ab["raster"] / 1e9 / (stage_ms["raster"] / 1e3)
        name = ctypes_name(lib)
        result = {
            "metric": "frames/sec, 1 M splats @ 1920x1080 (raster-kernel HBM GB/s in roofline)",
            "value": round(args.steps / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{N} splats (lilly_boquet-class synthetic PLY via parsePly), SH degree {args.sh_degree}, {W}x{H}, "
                            f"{views}-view orbit, tile-row bands over {world} GPU(s)",
                "splats": N, "width": W, "height": H, "sh_degree": args.sh_degree, "views": views,
                "pairs_per_frame": round(p_mean), "consumed_pairs_per_frame": round(pc_mean),
                "raster_pixels_per_lane": lib.gs_get_raster_pixels_per_lane(), "device": name, "frames_in_flight": depth, "captured_frames": not args.no_graph, "colours": "on demand in the raster kernel (no SH pass)" if renderer.colours_on_demand() else "SH pass over all splats", "footprint": "epanechnikov" if args.epanechnikov else "gaussian", "pipeline": {1: "two_level", 2: "binned"}.get(args.pipeline, str(args.pipeline)), "tile_row_bands": [list(b) for b in bands],
                "opacity_mean_raw": args.opacity_mean, "ingest_s": round(t_ingest, 2),
                "gathered_frame_equals_single_gpu_frame": gathered_ok,
            },
            "roofline": {
                "kernel": "rasterizeTilesKernel", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), **pmc_traffic(world, N, W, H, args),
                "note": "VALU/exp-bound kernel (44 B, or 236 B with on-demand SH colours, vs ~2800 VALU lane-ops per consumed pair); see DESIGN.md",
                "dominant_stage": dominant,
                "dominant_stage_GBps": stages[dominant]["GBps"],
                "dominant_stage_frac": round((stages[dominant]["GBps"] or 0.0) / HBM_PEAK_GBS, 4),
            },
            "stages": stages,
            "gpu_ms_sum_of_stages": round(sum(stage_ms.values()), 4),
        }
        if args.cpu_frames > 0 and world == 1:
            result["cpu_baseline"] = cpu_baseline(ply, cameras, W, H, args.cpu_frames, args.epanechnikov)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for pool in pools:
        pool.shutdown()
    for r in renderers:
        r.close()
    if rank == 0:
        print(json.dumps(result))


def pmc_traffic(world, n, w, h, args):
    """HBM bytes per raster launch from the committed PMC passes (profiles/r01_pmc_raster.json: rocprofv3 --pmc FETCH_SIZE and
    --pmc WRITE_SIZE of this very workload). Counters cannot be collected from inside the timed run, so the figure is only
    attached when the run IS that workload (default scene, full frame on one GPU); otherwise traffic is null."""
    default_scene = (world == 1 and n == 1_000_000 and (w, h) == (1920, 1080) and args.sh_degree == 3 and args.seed == 1234
                     and args.opacity_mean == 1.0)
    path = os.path.join(ROOT, "profiles", "r01_pmc_raster.json")
    if not default_scene or not os.path.exists(path):
        return {"traffic": None}
    with open(path) as f:
        d = json.load(f)
    return {"traffic": round(d["traffic_bytes_per_launch_low"]), "traffic_unit": "bytes per launch",
            "traffic_source": "profiles/r01_pmc_raster.json (FETCH_SIZE raw + WRITE_SIZE; gather widths are uncalibrated, "
                              "upper bound %d with the wide-read 2x correction)" % round(d["traffic_bytes_per_launch_high"])}


def ctypes_name(lib):
    import ctypes as C

    buf = C.create_string_buffer(128)
    cus = C.c_int()
    mem = C.c_size_t()
    lib.gs_device_info(buf, 128, C.byref(cus), C.byref(mem))
    return f"{buf.value.decode()} {cus.value} CUs"


def cpu_baseline(ply, cameras, W, H, frames, epanechnikov=False):
    """The CPU oracle (serial restatement of the reference pipeline, kind 'port') on the same workload:
    `frames` frames of the same scene/views on ONE host thread. Checker code: never part of the product."""
    import cudagaussianrenderer_amd as gs
    from oracle import gsref

    sh = gs.realign_spherical_harmonics(ply["sh"], ply["shCount"], ply["vertexCount"])
    gsref.set_footprint(epanechnikov)
    t = time.perf_counter()
    done = 0
    per_stage = {}
    for i in range(frames):
        _, st, _ = gsref.render(ply["position"], ply["scaleAndRotation"], ply["color"], sh, ply["shDegree"], ply["shCount"],
                                cameras[i % len(cameras)], W, H)
        done += 1
        for f, _ in st._fields_:
            if f.startswith("ms"):
                per_stage[f] = per_stage.get(f, 0.0) + getattr(st, f)
        if time.perf_counter() - t > 30.0:
            break
    dt = time.perf_counter() - t
    return {
        "value": round(done / dt, 4), "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"{done} full frames (views 0..{done - 1}) of the same 1 M-splat scene at {W}x{H}, single thread, "
                  f"g++ -O2 without -ffast-math; host has {gsref.hardware_concurrency()} hardware threads",
        "ms_per_frame_by_stage": {k: round(v / done, 1) for k, v in per_stage.items()},
    }


if __name__ == "__main__":
    main()
