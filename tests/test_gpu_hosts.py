"""The two host programs above the C ABI, run the way a user runs them (-m gpu):

  * bench.py as a command line: `--gpus 2` WITHOUT a launcher must start its own two ranks (rehearsal mode: both on cuda:0, gloo)
    and report n_gpus 2; the N = 1 line must carry `roofline`, `cpu_baseline` and a passing `parity` block;
  * the C++ host tools/gsplat_demo.cpp (the reference's frame loop src/Demo.cpp:394-479 over include/GaussianRender.hpp):
    built with `make demo`, its PPM (rows flipped, Demo.cpp:146-147 convention) must equal the Python Renderer's
    reference-pipeline frame byte for byte.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(*flags, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert proc.returncode == 0 and len(lines) == 1, (proc.returncode, proc.stdout[-2000:], proc.stderr[-4000:])
    return json.loads(lines[0])


def test_bench_gpus_2_without_launcher_runs_two_ranks():
    """plain `python bench.py --gpus 2` spawns two ranks (here both on the one GPU, gloo) and says so."""
    out = _run_bench("--gpus", "2", "--rehearse", "--steps", "6", "--warmup", "2", "--splats", "60000", "--width", "640", "--height", "368",
                     "--views", "3", "--cpu-frames", "0")
    assert out["n_gpus"] == 2 and out["config"]["dist_world_size"] == 2 and out["config"]["dist_backend"] == "gloo"
    assert out["config"]["rehearsal_on_one_gpu"] is True
    assert out["config"]["gathered_frame_equals_single_gpu_frame"] is True
    assert len(out["config"]["tile_row_bands"]) == 2 and out["scaling"] == "strong"


def test_bench_single_gpu_line_has_roofline_cpu_baseline_and_parity():
    """A reduced workload (200 k splats at 1280x720) so the CPU leg stays short; same code path as the default run."""
    out = _run_bench("--steps", "16", "--warmup", "4", "--splats", "200000", "--width", "1280", "--height", "720", "--cpu-frames", "2",
                     "--no-extra")
    assert out["n_gpus"] == 1 and out["unit"] == "frames/s" and out["value"] > 0 and out["dtype"] == "f32"
    rf = out["roofline"]
    pc, tiles = out["config"]["consumed_pairs_per_frame"], 80 * 45
    assert abs(rf["algorithmic_bytes"] - (44 * pc + 8 * tiles + 4 * 1280 * 720)) <= 44 * 8  # 44 P_c + 8 T + 4 W H
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4 and rf["bound"] == "hbm" and rf["traffic"] is None  # not the profiled workload
    assert abs(rf["achieved"] - rf["algorithmic_bytes"] / 1e9 / (rf["kernel_ms"] / 1e3)) <= 0.01 * rf["achieved"] + 0.1
    cpu, par = out["cpu_baseline"], out["parity"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0
    assert par["ok"] is True and par["views"] == 2 and par["max_lsb"] <= 3 and par["frac_gt1"] <= 1e-3
    assert all(abs(a - b) <= par["tolerance"]["pairs_abs"] for a, b in zip(par["pairs_gpu"], par["pairs_oracle"]))
    # the kernel the timed loop runs has a roofline entry of its own (this small scene: one list per frame -> the block above)
    tk = rf["timed_loop_kernel"]
    assert tk["launches_per_frame"] == 1 and "one list" in tk["kernel"] and abs(tk["frac"] - rf["frac"]) < 1e-4
    # no stage prints "reference bytes over our time" under a name that reads like a bandwidth of this machine
    for stage in out["stages"].values():
        assert "reference_bytes_over_our_time" in stage and "reference_formulation_GBps" not in stage


def test_bench_line_prices_the_layered_raster_kernel_the_timed_loop_runs():
    """Depth layers forced (two): roofline.timed_loop_kernel describes rasterizeTilesKernel<..., LAYERED> as the loop runs it -- two launches
    per frame timed with HIP events, SURVEY.md 8(d)'s bytes per launch plus the open tiles' 16 B/px state round trip."""
    out = _run_bench("--steps", "16", "--warmup", "8", "--splats", "300000", "--width", "1280", "--height", "720", "--cpu-frames", "0",
                     "--no-extra", "--depth-layers", "2")
    assert out["config"]["depth_layers"]["in_use"] == 2
    rf = out["roofline"]
    tk = rf["timed_loop_kernel"]
    tiles = 80 * 45
    assert "LAYERED" in tk["kernel"] and tk["launches_per_frame"] == 2 and len(tk["kernel_ms_by_layer"]) == 2
    assert abs(tk["kernel_ms_per_frame"] - sum(tk["kernel_ms_by_layer"])) < 1e-3 and tk["kernel_ms_per_frame"] > 0
    want = (44 * tk["consumed_pairs_per_frame"] + 8 * tiles * 2 + 4 * 256 * tk["tiles_run_per_frame"]
            + 16 * 256 * (tk["state_tile_loads_per_frame"] + tk["state_tile_stores_per_frame"]))
    assert abs(tk["algorithmic_bytes"] - want) <= 0.001 * want + 64 * 1024  # (the counts are rounded means over 32 frames)
    assert abs(tk["achieved"] - tk["algorithmic_bytes"] / 1e9 / (tk["kernel_ms_per_frame"] / 1e3)) <= 0.01 * tk["achieved"] + 0.1
    assert abs(tk["frac"] - tk["achieved"] / tk["peak"]) < 1e-4
    # the layered frame consumes what the one-list frame consumes (same votes at the same entries): within the rounding of the two means
    assert abs(tk["consumed_pairs_per_frame"] - out["config"]["consumed_pairs_per_frame"]) <= 2
    # every tile an earlier layer leaves open stores its state once, and loads it at most once
    assert 0 < tk["state_tile_loads_per_frame"] <= tk["state_tile_stores_per_frame"] <= tiles


def _read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = [int(v) for v in f.readline().split()]
        assert f.readline().strip() == b"255"
        data = np.frombuffer(f.read(), np.uint8)
    return data.reshape(h, w, 3)


@pytest.fixture(scope="module")
def demo_binary():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cudagaussianrenderer_amd", "csrc"), "demo"], stdout=subprocess.DEVNULL)
    path = os.path.join(ROOT, "tools", "gsplat_demo")
    assert os.path.exists(path)
    return path


def test_cpp_host_procedural_frame_equals_python_reference_pipeline(gs, demo_binary, tmp_path):
    """BASELINE.json configs[0] (10 k procedural splats, 256x256) through the C++ frame loop: eight staged calls, caller-owned
    DeviceBuffer<T>s, blocking timers, PPM writer."""
    out = str(tmp_path / "c1.ppm")
    log = subprocess.run([demo_binary, "--procedural", "10000", "256", "256", "3", out], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "Rasterize" in log and "Sort Tile List" in log
    pos, sr, col = gs.generate_random_gaussians(10000, seed=1234, min_scale=0.01, max_scale=0.5)
    cam = gs.camera_from_bounds([-4] * 3, [4] * 3, 256, 256)
    r = gs.Renderer(256, 256, pipeline=gs.PIPELINE_REFERENCE)
    r.set_scene(pos, sr, col)
    r.render(cam)
    want = r.read_image()
    r.close()
    got = _read_ppm(out)[::-1]  # the writer puts buffer row H-1 (clip y = +1) first
    assert got.shape == (256, 256, 3) and np.array_equal(got, want[..., :3]) and got.max() > 0


def test_cpp_host_ply_frame_equals_python_reference_pipeline(gs, demo_binary, tmp_path):
    """A degree-3 .ply through gs::parsePly, realignSphericalHarmonics and the SH stage of the C++ loop, 333x177 (partial tiles);
    the tile list grows past its initial 8 N capacity on the way (Demo.cpp:357-366)."""
    from cudagaussianrenderer_amd import scenes

    ply_path = str(tmp_path / "scene.ply")
    scenes.write_synthetic_ply(ply_path, 20000, sh_degree=3, seed=3, log_scale_mean=-1.0)  # ~14 pairs per splat
    out = str(tmp_path / "ply.ppm")
    subprocess.run([demo_binary, ply_path, "333", "177", "4", out], stdout=subprocess.DEVNULL, check=True)
    # the same loop with the (splat, local tile) emission order and the unconditional sort, and with warm-up frames the printed averages
    # leave out: the same bytes, and the reference's Stats table plus the loop's wall clock
    other = str(tmp_path / "ply_reference_order.ppm")
    log = subprocess.run([demo_binary, ply_path, "333", "177", "3", other, "--builder", "reference-order", "--warm-up", "2"], stdout=subprocess.PIPE,
                         text=True, check=True).stdout
    assert open(out, "rb").read() == open(other, "rb").read()
    assert "Frame loop wall clock" in log and "(reference-order builder)" in log and "Total" in log
    # ... and the loop body as ONE call from the same C++ host (gs_renderer_render, captured frames): the same bytes again
    one = str(tmp_path / "ply_one_call.ppm")
    log = subprocess.run([demo_binary, ply_path, "333", "177", "3", one, "--one-call", "--warm-up", "2"], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert open(out, "rb").read() == open(one, "rb").read() and "(one-call renderer)" in log
    ply = gs.parse_ply(ply_path)
    cam = gs.camera_from_bounds(ply["boundsMin"], ply["boundsMax"], 333, 177)
    r = gs.Renderer(333, 177, pipeline=gs.PIPELINE_REFERENCE)
    r.set_scene(ply["position"], ply["scaleAndRotation"], ply["color"], ply["sh"], ply["shDegree"], ply["shCount"])
    st = r.render(cam, stats=True)
    want = r.read_image()
    r.close()
    assert st.tileListSize > 8 * 20000  # the C++ loop had to double its buffers to get the complete frame
    assert np.array_equal(_read_ppm(out)[::-1], want[..., :3])


def test_cpp_host_replays_a_recorded_camera_session(gs, gsref, demo_binary, tmp_path, golden_dir):
    """SURVEY.md 8(f4), the interactive viewer without its window: the C++ frame loop moves gs::CameraControls with one recorded input
    line per frame (drag, orbit, pan, keys -- the first session of tests/golden/camera_controls.json) and renders every frame.  The pose
    it ends in is the oracle's pose for that session and the image it leaves is the frame the Python Renderer draws from the oracle's
    CameraData for that pose, byte for byte; the oracle's (view, projection, rays, roll removal ...) are held to the reference's own
    CameraControls.cpp in tests/test_oracle_golden.py."""
    sess = json.load(open(os.path.join(golden_dir, "camera_controls.json")))["sessions"][1]
    w, h, frames = sess["width"] // 4, sess["height"] // 4, sess["frames"][:48]  # 256 x 256 window: scale the recorded cursor
    frames = [(dt, b, k, x / 4.0, y / 4.0) for (dt, b, k, x, y) in frames]
    events = tmp_path / "events.txt"
    events.write_text("".join("%.9g %d %d %.9g %.9g\n" % tuple(f) for f in frames))
    out = str(tmp_path / "session.ppm")
    log = subprocess.run([demo_binary, "--procedural", "10000", str(w), str(h), str(len(frames)), out, "--events", str(events), "--dump-every", "16"],
                         stdout=subprocess.PIPE, text=True, check=True).stdout
    line = [ln for ln in log.splitlines() if ln.startswith("camera after")]
    assert len(line) == 1
    nums = [float(v) for v in line[0].replace("camera after %d frames:" % len(frames), "").replace("position", "").replace("rotation", "")
            .replace("anchor", "").replace("mode", "").split()]
    states, cams = gsref.camera_controls_session(w, h, [-4] * 3, [4] * 3, frames, with_camera_data=True)
    assert np.array_equal(np.array(nums[:10], np.float32), states[-1, :10]) and int(nums[10]) == int(states[-1, 15])
    assert not np.array_equal(states[-1, :7], states[0, :7])  # the camera did move
    pos, sr, col = gs.generate_random_gaussians(10000, seed=1234, min_scale=0.01, max_scale=0.5)
    cam = gs.CameraData.from_buffer_copy(bytes(cams[-1]))
    r = gs.Renderer(w, h, pipeline=gs.PIPELINE_REFERENCE)
    r.set_scene(pos, sr, col)
    r.render(cam)
    want = r.read_image()
    r.close()
    got = _read_ppm(out)[::-1]
    assert np.array_equal(got, want[..., :3]) and got.max() > 0
    # --dump-every 16: the session as an image sequence; its last image is the final frame, an earlier one is another view
    dumps = [out + ".%05d.ppm" % k for k in (16, 32, 48)]
    assert all(os.path.exists(d) for d in dumps)
    assert np.array_equal(_read_ppm(dumps[-1]), _read_ppm(out)) and not np.array_equal(_read_ppm(dumps[0]), _read_ppm(out))


def test_cpp_host_one_call_viewer_session_with_tile_start_order(gs, gsref, demo_binary, tmp_path, golden_dir):
    """The viewer's recorded session through the ONE-CALL renderer of the C++ host (gsplat_demo --one-call --events), with and without
    gs_renderer_set_tile_start_order (--tile-order: the raster's tiles started longest-first by the previous frame's consumption -- the
    interactive camera is the host the option is for): the same final pose, which is the oracle's, and the same image bytes, which are the
    staged frame loop's."""
    sess = json.load(open(os.path.join(golden_dir, "camera_controls.json")))["sessions"][1]
    w, h, frames = sess["width"] // 4, sess["height"] // 4, sess["frames"][:48]
    frames = [(dt, b, k, x / 4.0, y / 4.0) for (dt, b, k, x, y) in frames]
    events = tmp_path / "events.txt"
    events.write_text("".join("%.9g %d %d %.9g %.9g\n" % tuple(f) for f in frames))
    images, poses = {}, {}
    for name, flags in (("staged", []), ("one_call", ["--one-call"]), ("one_call_tile_order", ["--one-call", "--tile-order"]),
                        ("staged_tile_order", ["--tile-order"])):
        out = str(tmp_path / (name + ".ppm"))
        log = subprocess.run([demo_binary, "--procedural", "10000", str(w), str(h), str(len(frames)), out, "--events", str(events)] + flags,
                             stdout=subprocess.PIPE, text=True, check=True).stdout
        line = [ln for ln in log.splitlines() if ln.startswith("camera after")]
        assert len(line) == 1, log
        poses[name] = line[0]
        images[name] = _read_ppm(out)
        if "--tile-order" in flags:
            assert "tile start order from the previous frame" in log
    assert poses["one_call"] == poses["staged"] == poses["one_call_tile_order"] == poses["staged_tile_order"]
    states, _ = gsref.camera_controls_session(w, h, [-4] * 3, [4] * 3, frames, with_camera_data=True)
    nums = [float(v) for v in poses["staged"].replace("camera after %d frames:" % len(frames), "").replace("position", "").replace("rotation", "")
            .replace("anchor", "").replace("mode", "").split()]
    assert np.array_equal(np.array(nums[:10], np.float32), states[-1, :10])
    assert images["staged"].max() > 0
    assert np.array_equal(images["one_call"], images["staged"]) and np.array_equal(images["one_call_tile_order"], images["staged"])
    assert np.array_equal(images["staged_tile_order"], images["staged"])  # gs_set_staged_tile_start_order: the eight staged calls with the option


def test_gl_interop_without_a_context_is_refused_and_mapping_it_ends_the_process(gs):
    """gs_graphics_* is the reference's CudaGraphicsResource (Utilities.h:113-153) over HIP's GL interop.  This pool has no
    window system, so there is no GL context to register a buffer from: what can be run is the refusal -- a registration without a context
    returns NULL with a message, unregister / unmap of NULL do nothing, the library renders on -- and the reference's error convention:
    mapping a resource that was never registered prints and exits with EXIT_FAILURE (checkCudaErrors, Utilities.h:127), in a child."""
    lib = gs.lib()
    handle = lib.gs_graphics_gl_register_buffer(1, 0)
    assert not handle
    lib.gs_graphics_unmap(None)
    lib.gs_graphics_unregister(None)
    r = gs.Renderer(64, 64)
    pos, sr, col = gs.generate_random_gaussians(1000, seed=3)
    r.set_scene(pos, sr, col)
    r.render(gs.camera_from_bounds([-4] * 3, [4] * 3, 64, 64))
    assert int(r.read_image()[..., :3].sum()) > 0
    r.close()
    child = ("import sys; sys.path.insert(0, %r); import torch; import cudagaussianrenderer_amd as gs; lib = gs.lib(); lib.gs_set_device(0); "
             "lib.gs_graphics_map(None, None); print('survived')" % ROOT)
    proc = subprocess.run([sys.executable, "-c", child], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, cwd=ROOT)
    assert proc.returncode == 1 and "survived" not in proc.stdout and "gs_graphics_map" in proc.stderr, (proc.returncode, proc.stdout, proc.stderr[-500:])
