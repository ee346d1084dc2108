"""Synthetic inputs for tests and bench (numpy only; no GPU, no oracle).

* `write_synthetic_ply`  -- a 3DGS-format .ply FILE of the "lilly_boquet class" (BASELINE.json configs 3-5):
  the reference ships no data (README.md:8), so the scene is synthesised and then goes through the real
  ingest (gs_parse_ply) like a captured scene would.
* `orbit_camera_poses`   -- the default pose of CameraControls::setBounds orbited about the scene's up axis.
Procedural splats of configs 1-2 come from gs_generate_random_gaussians (Demo.cpp:119-143) in the library.
"""
import numpy as np

# Standard 3DGS vertex layout: 62 float properties, 248 bytes per vertex at SH degree 3.
def ply_property_names(sh_degree=3):
    rest = {0: 0, 1: 9, 2: 24, 3: 45, 4: 72}[sh_degree]
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    names += [f"f_rest_{i}" for i in range(rest)]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    return names


def synthetic_vertex_table(count, sh_degree=3, seed=1234, extent=4.0, clusters=64, log_scale_mean=-3.6,
                           log_scale_sigma=0.7, chunk=None):
    """Raw (pre-activation) attribute table [count, properties] float32, as a trained 3DGS model stores it.

    positions: mixture of `clusters` isotropic Gaussians inside the +-extent box; log-scales ~ N(mean, sigma);
    raw opacity ~ N(1, 2) (pre-sigmoid); f_dc ~ N(0, 1); f_rest ~ N(0, 0.1); rotations uniform unit quaternions.
    """
    rng = np.random.default_rng(seed)
    names = ply_property_names(sh_degree)
    col = {n: i for i, n in enumerate(names)}
    table = np.zeros((count, len(names)), np.float32)
    centres = rng.uniform(-0.75 * extent, 0.75 * extent, size=(clusters, 3)).astype(np.float32)
    spread = rng.uniform(0.15, 0.9, size=(clusters,)).astype(np.float32)
    which = rng.integers(0, clusters, size=count)
    pos = centres[which] + rng.standard_normal((count, 3)).astype(np.float32) * spread[which, None]
    pos = np.clip(pos, -extent, extent)
    table[:, col["x"]:col["x"] + 3] = pos
    table[:, col["f_dc_0"]:col["f_dc_0"] + 3] = rng.standard_normal((count, 3)).astype(np.float32)
    rest = len(names) - 17
    if rest:
        table[:, col["f_rest_0"]:col["f_rest_0"] + rest] = 0.1 * rng.standard_normal((count, rest)).astype(np.float32)
    table[:, col["opacity"]] = 1.0 + 2.0 * rng.standard_normal(count).astype(np.float32)
    table[:, col["scale_0"]:col["scale_0"] + 3] = (
        log_scale_mean + log_scale_sigma * rng.standard_normal((count, 3))).astype(np.float32)
    q = rng.standard_normal((count, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    table[:, col["rot_0"]:col["rot_0"] + 4] = q
    return names, table


def write_ply(path, names, table):
    """binary_little_endian PLY with `property float` columns only -- the only dialect the reference reads."""
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % table.shape[0]
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(table, dtype="<f4").tobytes())


def write_synthetic_ply(path, count, sh_degree=3, seed=1234, **kw):
    names, table = synthetic_vertex_table(count, sh_degree=sh_degree, seed=seed, **kw)
    write_ply(path, names, table)
    return names


def default_pose(bounds_min, bounds_max):
    """CameraControls.cpp:132-146: position and (x,y,z,w) rotation looking at the box centre from (0,1,1)."""
    bmin = np.asarray(bounds_min, np.float32)
    bmax = np.asarray(bounds_max, np.float32)
    size = bmax - bmin
    centre = bmin + size * 0.5
    max_size = float(size.max())
    offset = np.array([0.0, 1.0, 1.0], np.float32) / np.sqrt(2.0)
    dist = np.sqrt(2.0) * max_size * 0.5 / np.tan(np.radians(60.0) * 0.5)
    return centre, offset, float(dist)


def _quat_from_basis(right, up, back):
    m = np.stack([right, up, back], axis=1)  # columns
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        w = 0.25 * s
        x = (m[2, 1] - m[1, 2]) / s
        y = (m[0, 2] - m[2, 0]) / s
        z = (m[1, 0] - m[0, 1]) / s
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        w = (m[2, 1] - m[1, 2]) / s
        x = 0.25 * s
        y = (m[0, 1] + m[1, 0]) / s
        z = (m[0, 2] + m[2, 0]) / s
    elif m[1, 1] > m[2, 2]:
        s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        w = (m[0, 2] - m[2, 0]) / s
        x = (m[0, 1] + m[1, 0]) / s
        y = 0.25 * s
        z = (m[1, 2] + m[2, 1]) / s
    else:
        s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        w = (m[1, 0] - m[0, 1]) / s
        x = (m[0, 2] + m[2, 0]) / s
        y = (m[1, 2] + m[2, 1]) / s
        z = 0.25 * s
    return np.array([x, y, z, w], np.float32)


def orbit_camera_poses(bounds_min, bounds_max, views=8):
    """`views` poses: the default pose yawed by k*360/views degrees about +Y through the box centre.

    Returns a list of (position[3], rotation_xyzw[4]); feed to gs_camera_from_pose.
    """
    centre, offset, dist = default_pose(bounds_min, bounds_max)
    poses = []
    for k in range(views):
        a = 2.0 * np.pi * k / views
        rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float64)
        back = rot @ offset.astype(np.float64)
        position = centre + back * dist
        up0 = np.array([0.0, 1.0, 0.0])
        right = np.cross(up0, back)
        right /= np.linalg.norm(right)
        up = np.cross(back, right)
        poses.append((position.astype(np.float32), _quat_from_basis(right, up, back)))
    return poses


def tile_row_bands(tiles_y, world_size):
    """Contiguous tile-row bands, sizes differing by at most one (1080p, 8 ranks: 9,9,9,9,8,8,8,8)."""
    base, extra = divmod(tiles_y, world_size)
    bands, begin = [], 0
    for r in range(world_size):
        rows = base + (1 if r < extra else 0)
        bands.append((begin, begin + rows))
        begin += rows
    return bands


def balanced_tile_row_bands(row_costs, world_size):
    """Contiguous tile-row bands whose summed cost is as even as a prefix partition allows.

    `row_costs[y]` is the estimated work of tile row y (e.g. pairs in that row from a calibration frame). Every rank
    calls this with the same numbers, so all ranks agree on the bands without communicating. Each band gets at least
    one row while rows remain. Minimises the maximum band cost (binary search on the cap + greedy packing).
    """
    costs = [float(c) for c in row_costs]
    rows = len(costs)
    world_size = max(1, int(world_size))
    if world_size == 1:
        return [(0, rows)]
    if rows <= world_size:
        return [(min(r, rows), min(r + 1, rows)) for r in range(world_size)]

    def pack(cap):
        bands, begin, acc = [], 0, 0.0
        for y in range(rows):
            remaining_bands = world_size - len(bands) - 1
            must_close = rows - y <= remaining_bands  # keep one row for every later band
            if y > begin and (acc + costs[y] > cap or must_close):
                bands.append((begin, y))
                begin, acc = y, 0.0
                if len(bands) == world_size - 1:
                    break
            acc += costs[y]
        bands.append((begin, rows))
        return bands

    lo, hi = max(costs), sum(costs)
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        b = pack(mid)
        worst = max(sum(costs[a:e]) for a, e in b)
        if len(b) == world_size and worst <= mid * (1 + 1e-9):
            hi = mid
        else:
            lo = mid
    bands = pack(hi)
    while len(bands) < world_size:  # degenerate inputs: split the largest band
        i = max(range(len(bands)), key=lambda k: bands[k][1] - bands[k][0])
        a, e = bands[i]
        if e - a < 2:
            bands.append((rows, rows))
            continue
        m = (a + e) // 2
        bands[i:i + 1] = [(a, m), (m, e)]
    return bands
