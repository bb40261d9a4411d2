"""Seeded synthetic LiDAR sweeps shaped like the sensors A-LOAM supports.

No dataset ships with the reference (NSH bag / KITTI are external downloads, reference README.md:36,45)
and there is no network here, so every test / bench input is ray-cast from a small procedural world:
ground plane + enclosing walls + yawed boxes ("buildings") + vertical cylinders ("poles"), scanned from
poses on a closed trajectory.  The world gives both feature classes A-LOAM extracts (edges, planes) and a
ground-truth trajectory for ATE.

Written with torch ops only so the same code renders on the CPU (tests) and on the GPU (bench input
generation).  Nothing here is on the measured path.

Sensor models (inverse of the ring formulas at reference src/scanRegistration.cpp:166-200):
  VLP-16 : ring r at elevation -15 + 2 r deg, firing order (all 16 lasers per azimuth step)
  HDL-32 : ring r at elevation -92/3 + 4/3 (r + 0.5) deg, firing order
  HDL-64 : ring r at 2 - r/3 deg (r < 32) and -8.83 - (r - 32)/2 deg (r >= 32), ring-major order as in
           KITTI .bin files (reference src/kittiHelper.cpp:131-151 streams them unchanged)
  ROWS128: 128 evenly spaced rows, ring-major, ring index carried in the 4th float (config 4: the
           reference has no 128-line formula, src/scanRegistration.cpp:472-476)
Azimuth decreases with time (clockwise seen from above) so that `ori = -atan2(y, x)` increases through the
sweep as src/scanRegistration.cpp:141-153,208-236 assumes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

SENSOR_HEIGHT = 1.73


@dataclass
class SensorModel:
    name: str
    n_scans: int
    columns: int
    min_range: float
    ring_from_field: bool
    dirs: torch.Tensor      # [N, 3] unit directions in the sensor frame, in scan (message) order
    ring: torch.Tensor      # [N] int32 nominal ring of each direction


def _elevations(name: str) -> torch.Tensor:
    if name == "VLP-16":
        return torch.tensor([-15.0 + 2.0 * r for r in range(16)], dtype=torch.float64)
    if name == "HDL-32":
        return torch.tensor([-92.0 / 3.0 + (r + 0.5) * 4.0 / 3.0 for r in range(32)], dtype=torch.float64)
    if name == "HDL-64":
        return torch.tensor([2.0 - r / 3.0 if r < 32 else -8.83 - (r - 32) / 2.0 for r in range(64)], dtype=torch.float64)
    if name == "ROWS128":
        return torch.tensor([2.0 - r * (26.8 / 127.0) for r in range(128)], dtype=torch.float64)
    raise ValueError(name)


def sensor_model(name: str, columns: int | None = None, device="cpu", az_offset: float = 0.0) -> SensorModel:
    """`az_offset` (in columns) turns the start of the sweep: 0 starts half a column below +pi; 0.75 starts a quarter of a column beyond
    the +-pi wrap of atan2, which is where startOri / endOri / halfPassed of src/scanRegistration.cpp:141-153,208-236 take their other arms."""
    spec = {
        "VLP-16": (16, 1800, 0.3, False, "firing"),
        "HDL-32": (32, 2048, 0.3, False, "firing"),
        "HDL-64": (64, 2048, 5.0, False, "ring"),
        "ROWS128": (128, 2048, 5.0, True, "ring"),
    }[name]
    n_scans, cols, min_range, from_field, order = spec
    if columns is not None:
        cols = columns
    el = torch.deg2rad(_elevations(name))
    k = torch.arange(cols, dtype=torch.float64)
    az = math.pi - 2.0 * math.pi * (k + 0.5 - az_offset) / cols          # pi -> -pi, clockwise
    if order == "ring":
        el_g = el[:, None].expand(n_scans, cols)
        az_g = az[None, :].expand(n_scans, cols)
        ring = torch.arange(n_scans, dtype=torch.int32)[:, None].expand(n_scans, cols)
    else:
        el_g = el[None, :].expand(cols, n_scans)
        az_g = az[:, None].expand(cols, n_scans)
        ring = torch.arange(n_scans, dtype=torch.int32)[None, :].expand(cols, n_scans)
    ce = torch.cos(el_g)
    dirs = torch.stack([ce * torch.cos(az_g), ce * torch.sin(az_g), torch.sin(el_g)], dim=-1).reshape(-1, 3)
    return SensorModel(name, n_scans, cols, min_range, from_field, dirs.to(device), ring.reshape(-1).contiguous().to(device))


@dataclass
class World:
    half_extent: float
    box_c: torch.Tensor     # [B, 3] centres
    box_h: torch.Tensor     # [B, 3] half sizes
    box_cs: torch.Tensor    # [B, 2] cos / sin of yaw
    pole_c: torch.Tensor    # [P, 2] xy
    pole_r: torch.Tensor    # [P]
    pole_top: torch.Tensor  # [P] z of the top

    def to(self, device):
        return World(self.half_extent, *(t.to(device) for t in (self.box_c, self.box_h, self.box_cs, self.pole_c, self.pole_r, self.pole_top)))


def make_world(seed: int, n_boxes: int = 40, n_poles: int = 30, half_extent: float = 70.0, track_radius: float = 30.0) -> World:
    g = torch.Generator().manual_seed(seed)

    def place(n, clearance):
        out = []
        while len(out) < n:
            xy = (torch.rand(2, generator=g, dtype=torch.float64) * 2 - 1) * (half_extent - 8.0)
            rad = float(torch.linalg.norm(xy))
            if abs(rad - track_radius) < clearance:
                continue
            out.append(xy)
        return torch.stack(out)

    bxy = place(n_boxes, 9.0)
    size = torch.rand(n_boxes, 3, generator=g, dtype=torch.float64)
    half = torch.stack([1.5 + 4.0 * size[:, 0], 1.5 + 4.0 * size[:, 1], 1.5 + 5.0 * size[:, 2]], dim=1)
    cz = -SENSOR_HEIGHT + half[:, 2]
    yaw = torch.rand(n_boxes, generator=g, dtype=torch.float64) * math.pi
    pxy = place(n_poles, 3.0)
    pr = 0.12 + 0.25 * torch.rand(n_poles, generator=g, dtype=torch.float64)
    ptop = -SENSOR_HEIGHT + 3.0 + 6.0 * torch.rand(n_poles, generator=g, dtype=torch.float64)
    return World(half_extent, torch.cat([bxy, cz[:, None]], dim=1), half, torch.stack([torch.cos(yaw), torch.sin(yaw)], dim=1), pxy, pr, ptop)


STREET_AMP, STREET_WAVELENGTH, STREET_MAX_RANGE = 25.0, 320.0, 120.0


def _street_y(x):
    return STREET_AMP * torch.sin(2.0 * math.pi * x / STREET_WAVELENGTH)


def make_street_world(seed: int, length: float = 700.0) -> World:
    """A world to TRAVEL through (non-returning trajectories, `trajectory_travel`): a gently winding street of `length` metres along x
    (centre line y = 25 sin(2 pi x / 320)) with a yawed box every ~14 m on either side (9 - 22 m off the centre line) and a pole every
    ~11 m (4.5 - 7 m off it), so that a sensor with a 120 m range always sees both feature classes.  The map window of the reference's
    mapping node (21 x 21 x 11 cubes of 50 m around the FIRST pose, src/laserMapping.cpp:72-80) only shifts once the sensor is more
    than 375 m from where it started (:323-507): that is what the length is for."""
    g = torch.Generator().manual_seed(4000 + seed)
    dt = torch.float64
    xs_b, xs_p = [], []
    for side in (-1.0, 1.0):
        x = -length / 2 + 5.0 + 7.0 * float(torch.rand(1, generator=g, dtype=dt))
        while x < length / 2 - 5.0:
            xs_b.append((x, side))
            x += 11.0 + 6.0 * float(torch.rand(1, generator=g, dtype=dt))
    x = -length / 2 + 3.0
    k = 0
    while x < length / 2 - 3.0:
        xs_p.append((x, -1.0 if k % 2 else 1.0))
        x += 8.0 + 6.0 * float(torch.rand(1, generator=g, dtype=dt))
        k += 1
    nb, npole = len(xs_b), len(xs_p)
    bx = torch.tensor([v[0] for v in xs_b], dtype=dt)
    bs = torch.tensor([v[1] for v in xs_b], dtype=dt)
    size = torch.rand(nb, 3, generator=g, dtype=dt)
    half = torch.stack([1.5 + 4.0 * size[:, 0], 1.5 + 4.0 * size[:, 1], 1.5 + 5.0 * size[:, 2]], dim=1)
    off = 9.0 + torch.linalg.norm(half[:, :2], dim=1) + 8.0 * torch.rand(nb, generator=g, dtype=dt)      # clear of the centre line whatever the yaw
    by = _street_y(bx) + bs * off
    cz = -SENSOR_HEIGHT + half[:, 2]
    yaw = torch.rand(nb, generator=g, dtype=dt) * math.pi
    px = torch.tensor([v[0] for v in xs_p], dtype=dt)
    ps = torch.tensor([v[1] for v in xs_p], dtype=dt)
    py = _street_y(px) + ps * (4.5 + 2.5 * torch.rand(npole, generator=g, dtype=dt))
    pr = 0.12 + 0.25 * torch.rand(npole, generator=g, dtype=dt)
    ptop = -SENSOR_HEIGHT + 3.0 + 6.0 * torch.rand(npole, generator=g, dtype=dt)
    return World(length / 2 + 60.0, torch.stack([bx, by, cz], dim=1), half, torch.stack([torch.cos(yaw), torch.sin(yaw)], dim=1),
                 torch.stack([px, py], dim=1), pr, ptop)


def trajectory_travel(n_frames: int, step: float = 1.6, seed: int = 0, length: float = 700.0, ramp: int = 10):
    """Poses along the street of `make_street_world`, never returning: starts at x = -3/4 wavelength (where the street runs parallel to x, so the
    map frame of the reference - the first pose - is the world's up to the small roll / pitch) and advances `step` metres of arc
    per sweep (ramping up over the first `ramp` sweeps: the odometry's first solve starts from the identity, reference
    src/laserOdometry.cpp:97-98), heading along the tangent, with the small roll / pitch / height oscillations of `trajectory`."""
    g = torch.Generator().manual_seed(2000 + seed)
    ph = torch.rand(3, generator=g, dtype=torch.float64) * 2 * math.pi
    k = torch.arange(n_frames, dtype=torch.float64)
    ds = step * torch.clamp((k + 1.0) / float(ramp + 1), max=1.0)
    xs = [-0.75 * STREET_WAVELENGTH]
    assert xs[0] > -length / 2 + 60.0
    for i in range(1, n_frames):
        x = xs[-1]
        slope = STREET_AMP * 2.0 * math.pi / STREET_WAVELENGTH * math.cos(2.0 * math.pi * x / STREET_WAVELENGTH)
        xs.append(x + float(ds[i]) / math.sqrt(1.0 + slope * slope))
    x = torch.tensor(xs, dtype=torch.float64)
    assert float(x[-1]) < length / 2 - 20.0, "trajectory leaves the street: raise `length`"
    slope = STREET_AMP * 2.0 * math.pi / STREET_WAVELENGTH * torch.cos(2.0 * math.pi * x / STREET_WAVELENGTH)
    t = torch.stack([x, _street_y(x), 0.05 * torch.sin(0.31 * k + ph[0])], dim=1)
    return _rpy_matrices(torch.atan(slope), torch.deg2rad(torch.tensor(0.4, dtype=torch.float64)) * torch.sin(0.23 * k + ph[1]),
                         torch.deg2rad(torch.tensor(0.3, dtype=torch.float64)) * torch.sin(0.17 * k + ph[2])), t


def _rpy_matrices(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = torch.cos(yaw), torch.sin(yaw), torch.cos(pitch), torch.sin(pitch), torch.cos(roll), torch.sin(roll)
    return torch.stack([
        torch.stack([cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], dim=1),
        torch.stack([sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr], dim=1),
        torch.stack([-sp, cp * sr, cp * cr], dim=1)], dim=1)


def trajectory(n_frames: int, step: float = 1.0, radius: float = 30.0, seed: int = 0, start_angle: float = 0.0):
    """Poses (R [n,3,3], t [n,3], float64) of the sensor in the world: a circle of `radius` driven `step`
    metres per sweep with small roll / pitch / height oscillations (6-DoF motion)."""
    g = torch.Generator().manual_seed(1000 + seed)
    ph = torch.rand(3, generator=g, dtype=torch.float64) * 2 * math.pi
    k = torch.arange(n_frames, dtype=torch.float64)
    ang = start_angle + k * (step / radius)
    t = torch.stack([radius * torch.cos(ang), radius * torch.sin(ang), 0.05 * torch.sin(0.31 * k + ph[0])], dim=1)
    yaw = ang + math.pi / 2
    pitch = torch.deg2rad(torch.tensor(0.4, dtype=torch.float64)) * torch.sin(0.23 * k + ph[1])
    roll = torch.deg2rad(torch.tensor(0.3, dtype=torch.float64)) * torch.sin(0.17 * k + ph[2])
    cy, sy, cp, sp, cr, sr = torch.cos(yaw), torch.sin(yaw), torch.cos(pitch), torch.sin(pitch), torch.cos(roll), torch.sin(roll)
    R = torch.stack([
        torch.stack([cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], dim=1),
        torch.stack([sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr], dim=1),
        torch.stack([-sp, cp * sr, cp * cr], dim=1)], dim=1)
    return R, t


def render_scan(world: World, model: SensorModel, R: torch.Tensor, t: torch.Tensor, noise_sigma: float,
                generator: torch.Generator | None = None, nan_fraction: float = 0.0, rough: bool = False, max_range: float = 200.0,
                cull: bool = False) -> torch.Tensor:
    """Ray-cast one sweep.  Returns float32 [N, 4] (x, y, z, field) in the SENSOR frame and scan order; the 4th
    float is the ring index for ROWS128 and 0 otherwise.  Rays without a return are dropped (as a real
    driver does); `nan_fraction` replaces that share of the returns by NaN to exercise the NaN filter.

    `rough` makes the sweep irregular the way recorded KITTI sweeps are: 7 % of the rays return nothing at random, every ring loses
    one contiguous arc of its own length (rings of unequal length), a tenth of the azimuth range is "vegetation" (range noise of
    0.3 m: corner candidates everywhere), and a few returns are repeated verbatim 2 .. 14 times (runs of identical points give
    exactly equal — mostly zero — curvatures, i.e. ties in the sort of src/scanRegistration.cpp:288)."""
    dev = model.dirs.device
    dt = torch.float64
    R = R.to(dev, dt)
    o = t.to(dev, dt)
    if cull:       # only the obstacles that can lie within `max_range` (big worlds); leaves every return unchanged
        kb = torch.linalg.norm(world.box_c[:, :2] - o[None, :2], dim=1) < max_range + 12.0
        kp = torch.linalg.norm(world.pole_c - o[None, :2], dim=1) < max_range + 1.0
        world = World(world.half_extent, world.box_c[kb], world.box_h[kb], world.box_cs[kb], world.pole_c[kp], world.pole_r[kp], world.pole_top[kp])
    d = model.dirs @ R.T                                    # world-frame directions [N,3]
    inf = torch.full((d.shape[0],), float("inf"), dtype=dt, device=dev)
    best = inf.clone()

    # ground plane z = -SENSOR_HEIGHT
    tz = (-SENSOR_HEIGHT - o[2]) / d[:, 2]
    best = torch.where((d[:, 2] < 0) & (tz > 0), torch.minimum(best, tz), best)

    # enclosing walls (seen from inside): |x| = W, |y| = W
    W = world.half_extent
    for ax in (0, 1):
        tw = torch.where(d[:, ax] > 0, (W - o[ax]) / d[:, ax], (-W - o[ax]) / d[:, ax])
        tw = torch.where(d[:, ax] == 0, inf, tw)
        best = torch.minimum(best, torch.where(tw > 0, tw, inf))

    # yawed boxes: slab test in each box frame, chunked over boxes to bound memory
    c, s = world.box_cs[:, 0], world.box_cs[:, 1]
    for b0 in range(0, world.box_c.shape[0], 8):
        sl = slice(b0, b0 + 8)
        rel = o[None, :] - world.box_c[sl]                   # [b,3]
        ox = c[sl] * rel[:, 0] + s[sl] * rel[:, 1]
        oy = -s[sl] * rel[:, 0] + c[sl] * rel[:, 1]
        ob = torch.stack([ox, oy, rel[:, 2]], dim=1)         # [b,3]
        dx = d[:, None, 0] * c[sl][None] + d[:, None, 1] * s[sl][None]
        dy = -d[:, None, 0] * s[sl][None] + d[:, None, 1] * c[sl][None]
        db = torch.stack([dx, dy, d[:, None, 2].expand_as(dx)], dim=2)   # [N,b,3]
        inv = 1.0 / db
        t1 = (-world.box_h[sl][None] - ob[None]) * inv
        t2 = (world.box_h[sl][None] - ob[None]) * inv
        tn = torch.minimum(t1, t2).amax(dim=2)
        tf = torch.maximum(t1, t2).amin(dim=2)
        hit = (tf >= tn) & (tn > 0)
        tb = torch.where(hit, tn, torch.full_like(tn, float("inf"))).amin(dim=1)
        best = torch.minimum(best, tb)

    # vertical cylinders
    rel = o[None, :2] - world.pole_c                          # [P,2]
    a = (d[:, 0] ** 2 + d[:, 1] ** 2)[:, None]                # [N,1]
    bq = d[:, None, 0] * rel[None, :, 0] + d[:, None, 1] * rel[None, :, 1]
    cq = (rel ** 2).sum(dim=1)[None] - (world.pole_r ** 2)[None]
    disc = bq * bq - a * cq
    tc = (-bq - torch.sqrt(disc.clamp_min(0))) / a
    zc = o[2] + tc * d[:, None, 2]
    okc = (disc > 0) & (tc > 0) & (zc <= world.pole_top[None]) & (zc >= -SENSOR_HEIGHT)
    best = torch.minimum(best, torch.where(okc, tc, torch.full_like(tc, float("inf"))).amin(dim=1))

    valid = torch.isfinite(best) & (best < max_range)
    if generator is not None:
        noise = torch.randn(best.shape, generator=generator, dtype=dt, device=generator.device).to(dev)
    else:
        noise = torch.zeros_like(best)
    rng = best + noise_sigma * noise
    if rough:
        assert generator is not None
        gdev = generator.device
        n_ray = best.shape[0]
        cols = model.columns
        # position of every ray on its ring (0 .. columns-1), whatever the message order is
        ring = model.ring.to(torch.long)
        col = torch.zeros(n_ray, dtype=torch.long, device=dev)
        order = torch.argsort(ring, stable=True)
        col[order] = torch.arange(n_ray, device=dev) % cols if n_ray == model.n_scans * cols else torch.arange(n_ray, device=dev) % cols
        u = torch.rand(n_ray, generator=generator, dtype=dt, device=gdev).to(dev)
        valid = valid & (u > 0.07)                                              # random no-returns
        arc0 = (torch.rand(model.n_scans, generator=generator, dtype=dt, device=gdev).to(dev) * cols).to(torch.long)
        arcl = (torch.rand(model.n_scans, generator=generator, dtype=dt, device=gdev).to(dev) * 0.04 * cols + 4).to(torch.long)
        d_arc = (col - arc0[ring]) % cols
        valid = valid & (d_arc >= arcl[ring])                                    # one missing arc per ring
        veg0 = int(torch.rand(1, generator=generator, dtype=dt, device=gdev).item() * cols)
        veg = ((col - veg0) % cols) < cols // 10
        rough_noise = torch.randn(n_ray, generator=generator, dtype=dt, device=gdev).to(dev)
        rng = torch.where(veg & (ring % 3 != 0), rng + 0.3 * rough_noise, rng)
    pts = (model.dirs * rng[:, None]).to(torch.float32)
    field = model.ring.to(torch.float32) if model.ring_from_field else torch.zeros(pts.shape[0], dtype=torch.float32, device=dev)
    out = torch.cat([pts, field[:, None]], dim=1)
    if nan_fraction > 0 and generator is not None:
        drop = torch.rand(best.shape, generator=generator, dtype=dt, device=generator.device).to(dev) < nan_fraction
        out[drop, :3] = float("nan")
    out = out[valid].contiguous()
    if rough and out.shape[0] > 64:
        n_dup = 24
        pos = (torch.rand(n_dup, generator=generator, dtype=dt, device=generator.device) * (out.shape[0] - 16)).to(torch.long).tolist()
        run = (torch.rand(n_dup, generator=generator, dtype=dt, device=generator.device) * 13 + 2).to(torch.long).tolist()
        for i, l in zip(pos, run):
            out[i + 1:i + l, :3] = out[i, :3]                                    # the same return repeated (the ring field stays)
    return out


def make_sequence(model_name: str, n_frames: int, seed: int, noise_sigma: float | None = None, device="cpu",
                  world_seed: int | None = None, step: float = 1.0, nan_fraction: float = 0.0, columns: int | None = None, rough: bool = False,
                  az_offset: float = 0.0, travel: bool = False, street_length: float = 700.0):
    """Returns (scans: list of float32 [N_i,4] tensors, R [n,3,3], t [n,3], model).  `travel`: a non-returning drive down the street of
    `make_street_world` (`step` metres per sweep, 120 m sensor range) instead of laps of the 30 m circle."""
    model = sensor_model(model_name, columns=columns, device=device, az_offset=az_offset)
    if noise_sigma is None:
        noise_sigma = 0.01 if model_name == "VLP-16" else 0.02
    gen = torch.Generator(device=device).manual_seed(77 + seed)
    if travel:
        world = make_street_world(seed if world_seed is None else world_seed, street_length).to(device)
        R, t = trajectory_travel(n_frames, step=step, seed=seed, length=street_length)
        scans = [render_scan(world, model, R[k], t[k], noise_sigma, gen, nan_fraction, rough, max_range=STREET_MAX_RANGE, cull=True) for k in range(n_frames)]
        return scans, R, t, model
    world = make_world(seed if world_seed is None else world_seed).to(device)
    R, t = trajectory(n_frames, step=step, seed=seed, start_angle=0.37 * seed)
    scans = [render_scan(world, model, R[k], t[k], noise_sigma, gen, nan_fraction, rough) for k in range(n_frames)]
    return scans, R, t, model


def relative_pose(R, t, k):
    """Ground-truth (R_lc, t_lc) with p_last = R_lc p_curr + t_lc between frames k-1 and k."""
    Rl, tl, Rc, tc = R[k - 1], t[k - 1], R[k], t[k]
    return Rl.T @ Rc, Rl.T @ (tc - tl)
