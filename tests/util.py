"""Shared scene / camera presets of the test-suite."""
import numpy as np

SCENE_SEED = 0x5EED0001


def record_bar(line):
    """Measured f32 figures behind the asserted bounds (ids that differ from the oracle, pixels over the
    literal 1e-5 bar): printed, and appended to gpurun_out/f32_bars.txt when that directory exists, so
    that one run of the GPU suite leaves a record that can be kept under profiles/."""
    import os
    print(line)
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "f32_bars.txt"), "a") as f:
            f.write(line + "\n")


def camera_args(name, w, h, rt):
    """Camera::new arguments per scene: the headless camera (raytracer_headless/src/main.rs:20-28)
    for the two grid scenes, the wgpu app's start pose (raytracer_gpu/src/main.rs:70) for the rest."""
    headless = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6, w, h)
    table = {
        "default": headless,
        "sample": headless,
        "simple": ((1.0, 1.5, 3.0), rt.get_camera_rotation(-90.0, 0.0), 90.0, 10.0, 0.6, w, h),
        "glass": ((1.0, 1.5, 3.0), rt.get_camera_rotation(-90.0, -20.0), 90.0, 10.0, 0.6, w, h),
        "metal": ((1.0, 1.5, 3.0), rt.get_camera_rotation(-90.0, -20.0), 90.0, 10.0, 0.6, w, h),
        "wallpaper": ((0.0, 0.0, 0.0), rt.get_camera_rotation(-90.0, 0.0), 90.0, 10.0, 0.0, w, h),
    }
    return table[name]


SCENES = ["default", "sample", "simple", "glass", "metal", "wallpaper"]


def unpack_accum_words(words, width, height):
    """Host-side decoding of the accumulation buffer ([tile][channel][64 px], 2^-36 fixed point)."""
    tiles_x, tiles_y = (width + 7) // 8, (height + 7) // 8
    a = np.asarray(words, dtype=np.uint64).reshape(tiles_y, tiles_x, 3, 8, 8)
    img = a.transpose(0, 3, 1, 4, 2).reshape(tiles_y * 8, tiles_x * 8, 3)[:height, :width]
    return img.astype(np.float64) / 2.0 ** 36


def pack_accum_words(img, width, height):
    """Inverse of unpack_accum_words for an [H, W, 3] array of sums."""
    tiles_x, tiles_y = (width + 7) // 8, (height + 7) // 8
    full = np.zeros((tiles_y * 8, tiles_x * 8, 3), dtype=np.uint64)
    full[:height, :width] = np.floor(np.asarray(img) * 2.0 ** 36 + 0.5).astype(np.uint64)
    return full.reshape(tiles_y, 8, tiles_x, 8, 3).transpose(0, 2, 4, 1, 3).reshape(-1).copy()


def random_scene(rt, seed):
    """A small adversarial scene for differential tests: spheres and cubes of all three
    materials, with the cases that make traversal order matter: concentric glass shells (negative
    radius), stand-alone inverted spheres, cubes with coplanar faces and shared centres, flat and
    zero-size cubes, negative half extents, objects around and behind the camera."""
    g = np.random.default_rng(seed)
    s = rt.Scene()

    def material():
        k = g.integers(0, 3)
        if k == 0:
            return rt.Material.lambertian(tuple(g.uniform(0.1, 0.9, 3)), float(g.choice([0.0, 0.0, 2.0])))
        if k == 1:
            return rt.Material.metal(tuple(g.uniform(0.5, 1.0, 3)), float(g.choice([0.0, 0.2])))
        return rt.Material.dielectric(1.5)

    n = int(g.integers(6, 60))
    grid = g.integers(0, 2) == 1  # snap centres and sizes to a coarse grid: exact ties
    for _ in range(n):
        c = g.uniform(-4, 4, 3)
        if grid:
            c = np.round(c * 2) / 2
        kind = g.integers(0, 10)
        if kind < 4:
            r = float(g.choice([0.25, 0.5, 1.0])) if grid else float(g.uniform(0.1, 1.2))
            s.add_sphere(tuple(c), r, material())
            if g.random() < 0.3:   # hollow shell, src/scene.rs:149
                s.add_sphere(tuple(c), -0.9 * r, rt.Material.dielectric(1.5))
        elif kind == 4:
            s.add_sphere(tuple(c), -float(g.uniform(0.2, 1.0)), material())
        else:
            he = g.choice([0.25, 0.5, 1.0], 3) if grid else g.uniform(0.05, 1.0, 3)
            if g.random() < 0.1:
                he[g.integers(0, 3)] = 0.0
            if g.random() < 0.05:
                he = -he
            s.add_cube(tuple(c), tuple(he), material())
    if g.random() < 0.7:
        s.add_cube((0.0, -5.0 - 100.0, 0.0), (100.0, 100.0, 100.0), rt.Material.lambertian((0.5, 0.5, 0.5), 0.0))
    origin = tuple(g.uniform(-6, 6, 3))
    direction = tuple(-np.asarray(origin) + g.uniform(-1, 1, 3))
    cam = (origin, direction, float(g.uniform(30, 90)), float(g.uniform(2, 10)), float(g.choice([0.0, 0.6])))
    return s, cam
