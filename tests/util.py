"""Shared scene / camera presets of the test-suite."""
import numpy as np

SCENE_SEED = 0x5EED0001


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
