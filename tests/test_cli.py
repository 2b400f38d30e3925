"""rt_headless, the C++ counterpart of raytracer_headless with --backend hip."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from util import camera_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "raytracer_amd", "rt_headless")


def read_png_rgb8(path):
    """Decoder for what the image crate's RgbImage::save produces in the reference
    (raytracer_headless/src/main.rs:64-72) and rt_headless writes: 8-bit RGB, non-interlaced."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(tag + body) == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0], "chunk CRC"
        if tag == b"IHDR":
            w, h, depth, colour, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert (depth, colour, interlace) == (8, 2, 0)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, w * 3 + 1)
    assert not raw[:, 0].any(), "filter type none on every row"
    return raw[:, 1:].reshape(h, w, 3)


def test_cli_fails_loudly_without_a_gpu_or_with_bad_arguments():
    import torch
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    bad = subprocess.run([EXE, "--backend", "cpu"], capture_output=True, text=True)
    assert bad.returncode != 0 and "only --backend hip" in bad.stderr
    bad = subprocess.run([EXE, "--backend", "hip", "--bogus"], capture_output=True, text=True)
    assert bad.returncode != 0 and "usage" in bad.stderr
    if not torch.cuda.is_available():   # there is no CPU path to fall back to
        out = subprocess.run([EXE, "--backend", "hip", "--width", "16", "--height", "16", "--spp", "1", "--out", os.devnull],
                             capture_output=True, text=True)
        assert out.returncode != 0 and "no HIP device" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("scene,sky", [("default", "reference"), ("wallpaper", "wallpaper")])
def test_cli_png_equals_library_readback(rt, tmp_path, scene, sky):
    w, h, spp, seed = 96, 54, 6, 77
    png = str(tmp_path / "out.png")
    out = subprocess.run([EXE, "--backend", "hip", "--scene", scene, "--width", str(w), "--height", str(h), "--spp", str(spp),
                          "--seed", str(seed), "--sky", sky, "--chunk-spp", "4", "--out", png], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "Rendering took" in out.stdout   # raytracer_headless/src/main.rs:59-62
    img = read_png_rgb8(png)
    cam = camera_args(scene, w, h, rt)
    if scene == "wallpaper":   # the CLI keeps the headless lens (focus 10, defocus 0.6) for every scene
        cam = cam[:3] + (10.0, 0.6, w, h)
    else:                      # raytracer_headless/src/main.rs:22: the direction is normalised before Camera::new
        n = (13.0 * 13.0 + 1.5 * 1.5 + 3.0 * 3.0) ** 0.5
        cam = (cam[0], (-(13.0 / n), -(1.5 / n), -(3.0 / n))) + cam[2:]
    r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*cam), seed=seed, sky_mode=1 if sky == "wallpaper" else 0)
    r.accumulate(spp)
    assert np.array_equal(img, r.read_rgb8())
    r.close()


@pytest.mark.gpu
def test_cfg1_headless_default_render(rt, oracle, tmp_path):
    """BASELINE configs[0]: raytracer_headless's own run (raytracer_headless/src/main.rs:15-28: default
    scene, 2560x1440, depth 10, 800 spp) through the CLI with no size arguments.  The PNG equals the
    library's read-back of the same 800 samples; the first two samples of that frame, at the full
    2560x1440, are checked against the reference algorithm on the CPU (same stream: RMSE <= 2e-3);
    and the 800-sample image agrees with that 2-sample CPU image within Monte-Carlo noise."""
    seed = 0x5EED0001
    png = str(tmp_path / "output.png")
    out = subprocess.run([EXE, "--backend", "hip", "--out", png], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Rendering took" in out.stdout
    print("[cfg1]", [l for l in out.stdout.splitlines() if "took" in l or "paths/s" in l])
    img = read_png_rgb8(png)
    assert img.shape == (1440, 2560, 3)
    assert os.path.getsize(png) < 2560 * 1440 * 3           # deflated, as the image crate writes it
    n = (13.0 * 13.0 + 1.5 * 1.5 + 3.0 * 3.0) ** 0.5
    cam = ((13.0, 1.5, 3.0), (-(13.0 / n), -(1.5 / n), -(3.0 / n)), 20.0, 10.0, 0.6, 2560, 1440)
    sc = rt.Scene.generate("default")
    r = rt.Renderer(sc, rt.Camera(*cam), seed=seed, max_depth=10)
    r.accumulate(2)
    first2 = r.read_accum() / 2
    for _ in range(798 // 100 + 1):
        r.accumulate(min(100, 800 - r.spp_done))
    assert r.spp_done == 800 and np.array_equal(img, r.read_rgb8())
    full = r.read_accum() / 800
    r.close()
    osc = oracle.OracleScene(sc.objects)
    ref2, _, cnt, _ = osc.render(oracle.camera_desc(*cam), 2, 10, seed, oracle.FAITHFUL)
    ref2 = ref2 / 2
    rmse = float(np.sqrt(((np.clip(first2, 0, 1) - np.clip(ref2, 0, 1)) ** 2).mean()))
    assert rmse <= 2e-3, rmse
    # converged GPU image against the noisy CPU one: no bias (3.7 M pixels: the frame means agree to 0.5 %;
    # unclipped, since clipping a 2-sample mean and an 800-sample mean are different estimators)
    a, b = full, ref2
    assert abs(a.mean() - b.mean()) <= 5e-3 * b.mean(), (a.mean(), b.mean())
    print(f"[cfg1] 2560x1440: first two samples RMSE {rmse:.2e} against the CPU reference algorithm ({cnt['rays']} rays); "
          f"800-spp image mean {a.mean():.4f} against {b.mean():.4f}")


@pytest.mark.gpu
@pytest.mark.parametrize("split", ["spp", "tile"])
@pytest.mark.parametrize("shares", [2, 3])
def test_cli_several_gpus_from_one_process(tmp_path, split, shares):
    """rt_headless --gpus N: N contexts in one process (here all on GPU 0, taking turns), every one
    renders its share, rthip_merge_local sums them on the first: the PNG equals the one-GPU PNG
    byte for byte, for both ways of sharing and a sample count the shares do not divide."""
    w, h, spp = 200, 120, 7
    common = ["--backend", "hip", "--scene", "sample", "--width", str(w), "--height", str(h), "--spp", str(spp), "--seed", "5", "--chunk-spp", "5"]
    one, many = str(tmp_path / "one.png"), str(tmp_path / "many.png")
    a = subprocess.run([EXE] + common + ["--out", one], capture_output=True, text=True, timeout=120)
    b = subprocess.run([EXE] + common + ["--gpus", str(shares), "--devices", ",".join(["0"] * shares), "--split", split, "--out", many],
                       capture_output=True, text=True, timeout=120)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert np.array_equal(read_png_rgb8(one), read_png_rgb8(many))
    bad = subprocess.run([EXE] + common + ["--gpus", "2", "--devices", "0", "--out", many], capture_output=True, text=True)
    assert bad.returncode != 0 and "usage" in bad.stderr
