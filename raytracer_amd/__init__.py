"""raytracer_amd — MI355X-native path-tracing hot path behind the API of Peanutt42/raytracer.

Host-side mirror of the reference's public interface (src/lib.rs:1-19): `Scene`, `Material`,
`Camera`, `BVH`, the scene factories and `get_camera_rotation`, plus `Renderer`, which
replaces the callers' frame loops (raytracer_headless/src/main.rs:34-57,
raytracer_runtime/src/main.rs:69-94).  All computing happens in librthip.so (HIP, gfx950).
"""
from .api import (BVH, Camera, Material, Renderer, Scene, combine_spheres_and_cubes, create_10_metallic_scene,
                  create_glass_scene, create_sample_scene, create_simple_scene, create_synthetic_scene,
                  create_wallpaper_scene, get_camera_rotation, bvh_gate_count, merge_local)
from ._native import F32, F64, F32_RAW, MEGAKERNEL, SKY_REFERENCE, SKY_WALLPAPER, WAVEFRONT, RthipError

__all__ = ["BVH", "Camera", "Material", "Renderer", "Scene", "combine_spheres_and_cubes", "create_10_metallic_scene",
           "create_glass_scene", "create_sample_scene", "create_simple_scene", "create_synthetic_scene",
           "create_wallpaper_scene", "get_camera_rotation", "bvh_gate_count", "merge_local", "F32", "F64", "F32_RAW", "MEGAKERNEL", "WAVEFRONT", "SKY_REFERENCE", "SKY_WALLPAPER",
           "RthipError"]
