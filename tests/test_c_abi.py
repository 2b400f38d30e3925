"""The C ABI from plain C: tests/c_abi.c is built with gcc against include/rthip.h and
librthip.so (no Python in between) and calls every declared function."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_build", "c_abi")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "raytracer_amd")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-o", EXE,
                           os.path.join(ROOT, "tests", "c_abi.c"), "-L", libdir, "-lrthip", f"-Wl,-rpath,{libdir}"])


def test_c_program_calls_every_declared_function(rt):
    header = open(os.path.join(ROOT, "include", "rthip.h")).read()
    declared = set(re.findall(r"\b(rthip_[a-z0-9_]+)\s*\(", header))
    source = open(os.path.join(ROOT, "tests", "c_abi.c")).read()
    missing = {name for name in declared if not re.search(r"\b" + name + r"\s*\(", source)}
    assert not missing, f"tests/c_abi.c does not call {sorted(missing)}"


def test_c_abi_host_functions(rt):
    build()
    out = subprocess.run([EXE, "host"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "host: ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_c_abi_on_the_gpu(rt):
    if not os.path.exists(EXE):
        build()
    out = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "gpu: ok" in out.stdout, out.stdout + out.stderr
