#!/usr/bin/env python3
"""Build every native extension in-tree for gfx950.

* ``areal_vllm_amd/_C/libareal_hip.so``   - all HIP kernels (``csrc/hip/*.hip``), C ABI, loaded with ctypes.
* ``areal_vllm_amd/_C/_runtime*.so``      - C++ host runtime (``csrc/runtime/*.cpp``): inference scheduler /
  paged block manager, sequence packing, interval merging; pybind11 module.

``hipcc`` cross-compiles for gfx950 without a GPU, so this runs on a CPU-only box. Objects are
cached by source mtime under ``build/``.
"""
import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent
HIP_DIR = ROOT / "csrc" / "hip"
RT_DIR = ROOT / "csrc" / "runtime"
OUT_DIR = ROOT / "areal_vllm_amd" / "_C"
BUILD_DIR = ROOT / "build"
ARCH = os.environ.get("AREAL_HIP_ARCH", "gfx950")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(src: Path, dst: Path, extra=()) -> bool:
    if not dst.exists():
        return True
    t = dst.stat().st_mtime
    return any(p.stat().st_mtime > t for p in (src, *extra))


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(map(str, cmd)) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]}")
    return r


def build_hip(verbose=True) -> Path:
    BUILD_DIR.mkdir(exist_ok=True)
    OUT_DIR.mkdir(exist_ok=True)
    headers = list(HIP_DIR.glob("*.hpp"))
    srcs = sorted(HIP_DIR.glob("*.hip"))
    objs = []
    jobs = []
    for s in srcs:
        o = BUILD_DIR / (s.stem + ".o")
        objs.append(o)
        if _newer(s, o, headers):
            jobs.append([HIPCC, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-c", str(s), "-o", str(o)])
    if jobs:
        if verbose:
            print(f"[build_ext] compiling {len(jobs)} HIP source(s) for {ARCH}")
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    lib = OUT_DIR / "libareal_hip.so"
    manifest = BUILD_DIR / "libareal_hip.objs"  # what the library was linked from: a deleted source must relink it
    linked_from = "\n".join(o.name for o in objs)
    stale = not manifest.exists() or manifest.read_text() != linked_from
    if jobs or stale or not lib.exists() or any(o.stat().st_mtime > lib.stat().st_mtime for o in objs):
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(lib), *map(str, objs)])
        manifest.write_text(linked_from)
        if verbose:
            print(f"[build_ext] linked {lib}")
    return lib


def build_runtime(verbose=True) -> Path:
    import pybind11

    OUT_DIR.mkdir(exist_ok=True)
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    out = OUT_DIR / f"_runtime{suffix}"
    srcs = sorted(RT_DIR.glob("*.cpp"))
    headers = list(RT_DIR.glob("*.hpp"))
    if not srcs:
        return out
    if any(_newer(s, out, headers) for s in srcs):
        cmd = [
            os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden",
            f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}",
            *map(str, srcs), "-o", str(out), "-pthread",
        ]
        if verbose:
            print("[build_ext] compiling C++ runtime")
        _run(cmd)
    return out


def build_stress(sanitize: str = "address,undefined", verbose=True) -> Path:
    """Stand-alone stress test of the scheduler / block manager, built with host sanitizers
    (``address,undefined`` by default; the runtime is driven by one thread, so no thread sanitizer)."""
    BUILD_DIR.mkdir(exist_ok=True)
    src = RT_DIR / "tests" / "stress_test.cpp"
    out = BUILD_DIR / f"sched_stress_{sanitize.replace(',', '_') or 'plain'}"
    if _newer(src, out, list(RT_DIR.glob("*.hpp"))):
        flags = [f"-fsanitize={sanitize}", "-fno-omit-frame-pointer"] if sanitize else []
        if verbose:
            print(f"[build_ext] compiling scheduler stress test ({sanitize or 'no sanitizer'})")
        _run([os.environ.get("CXX", "g++"), "-std=c++17", "-O1", "-g", *flags, f"-I{RT_DIR}", str(src), "-o", str(out)])
    return out


def build_all(verbose=True):
    return build_hip(verbose), build_runtime(verbose)


if __name__ == "__main__":
    build_all()
