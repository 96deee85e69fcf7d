"""In-tree builds of the native libraries (no cmake, no JIT cache: the .so files travel with the repo snapshot).

* ``libalfi_hip.so``   -- the product: hand-written HIP kernels for gfx950 behind the C ABI of include/alfi_hip.h.
* ``libalfi_host.so``  -- CPU/OpenMP operator generator (input generation only, csrc/host_assemble.cpp).

``python -m alfi_amd.build`` builds both; ``__graft_entry__.build()`` calls the same functions.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
HIP_LIB = os.path.join(HERE, "libalfi_hip.so")
HOST_LIB = os.path.join(HERE, "libalfi_host.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("[alfi_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def hip_sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def hip_deps():
    inc = os.path.join(ROOT, "include")
    deps = hip_sources()
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return deps


def build_hip(force=False):
    """One object per .hip file (compiled in parallel, only when the file or a header changed), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libalfi_hip.so")
    inc = os.path.join(ROOT, "include")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    objdir = os.path.join(CSRC, ".obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I", inc, "-I", CSRC]
    jobs, objs = [], []
    for src in hip_sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + headers):
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])
    stale = set(os.listdir(objdir)) - set(os.path.basename(o) for o in objs)
    for f in stale:
        os.remove(os.path.join(objdir, f))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if jobs or stale or force or _newer(HIP_LIB, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", HIP_LIB] + objs)
    return HIP_LIB


def build_host(force=False):
    src = os.path.join(CSRC, "host_assemble.cpp")
    if force or _newer(HOST_LIB, [src]):
        _run(["g++", "-O3", "-march=x86-64-v3", "-fopenmp", "-std=c++17", "-fPIC", "-shared", "-o", HOST_LIB, src])
    return HOST_LIB


def build_all(force=False):
    build_host(force)
    build_hip(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
