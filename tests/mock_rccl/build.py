"""Builds tests/mock_rccl/libmock_rccl.so (TEST INFRASTRUCTURE: a shared-memory stand-in for librccl, see mock_rccl.cpp)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    src, out = os.path.join(HERE, "mock_rccl.cpp"), os.path.join(HERE, "libmock_rccl.so")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    hipcc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950",
                           "-D__HIP_PLATFORM_AMD__", src, "-o", out, "-lrt"])
    return out


if __name__ == "__main__":
    print(build())
