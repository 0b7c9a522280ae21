"""In-tree native builds (gfx950 only).

`build_hip()`  -> ray_tracing_octrees_amd/librto_hip.so   (hipcc, device + C ABI)
`build_host()` -> ray_tracing_octrees_amd/librto_host.so  (g++, the C++ drop-in host layer + its C shim)

The .so files are git-ignored but travel with the tree to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
LIB_HIP = os.path.join(PKG, "librto_hip.so")
LIB_HOST = os.path.join(PKG, "librto_host.so")

# -ffp-contract=off: one IEEE operation per source operator (the exactness contract).
# HIP's default -fhip-fp32-correctly-rounded-divide-sqrt stays on; no fast-math anywhere.
# -fno-slp-vectorize (round 5): the packed operations these kernels want are written as 2-vectors in the source; what the SLP
# vectoriser adds on top (pairs of unrelated scalars packed through v_mov shuffles into v_pk_* at 4.4 cycles against 2 x 2.4) costs
# config 5 5 %, configs 2 and 4 2 % (A/B, same box).  No effect on results: it only groups identical IEEE operations.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17"]
HOST_FLAGS = ["-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall", "-Wextra"]


def preload_torch_runtime() -> bool:
    """Load torch's bundled ROCm runtime BEFORE librto_hip.so when torch is installed.

    PyTorch wheels ship their own libamdhip64.so/libhsa-runtime64.so and libtorch_hip.so asks for the
    unversioned name, so a system copy that is already mapped is not reused: the process ends up with two
    HIP runtimes and the second one finds no GPU.  In the other order glibc resolves our
    NEEDED libamdhip64.so.7 to torch's already-loaded copy by SONAME and both sides share one runtime --
    which is also what makes torch tensors' data_ptr()/streams valid arguments of the C ABI.
    Set RTO_NO_TORCH=1 to skip (pure C/C++ processes never need this)."""
    if os.environ.get("RTO_NO_TORCH") == "1":
        return False
    try:
        import torch  # noqa: F401
        return True
    except Exception:
        return False


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _sources(d: str, exts: tuple[str, ...]) -> list[str]:
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(exts))


def build_hip(force: bool = False, verbose: bool = False) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    deps = _sources(CSRC, (".hip", ".h", ".inc")) + [os.path.join(HOST, "rtmath.h"), os.path.join(ROOT, "include", "rto_hip.h")]
    if not force and _newer(LIB_HIP, deps):
        return LIB_HIP
    extra = os.environ.get("RTO_HIP_EXTRA_FLAGS", "").split()      # developer aid: A/B builds such as -DRTO_STAMP
    cmd = [hipcc, *HIP_FLAGS, *extra, os.path.join(CSRC, "rto_api.hip"), "-o", LIB_HIP]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_HIP


def build_host(force: bool = False, verbose: bool = False) -> str:
    srcs = _sources(HOST, (".cpp",))
    deps = srcs + _sources(HOST, (".h",)) + [os.path.join(ROOT, "include", "rto_hip.h")]
    if not force and _newer(LIB_HOST, deps):
        return LIB_HOST
    cmd = ["g++", *HOST_FLAGS, "-I", os.path.join(ROOT, "include"), *srcs, "-o", LIB_HOST, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_HOST


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip(force, verbose)
    if os.path.isdir(HOST) and _sources(HOST, (".cpp",)):
        build_host(force, verbose)


if __name__ == "__main__":
    build_all(force=True, verbose=True)
