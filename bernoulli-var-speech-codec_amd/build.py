"""Builds csrc/*.hip into csrc/libbvcodec_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m bvcodec.build            # or: __graft_entry__.build()
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libbvcodec_hip.so")
SOURCES = ["bvcodec_abi.hip", "k_gemm.hip", "k_flow.hip", "k_frontend.hip", "k_vocoder.hip"]
HEADERS = ["bvc_internal.h", os.path.join("..", "..", "include", "bvcodec.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            extra = (["-mllvm", "-amdgpu-kernarg-preload-count=16"]
                     if (src == "k_gemm.hip" and os.environ.get("BVC_KERNARG_PRELOAD", "0") == "1") else [])
            if src == "k_flow.hip" and os.environ.get("BVC_FLOW_WAVES"):       # experiments: register budget of the persistent kernel
                extra = extra + ["-DBVC_FLOW_WAVES_PER_SIMD=" + os.environ["BVC_FLOW_WAVES"]]
            cmd = [hipcc] + FLAGS + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
