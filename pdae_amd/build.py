"""Builds pdae_amd/lib/libpdae_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m pdae_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libpdae_hip.so")
SOURCES = ["api.hip", "igemm.hip", "conv3x3p.hip", "conv3x3r.hip", "conv3x3y.hip", "wprep.hip", "conv3x3w.hip", "conv3x3v.hip", "conv1x1.hip", "convhead.hip", "convedge.hip", "skinny.hip", "norm.hip", "mlp.hip", "elementwise.hip", "metric.hip", "image.hip", "attention.hip", "comm.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    tt = os.path.getmtime(target)
    return any(os.path.getmtime(s) > tt for s in src_list)


def build_library(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "pdae_hip.h"))
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer([src] + hdrs, obj):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
    if force or procs or _newer(objs, LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
