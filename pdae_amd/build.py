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
# sources whose device assembly is scanned for the store-operand hazards of isa_hazard.py after compilation: a hit FAILS the build
HAZARD_SCAN = ["conv3x3y.hip"]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    tt = os.path.getmtime(target)
    return any(os.path.getmtime(s) > tt for s in src_list)


def _hazard_scan(src):
    """The device assembly hipcc left next to the object (--save-temps=obj) must be free of both store-operand hazard patterns; all temporaries
    are removed afterwards."""
    from .isa_hazard import lds_store_hazard_sites, output_store_hazard_sites
    stem = src.replace(".hip", "")
    asm_file = None
    for fn in os.listdir(LIBDIR):
        if fn.startswith(stem + "-") and (fn.endswith(".s") and "amdgcn" in fn):
            asm_file = os.path.join(LIBDIR, fn)
        elif fn.startswith(stem + "-") or fn.startswith(stem + ".hip-"):
            os.remove(os.path.join(LIBDIR, fn))
    if asm_file is None:
        raise RuntimeError(f"{src}: no device assembly next to the object (hipcc --save-temps=obj): the hazard scan cannot run")
    asm = open(asm_file).read()
    os.remove(asm_file)
    sites = lds_store_hazard_sites(asm) + output_store_hazard_sites(asm)
    if sites:
        os.remove(os.path.join(LIBDIR, stem + ".o"))
        raise RuntimeError(f"{src}: store-operand hazard pattern in the compiled kernel (pdae_amd/isa_hazard.py): {sites[:3]}")


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
            cmd = [HIPCC] + FLAGS + (["--save-temps=obj"] if s in HAZARD_SCAN else []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
        if s in HAZARD_SCAN:
            _hazard_scan(s)
    if force or procs or _newer(objs, LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
