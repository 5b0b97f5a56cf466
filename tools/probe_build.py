"""Builds timing-probe variants of libpdae_hip.so (WRONG RESULTS by design: pieces of the conv3x3p main loop are compiled out to see what bounds
it).  Usage: python tools/probe_build.py  ->  pdae_amd/lib/probe_<name>/libpdae_hip.so;  select with PDAE_HIP_LIB=<path> (tools only)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pdae_amd.build import CSRC, LIBDIR, SOURCES, HIPCC, FLAGS
VARIANTS = {"nob": ["-DPDAE_PROBE_NOB"], "noa": ["-DPDAE_PROBE_NOA"], "nostage": ["-DPDAE_PROBE_NOSTAGE"],
            "mfma": ["-DPDAE_PROBE_NOB", "-DPDAE_PROBE_NOA", "-DPDAE_PROBE_NOSTAGE"], "clustered": ["-DPDAE_P3_CLUSTERED"]}
for name, defs in VARIANTS.items():
    d = os.path.join(LIBDIR, "probe_" + name)
    os.makedirs(d, exist_ok=True)
    obj = os.path.join(d, "conv3x3p.o")
    subprocess.check_call([HIPCC] + FLAGS + defs + ["-c", os.path.join(CSRC, "conv3x3p.hip"), "-o", obj])
    objs = [obj if s == "conv3x3p.hip" else os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in SOURCES]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libpdae_hip.so")] + objs)
    print("built", d)
