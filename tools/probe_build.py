"""Builds timing-probe variants of libpdae_hip.so with -DPDAE_PROBE_BUILD (WRONG RESULTS by design: pieces of the conv3x3p / conv3x3w main loops are compiled out to see what bounds
it).  Usage: python tools/probe_build.py  ->  pdae_amd/lib/probe_<name>/libpdae_hip.so;  select with PDAE_HIP_LIB=<path> (tools only)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pdae_amd.build import CSRC, LIBDIR, SOURCES, HIPCC, FLAGS
P3, W3, R3, AT, Y3, C1, V3 = "conv3x3p.hip", "conv3x3w.hip", "conv3x3r.hip", "attention.hip", "conv3x3y.hip", "conv1x1.hip", "conv3x3v.hip"
# (the wn_* / x_* variants of round 4 built winograd.hip / conv3x3x.hip, which left the library in round 5: tools/probes/r04_winograd/README.md)
VARIANTS = {"nob": (P3, ["-DPDAE_PROBE_NOB"]), "noa": (P3, ["-DPDAE_PROBE_NOA"]), "nostage": (P3, ["-DPDAE_PROBE_NOSTAGE"]),
            "mfma": (P3, ["-DPDAE_PROBE_NOB", "-DPDAE_PROBE_NOA", "-DPDAE_PROBE_NOSTAGE"]), "clustered": (P3, ["-DPDAE_P3_CLUSTERED"]),
            "w3_nomma": (W3, ["-DPDAE_W3_PROBE_NOMMA"]), "w3_nostage": (W3, ["-DPDAE_W3_PROBE_NOSTAGE"]), "w3_noload": (W3, ["-DPDAE_W3_PROBE_NOLOAD"]),
            "w3_6taps": (W3, ["-DPDAE_W3_PROBE_6TAPS"]),
            "v_nomma": (V3, ["-DPDAE_V_PROBE_NOMMA"]), "v_nostage": (V3, ["-DPDAE_V_PROBE_NOSTAGE"]), "v_noload": (V3, ["-DPDAE_V_PROBE_NOLOAD"]),
            "v_mmaonly": (V3, ["-DPDAE_V_PROBE_NOSTAGE", "-DPDAE_V_PROBE_NOLOAD"]),
            "w3_mmaonly": (W3, ["-DPDAE_W3_PROBE_NOSTAGE", "-DPDAE_W3_PROBE_NOLOAD"]),
            "r_noa": (R3, ["-DPDAE_R_PROBE_NOA"]), "r_nob": (R3, ["-DPDAE_R_PROBE_NOB"]), "r_nogload": (R3, ["-DPDAE_R_PROBE_NOGLOAD"]),
            "r_noconv": (R3, ["-DPDAE_R_PROBE_NOCONV"]), "r_nodrain": (R3, ["-DPDAE_R_PROBE_NODRAIN"]),
            "at_nonn": (AT, ["-DPDAE_AT_PROBE_NONN"]), "at_nont": (AT, ["-DPDAE_AT_PROBE_NONT"]), "at_nnnoload": (AT, ["-DPDAE_AT_PROBE_NNNOLOAD"]),
            "at_nnnomma": (AT, ["-DPDAE_AT_PROBE_NNNOMMA"]), "at_nosched": (AT, ["-DPDAE_AT_PROBE_NOSCHED"]),
            "y_noa": (Y3, ["-DPDAE_Y_PROBE_NOA"]), "y_nob": (Y3, ["-DPDAE_Y_PROBE_NOB"]), "y_nostage": (Y3, ["-DPDAE_Y_PROBE_NOCONV", "-DPDAE_Y_PROBE_NOGLOAD"]),
            "y_nogload": (Y3, ["-DPDAE_Y_PROBE_NOGLOAD"]), "y_noconv": (Y3, ["-DPDAE_Y_PROBE_NOCONV"]),
            "y_nobgl": (Y3, ["-DPDAE_Y_PROBE_NOB", "-DPDAE_Y_PROBE_NOGLOAD"]), "y_noabgl": (Y3, ["-DPDAE_Y_PROBE_NOA", "-DPDAE_Y_PROBE_NOB", "-DPDAE_Y_PROBE_NOGLOAD"]),
            "y_noepi": (Y3, ["-DPDAE_Y_PROBE_NOEPI"]),
            "y_mfma": (Y3, ["-DPDAE_Y_PROBE_NOA", "-DPDAE_Y_PROBE_NOB", "-DPDAE_Y_PROBE_NOCONV", "-DPDAE_Y_PROBE_NOGLOAD", "-DPDAE_Y_PROBE_NOEPI"]),
            "c1_nob": (C1, ["-DPDAE_C1_PROBE_NOB"]), "c1_nomma": (C1, ["-DPDAE_C1_PROBE_NOMMA"]), "c1_noconv": (C1, ["-DPDAE_C1_PROBE_NOCONV"]),
            "c1_nostore": (C1, ["-DPDAE_C1_PROBE_NOSTORE"]), "c1_stream": (C1, ["-DPDAE_C1_PROBE_NOB", "-DPDAE_C1_PROBE_NOMMA", "-DPDAE_C1_PROBE_NOCONV"]),
            "c1_loadonly": (C1, ["-DPDAE_C1_PROBE_NOB", "-DPDAE_C1_PROBE_NOMMA", "-DPDAE_C1_PROBE_NOCONV", "-DPDAE_C1_PROBE_NOSTORE"]),
            "r_24u": (R3, ["-DPDAE_R_PROBE_24U"]),
            "r_mfma": (R3, ["-DPDAE_R_PROBE_NOA", "-DPDAE_R_PROBE_NOB", "-DPDAE_R_PROBE_NOGLOAD", "-DPDAE_R_PROBE_NOCONV", "-DPDAE_R_PROBE_NODRAIN"])}
only = sys.argv[1:]
for name, (src, defs) in VARIANTS.items():
    if only and not any(name.startswith(o) for o in only): continue
    d = os.path.join(LIBDIR, "probe_" + name)
    os.makedirs(d, exist_ok=True)
    obj = os.path.join(d, src.replace(".hip", ".o"))
    subprocess.check_call([HIPCC] + FLAGS + ["-DPDAE_PROBE_BUILD"] + defs + ["-c", os.path.join(CSRC, src), "-o", obj])
    objs = [obj if s == src else os.path.join(LIBDIR, s.replace(".hip", ".o")) for s in SOURCES]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libpdae_hip.so")] + objs)
    print("built", d)
