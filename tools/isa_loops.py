"""Per kernel of a --save-temps .s file: instruction mix of the innermost (Depth=2) loop and of everything else -- scratch traffic inside the chunk
loop is what matters (tests/test_kernel_resources_cpu.py allows spills around it).  Usage: python tools/isa_loops.py file.s [name-substring]"""
import re, sys
f = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
L = open(f).read().split("\n")
starts = [i for i, l in enumerate(L) if re.match(r"^_Z\w+:", l) and sub in l]
for st in starts:
    en = [i for i, l in enumerate(L) if i > st and "s_endpgm" in l][0]
    inner = {"mfma": 0, "scratch": 0, "valu": 0, "vmem": 0, "ds": 0}; outer = dict(inner)
    depth2 = False
    for i in range(st, en):
        t = L[i].strip()
        m = re.match(r"^\.LBB\d+_\d+:", t)
        if m:
            j, txt = i + 1, t
            while j < en and L[j].strip().startswith(";"):
                txt += L[j]; j += 1
            depth2 = "Depth=2" in txt
            continue
        d = inner if depth2 else outer
        if t.startswith("v_mfma"): d["mfma"] += 1
        elif t.startswith("scratch_"): d["scratch"] += 1
        elif t.startswith("buffer_") or t.startswith("global_"): d["vmem"] += 1
        elif t.startswith("ds_"): d["ds"] += 1
        elif t.startswith("v_"): d["valu"] += 1
    print(L[st].rstrip(":"), "\n   inner loop:", inner, "\n   elsewhere: ", outer)
