"""Compressed instruction stream of one kernel from a --save-temps .s file: M mfma, v VALU, s SALU, r ds_read, w ds_write, b buffer/global load-store, W s_waitcnt, N s_nop,
| s_barrier, a accvgpr move, S scratch.  Usage: python tools/isa_stream.py file.s kernel_symbol [first_line last_line]"""
import sys
f, name = sys.argv[1], sys.argv[2]
lines = open(f).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(name + ":")][0]
end = [i for i, l in enumerate(lines) if i > start and "s_endpgm" in l][0]
seg = lines[start:end]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(seg)
out = []
for i, l in enumerate(seg):
    if i < lo or i >= hi:
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        if t.endswith(":") and not t.startswith(";"):
            out.append("\n[%d %s]" % (i, t))
        continue
    if t.startswith("v_mfma"): out.append("M")
    elif t.startswith("s_barrier"): out.append("|\n")
    elif t.startswith("s_waitcnt"): out.append("W")
    elif t.startswith("s_nop"): out.append("N")
    elif t.startswith("scratch_"): out.append("S")
    elif t.startswith("v_accvgpr"): out.append("a")
    elif t.startswith("ds_read") or t.startswith("ds_load"): out.append("r")
    elif t.startswith("ds_write") or t.startswith("ds_store"): out.append("w")
    elif t.startswith("buffer_") or t.startswith("global_"): out.append("b")
    elif t.startswith("v_"): out.append("v")
    elif t.startswith("s_"): out.append("s")
    else: out.append("?")
print("".join(out))
