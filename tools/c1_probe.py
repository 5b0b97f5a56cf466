"""What bounds conv1x1 on the 128^2 skip shape: times the product library and the probe builds of tools/probe_build.py (c1_*: pieces of the
kernel compiled out, WRONG results by design) in one subprocess each.  Usage (GPU box): python tools/probe_build.py c1_ && python tools/c1_probe.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from pdae_amd import hip as H
    out = []
    for (N, S, C0, C1, Cout) in [(32, 128, 128, 128, 128), (32, 64, 256, 128, 256), (32, 32, 384, 256, 256)]:
        x0 = torch.randn(N, S, S, C0, device="cuda"); x1 = torch.randn(N, S, S, C1, device="cuda")
        w = torch.randn(Cout, 1, 1, C0 + C1, device="cuda") / (C0 + C1) ** 0.5; b = torch.randn(Cout, device="cuda")
        c = H.Conv(N, S, S, C0, C1, Cout, k=1, math=4)
        wp = torch.empty(max(c.wprep_bytes(0), 4) // 4, device="cuda")
        H.run(H.op_conv_wprep(c, w, 0, wp))
        y = torch.empty(N, S, S, Cout, device="cuda")
        op = H.op_conv_fwd(c, x0, x1, w, b, y, wp=wp)
        for _ in range(30): H.run(op)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): H.run(op)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10)
        gb = 4.0 * N * S * S * (C0 + C1 + Cout) / 1e9
        out.append(f"{S}^2 {C0}+{C1}->{Cout}: {best*1e3:6.1f} us ({gb/best:4.2f} TB/s of the full kernel's bytes)")
    print(" | ".join(out))
    sys.exit(0)
libdir = os.path.join(ROOT, "pdae_amd", "lib")
for name in ["product"] + sorted(d[6:] for d in os.listdir(libdir) if d.startswith("probe_c1_")):
    env = dict(os.environ)
    if name != "product": env["PDAE_HIP_LIB"] = os.path.join(libdir, "probe_" + name, "libpdae_hip.so")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    print(f"{name:12s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
