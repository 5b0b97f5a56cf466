"""Times a few 3x3 launches (B=32) under the product library and the probe variants (tools/probe_build.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("P3_CHILD"):
    sys.path.insert(0, ROOT)
    import torch
    from pdae_amd import hip as H
    def t(op, n=10):
        H.run(op); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): H.run(op)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    out = []
    for (N, S, C, Cout) in [(32, 128, 256, 128), (32, 128, 128, 128), (32, 64, 128, 128), (32, 32, 256, 256), (32, 16, 384, 384)]:
        x = torch.randn(N, S, S, C, device="cuda"); w = torch.randn(Cout, 3, 3, C, device="cuda") / (C * 9) ** 0.5
        y = torch.empty(N, S, S, Cout, device="cuda")
        c = H.Conv(N, S, S, C, 0, Cout, math=4)
        wp = torch.empty(c.wprep_bytes(0) // 4 + 16, device="cuda")
        H.run(H.op_conv_wprep(c, w, 0, wp))
        ms = t(H.op_conv_fwd(c, x, None, w, None, y, wp=wp))
        out.append(f"{C}->{Cout}@{S}: {ms:.3f} ms {2.0*N*S*S*Cout*9*C/ms/1e9:6.1f} TF")
    print((os.environ.get("PDAE_HIP_LIB") or "x/product/x").split("/")[-2], " | ".join(out), flush=True)
else:
    for v in (sys.argv[1:] or ["", "nob", "noa", "nostage", "mfma", "ilv"]):
        env = dict(os.environ)
        if v:
            env["PDAE_HIP_LIB"] = os.path.join(ROOT, "pdae_amd", "lib", "probe_" + v, "libpdae_hip.so")
        env["P3_CHILD"] = "1"
        subprocess.call([sys.executable, os.path.abspath(__file__)], env=env)
