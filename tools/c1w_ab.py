"""conv1x1w (1x1 weight gradient) timing on the step's shapes, product library against another build (PDAE_HIP_LIB of the baseline as argv[1]),
one subprocess each.  Usage (GPU box): python tools/c1w_ab.py pdae_amd/lib/<dir>/libpdae_hip.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from pdae_amd import hip as H
    out = []
    for (N, S, C0, C1, Cout) in [(32, 128, 128, 128, 128), (32, 64, 256, 128, 256), (32, 32, 384, 256, 256), (32, 16, 384, 0, 1152), (32, 16, 512, 384, 384), (32, 8, 512, 512, 512)]:
        x0 = torch.randn(N, S, S, C0, device="cuda"); x1 = torch.randn(N, S, S, C1, device="cuda") if C1 else None
        dy = torch.randn(N, S, S, Cout, device="cuda")
        c = H.Conv(N, S, S, C0, C1, Cout, k=1, math=int(os.environ.get("C1W_MATH", "4")))
        wsb = c.wgrad_ws_bytes()
        ws = torch.empty(max(wsb, 4) // 4, device="cuda"); dw = torch.empty(Cout, 1, 1, C0 + C1, device="cuda"); db = torch.empty(Cout, device="cuda")
        am = torch.zeros(4, device="cuda"); H.run(H.op_amax(dy, dy.numel(), am))
        op = H.op_conv_wgrad(c, x0, x1, dy, dw, ws, wsb, db=db, dy_amax=am)
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
        out.append(f"{S}^2 {C0}+{C1}->{Cout}: {best*1e3:6.1f} us {gb/best:4.2f} TB/s  |dw| {float(dw.abs().sum()):.6e}")
    print("\n".join(out))
    sys.exit(0)
for name, lib in [("baseline", sys.argv[1] if len(sys.argv) > 1 else None), ("product", None)]:
    if name == "baseline" and lib is None: continue
    env = dict(os.environ)
    if lib: env["PDAE_HIP_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    print(f"--- {name}\n{r.stdout.strip() or r.stderr.strip()[-600:]}", flush=True)
