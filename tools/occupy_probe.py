"""What a few busy CUs cost the persistent conv3x3y launch (the N > 1 risk): a spin kernel of k workgroups (tools/micro/occupy.hip,
built by this script's docstring command) on a side stream holds k CUs while one 128^2 128->128 Winograd-form convolution runs on the main stream.
conv3x3y needs every register of its four SIMDs, so a CU with a spinner on it cannot take a workgroup; tiles are assigned statically.
Build first: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/micro/libpdae_occupy.so tools/micro/occupy.hip"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PDAE_W1", "2")
import torch
from pdae_amd import hip as H
N, S, Cin, Cout = 32, 128, 128, 128
x = torch.randn(N, S, S, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda") / (Cin * 9) ** 0.5; b = torch.randn(Cout, device="cuda")
y = torch.empty(N, S, S, Cout, device="cuda")
c = H.Conv(N, S, S, Cin, 0, Cout, k=3, math=4)
wp = torch.empty(c.wprep_bytes(0) // 4, device="cuda"); H.run(H.op_conv_wprep(c, w, 0, wp))
op = H.op_conv_fwd(c, x, None, w, b, y, wp=wp)
for _ in range(40): H.run(op)
torch.cuda.synchronize()
import ctypes
occ = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libpdae_occupy.so"))
occ.pdae_occupy.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
sink = torch.zeros(4, device="cuda", dtype=torch.int32)
side = torch.cuda.Stream()
def timed(k, threads=256, us=8000, lds=0):
    torch.cuda.synchronize()
    if k: assert occ.pdae_occupy(k, threads, us, lds, sink.data_ptr(), side.cuda_stream) == 0
    time.sleep(0.002)                                                   # let them become resident
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): H.run(op)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5
for k in (0, 1, 4, 16, 32, 64, 0):
    print(f"{k:3d} spinning workgroups of 256 threads: {timed(k):.4f} ms per launch", flush=True)
for k in (16, 32):
    print(f"{k:3d} spinning workgroups of 64 threads:  {timed(k, threads=64):.4f} ms per launch", flush=True)
for k, th, lds in ((256, 1024, 0), (512, 1024, 0), (32, 256, 65536), (64, 256, 65536), (256, 256, 65536)):
    print(f"{k:3d} spinning workgroups of {th} threads, {lds} B of LDS: {timed(k, threads=th, lds=lds):.4f} ms per launch", flush=True)
# the spinner's own duration (is it resident for as long as asked?) and whether the convolution ran inside it
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(side):
    e0.record(side); occ.pdae_occupy(32, 256, 8000, 0, sink.data_ptr(), side.cuda_stream); e1.record(side)
c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
time.sleep(0.002)
c0.record(); H.run(op); c1.record()
torch.cuda.synchronize()
print(f"spinner alone: {e0.elapsed_time(e1):.3f} ms; convolution started {e0.elapsed_time(c0):.3f} ms and ended {e0.elapsed_time(c1):.3f} ms after the spinner's start", flush=True)
