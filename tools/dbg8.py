import sys, os, math
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
from pdae_amd import hip as H
def nhwc(t): return t.permute(0,2,3,1).contiguous()
torch.manual_seed(0)
N,Hh,W,Cin,Cout = 1,64,48,96,256
x=torch.randn(N,Cin,Hh,W); w=torch.randn(Cout,Cin,3,3)/math.sqrt(9*Cin); b=torch.randn(Cout)*0.1; res=torch.randn(N,Cout,Hh,W)
yc=F.conv2d(x.double(),w.double(),b.double(),padding=1)
c=H.Conv(N,Hh,W,Cin,0,Cout,k=3,math=4)
xd,wd,bd,resd=nhwc(x).cuda(),nhwc(w).cuda(),b.cuda(),nhwc(res).cuda()
for mode in (2,3):
    H.set_knob("PDAE_W1", mode)
    wp=torch.empty(c.wprep_bytes(0,force=True)//4,device="cuda"); H.run(H.op_conv_wprep(c,wd,0,wp))
    for rm in (0,1):
        y=torch.full((N,Hh,W,Cout),float("nan"),device="cuda")
        H.run(H.op_conv_fwd(c,xd,None,wd,bd,y,res=resd if rm else None,res_mode=rm,wp=wp))
        yy=y.permute(0,3,1,2).double().cpu()
        d0=(yy-yc); d1=(yy-yc-res.double())
        print("mode",mode,"res",rm,"err vs conv",float(d0.abs().max()),"vs conv+res",float(d1.abs().max()))
        if rm and float(d1.abs().max())>1e-3:
            bad=(d1.abs()>1e-3)[0]          # [C][H][W]
            print(" bad fraction",float(bad.float().mean()),"by row%8",[round(float(bad[:,r::8,:].float().mean()),3) for r in range(8)],"by col%16",[round(float(bad[:,:,cx::16].float().mean()),2) for cx in range(16)])
            print(" by channel block of 32:",[round(float(bad[k*32:(k+1)*32].float().mean()),2) for k in range(Cout//32)])
            # is the added residual some other pixel's?
            add=(yy-yc)[0]; r0=res.double()[0]
            for dy in (0,8,-8,16):
                for dx in (0,):
                    rs=torch.roll(r0,shifts=(dy,dx),dims=(1,2))
                    print("  shift rows",dy,"match frac",float(((add-rs).abs()<1e-3).float().mean()))
            idx = bad.nonzero()[:400]
            import collections
            cnt = collections.Counter()
            for (cc, yy_, xx_) in idx.tolist():
                a_ = float(add[cc, yy_, xx_]); found = "?"
                for dy in range(-8, 9):
                    for dx in range(-16, 17):
                        y2, x2 = yy_ + dy, xx_ + dx
                        if 0 <= y2 < Hh and 0 <= x2 < W and abs(float(r0[cc, y2, x2]) - a_) < 1e-4: found = (dy, dx)
                if found == "?" and abs(a_) < 1e-4: found = "zero"
                cnt[found] += 1
            print("  what was added instead (dy, dx):", cnt.most_common(8))
            tiles = collections.Counter(((yy_ // 8), (xx_ // 16)) for (cc, yy_, xx_) in bad.nonzero().tolist())
            print("  bad per tile:", sorted(tiles.items())[:30])
            lanes = collections.Counter((cc % 32 // 4, yy_ % 8, xx_ % 16) for (cc, yy_, xx_) in bad.nonzero().tolist())
            print("  bad by (channel quad in wave, row, col):", sorted(lanes.items(), key=lambda kv: -kv[1])[:20])
