import subprocess, sys, re
out = subprocess.run(["/opt/rocm/bin/hipcc","--offload-arch=gfx950","-O3","-std=c++17","-fPIC","-fno-gpu-rdc","-Rpass-analysis=kernel-resource-usage","-c",sys.argv[1],"-o","/tmp/res_tmp.o"],capture_output=True,text=True).stderr
cur={}
for line in out.splitlines():
    m=re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur={"name":t.split(":",1)[1].strip()}
    elif ":" in t:
        k,v=t.split(":",1); cur[k.strip()]=v.strip()
        if k.strip().startswith("LDS Size"):
            n=cur["name"]
            if len(sys.argv)<3 or re.search(sys.argv[2],n):
                print(n[:60], "sgpr",cur.get("TotalSGPRs"),"vgpr",cur.get("VGPRs"),"agpr",cur.get("AGPRs"),"occ",cur.get("Occupancy [waves/SIMD]"),"sspill",cur.get("SGPRs Spill"),"vspill",cur.get("VGPRs Spill"))
