import os, sys, time, re, subprocess, threading, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import self_forcing_amd as sfa
from self_forcing_amd import vae_weights as vw
samples=[]; stop=False
def sampler():
    while not stop:
        out=subprocess.run(["rocm-smi","--showclocks","--showpower","-d","0"],capture_output=True,text=True).stdout
        c=re.search(r"sclk clock level[^\n]*\((\d+)Mhz\)",out); w=re.search(r"Graphics Package Power \(W\): ([0-9.]+)",out)
        if c and w: samples.append((time.perf_counter(),int(c.group(1)),float(w.group(1))))
vae=sfa.WanVAEWrapper(vw.synth_vae_state_dict(vw.WAN_VAE,seed=0),device="cuda:0")
lat=torch.randn(1,21,16,60,104).to(torch.bfloat16).cuda()
vae.decode_to_pixel(lat); torch.cuda.synchronize()
th=threading.Thread(target=sampler); th.start()
t0=time.perf_counter(); n=0
while time.perf_counter()-t0<8:
    vae.decode_to_pixel(lat); n+=1
torch.cuda.synchronize(); el=time.perf_counter()-t0
stop=True; th.join()
seg=[s for s in samples if s[0]>t0+1]
clk=sorted(s[1] for s in seg); pw=sorted(s[2] for s in seg)
print(f"VAE decode loop: {el/n*1e3:.1f} ms per clip; sclk median {clk[len(clk)//2]} MHz; power median {pw[len(pw)//2]:.0f} W (max {pw[-1]:.0f}), {len(seg)} samples")
