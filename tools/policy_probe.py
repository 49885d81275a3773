import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
from tests import helpers as Hh
from tests.golden import common
import dreamer
from dv3hip import ops
class L:
    step = 0
    def scalar(self,*a): pass
    def video(self,*a,**k): pass
    def write(self,fps=False): pass
name="cfg2"; cfg=Hh.make_config(name); cfg.pretrain=0
def ds():
    while True: yield common.make_batch(name)
agent=dreamer.Dreamer(Hh.obs_space(name),None,cfg,L(),ds()).to(cfg.device); agent.requires_grad_(False)
rs=np.random.RandomState(0)
def med(tag, f, n=40, sync=False):
    ts=[]
    for _ in range(n):
        t0=time.perf_counter(); f()
        if sync: torch.cuda.synchronize()
        ts.append((time.perf_counter()-t0)*1e3)
    torch.cuda.synchronize()
    print(f"{tag:28s} median {np.median(ts):.3f} ms  min {np.min(ts):.3f}", flush=True)
for E in (1,16):
    obs={"image":rs.randint(0,256,(E,64,64,3)).astype(np.uint8),"is_first":np.zeros((E,),bool),"is_terminal":np.zeros((E,),bool)}
    out,state=agent._policy(dict(obs,is_first=np.ones((E,),bool)),None,True)
    for _ in range(3): out,state=agent._policy(obs,state,True)
    torch.cuda.synchronize()
    if E == 1:
        ops.PROFILE.by_shape=True; ops.PROFILE.start()
        agent._policy_eager(obs,state,True)
        prof=ops.PROFILE.stop()
        tot=0
        for k,v in prof.items():
            print(f"   {v['launches']:3d} x {v['ms']*1e3/v['launches']:7.1f} us  {k}"); tot+=v['launches']
        print("   dv3 launches:", tot)
    pr=agent._policy_runner; key=[k for k in pr._sig if k[0]==E][0]; st=pr._sig[key]
    med(f"E={E} step+cpu (wall)", lambda: agent._policy(obs,state,True)[0]["action"].cpu())
    med(f"E={E} load host", lambda: pr._load(st,obs,state))
    med(f"E={E} load +sync", lambda: pr._load(st,obs,state), sync=True)
    med(f"E={E} replay host", lambda: st["graph"].replay())
    med(f"E={E} replay +sync", lambda: st["graph"].replay(), sync=True)
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    ds_=[]
    for _ in range(20):
        a.record(); st["graph"].replay(); b.record(); torch.cuda.synchronize(); ds_.append(a.elapsed_time(b))
    print(f"E={E} replay device (events) median {np.median(ds_):.3f} ms")
    med(f"E={E} clone+cpu +sync", lambda: st["packed"].clone()[:6].cpu())
