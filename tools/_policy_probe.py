import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
from tests import helpers as Hh
from tests.golden import common
import dreamer
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
def series(tag, f, n=10, sync=True):
    ts=[]
    for _ in range(n):
        t0=time.perf_counter(); r=f();
        if sync: torch.cuda.synchronize()
        ts.append((time.perf_counter()-t0)*1e3)
    print(tag, " ".join("%.2f"%t for t in ts), flush=True)
for E in (1,4):
    obs={"image":rs.randint(0,256,(E,64,64,3)).astype(np.uint8),"is_first":np.zeros((E,),bool),"is_terminal":np.zeros((E,),bool)}
    out,state=agent._policy_eager(dict(obs,is_first=np.ones((E,),bool)),None,True)
    for _ in range(3): out,state=agent._policy_eager(obs,state,True)
    torch.cuda.synchronize()
    series(f"E={E} eager-before", lambda: agent._policy_eager(obs,state,True))
    out,_=agent._policy(obs,state,True); torch.cuda.synchronize()
    series(f"E={E} graph", lambda: agent._policy(obs,state,True))
    series(f"E={E} graph+cpu", lambda: agent._policy(obs,state,True)[0]["action"].cpu(), sync=False)
    series(f"E={E} eager-after", lambda: agent._policy_eager(obs,state,True))
    pr=agent._policy_runner; key=[k for k in pr._sig if k[0]==E][0]; st=pr._sig[key]
    series(f"E={E} load-only", lambda: pr._load(st,obs,state))
    series(f"E={E} replay-only", lambda: st["graph"].replay())
    print("mem allocated MB", torch.cuda.memory_allocated()/2**20, "reserved", torch.cuda.memory_reserved()/2**20)
