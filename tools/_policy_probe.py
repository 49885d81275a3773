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
for E in (1,4,16):
    obs={"image":rs.randint(0,256,(E,64,64,3)).astype(np.uint8),"is_first":np.zeros((E,),bool),"is_terminal":np.zeros((E,),bool)}
    out,state=agent._policy(dict(obs,is_first=np.ones((E,),bool)),None,True)
    for _ in range(3): out,state=agent._policy(obs,state,True)
    pr=agent._policy_runner; key=[k for k in pr._sig if k[0]==E][0]; st=pr._sig[key]
    torch.cuda.synchronize()
    def tm(f,n=20):
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
    print(E, "load %.3f"%tm(lambda: pr._load(st,obs,state)), "replay %.3f"%tm(lambda: st["graph"].replay()),
          "clone %.3f"%tm(lambda: st["packed"].clone()), "full %.3f"%tm(lambda: agent._policy(obs,state,True)),
          "eager %.3f"%tm(lambda: agent._policy_eager(obs,state,True)), flush=True)
