"""Host-side cost of dv3hip.staging.BatchStager on a cfg-2 batch (MI355X box): pinned memcpy, H2D, stage()."""
import sys, time, os
sys.path.insert(0, "dreamerv3-torch_amd"); sys.path.insert(0, ".")
import torch, numpy as np
from dv3hip.staging import BatchStager
from tests.golden import common
b = common.make_batch("cfg2")
st = BatchStager("cuda:0")
for _ in range(3): st.stage(b)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(10): st.stage(b)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("host ms/stage", (t1-t0)*100, "incl sync", (t2-t0)*100)
h = torch.empty(b["image"].shape, dtype=torch.uint8).pin_memory(); d = torch.empty(b["image"].shape, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); print("H2D 12.6MB pinned ms", (time.perf_counter()-t0)*100)
t0=time.perf_counter()
for _ in range(10): np.copyto(h.numpy(), b["image"])
print("host memcpy ms", (time.perf_counter()-t0)*100)
