import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "dreamerv3-torch_amd")); sys.path.insert(0, REPO)
import numpy as np, torch
from tests import helpers as Hh
from tests.golden import common
from dv3hip.graph import UpdateRunner
from dv3hip.staging import BatchStager
name = "cfg2"
cfg, wm, beh = Hh.build_models(name)
data = {k: torch.from_numpy(v).cuda() for k, v in common.make_batch(name).items()}
runner = UpdateRunner(wm, beh)
for _ in range(5): runner.step(data)
torch.cuda.synchronize()
def loop(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    h = (time.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    return h, (time.perf_counter() - t0) / n * 1e3
print("resident: host %.2f ms/step, total %.2f" % loop(lambda: runner.step(data)))
host = {k: v.cpu().numpy() for k, v in data.items()}
st = BatchStager("cuda")
for _ in range(3): runner.step(st.stage(host))
print("staged  : host %.2f ms/step, total %.2f" % loop(lambda: runner.step(st.stage(host))))
for depth in (3, 4):
    st2 = BatchStager("cuda", depth=depth)
    for _ in range(5): runner.step(st2.stage(host))
    h, t = loop(lambda: runner.step(st2.stage(host)))
    print(f"staged depth {depth}: host {h:.2f} ms/step, total {t:.2f}")
# same data every step, but through the stager's device buffers without any H2D in the loop
fixed = st.stage(host)
torch.cuda.synchronize()
print("resident (stager buffers): host %.2f ms/step, total %.2f" % loop(lambda: runner.step(fixed)))
st3 = BatchStager("cuda", overlap=False)
for _ in range(5): runner.step(st3.stage(host))
h, t = loop(lambda: runner.step(st3.stage(host)))
print(f"staged on the update's own stream: host {h:.2f} ms/step, total {t:.2f}")
ts, tr = [], []
torch.cuda.synchronize()
for _ in range(20):
    a = time.perf_counter(); d = st.stage(host); b = time.perf_counter(); runner.step(d); c = time.perf_counter()
    ts.append((b - a) * 1e3); tr.append((c - b) * 1e3)
torch.cuda.synchronize()
print("staged loop host split: stage %.2f ms, runner.step %.2f ms (medians)" % (np.median(ts), np.median(tr)))
print("   stage", " ".join("%.1f" % x for x in ts))
print("   step ", " ".join("%.1f" % x for x in tr))
print("stage only: host %.2f ms/step, total %.2f" % loop(lambda: st.stage(host)))
t0 = time.perf_counter()
for _ in range(20):
    for k, v in host.items(): np.copyto(np.empty_like(v), v)
print("np.copyto of the batch: %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
print({k: (v.dtype, v.shape) for k, v in host.items()})
