import torch, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from kurosiwo_amd.snunet import SNUNet_ECAM
from kurosiwo_amd.trainer import CDTrainStep
torch.manual_seed(0)
B, S = 32, 224
m = SNUNet_ECAM(2, 3, base_channel=32, precision="bf16").cuda().train()
st = CDTrainStep(m, B, S, S, "ce+dice", (1.0, 1.0, 1.0), lr=1e-3)
xA, xB = torch.randn(B, 2, S, S).cuda(), torch.randn(B, 2, S, S).cuda()
y = torch.randint(0, 3, (B, S, S)).cuda()
for _ in range(5):
    st.step(xA, xB, y)
torch.cuda.synchronize()
lane = st._ss
orig_join = lane.join
rec = []
def join():
    if lane.dirty:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream()); b.record(lane.side)
        rec.append((a, b))
    orig_join()
lane.join = join
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    st.step(xA, xB, y)
e1.record()
torch.cuda.synchronize()
print("step ms", e0.elapsed_time(e1) / 10)
print("side finishes after main by (ms):", [round(a.elapsed_time(b), 3) for a, b in rec])
import time
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    st.step(xA, xB, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue ms/step", (t1 - t0) * 100, " wall ms/step", (t2 - t0) * 100)
