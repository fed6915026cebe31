import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gantts_amd import models
B, T = 32, 256
m = models.LSTMRNN(in_dim=425, out_dim=187, num_hidden=1, hidden_dim=256, bidirectional=True).cuda().eval()
x = torch.rand(B, T, 425).cuda()
L = [T] * B
def run(tag):
    m(x, L); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): m(x, L)
    torch.cuda.synchronize()
    print("%s: %.2f us per step" % (tag, (time.perf_counter()-t0)/5/T*1e6))
run("default stream")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run("side stream")
print(os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), {k: v for k, v in os.environ.items() if k.startswith(("HIP", "HSA", "AMD", "ROC", "GPU"))})
