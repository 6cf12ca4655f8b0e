"""Dev probe: does hipGraphDebugDotPrint work here, and what do kernel nodes look like (to count launches per frame)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
x = torch.zeros(64, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    x.add_(1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
g.enable_debug_mode()
with torch.cuda.graph(g):
    x.add_(1); x.mul_(2); y = x.clone()
g.debug_dump("/tmp/g.dot")
print(open("/tmp/g.dot").read()[:3000])
