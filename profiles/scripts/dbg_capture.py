import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import multimodal_learning_amd as m
m.set_precision("bf16")
from oracle.step import default_opt
net = m.define_net(default_opt(), 1, path_only=True).cuda().train()
net._no_bwd_overlap = len(sys.argv) > 1 and sys.argv[1] == "one"
warm_side = len(sys.argv) > 2
x = torch.rand(4, 3, 128, 128, device="cuda") * 2 - 1
ss = torch.cuda.Stream()
ss.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(ss if warm_side else torch.cuda.current_stream()):
    for rep in range(2):
        out = net(x_path=x)
        (out[1].square().mean() + out[2].sum()).backward()
torch.cuda.current_stream().wait_stream(ss)
torch.cuda.synchronize()
del out
print("eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = net(x_path=x)
    (out[1].square().mean() + out[2].sum()).backward()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed", flush=True)
