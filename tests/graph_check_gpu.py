# eager vs graph-replayed steps must produce identical trajectories (dropout 0)
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
import multimodal_learning_amd as m
from tests.test_gpu_step import _mk_step, _tuple
from oracle.step import default_opt, synthetic_batch
m.set_precision("bf16")
res=[]
for use_graph in (False, True):
    step=_mk_step(default_opt(), 1024, seed=0)
    if use_graph: step.enable_graph()
    outs=[]
    for it in range(6):
        bt=synthetic_batch(8,128,seed=50+it)
        ranks=[np.random.RandomState(10*it+i).choice(np.arange(30,100),20,replace=False) for i in range(2)]
        o=step.step(_tuple(bt), epoch=1, ranks=ranks)
        outs.append((o['loss'].item(), o['logit_path'].cpu().clone()))
    res.append(outs)
for it in range(6):
    a,b=res[0][it],res[1][it]
    print(it, a[0], b[0], (a[1]-b[1]).abs().max().item())
assert all(res[0][i][0]==res[1][i][0] for i in range(6)), "graph replay differs from eager"
print("graph == eager: OK")
