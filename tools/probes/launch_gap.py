import torch
x=torch.zeros(64,device="cuda")
y=torch.zeros(4680*1536,device="cuda",dtype=torch.bfloat16)
def g_of(fn,n):
    s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    return g
def t(g,n):
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)*1e3/n
g1=g_of(lambda: x.add_(1),500)
print("tiny dependent kernel chain (graph): %.2f us per kernel"%t(g1,500))
g2=g_of(lambda: y.add_(1),200)
print("28.8 MB rw elementwise (graph): %.2f us per kernel"%t(g2,200))
# eager
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(500): x.add_(1)
e1.record(); e1.synchronize()
print("tiny eager: %.2f us per kernel"%(e0.elapsed_time(e1)*1e3/500))
