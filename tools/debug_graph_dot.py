import torch, tempfile, os, re
dev = torch.device("cuda:0")
x = torch.randn(1920, 128, device=dev)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x.sum(0)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
g.enable_debug_mode()
with torch.cuda.graph(g):
    y = x.sum(0)
    z = x * 2
path = os.path.join(tempfile.gettempdir(), "g.dot")
g.debug_dump(path)
t = open(path).read()
print(len(t)); print(t[:3000])
