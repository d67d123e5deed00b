import torch, sys
dev = torch.device("cuda:0")
a = torch.randn(1 << 20, device=dev); b = torch.randn(1 << 20, device=dev)
mode = sys.argv[1]
cap = torch.cuda.Stream(); side = torch.cuda.Stream(priority=-1 if mode == "prio" else 0); side1 = torch.cuda.Stream()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=cap):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        c = a * 2
        if mode == "nested":
            side1.wait_stream(side)
            with torch.cuda.stream(side1):
                e = c + 1
            f = c * 3
            side.wait_stream(side1)
            c = e + f
    d = b + 1
    cur.wait_stream(side)
    out = c + d
g.replay(); torch.cuda.synchronize()
print(mode, "ok", float(out.sum()))
