"""does a replayed HIP graph run two independent branches side by side?  Two chains of N small kernels, captured (a) one after
the other on one stream, (b) forked onto a second stream and joined: replay time of each"""
import time, torch
N = 200
a = torch.randn(1 << 14, device="cuda"); b = torch.randn(1 << 14, device="cuda")
def chain(t):
    for _ in range(N):
        t = t * 1.0001 + 0.5
    return t
res = {}
for mode in ("serial", "forked"):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); side = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(a); chain(b)  # warm-up
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        if mode == "serial":
            ra = chain(a); rb = chain(b)
        else:
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
            side.wait_event(ev)
            with torch.cuda.stream(side):
                rb = chain(b)
                ev2 = torch.cuda.Event(); ev2.record(side)
            ra = chain(a)
            torch.cuda.current_stream().wait_event(ev2)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    res[mode] = (time.perf_counter() - t0) / 20 * 1e3
    print("%s: %.3f ms per replay of 2 x %d kernels" % (mode, res[mode], N), flush=True)
