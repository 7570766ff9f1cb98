"""torch.mm (hipBLASLt/rocBLAS) on the weight-gradient shapes of the coupling conditioner: [M x NPX] @ [NPX x N], long
reduction, small outputs -- time per call in a graph replay"""
import torch, time
shapes = [(512, 19600, 18), (4, 19600, 512), (4, 19600, 36), (512, 4900, 36), (8, 4900, 512), (8, 4900, 72),
          (128, 8192, 54), (12, 8192, 128), (12, 8192, 108), (128, 2048, 108), (24, 2048, 128), (24, 2048, 216),
          (256, 3328, 54), (12, 3328, 256), (256, 208, 216), (48, 208, 256), (48, 208, 432)]
for dt in (torch.bfloat16, torch.float32):
    for (M, K, N) in shapes:
        a = torch.randn(M, K, device="cuda", dtype=dt)
        bt = torch.randn(N, K, device="cuda", dtype=dt)
        kw = {}
        try:
            if dt == torch.bfloat16:
                torch.mm(a, bt.t(), out_dtype=torch.float32); kw = {"out_dtype": torch.float32}
        except Exception as e:
            kw = {}
        for _ in range(3):
            c = torch.mm(a, bt.t(), **kw)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(20):
                    c = torch.mm(a, bt.t(), **kw)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 200 * 1e6
        print("%s M%d K%d N%d out %s: %.1f us (%.1f GFLOP/s)" % (str(dt)[6:], M, K, N, c.dtype, us, 2 * M * K * N / us / 1e3), flush=True)
