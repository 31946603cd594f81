"""How long does a chain of dependent tiny kernels take inside one hipGraph?  (per-launch floor of the step graph)"""
import torch, time
dev = torch.device("cuda:0")
s = torch.cuda.Stream()
for n_elem in (1, 4096, 1 << 20):
    x = torch.zeros(n_elem, device=dev)
    for K in (10, 40, 160):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            for _ in range(3):
                x.add_(1.0)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(K):
                    x.add_(1.0)
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            R = 200
            for _ in range(R):
                g.replay()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / R
        print(f"n_elem {n_elem:8d}  K {K:4d}  replay {dt*1e6:8.1f} us  per kernel {dt*1e6/K:6.2f} us")
