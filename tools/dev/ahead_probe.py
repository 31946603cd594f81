"""developer probe (round 6): what does it cost the main stream to WAIT for an event of a side stream between two hipGraph replays?
Main graph A = 27 dependent small kernels (~the one-pair step's launch structure); side graph B = 3 kernels (~ the graph build).
  serial : A contains B's kernels in front (today's step)
  ahead  : B replayed on a side stream one iteration ahead, event recorded behind it, main stream waits for the event, then replays A
  nowait : like ahead without the wait (unsafe; the floor)"""
import time, torch
dev = "cuda"
x = torch.randn(4096, 32, device=dev); w = torch.randn(32, 32, device=dev) * 0.1
y = torch.randn(200000, device=dev)
def body_a(t):
    for _ in range(27):
        t = torch.tanh(t @ w)
    return t
def body_b():
    z = y
    for _ in range(3):
        z = torch.sin(z) * 1.0001
    return z
main, side = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(main):
    body_a(x); body_b(); torch.cuda.synchronize()
    gA, gAB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gA): oa = body_a(x)
    with torch.cuda.graph(gAB, pool=gA.pool()):
        ob = body_b(); oab = body_a(x)
with torch.cuda.stream(side):
    gB = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gB): ob2 = body_b()
torch.cuda.synchronize()
def timed(fn, n=400):
    with torch.cuda.stream(main):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
evs = [torch.cuda.Event() for _ in range(4)]
state = {"i": 0}
def serial(): gAB.replay()
def only_a(): gA.replay()
def ahead():
    i = state["i"]; state["i"] += 1
    ev = evs[i % 4]
    with torch.cuda.stream(side):
        gB.replay(); ev.record(side)
    main.wait_event(evs[(i - 1) % 4]) if i > 0 else None
    gA.replay()
def nowait():
    with torch.cuda.stream(side):
        gB.replay()
    gA.replay()
for name, fn in (("A alone", only_a), ("serial B+A in one graph", serial), ("ahead: B on side stream, main waits on its event", ahead), ("nowait", nowait), ("A alone", only_a)):
    print(f"{name:50s} {timed(fn):8.1f} us per iteration")
