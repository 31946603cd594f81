"""Where does the host time of train.GraphedStep.run go?  (developer probe)"""
import itertools, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import se3_equi_graph_registration_amd as pkg
from se3_equi_graph_registration_amd import train
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
egnn = pkg.EGNN(32, 32, 32, in_edge_nf=1, device=dev, n_layers=4)
model = pkg.CrossAttentionPoseRegression(egnn, num_nodes=2048, hidden_nf=32, device=dev).to(dev)
opt = train.make_optimizer(model, lr=1e-4)
ring = []
for j in range(max(2, 64 // B)):
    b = pkg.synthetic_batch(1234 + j * B, B)
    ring.append((b["corr"], b["labels"], b["src"], b["tgt"], b["feat_src"], b["feat_tgt"], b["gt_pose"]))
ws = torch.cuda.Stream(); torch.cuda.set_stream(ws)
st = train.GraphedStep(model, opt, k=20)
st.run(itertools.islice(itertools.cycle(ring), 20), log_interval=0, skip_nonfinite=True)
for n in (200, 200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.run(itertools.islice(itertools.cycle(ring), n), log_interval=0, skip_nonfinite=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B}: {1e3 * dt / n:.4f} ms/step  mode={st.graph_mode}  timing={getattr(st, 'timing', None)}")


def raw(label, fn, n=400):
    with torch.cuda.stream(st.stream):
        for _ in range(20):
            fn(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"   {label}: {1e3 * dt / n:.4f} ms/step")


raw("replay slot 0 only, no events", lambda i: st.graphs[0][0].replay())
raw("replay alternating slots, no events", lambda i: st.graphs[i % len(st.graphs)][0].replay())
adam = opt.adam
pin = train._Slot(B, 2048, 32, dev, pinned=True)
pin.fill(ring[0])
NS = len(st.graphs)


def reset():
    torch.cuda.synchronize()
    with torch.cuda.stream(st.stream):
        adam.stats_dev.zero_()
    torch.cuda.synchronize()
    adam.progress[0] = 0.0
    st._n = 0


def throttled(depth):
    def fn(i):
        while adam.progress[0] < st._n - depth:
            pass
        st.graphs[i % NS][0].replay()
        st._n += 1
    return fn


for d in (1, 2, 4):
    reset(); raw(f"replay throttled to {d} step(s) in flight, no H2D", throttled(d))


def h2d_unthrottled(i):
    with torch.cuda.stream(st.copy_stream):
        st.slots[(i + 1) % NS].packet.copy_(pin.packet, non_blocking=True)
    st.graphs[i % NS][0].replay()


reset(); raw("replay + H2D into the next slot, no throttle, no events", h2d_unthrottled)


def h2d_throttled(depth):
    def fn(i):
        while adam.progress[0] < st._n - depth:
            pass
        with torch.cuda.stream(st.copy_stream):
            st.slots[(i + 1) % NS].packet.copy_(pin.packet, non_blocking=True)
        st.graphs[i % NS][0].replay()
        st._n += 1
    return fn


for d in (1, 2, 4):
    reset(); raw(f"replay + H2D, throttled to {d} in flight", h2d_throttled(d))

import threading, queue as _q, numpy as np


def h2d_variant(depth, ev_record=False, stream_sync=False):
    def fn(i):
        while adam.progress[0] < st._n - depth:
            pass
        if stream_sync:
            st.copy_stream.synchronize()
        with torch.cuda.stream(st.copy_stream):
            st.slots[(i + 1) % NS].packet.copy_(pin.packet, non_blocking=True)
            if ev_record:
                e = torch.cuda.Event(); e.record(st.copy_stream)
        st.graphs[i % NS][0].replay()
        st._n += 1
    return fn


reset(); raw("H2D throttled(1) + event record on the copy stream", h2d_variant(1, ev_record=True))
reset(); raw("H2D throttled(1) + copy_stream.synchronize() first", h2d_variant(1, stream_sync=True))
# with a busy producer thread beside it
stop = threading.Event()
pins = [train._Slot(B, 2048, 32, dev, pinned=True) for _ in range(4)]


def busy():
    k = 0
    while not stop.is_set():
        pins[k % 4].fill(ring[k % len(ring)]); k += 1
        time.sleep(3e-4)


th = threading.Thread(target=busy, daemon=True); th.start()
reset(); raw("H2D throttled(1), a producer thread packing beside it", h2d_variant(1))
reset(); raw("H2D throttled(1) + stream sync, producer thread beside it", h2d_variant(1, stream_sync=True))
stop.set(); th.join()
