"""Times SO3TensorProductLayer forward and backward (A15) on the GPU: python tools/bench_so3.py [rows]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import se3_equi_graph_registration_amd as pk

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
layer = pk.SO3TensorProductLayer(32, 32).cuda()
x = torch.randn(rows, 32, device="cuda")
with torch.no_grad():
    for _ in range(5):
        y = layer(x)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50):
        y = layer(x)
    ev[1].record()
    torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1e3 / 50
flops = 2.0 * rows * (1024 * 64 + 64 * 32)
print(f"rows={rows} {us:.1f} us/call  {flops / us * 1e-6:.1f} TFLOP/s fp32 (peak 157.3)")

xg = x.clone().requires_grad_(True)
gy = torch.randn(rows, 32, device="cuda")
for _ in range(3):
    layer.zero_grad(); y = layer(xg); y.backward(gy)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(20):
    layer.zero_grad(set_to_none=True); xg.grad = None
    y = layer(xg); y.backward(gy)
ev[1].record()
torch.cuda.synchronize()
us_fb = ev[0].elapsed_time(ev[1]) * 1e3 / 20
print(f"rows={rows} forward + backward (all gradients: x, W1, b1, W2, b2) {us_fb:.1f} us/call -> backward ~{us_fb - us:.1f} us "
      f"({3 * flops / (us_fb - us) * 1e-6:.1f} TFLOP/s fp32 on 3x the forward's flops)")
