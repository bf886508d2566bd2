"""DeepLabV3-R101 eval forward on one 240x320 image: eager vs HIP-graph replay
(wall time per call, host-side enqueue time per call)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ucsa_neural_rendering_amd.network import DeepLabV3
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "1"))
m = DeepLabV3({"pretrained": False, "pretrained_backbone": False, "num_classes": 40}).to(dev).eval()
x = torch.rand(B, 3, 240, 320, device=dev)
with torch.no_grad():
    for _ in range(5):
        y = m(x)["out"]
    torch.cuda.synchronize()

    def bench(fn, n=50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        return th / n * 1e3, (time.perf_counter() - t0) / n * 1e3
    print("eager        host %.2f ms, wall %.2f ms" % bench(lambda: m(x)["out"]))
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            m(x)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        yg = m(x)["out"]
    print("graph replay host %.2f ms, wall %.2f ms" % bench(g.replay))
    print("max |graph - eager|", float((yg - m(x)["out"]).abs().max()))
