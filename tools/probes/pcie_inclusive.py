"""The PCIe-inclusive rate of the headline configuration: the same captured step, but every step's observation batch (64 x 50 x 50 fp32 =
640 KB) handed over as a pinned HOST buffer and copied on the engine's stream in front of the replay (what a host-side feeder such as the
reference's tf.py_func one, data.py:121-158, would do), against the batch resident in HBM (bench.py's `value`).
    python tools/probes/pcie_inclusive.py [c2|c4] [batch] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from attend_infer_repeat_amd.data import synthetic_multi_mnist  # noqa: E402
from attend_infer_repeat_amd.engine import AIREngine, EngineConfig  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c2"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
    dev = torch.device("cuda:0")
    kw = dict(img_size=(100, 100), crop_size=(28, 28), max_steps=5) if name == "c4" else {}
    cfg = EngineConfig(**kw)
    eng = AIREngine(cfg, B, device=dev, seed=1000004, keep_canvas_steps=True)
    host = [torch.from_numpy(synthetic_multi_mnist(B, cfg.img_size, max_objects=4 if name == "c4" else 2, seed=s)[0]).pin_memory() for s in range(8)]
    eng.set_obs(host[0].to(dev))
    eng.capture()

    def run(feed):
        for _ in range(200):
            eng.train_step(host[0] if feed else None)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            eng.train_step(host[i & 7] if feed else None)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps * 1e3
    res = [(run(False), run(True)) for _ in range(2)]
    nbytes = host[0].numel() * 4
    for r, f in res:
        print("%s batch %d: resident %.4f ms/step (%.1f images/s) | pinned host batch per step (%d KB over PCIe) %.4f ms/step (%.1f images/s) | +%.1f us" % (
            name, B, r, B / r * 1e3, nbytes // 1024, f, B / f * 1e3, (f - r) * 1e3), flush=True)
    # the copy alone, back to back on the engine stream
    dst = eng.obs
    with torch.cuda.stream(eng.stream):
        for _ in range(50):
            dst.copy_(host[0].reshape(dst.shape), non_blocking=True)
        eng.stream.synchronize()
        t0 = time.perf_counter()
        for i in range(500):
            dst.copy_(host[i & 7].reshape(dst.shape), non_blocking=True)
        eng.stream.synchronize()
        us = (time.perf_counter() - t0) / 500 * 1e6
    print("H2D copy alone: %.1f us per %d KB batch = %.1f GB/s" % (us, nbytes // 1024, nbytes / us / 1e3))


if __name__ == "__main__":
    main()
