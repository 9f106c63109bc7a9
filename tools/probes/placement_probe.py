"""Round 5, last session: does the PLACEMENT of the engine's buffers explain why configs[3] reads 0.304 ms as `bench.py --config c4`
and 0.324 ms as `other_configs.c4` of the default line (same box, same binary, same plan: tools/runs/r05_y.sh)?

    python tools/probes/placement_probe.py <c2|c4|c5> <fresh|cached|second> [steps]

fresh  : the engine is the first thing the process allocates (the `--config` run)
cached : a 1 GiB block is allocated and freed first, so the caching allocator carves the engine's buffers out of it back to back
second : a configs[1] engine is built, stepped and kept alive first (what `other_configs` sees)
AIR_FLAT_LAYOUT / AIR_GEMM_SHORTK pass through the environment; AIR_PROBE_SEED = the engine's seed (initial parameters + noise stream:
`bench.py --config` uses distributed.rank_seed(1, 0) = 1000004, `other_configs` uses 1).  Prints ms per step and where the five flat arrays landed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    name, ctx = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    dev = torch.device("cuda:0")
    keep = None
    if ctx == "cached":
        x = torch.empty(1 << 30, dtype=torch.uint8, device=dev); del x
    elif ctx == "second":
        from attend_infer_repeat_amd.engine import AIREngine, EngineConfig
        keep = AIREngine(EngineConfig(), 64, device=dev, seed=1, keep_canvas_steps=True)
        keep.capture()
        for _ in range(50):
            keep.train_step()
        keep.synchronize()
    # where the flat arrays of the engine under test land: patch the allocator hook to report
    from attend_infer_repeat_amd import engine as E
    orig = E.AIREngine._alloc_flat
    where = {}

    def spy(count, n, d):
        out = orig(count, n, d)
        where["ptr"] = [t.data_ptr() for t in out]; where["n"] = n
        return out
    E.AIREngine._alloc_flat = staticmethod(spy)
    seed = int(os.environ.get("AIR_PROBE_SEED", "1"))
    rec = bench.run_other_config(name, dev, steps=steps, warmup=100, seed=seed)
    p = where["ptr"]
    print("state at end: %s" % rec.get("model_state_at_end"))
    print("%s %-6s seed=%d layout=%-14s shortk=%s  %.4f ms  %9.1f images/s  launches %d | n_total %d | offset in 2 MiB: %s | deltas to params (MiB): %s" % (
        name, ctx, seed, os.environ.get("AIR_FLAT_LAYOUT", "-"), os.environ.get("AIR_GEMM_SHORTK", "1"), rec["ms_per_step"], rec["value"],
        rec["kernel_launches_per_step"], where["n"], [q % (2 << 20) for q in p], ["%.4f" % ((q - p[0]) / 2 ** 20) for q in p]), flush=True)


if __name__ == "__main__":
    main()
