"""CPU legs of bench.py that need no GPU: the all-host-cores replica baseline (SURVEY 8d's "N = all cores")."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_all_cores_replica_leg_adds_the_rates_of_independent_oracle_replicas():
    import bench
    threads = max(1, (os.cpu_count() or 2) // 2)              # two replicas on this host
    rec = bench.cpu_all_cores_replicas({}, 8, threads, seconds=0.5, limit_s=120.0)
    assert rec is not None, "the replica leg failed on a healthy host"
    assert rec["replicas"] == max(1, (os.cpu_count() or 1) // threads) and rec["threads_per_replica"] == threads
    assert rec["cores"] == rec["replicas"] * threads <= (os.cpu_count() or 1)
    assert rec["value"] > 0 and "independent replicas" in rec["sample"]


def test_all_cores_replica_leg_never_blocks_the_line():
    import bench
    # a limit no interpreter start-up can meet: the leg gives up, cleans up its children and reports nothing
    assert bench.cpu_all_cores_replicas({}, 8, max(1, (os.cpu_count() or 2) // 2), seconds=0.5, limit_s=0.05) is None


def test_configs3_line_is_the_aggregate_over_engine_seeds(monkeypatch):
    """other_configs.c4 = total images / total time over bench.C4_SEEDS (fresh batch per step from the HBM feeder) with each seed's own
    figure and the spread beside it, the fixed-batch figure of rounds 1-5 for continuity and the configs[3]-shape ST sweeps; configs[4]
    stays on one seed and carries the fixed-batch figure and the all-steps-present probe."""
    import bench
    ms = {1: 0.3025, 1000004: 0.3026, 7: 0.3075}
    calls = []

    def fake(name, device, steps=400, warmup=100, seed=1, feeder=True, step_bias=None):
        calls.append((name, seed, feeder, step_bias))
        return {"workload": name, "value": round(64 / (ms.get(seed, 0.4) * 1e-3), 1), "unit": "images/sec", "ms_per_step": ms.get(seed, 0.4),
                "steps": steps, "warmup": warmup, "kernel_launches_per_step": 36, "params_finite_after_run": seed != 7,
                "input": "feeder" if feeder else "fixed", "step_bias": 0.75 if step_bias is None else step_bias,
                "model_state_at_end": {"steps_present_per_image": float(seed % 3), "mean_abs_where": [1.0] * 4}, "roofline": {"frac": 0.1}}
    monkeypatch.setattr(bench, "run_other_config", fake)
    monkeypatch.setattr(bench, "c4_shape_sweeps", lambda device: {"roofline_sweep_canvas_write_bwd": [{"batch": 65536, "frac": 0.2}]})
    rec = bench.run_other_config_seeds("c4", None)
    assert calls[:3] == [("c4", s, True, None) for s in bench.C4_SEEDS] and 1000004 in bench.C4_SEEDS    # the seed of `bench.py --config c4` is one of them
    assert calls[3] == ("c4", 1, False, None) and rec["fixed_batch"]["input"] == "fixed"
    assert rec["steps"] == 1200 and abs(rec["ms_per_step"] - sum(ms.values()) / 3) < 1e-4
    assert abs(rec["value"] * rec["ms_per_step"] * 1e-3 / 64 - 1.0) < 1e-3 and rec["params_finite_after_run"] is False
    assert [p["engine_seed"] for p in rec["per_seed"]] == list(bench.C4_SEEDS) and all("model_state_at_end" in p for p in rec["per_seed"])
    assert abs(rec["per_seed_spread"] - (0.3075 - 0.3025) / 0.3025) < 1e-3 and rec["roofline_sweep_canvas_write_bwd"][0]["frac"] == 0.2
    calls.clear()
    one = bench.run_other_config_seeds("c5", None)
    assert calls == [("c5", 1, True, None), ("c5", 1, False, None), ("c5", 1, True, 20.0)] and "per_seed" not in one
    assert one["fixed_batch"]["input"] == "fixed" and one["all_steps_present_probe"]["step_bias"] == 20.0
