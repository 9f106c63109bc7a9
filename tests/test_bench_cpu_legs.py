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
