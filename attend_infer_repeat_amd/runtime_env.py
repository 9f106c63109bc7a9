"""HIP runtime settings the engine is measured with.  They are read by libamdhip64 when the runtime initialises (the first HIP call of
the process), so they are put into the environment when this package is imported -- `setdefault`: anything the user exported wins.
The setting is PROCESS-WIDE: it changes how every hipGraph of the process is replayed, torch's own CUDA graphs included (same
results, different submission path).  `AIR_RUNTIME_ENV=0` leaves the runtime's defaults alone; importing the package after the
runtime is up raises a RuntimeWarning (the setting can no longer apply) and `runtime_env.late` says so.

DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
    ROCm 7 replays a captured graph from pre-recorded AQL packets ("graph packet capture").  For the train step -- a chain of 34-40
    short dependent kernels -- the ordinary submission path is FASTER on the GPU side: 0.2043 against 0.2084 ms per step at BASELINE
    configs[1] (three runs each on one box), 0.3253 / 0.3292 ms at configs[3], 0.4226 / 0.4289 ms at configs[4]
    (tools/runs/r03_env.sh, profiles/r03_env_sweep.txt).  Same kernels, same results; only how the runtime feeds them to the queue.
    The price is on the HOST: the ordinary path spends ≈ 2.8 us of CPU per node against ≈ 1.9 us of pre-recorded packets
    (tools/kbench/graph_replay_gap.cpp, profiles/r03_kbench_graph_replay_gap.txt: a chain of 34 EMPTY kernels replays in 96 us
    instead of 64).  The train step's kernels average 6 us, so the host stays well ahead of the GPU; a workload whose kernels
    average less than ~3 us would be host-bound on this path and should export DEBUG_CLR_GRAPH_PACKET_CAPTURE=1.

Measured in the same sweep and NOT set: AMD_OPT_FLUSH=0 (system-scope fences between kernels: 0.2737 ms), HIP_FORCE_DEV_KERNARG=0
(0.2678 ms), AMD_DIRECT_DISPATCH=0 (0.2066 ms alone: within reach of the setting above, changes the host threading model),
DEBUG_HIP_GRAPH_BATCH_SIZE / DEBUG_CLR_MAX_BATCH_SIZE / DEBUG_HIP_FORCE_GRAPH_QUEUES / ROC_USE_FGS_KERNARG (no effect),
ROC_SYSTEM_SCOPE_SIGNAL=0 (the process hangs).
"""
import os
import sys

SETTINGS = {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}

applied = {}          # name -> value in effect for this process (the user's own export if there was one)
late = False          # True when torch had already initialised the HIP runtime: the settings cannot take effect any more


def apply():
    global late
    if os.environ.get("AIR_RUNTIME_ENV", "1") == "0":        # opt out: leave the runtime's defaults alone
        applied.clear()
        return applied
    torch = sys.modules.get("torch")
    try:
        late = bool(torch is not None and torch.cuda.is_initialized())
    except Exception:                                         # noqa: BLE001 -- a torch without the cuda module: nothing to be late for
        late = False
    # `late` only counts when THIS package is the one trying to set something too late: a key the user exported before the process
    # started is in effect whatever the import order
    missing = [k for k in SETTINGS if k not in os.environ]
    for k, v in SETTINGS.items():
        os.environ.setdefault(k, v)
        applied[k] = os.environ[k]
    late = late and bool(missing)
    if late:
        import warnings
        warnings.warn("attend_infer_repeat_amd was imported after the HIP runtime was initialised: %s cannot take effect in this "
                      "process (the documented step times assume it; import the package, or export the setting, before the first "
                      "HIP call)" % ", ".join("%s=%s" % kv for kv in SETTINGS.items()), RuntimeWarning, stacklevel=3)
    return applied
