"""Data-parallel training of the AIR engine: one process per GPU, one RCCL all-reduce per step.

The reference is single-process (no NCCL / MPI / tf.device anywhere).  Images in a batch are independent
through the whole forward / backward, so the path shards on the batch dimension: every rank runs the full step on its
own 64 images with its own Philox stream, the flat fp32 gradient bucket (model + baseline variables, 10.5 MB at the
50x50 config) is summed with ONE all-reduce over RCCL/xGMI, and the centred-RMSProp kernel applies grad_scale =
1/world_size.  Nothing else is exchanged (no activations, no canvases).  Semantics: the averaged gradient equals the
mean of `world_size` independent B=64 reference steps (the NVIL mean-baseline quirk stays per rank).

Where the collective runs:
  "rccl-captured" (default on GPUs, world > 1): the engine's own stream calls ncclAllReduce through the C ABI
                  (air_allreduce_sum) INSIDE the captured step, so the step is still one hipGraph replay.  The
                  communicator is created once from a unique id that rank 0 broadcasts through torch.distributed.
  "torch-split"   (fallback; CPU / gloo tests): graph A (forward + backward) -> torch.distributed.all_reduce on the
                  engine stream -> graph B (update).
torch.distributed is plumbing here: rendezvous, the parameter broadcast and the fallback collective; backend "nccl" is
RCCL on ROCm, "gloo" is used by the CPU tests.
"""
import contextlib
import ctypes
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun-style environment variables.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shard_batch(global_batch: int, rank: int, world: int):
    """[start, stop) of this rank's images; the global batch must divide evenly (fixed 64 images per GPU)."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} does not divide over {world} ranks")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def rank_seed(base_seed: int, rank: int) -> int:
    """Distinct, reproducible Philox seed per rank (independent noise per replica)."""
    return (int(base_seed) * 1000003 + 7919 * int(rank) + 1) & 0x7FFFFFFFFFFFFFFF


def allreduce_gradients(flat_grads: torch.Tensor, group=None, average: bool = False):
    """The single collective of a data-parallel step: in-place SUM (optionally mean) of the flat gradient bucket."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return flat_grads
    dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat_grads.div_(dist.get_world_size(group))
    return flat_grads


def broadcast_parameters(flat_params: torch.Tensor, src: int = 0, group=None):
    """Make every replica start from rank `src`'s parameters (replicated weights and optimiser state)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
    return flat_params


def create_rccl_comm(device, group=None):
    """One RCCL communicator for the engine's own collective calls (air_allreduce_sum): rank 0 draws the unique id, it
    travels to the other ranks through torch.distributed (any backend), every rank joins with its GPU current."""
    from . import _lib
    from . import hip as H
    L = H.lib()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ident = [None]
    if rank == 0:
        buf = ctypes.create_string_buffer(128)
        st = L.air_comm_unique_id(buf)
        if st != 0:
            ident = [RuntimeError((L.air_comm_last_error() or b"").decode())]
        else:
            ident = [bytes(buf.raw)]
    dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if not isinstance(ident[0], (bytes, bytearray)):
        raise _lib.AirHipError("rank 0 could not create an RCCL unique id: %r" % (ident[0],))
    comm = ctypes.c_void_p()
    with torch.cuda.device(device):
        st = L.air_comm_init(ctypes.byref(comm), world, rank, ctypes.create_string_buffer(bytes(ident[0]), 128))
    if st != 0:
        raise _lib.AirHipError("air_comm_init failed: %s" % (L.air_comm_last_error() or b"").decode())
    return comm


class DataParallelEngine(object):
    """Wraps an AIREngine for multi-GPU data parallelism (one instance per process / GPU).

    `engine` needs: flat_params, flat_grads, world_size, capture(...), train_step(obs, allreduce), synchronize() and
    (optionally) stream_context() / device -- the CPU tests drive this class with a stand-in engine over gloo."""

    def __init__(self, engine, group=None, capture_graph=True, collective=None, overlap=None, steps_per_replay=1):
        self.engine = engine
        self.steps_per_replay = 1
        self.group = group
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.comm = None
        engine.world_size = self.world
        with self._stream():
            broadcast_parameters(engine.flat_params, 0, group)
        engine.synchronize()
        want = collective or os.environ.get("AIR_DP_COLLECTIVE", "").strip().lower() or None
        if overlap is None:
            overlap = os.environ.get("AIR_DP_OVERLAP", "0") == "1"
        on_gpu = getattr(getattr(engine, "device", None), "type", "cpu") == "cuda"
        self.collective = "none"
        if self.world > 1:
            self.collective = "torch-split"
            if capture_graph and on_gpu and want in (None, "captured", "rccl-captured"):
                # ONE graph per step even with world > 1: the RCCL call is a captured node.  If the library cannot be bound,
                # the communicator cannot be built or the capture is refused, fall back (on every rank alike: the outcome is
                # agreed on with a tiny all-reduce so that no rank ends up in the other protocol).
                ok = 1
                try:
                    self.comm = create_rccl_comm(engine.device, group)
                    engine.capture(comm=self.comm, overlap=bool(overlap))
                except Exception as e:                              # noqa: BLE001 -- any failure means "use the fallback"
                    ok, self._captured_error = 0, repr(e)
                flag = torch.tensor([ok], dtype=torch.int32, device=engine.device if dist.get_backend(group) == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
                if int(flag.item()) == 1:
                    self.collective = "rccl-captured" + ("+overlap" if overlap else "")
                    return
            if capture_graph:
                engine.capture(split_optimizer=True)
        elif capture_graph:
            if steps_per_replay > 1:        # single GPU only: several updates per graph replay (engine.capture)
                engine.capture(steps_per_replay=steps_per_replay)
                self.steps_per_replay = int(steps_per_replay)
            else:
                engine.capture()

    def _stream(self):
        ctx = getattr(self.engine, "stream_context", None)
        return ctx() if ctx is not None else contextlib.nullcontext()

    def _allreduce(self, grads):
        allreduce_gradients(grads, self.group, average=False)

    def train_step(self, obs=None):
        host_collective = self.world > 1 and not self.collective.startswith("rccl-captured")
        self.engine.train_step(obs, allreduce=self._allreduce if host_collective else None)

    def close(self):
        if self.comm is not None:
            from . import hip as H
            self.engine.synchronize()
            self.engine.release_graphs()
            H.lib().air_comm_destroy(self.comm)
            self.comm = None
