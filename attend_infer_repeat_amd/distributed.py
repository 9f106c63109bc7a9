"""Data-parallel training of the AIR engine: one process per GPU, one RCCL all-reduce per step.

The reference is single-process (no NCCL / MPI / tf.device anywhere).  Images in a batch are independent
through the whole forward / backward, so the path shards on the batch dimension: every rank runs the full step on its
own 64 images with its own Philox stream, the flat fp32 gradient bucket (model + baseline variables, 10.5 MB at the
50x50 config) is summed with ONE all-reduce over RCCL/xGMI, and the centred-RMSProp kernel applies grad_scale =
1/world_size.  Nothing else is exchanged (no activations, no canvases).  Semantics: the averaged gradient equals the
mean of `world_size` independent B=64 reference steps (the NVIL mean-baseline quirk stays per rank).

Where the collective runs (AIR_DP_COLLECTIVE / the `collective` argument):
  "torch-overlap" (default where the engine's backward has a gradient bucket that is final early -- the latency regime):
                  graph A1 (forward + backward up to the cut) -> async torch.distributed.all_reduce of the TAIL bucket (decoder /
                  baseline / what / glimpse-encoder gradients, ~half of the bytes) -> graph A2 (rest of the backward, running
                  underneath the collective) -> all_reduce of the head -> wait -> graph B (update).  Eager RCCL calls only.
  "torch-split"   (default otherwise): graph A (forward + backward) -> torch.distributed.all_reduce on the engine stream (backend
                  "nccl" = RCCL on ROCm; "gloo" in the CPU tests) -> graph B (update).  The plain, eager use of RCCL.
  "rccl-split"    : the same two graphs, the all-reduce issued by the engine itself through the C ABI (air_allreduce_sum ->
                  ncclAllReduce on the engine stream, eager) on a communicator of its own.
  "rccl-captured" : the ncclAllReduce call is a NODE of the step's hipGraph, so the step stays one graph replay with no host
                  round trip (+ AIR_DP_OVERLAP=1: the tail of the gradient buffer is reduced on a forked captured stream, on
                  a second communicator, while the backward finishes).  Opt-in: bit-equal to the single-GPU graph with one
                  rank on the GPU, but no multi-GPU node has been available to validate it on more than one.
  "ipc-rsag"      (round 5, opt-in): NO library collective.  The ranks of one node map each other's flat gradient / parameter
                  buffers (hipIpc, exchanged through torch.distributed) and the step's graph ends with three kernel nodes
                  (csrc/comm_ipc.hip): barrier | rank r sums ITS 1/world shard of all ranks' gradients in rank order, runs centred
                  RMSProp on that shard and writes the new parameters into every rank's buffer | barrier.  Replicas are
                  bit-identical by construction (one rank computes each element), every xGMI link carries 1/world of the bucket
                  at once, optimiser traffic per GPU drops by the world size, and the step stays ONE graph replay.  Validated
                  with two processes on one GPU only -- hence not a default.  flat_grads keeps the rank's LOCAL gradient.
The own communicator is created once from a unique id that rank 0 broadcasts through torch.distributed, after every rank has
AGREED that RCCL can be bound (a rank that failed early would otherwise leave the others blocked inside ncclCommInitRank); the
first use is an eager all-reduce of a small known buffer checked against the analytic sum, and any failure -- on any rank --
sends every rank to "torch-split" together.  torch.distributed is plumbing here: rendezvous, the parameter broadcast, the
agreement flags and the default collective.
"""
import contextlib
import ctypes
import os

import torch
import torch.distributed as dist


def free_rendezvous_port(lo: int = 20000, hi: int = 32000) -> int:
    """A free TCP port for a local rendezvous, taken BELOW the kernel's ephemeral range (ip_local_port_range, 32768-60999 by default).
    `bind(('127.0.0.1', 0))` hands out an ephemeral port; a rank that starts before rank 0 listens retries its connect from
    ephemeral SOURCE ports, and when the kernel picks the destination port itself the connect succeeds against itself (TCP
    simultaneous open) and rank 0's listen fails with EADDRINUSE -- seen once in ~140 spawned two-rank tests on a GPU box."""
    import random
    import socket
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            eph_lo = int(f.read().split()[0])
        if eph_lo > lo + 1000:
            hi = min(hi, eph_lo)
    except (OSError, ValueError, IndexError):
        pass
    rng = random.SystemRandom()
    for _ in range(200):
        port = rng.randrange(lo, hi)
        with socket.socket() as s:
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
        return port
    raise RuntimeError("no free rendezvous port in [%d, %d)" % (lo, hi))


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun-style environment variables.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and "OMP_NUM_THREADS" not in os.environ:
        # one process per GPU: the ranks of a node share its host cores (torchrun exports OMP_NUM_THREADS = 1 itself; a plain spawn
        # does not, and eight ranks x every core made the host side of an engine build take tens of seconds -- measured in the tests)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        torch.set_num_threads(max(1, min(torch.get_num_threads(), (os.cpu_count() or 1) // max(1, local_world))))
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


def _agree(ok: bool, device, group=None) -> bool:
    """every rank contributes one int; the answer is the minimum (all ranks take the same branch afterwards)"""
    backend = dist.get_backend(group)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item()) == 1


def create_rccl_comm(device, group=None):
    """One RCCL communicator for the engine's own collective calls (air_allreduce_sum).  Collective over `group`.
    Protocol (no rank may block alone): (1) every rank binds the library -- not collective -- and the outcome is agreed on;
    (2) rank 0 draws the unique id and broadcasts it (or the failure); (3) every rank joins with its GPU current
    (ncclCommInitRank, blocking, all ranks are known to arrive); (4) the outcome and RCCL's own rank count are agreed on.
    Returns the handle, or raises AirHipError ON EVERY RANK ALIKE."""
    from . import _lib
    from . import hip as H
    L = H.lib()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if not _agree(L.air_comm_available() == 0, device, group):
        raise _lib.AirHipError("RCCL cannot be bound on every rank: %s" % (L.air_comm_last_error() or b"").decode())
    ident = [None]
    if rank == 0:
        buf = ctypes.create_string_buffer(128)
        st = L.air_comm_unique_id(buf)
        ident = [bytes(buf.raw) if st == 0 else RuntimeError((L.air_comm_last_error() or b"").decode())]
    dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if not isinstance(ident[0], (bytes, bytearray)):                 # the same object on every rank: all raise together
        raise _lib.AirHipError("rank 0 could not create an RCCL unique id: %r" % (ident[0],))
    comm = ctypes.c_void_p()
    st, ok = -1, False
    try:                      # rank-local failures are folded into `ok`: the agreement below is reached by every rank
        with torch.cuda.device(device):
            st = L.air_comm_init(ctypes.byref(comm), world, rank, ctypes.create_string_buffer(bytes(ident[0]), 128))
        n = ctypes.c_int(0)
        ok = st == 0 and L.air_comm_count(comm, ctypes.byref(n)) == 0 and n.value == world
    except Exception:         # noqa: BLE001
        ok = False
    if not _agree(ok, device, group):
        if st == 0:
            L.air_comm_destroy(comm)
        raise _lib.AirHipError("air_comm_init failed on some rank: %s" % (L.air_comm_last_error() or b"").decode())
    return comm


def comm_count(comm) -> int:
    """ranks of an air_comm handle as RCCL reports them (ncclCommCount)"""
    from . import hip as H
    n = ctypes.c_int(0)
    if H.lib().air_comm_count(comm, ctypes.byref(n)) != 0:
        return -1
    return int(n.value)


def selftest_comm(comm, device, stream, group=None) -> bool:
    """First use of a fresh communicator: an EAGER all-reduce of a small buffer whose sum is known in closed form
    (rank r contributes r + 1 everywhere), checked on every rank and agreed on."""
    from . import hip as H
    L = H.lib()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ok = True
    try:                      # anything rank-local that fails (allocation, a HIP error) must still reach the agreement below:
        with torch.cuda.device(device), torch.cuda.stream(stream):      # a rank that skipped it would leave its peers blocked in it
            probe = torch.full((4096,), float(rank + 1), dtype=torch.float32, device=device)
            st = L.air_allreduce_sum(ctypes.c_void_p(probe.data_ptr()), ctypes.c_size_t(probe.numel()), comm,
                                     ctypes.c_void_p(stream.cuda_stream))
            stream.synchronize()
            ok = st == 0 and bool((probe == world * (world + 1) / 2.0).all().item())
    except Exception:         # noqa: BLE001 -- reported as "not ok"; every rank then takes the host-issued protocol together
        ok = False
    return _agree(ok, device, group)


class IpcPeerBuffers(object):
    """The peer mapping of the "ipc-rsag" protocol: every rank exports its flat gradient buffer, its flat parameter buffer and a
    block of flag words (torch's CUDA-IPC reductions: hipIpcGetMemHandle on the sender, hipIpcOpenMemHandle on the receiver; the
    handles travel through torch.distributed), and keeps the opened peer tensors alive for as long as the captured graph holds
    their addresses.  Collective over `group`; raises on the calling rank only -- the caller agrees on the outcome."""

    def __init__(self, engine, group=None):
        from torch.multiprocessing.reductions import reduce_tensor
        from . import _lib
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        dev = engine.device
        # a rank-local failure of the export (world > 8, a tensor torch cannot export, e.g. under expandable_segments) must still reach
        # the collective below -- a rank that raised in front of it would leave its peers blocked in all_gather_object --
        # so it travels as None and EVERY rank raises afterwards
        from . import hip as H
        L = H.lib()
        mine, local_error = None, None
        self._fg_ptr, self._fg_open, self.flags_kind = None, [], "torch (coarse-grained)"
        try:
            if self.world > 8:
                raise _lib.AirHipError("ipc-rsag maps the ranks of ONE node: at most 8")
            self.flags = torch.zeros(16, dtype=torch.int64, device=dev)       # [2 barriers][8 ranks]
            self.local = torch.zeros(8, dtype=torch.int64, device=dev)        # (comm_ipc.hip: epochs, arrivals, go, XCC masks)
            self.err = torch.zeros(1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize(dev)
            # the flag block in FINE-GRAINED memory where the runtime gives it (in-kernel visibility of a peer's store is
            # only defined there), exported with the HIP IPC call itself; AIR_IPC_FINEGRAINED=0 keeps the torch allocation.  Both
            # forms travel: every rank uses the fine-grained blocks only if EVERY rank has one and can open all the others'.
            fg = None
            if os.environ.get("AIR_IPC_FINEGRAINED", "1") == "1":
                with torch.cuda.device(dev):
                    ptr = ctypes.c_void_p()
                    if L.air_ipc_flags_alloc(ctypes.byref(ptr), 16 * 8) == 0:
                        buf = ctypes.create_string_buffer(64)
                        if L.air_ipc_handle_get(ptr, buf) == 0:
                            self._fg_ptr, fg = ptr, bytes(buf.raw)
                        else:
                            L.air_ipc_flags_free(ptr)
            mine = tuple(reduce_tensor(t) for t in (engine.flat_grads, engine.flat_params, self.flags)) + (fg,)
        except Exception as e:                                                 # noqa: BLE001
            local_error = e
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=group)
        if local_error is not None or any(x is None for x in everyone):
            raise _lib.AirHipError("ipc-rsag: a rank could not export its buffers (%r)" % (local_error,))
        # fine-grained flags: open every peer's block; agreed on (one more exchange of a boolean per rank)
        fg_ptrs, fg_ok = [None] * self.world, all(x[3] is not None for x in everyone)
        if fg_ok:
            with torch.cuda.device(dev):
                for q in range(self.world):
                    if q == self.rank:
                        fg_ptrs[q] = self._fg_ptr.value
                        continue
                    pp = ctypes.c_void_p()
                    if L.air_ipc_handle_open(ctypes.create_string_buffer(everyone[q][3], 64), ctypes.byref(pp)) != 0:
                        fg_ok = False
                        break
                    self._fg_open.append(pp); fg_ptrs[q] = pp.value
        oks = [None] * self.world
        dist.all_gather_object(oks, bool(fg_ok), group=group)
        fg_ok = all(oks)
        if not fg_ok:
            self._release_finegrained()
        else:
            self.flags_kind = "fine-grained (hipExtMallocWithFlags), HIP IPC handles"
        self._keep = []
        self.struct = _lib.AirIpcPeers()
        self.struct.world, self.struct.rank = self.world, self.rank
        for q in range(self.world):
            if q == self.rank:
                g, p_, f = engine.flat_grads, engine.flat_params, self.flags
            else:
                g, p_, f = (fn(*args) for fn, args in everyone[q][:3])
                if g.device != dev:                                             # another GPU of the node: peer access both ways
                    if not torch.cuda.can_device_access_peer(dev.index, g.device.index):
                        raise _lib.AirHipError("GPU %d cannot map GPU %d's memory" % (dev.index, g.device.index))
                    _ = f[:1].to(dev)                                           # (torch enables peer access on the first P2P copy)
                self._keep += [g, p_, f]
            self.struct.grads[q], self.struct.params[q] = g.data_ptr(), p_.data_ptr()
            self.struct.flags[q] = fg_ptrs[q] if fg_ok else f.data_ptr()
        torch.cuda.synchronize(dev)

    def _release_finegrained(self):
        from . import hip as H
        L = H.lib()
        for pp in self._fg_open:
            L.air_ipc_handle_close(pp)
        self._fg_open = []
        if self._fg_ptr is not None:
            L.air_ipc_flags_free(self._fg_ptr)
            self._fg_ptr = None

    def zero_flags(self, engine):
        """this rank's flag block back to zero (self-test; collective callers synchronise around it)"""
        if self._fg_ptr is not None:
            from . import hip as H
            _ = H.lib().air_ipc_flags_zero(self._fg_ptr, 16 * 8, engine._sp())
            engine.synchronize()
        else:
            self.flags.zero_()

    def release(self):
        """unmap the peers' fine-grained blocks and free this rank's (after a barrier: nobody may still be spinning on them)"""
        self._release_finegrained()

    def plan(self, engine):
        """the three launches that follow the backward in the step's graph"""
        from . import hip as H
        L, p, cfg = H.lib(), H._p, engine.cfg
        peers = ctypes.byref(self.struct)
        tail_mult = cfg.baseline_lr_mult if cfg.use_reinforce else 0.0
        upd = (peers, p(engine.flat_ms), p(engine.flat_mg), p(engine.flat_mom), ctypes.c_size_t(engine.n_model),
               ctypes.c_size_t(engine.n_total), p(engine.lr_dev), tail_mult, cfg.rms_decay, cfg.rms_momentum, cfg.rms_eps,
               p(engine.step_dev), p(engine.rng_state), ctypes.c_uint64(engine._rng_inc))
        return [(L.air_dp_ipc_barrier, (peers, 0, p(self.local), p(self.err)), "air_dp_ipc_barrier"),
                (L.air_dp_ipc_rs_update_ag, upd, "air_dp_ipc_rs_update_ag"),
                (L.air_dp_ipc_barrier, (peers, 1, p(self.local), p(self.err)), "air_dp_ipc_barrier")]

    def timed_out(self) -> bool:
        """a barrier of this rank gave up: a peer did not arrive within the bounded spin (err 1), or the barrier's workgroups did not
        cover every XCD of the device (err 2, comm_ipc.hip)"""
        return bool(self.err.item() != 0)

    def shard(self, engine):
        """[lo, hi) of the flat buffers this rank's update owns (ipc_rs_update_ag_kernel: ceil(n/4 / world) float4 per rank)"""
        nq = engine.n_total // 4
        per = (nq + self.world - 1) // self.world
        lo = min(per * self.rank, nq)
        return 4 * lo, 4 * min(lo + per, nq)

    def selftest(self, engine, group=None) -> bool:
        """Known-answer check before the protocol is adopted: ONE eager barrier | shard update | barrier on known
        gradients -- rank r contributes (r + 1) * pattern, so the rank-order sum is pattern * world (world + 1) / 2 exactly in fp32 for
        the small integers used -- against centred RMSProp evaluated in torch on that sum, on every element of every replica; then
        every buffer the step touched is restored.  Collective; agreed on."""
        ok = True
        try:
            dev = engine.device
            names = ("flat_params", "flat_grads", "flat_ms", "flat_mg", "flat_mom", "step_dev", "rng_state")
            with engine.stream_context():
                saved = {k: getattr(engine, k).clone() for k in names}
                n = engine.n_total
                pattern = ((torch.arange(n, device=dev) % 13) - 6).to(torch.float32) * 0.125        # exact small dyadic values
                engine.flat_grads.copy_(pattern * float(self.rank + 1))
                for k in ("flat_ms", "flat_mg", "flat_mom"):
                    getattr(engine, k).zero_()
            engine.synchronize()
            dist.barrier(group=group)                                # every rank's known gradients are in place
            engine._run(self.plan(engine), engine._sp())
            engine.synchronize()
            cfg = engine.cfg
            with engine.stream_context():
                g = pattern * (self.world * (self.world + 1) / 2.0) * (1.0 / self.world)
                lr = torch.full((n,), float(engine.lr_dev.item()), device=dev)
                lr[engine.n_model:] *= (cfg.baseline_lr_mult if cfg.use_reinforce else 0.0)
                ms = (1.0 - cfg.rms_decay) * g * g
                mg = (1.0 - cfg.rms_decay) * g
                mom = lr * g / torch.sqrt(ms - mg * mg + cfg.rms_eps)
                want = saved["flat_params"] - mom
                err = (engine.flat_params - want).abs().max().item()
                scale = mom.abs().max().item() + 1e-30
                lo, hi = self.shard(engine)
                slots_ok = bool(torch.allclose(engine.flat_mom[lo:hi], mom[lo:hi], rtol=1e-5, atol=1e-6 * scale))
                ok = err <= 1e-5 * scale + 1e-7 and slots_ok and not self.timed_out()
            dist.barrier(group=group)                                # nobody restores while a peer may still be reading / pushing
            with engine.stream_context():
                for k in names:
                    getattr(engine, k).copy_(saved[k])
            engine.synchronize()
            self.local.zero_(); self.err.zero_(); self.zero_flags(engine)
            torch.cuda.synchronize(dev)
            dist.barrier(group=group)
        except Exception as e:                                        # noqa: BLE001
            ok, self.selftest_error = False, repr(e)
        return _agree(ok, engine.device, group)


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
        sync_shadow = getattr(engine, "_sync_param_shadow", None)
        if sync_shadow is not None and self.world > 1:
            sync_shadow()                                   # the bf16 shadow of the parameters follows whatever wrote them
        engine.synchronize()
        want = (collective or os.environ.get("AIR_DP_COLLECTIVE", "").strip().lower() or "torch-overlap")
        want = {"captured": "rccl-captured", "split": "torch-split", "torch": "torch-split", "rccl": "rccl-split",
                "overlap": "torch-overlap"}.get(want, want)
        want = {"ipc": "ipc-rsag"}.get(want, want)
        if want not in ("torch-overlap", "torch-split", "rccl-split", "rccl-captured", "ipc-rsag"):
            raise ValueError("AIR_DP_COLLECTIVE / collective must be torch-overlap, torch-split, rccl-split, rccl-captured or "
                             "ipc-rsag, got %r" % want)
        if overlap is None:
            overlap = os.environ.get("AIR_DP_OVERLAP", "0") == "1"
        on_gpu = getattr(getattr(engine, "device", None), "type", "cpu") == "cuda"
        self.collective = "none"
        self.rccl_nranks = None
        self._comm_side = None
        self._ipc = self._ipc_plan = None
        if self.world > 1:
            self.collective = "torch-split"
            if on_gpu and want == "ipc-rsag":
                # peer mapping + capture, each followed by an agreement: every rank ends up in the same protocol
                ok = (getattr(engine, "_use16", False) is False and getattr(engine.cfg, "rms_centered", True)
                      and engine.n_total % 4 == 0 and engine.n_model % 4 == 0)
                try:
                    if ok:
                        self._ipc = IpcPeerBuffers(engine, group)
                except Exception as e:                              # noqa: BLE001
                    ok, self._ipc_error = False, repr(e)
                if _agree(ok, engine.device, group):
                    dist.barrier(group=group)                       # every rank's flags are mapped everywhere before anyone launches
                    good = self._ipc.selftest(engine, group)        # known answer first: the protocol has never met a second GPU
                    if not good:
                        self._ipc_error = "ipc-rsag self-test failed: %s" % getattr(self._ipc, "selftest_error", "wrong result")
                    if good:
                        self._ipc_plan = self._ipc.plan(engine)
                        try:
                            self._capture_ipc(capture_graph)
                        except Exception as e:                      # noqa: BLE001
                            good, self._ipc_error = False, repr(e)
                        good = _agree(good, engine.device, group)
                    if good:
                        dist.barrier(group=group)
                        self.collective = "ipc-rsag"
                        self._ipc_capture_graph = bool(capture_graph)
                        # the slots of (world - 1) / world of the parameters are stale on this rank from now on: the engine's own
                        # state_dict() refuses until gather_optimizer_state() has run; update_config() re-captures through this wrapper
                        engine._slots_sharded = True
                        engine._recapture_hook = self._recapture_ipc
                        self._ipc_steps = 0
                        return
                    engine.release_graphs()
                if self._ipc is not None:
                    dist.barrier(group=group)                       # (nobody frees a flag block a peer's self-test may still touch)
                    self._ipc.release()
                self._ipc = self._ipc_plan = None
                want = "torch-split"
            if on_gpu and want in ("rccl-split", "rccl-captured"):
                # Every step below is collective and ends in an agreement, so all ranks leave it in the same protocol.
                try:
                    self.comm = create_rccl_comm(engine.device, group)
                    if overlap and want == "rccl-captured":
                        self._comm_side = create_rccl_comm(engine.device, group)     # the forked stream's own communicator
                    good = selftest_comm(self.comm, engine.device, engine.stream, group)
                    if good and self._comm_side is not None:
                        good = selftest_comm(self._comm_side, engine.device, engine.stream, group)
                except Exception as e:                              # noqa: BLE001 -- raised on every rank alike (see create_rccl_comm)
                    good, self._captured_error = False, repr(e)
                if good:
                    self.rccl_nranks = comm_count(self.comm)
                    if want == "rccl-captured" and capture_graph:
                        ok = True
                        try:
                            engine.capture(comm=self.comm, overlap=bool(overlap), comm_side=self._comm_side)
                        except Exception as e:                      # noqa: BLE001 -- a refused capture: agreed fallback below
                            ok, self._captured_error = False, repr(e)
                        if _agree(ok, engine.device, group):
                            self.collective = "rccl-captured" + ("+overlap" if overlap else "")
                            return
                        want = "rccl-split"
                    self.collective = "rccl-split"
            bucketed = (want == "torch-overlap" and capture_graph and getattr(engine, "supports_bucketed_backward", False)
                        and engine.tail_bucket() is not None)
            if bucketed:
                engine.capture(split_optimizer=True, split_backward=True)
                self.collective = "torch-overlap"
            elif capture_graph:
                engine.capture(split_optimizer=True)
        elif capture_graph:
            if steps_per_replay > 1:        # single GPU only: several updates per graph replay (engine.capture)
                engine.capture(steps_per_replay=steps_per_replay)
                self.steps_per_replay = int(steps_per_replay)
            else:
                engine.capture()

    def _capture_ipc(self, capture_graph=True):
        """the ipc-rsag step as ONE graph: forward | backward | barrier | shard update + push | barrier"""
        eng = self.engine
        eng.release_graphs()
        eng.synchronize()
        eng._capture_kwargs = None                           # (a stale capture() of another protocol must not be replayed by update_config)
        if capture_graph:
            eng._graph = eng._capture_plans([eng._plan_fwd_train, eng._plan_bwd, self._ipc_plan])
            eng._graph_has_opt, eng._graph_b2, eng._steps_per_replay = True, None, 1

    def _recapture_ipc(self):
        """engine.update_config() rebuilt the plans (use_prior, explore_eps, bias switches of mnist_model): the ipc graph follows"""
        self._ipc_plan = self._ipc.plan(self.engine)
        self._capture_ipc(self._ipc_capture_graph)

    def gather_optimizer_state(self):
        """ipc-rsag shards the RMSProp slots (rank r updates only its 1/world slice): all-gather the slices so that every rank holds
        the complete flat_ms / flat_mg / flat_mom again -- before a checkpoint, and before the ranks go back to a protocol in which
        every rank updates everything.  Collective.  A no-op for the other protocols."""
        if self._ipc is None or self.world == 1:
            return
        eng = self.engine
        eng.synchronize()
        nq = eng.n_total // 4
        per = 4 * ((nq + self.world - 1) // self.world)
        lo, hi = self._ipc.shard(eng)
        with self._stream():
            for name in ("flat_ms", "flat_mg", "flat_mom"):
                buf = getattr(eng, name)
                mine = torch.zeros(per, dtype=buf.dtype, device=buf.device)
                mine[:hi - lo] = buf[lo:hi]
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.group)
                full = torch.cat(parts)[:eng.n_total]
                buf.copy_(full)
        eng.synchronize()

    def state_dict(self):
        """the engine's checkpoint with COMPLETE optimiser slots whatever the protocol (collective: every rank calls it)"""
        self.gather_optimizer_state()
        eng = self.engine
        sharded = getattr(eng, "_slots_sharded", False)
        eng._slots_sharded = False
        try:
            return eng.state_dict()
        finally:
            eng._slots_sharded = sharded

    def _stream(self):
        ctx = getattr(self.engine, "stream_context", None)
        return ctx() if ctx is not None else contextlib.nullcontext()

    def _allreduce(self, grads):
        if self.collective == "rccl-split":                 # the engine's own eager ncclAllReduce on its stream
            from . import _lib
            from . import hip as H
            st = H.lib().air_allreduce_sum(ctypes.c_void_p(grads.data_ptr()), ctypes.c_size_t(grads.numel()), self.comm,
                                           ctypes.c_void_p(self.engine.stream.cuda_stream))
            if st != 0:
                raise _lib.AirHipError("air_allreduce_sum failed: %s" % (H.lib().air_comm_last_error() or b"").decode())
            return
        allreduce_gradients(grads, self.group, average=False)

    def _allreduce_async(self, grads):
        return dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def train_step(self, obs=None):
        if self.collective == "ipc-rsag":
            eng = self.engine
            if eng._graph is not None:
                eng.train_step(obs)                         # ONE replay: forward, backward, barrier, shard update + push, barrier
            else:
                if obs is not None:
                    eng.set_obs(obs)
                for pl in (eng._plan_fwd_train, eng._plan_bwd, self._ipc_plan):
                    eng._run(pl, eng._sp())
                eng.global_step += 1
            # a barrier that gave up leaves garbage behind it: look at the error word every IPC_CHECK_EVERY steps (one host
            # synchronisation) instead of training on until somebody asks replicas_in_sync()
            self._ipc_steps += 1
            if self._ipc_steps % self.IPC_CHECK_EVERY == 0:
                eng.synchronize()
                if self._ipc.timed_out():
                    from . import _lib
                    raise _lib.AirHipError("ipc-rsag: a barrier gave up (err %d: 1 = a peer never arrived, 2 = XCDs not covered); "
                                           "the replicas are no longer valid" % int(self._ipc.err.item()))
            return
        host_collective = self.world > 1 and not self.collective.startswith("rccl-captured")
        if self.collective == "torch-overlap":
            self.engine.train_step(obs, allreduce=self._allreduce, allreduce_async=self._allreduce_async)
        else:
            self.engine.train_step(obs, allreduce=self._allreduce if host_collective else None)

    IPC_CHECK_EVERY = 256

    def replicas_in_sync(self) -> bool:
        """True when every rank holds bit-identical parameters (collective: every rank must call it).  Data-parallel replicas start
        from a broadcast and apply the same averaged gradient, so any difference means a collective reduced a buffer that was not
        final yet, or a rank skipped one -- the check a new overlap protocol has to pass on real multi-GPU hardware."""
        if self.world == 1:
            return True
        self.engine.synchronize()
        if self._ipc is not None and self._ipc.timed_out():            # a peer never arrived at a barrier: nothing after it is valid
            ok = False
        else:
            ok = True
        p = self.engine.flat_params
        # two order-independent 64-bit digests of the raw bits, compared through MIN / MAX over the ranks
        bits = p.view(torch.int32).to(torch.int64)
        idx = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64)
        dig = torch.stack([bits.sum(), (bits * (idx % 65521)).sum()])          # int64, wrapping: equal bits => equal digests
        lo, hi = dig.clone(), dig.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        return bool(torch.equal(lo, hi)) and _agree(ok, self.engine.device, self.group)

    def close(self):
        if self._ipc is not None:
            self.gather_optimizer_state()                   # every rank leaves with complete slots (whatever protocol comes next)
            self.engine.synchronize()
            self.engine.release_graphs()
            dist.barrier(group=self.group)                  # nobody unmaps while a peer may still be pushing
            self._ipc.release()
            self._ipc = self._ipc_plan = None
            self.engine._slots_sharded = False
            self.engine._recapture_hook = None
        if self.comm is not None:
            from . import hip as H
            self.engine.synchronize()
            self.engine.release_graphs()
            for c in (self._comm_side, self.comm):
                if c is not None:
                    H.lib().air_comm_destroy(c)
            self.comm = self._comm_side = None
