"""AIREngine -- the fused MI355X train step of AIR on multi-MNIST-shaped inputs.

One object owns every byte the step touches: a flat fp32 parameter buffer (Sonnet layouts), a flat gradient buffer
(=> a single RCCL all-reduce in data-parallel runs), flat centred-RMSProp slots and a static activation arena.  The
forward unroll, the hand-derived backward and the optimiser are a fixed list of C-ABI launches (include/air_hip.h)
over those buffers, so the whole step is hipGraph-capturable and replays with zero Python in the loop.

What is restated from the reference (file:line under attend_infer_repeat/):
  forward   cell.py:116-171 unrolled by model.py:83-84, re-scheduled: the input encoder and x.W_x of the LSTM are
            computed once (the image never changes, cell.py:121-125), the T tiny recurrences run, and everything
            downstream of h_t is batched over T*B rows.  Same numbers up to fp32 summation order.
  objective model.py:126-259,319-343 (ELBO with analytic step weights, NVIL with the [B,B] broadcast quirk).
  update    model.py:355-367 (centred RMSProp, baseline at 10x lr).
torch supplies device memory, streams and (optionally) torch.distributed; all arithmetic is in libair_hip.so.
"""
import ctypes
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import hip as H
from .engine_config import (EngineConfig, _Mlp, _mlp_shapes, anneal_weight, geometric_prior_f64,  # noqa: F401 (re-exported)
                            param_shapes)
from .engine_plan import PlanMixin


_ROCTX = [False, None]


def _roctx():
    """libroctx64 when AIR_ROCTX=1 (and the library is there), else None; resolved once per process"""
    if not _ROCTX[0]:
        _ROCTX[0] = True
        if os.environ.get("AIR_ROCTX", "0") == "1":
            for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
                try:
                    lib = ctypes.CDLL(name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]; lib.roctxRangePushA.restype = ctypes.c_int
                    lib.roctxRangePop.restype = ctypes.c_int
                    _ROCTX[1] = lib
                    break
                except (OSError, AttributeError):
                    continue
    return _ROCTX[1]


class AIREngine(PlanMixin):
    def __init__(self, cfg: EngineConfig, batch_size: int, device=None, seed: int = 0, keep_canvas_steps: bool = True):
        H.lib()
        self.cfg = cfg
        self.B = int(batch_size)
        self.T = int(cfg.max_steps)
        self.M = self.T * self.B
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise _lib.AirHipError("AIREngine needs a HIP device; there is no CPU fallback")
        self.keep_canvas_steps = keep_canvas_steps
        self.global_step = 0
        self.world_size = 1
        self._graph = None
        self._graph_opt = None
        self._graph_has_opt = True
        self._side_stream = None
        self._side_events = None
        self._bufs: Dict[str, torch.Tensor] = {}
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream(device=self.device)
            self._alloc_params(seed)
            self._alloc_activations()
            # private split-K workspace: engines on different streams must not share slabs
            self.ws = torch.empty((32 << 20) // 4, dtype=torch.float32, device=self.device)
            self._build_plans()
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------------------------------------------------
    # memory
    # ------------------------------------------------------------------------------------------------------------
    def _alloc_params(self, seed):
        cfg = self.cfg
        shapes = param_shapes(cfg)
        sizes = {k: int(math.prod(s)) for k, s in shapes.items()}
        # every tensor starts on a 16-byte boundary so vectorised operand loads apply
        offs, off = {}, 0
        self.n_model = None
        for k, n in sizes.items():
            if k.startswith("baseline/") and self.n_model is None:
                off = (off + 63) // 64 * 64
                self.n_model = off
            offs[k] = off
            off += (n + 3) // 4 * 4
        if self.n_model is None:
            self.n_model = off
        self.n_total = off
        dev = self.device
        flat = self._alloc_flat(5, self.n_total, dev)
        self.flat_params, self.flat_grads, self.flat_ms, self.flat_mg, self.flat_mom = flat
        self.flat_ms.fill_(1.0)
        self.params = {k: self.flat_params[offs[k]:offs[k] + sizes[k]].view(shapes[k]) for k in shapes}
        self.grads = {k: self.flat_grads[offs[k]:offs[k] + sizes[k]].view(shapes[k]) for k in shapes}
        self.param_offsets, self.param_sizes, self.param_shapes = offs, sizes, shapes
        self.lr_dev = torch.tensor([cfg.learning_rate], dtype=torch.float32, device=dev)
        self.rng_state = torch.tensor([seed, 0], dtype=torch.int64, device=dev)
        self.init_parameters(seed)

    @staticmethod
    def _alloc_flat(count, n, dev):
        """The flat parameter / gradient / RMSProp-slot buffers.  AIR_FLAT_LAYOUT (developer switch, placement probe of round 5:
        the same step ran 4-7 % faster or slower depending on WHERE the caching allocator happened to put these five arrays):
          unset            one torch allocation each (what the allocator gives: 2 MB-aligned segments of their own in a fresh
                           process once they exceed 10 MB, carved back to back out of a cached block otherwise)
          packed           one arena, the arrays back to back (512-byte granules)
          stagger:<bytes>  one arena, array k at a 2 MB-aligned slot + k * <bytes>"""
        mode = os.environ.get("AIR_FLAT_LAYOUT", "")
        if not mode:
            return [torch.zeros(n, dtype=torch.float32, device=dev) for _ in range(count)]
        nbytes = n * 4
        if mode == "packed":
            pitch, stagger = (nbytes + 511) // 512 * 512, 0
        elif mode.startswith("stagger:"):
            stagger = int(mode.split(":", 1)[1])
            if stagger % 16:
                raise ValueError("AIR_FLAT_LAYOUT=stagger:<bytes>: a multiple of 16 (vectorised operand loads)")
            pitch = ((nbytes + count * stagger) + (2 << 20) - 1) // (2 << 20) * (2 << 20)
        else:
            raise ValueError(f"AIR_FLAT_LAYOUT={mode!r}: expected 'packed' or 'stagger:<bytes>'")
        arena = torch.zeros(count * pitch + (2 << 20), dtype=torch.uint8, device=dev)
        base = (-arena.data_ptr()) % (2 << 20)                # first 2 MB boundary inside the arena
        out = []
        for k in range(count):
            lo = base + k * pitch + k * stagger
            out.append(arena[lo:lo + nbytes].view(torch.float32))
        return out

    def init_parameters(self, seed: int = 0):
        """Sonnet defaults: w ~ TruncNormal(0, 1/sqrt(fan_in)) (+-2 sigma), b = 0, LSTM initial state = 0."""
        gen = torch.Generator(device="cpu").manual_seed(int(seed))
        for k, shape in self.param_shapes.items():
            if len(shape) == 2 and not k.endswith(("/h0", "/c0")):
                w = torch.empty(shape, dtype=torch.float32)
                torch.nn.init.trunc_normal_(w, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
                self._copy_in(self.params[k], w * (1.0 / math.sqrt(shape[0])))
            else:
                self._fill_in(self.params[k], 0.0)
        if getattr(self, "flat_params16", None) is not None:
            self._sync_param_shadow()

    # ---- stream discipline ------------------------------------------------------------------------------------------
    # The engine runs on its own stream.  Everything that enters its buffers from outside (a batch gathered on the default
    # stream, checkpoint tensors, injected noise) is ordered explicitly: the engine stream first waits for the producer's
    # stream, the copy runs ON the engine stream, and a device-side source is marked as in use by the engine stream so the
    # caching allocator cannot hand its block to a later allocation while the copy is still pending.
    def _copy_in(self, dst: torch.Tensor, src):
        src_t = src if torch.is_tensor(src) else torch.as_tensor(src)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            dst.copy_(src_t.reshape(dst.shape), non_blocking=True)
        if src_t.is_cuda:
            src_t.record_stream(self.stream)

    def _fill_in(self, dst: torch.Tensor, value):
        with torch.cuda.stream(self.stream):
            dst.fill_(value)

    def wait_for_engine(self):
        """Order the CALLER's current stream after everything queued on the engine stream (before torch code on another
        stream reads buffers the engine writes: the shared parameters, outputs)."""
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def wait_for_caller(self):
        """Order the engine stream after the caller's current stream (after torch code on another stream touched buffers
        the next engine launch reads or overwrites)."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def load_parameters(self, named: Dict[str, torch.Tensor]):
        for k, v in named.items():
            self._copy_in(self.params[k], torch.as_tensor(v).to(torch.float32))
        self._sync_param_shadow()

    def load_optimizer_slots(self, ms=None, mg=None, mom=None):
        """RMSProp slots by parameter name (tf_checkpoint.import_tf_optimizer_slots): mean square, mean gradient (centred form), momentum.
        Parameters a dict does not name keep what they have."""
        for flat, named in ((self.flat_ms, ms), (self.flat_mg, mg), (self.flat_mom, mom)):
            for k, v in (named or {}).items():
                off, n = self.param_offsets[k], self.param_sizes[k]
                self._copy_in(flat[off:off + n], torch.as_tensor(v).to(torch.float32).reshape(-1))

    def reset_optimizer(self):
        self._fill_in(self.flat_ms, 1.0); self._fill_in(self.flat_mg, 0.0); self._fill_in(self.flat_mom, 0.0)

    def _buf(self, name, shape, dtype=torch.float32):
        if name not in self._bufs:
            self._bufs[name] = torch.zeros(shape, dtype=dtype, device=self.device)
        return self._bufs[name]

    def _alloc_activations(self):
        cfg, B, T, M = self.cfg, self.B, self.T, self.M
        Hd, A, P, hw = cfg.n_hidden, cfg.n_appearance, cfg.n_pix, cfg.n_crop
        b = self._buf
        self.obs = b("obs", (B, P))
        # noise: one flat normal buffer [eps_where | eps_what], one uniform buffer
        self.noise_normal = b("noise_normal", (M * 4 + M * A,))
        self.eps_where = self.noise_normal[:M * 4].view(T, B, 4)
        self.eps_what = self.noise_normal[M * 4:].view(T, B, A)
        self.u_pres = b("u_pres", (T, B))
        self.prior_dev = b("prior", (T + 1,), torch.float64)
        self.step_dev = b("global_step", (1,), torch.int64)
        self.enc = _Mlp(self, "input_encoder", B, P, cfg.inpt_encoder_hidden, None)
        self.gx = b("gx", (B, 4 * Hd))
        self.gate_act = b("gate_act", (T, B, 4 * Hd))
        self.h_seq = b("h_seq", (T + 1, B, Hd)); self.c_seq = b("c_seq", (T + 1, B, Hd))
        self.tr = _Mlp(self, "transform", M, Hd, cfg.transform_estimator_hidden, 8)
        self.st = _Mlp(self, "steps", M, Hd, cfg.steps_pred_hidden, 1)
        self.where_loc = b("where_loc", (T, B, 4)); self.where_scale = b("where_scale", (T, B, 4))
        self.where = b("where", (T, B, 4)); self.kl_where_row = b("kl_where_row", (T, B))
        self.presence_prob = b("presence_prob", (T, B)); self.presence = b("presence", (T, B))
        self.glimpse_in = b("glimpse_in", (T, B, hw))
        self.ge = _Mlp(self, "glimpse_encoder", M, hw, cfg.glimpse_encoder_hidden, None)
        self.q = b("what_pre", (M, 2 * A)); self.dq = b("dwhat_pre", (M, 2 * A))
        self.what_loc = b("what_loc", (T, B, A)); self.what_scale = b("what_scale", (T, B, A))
        self.what = b("what", (T, B, A)); self.kl_what_row = b("kl_what_row", (T, B))
        self.gd = _Mlp(self, "glimpse_decoder", M, A, cfg.glimpse_decoder_hidden, hw)
        self.canvas_steps = b("canvas_steps", (T, B, P)) if self.keep_canvas_steps else None
        self.final_canvas = b("final_canvas", (B, P)); self.rec = b("rec", (B,))
        # the canvas unroll runs one workgroup per (image, row band) so that a small batch fills the chip; each band leaves
        # its share of the reconstruction term here and the consumer (NVIL, or a plain sum) adds the shares in band order
        self.n_bands = int(H.lib().air_canvas_unroll_bands(B, int(cfg.img_size[0])))
        self.rec_parts = b("rec_parts", (self.n_bands, B))
        self.q_n = b("q_n", (B, T + 1)); self.kl_n = b("kl_n", (B,)); self.logp = b("logp", (B,))
        self.step_w = b("step_w", (T, B))
        # baseline input [obs | what | where | presence | h | c] (modules.py:131-139) is never materialised: the obs columns
        # of the first layer are multiplied straight from obs (in the same launch as the input encoder's first layer),
        # only the latent columns are packed
        self.base_lat = b("base_lat", (B, cfg.baseline_in - P))
        self.bl_obs = b("bl_obs", (B, _mlp_shapes(cfg.baseline_in, cfg.baseline_hidden, 1)[0][1]))
        self.bl = _Mlp(self, "baseline", B, cfg.baseline_in, cfg.baseline_hidden, 1)
        self.nvil_out = b("nvil_out", (4,)); self.dlogp = b("dlogp", (B,)); self.dbase = b("dbase", (B,))
        # backward scratch
        self.d_what = b("d_what", (M, A)); self.d_glimpse_in = b("d_glimpse_in", (M, hw))
        self.dwhere_w = b("dwhere_w", (4 * M, 4));    # (up to four partial slabs: air_canvas_unroll_fwd_bwd's n_split)
        self.dwhere_r = b("dwhere_r", (M, 4)); self.dwhere = b("dwhere", (M, 4))
        self.dkl_row = b("dkl_row", (M,)); self.dstep_w = b("dstep_w", (T, B)); self.dprob = b("dprob", (T, B))
        self.dH = b("dH", (T, B, Hd)); self.dH_b = b("dH_b", (T, B, Hd)); self.dh_init = b("dh_init", (B, Hd))
        self.ones_b = b("ones_b", (B, 1)); self.ones_b.fill_(1.0)
        self.dgates = b("dgates", (T, B, 4 * Hd)); self.dc_a = b("dc_a", (B, Hd)); self.dc_b = b("dc_b", (B, Hd))
        self.dgx = b("dgx", (B, 4 * Hd))

    # ------------------------------------------------------------------------------------------------------------
    # launch plans
    # ------------------------------------------------------------------------------------------------------------

    def _run(self, plan, stream_ptr):
        """Issue a plan: entries (fn, args, name[, ...]) on the engine stream, in order.  AIR_ROCTX=1: every entry inside a roctx range
        named "<position> <C-ABI entry>" (rocprofv3 --marker-trace shows the plan next to the kernel trace of an EAGER
        step -- `bench.py --no-graph`; a captured graph replays kernel nodes only, its positions are tools/probes/plan_dump.py's)."""
        rx = _roctx()
        for i, e in enumerate(plan):
            if rx is not None:
                rx.roctxRangePushA(("%02d %s" % (i, e[2])).encode())
            st = e[0](*e[1], stream_ptr)
            if rx is not None:
                rx.roctxRangePop()
            if st != 0:
                _lib.check(st, e[2])

    def _opt_slice_entry(self, lo, hi, lane, counters=False):
        """centred RMSProp over elements [lo, hi) of the flat buffers (two learning rates around n_model) as one launch;
        counters=True: the closing launch, which also advances the device step counter and the Philox offset"""
        L, p, cfg = H.lib(), H._p, self.cfg
        assert lo % 4 == 0 and hi % 4 == 0 and lo < hi
        off = lambda t: ctypes.c_void_p(t.data_ptr() + 4 * lo)
        n_model = min(max(self.n_model - lo, 0), hi - lo)
        tail_mult = cfg.baseline_lr_mult if cfg.use_reinforce else 0.0
        args = (off(self.flat_params), off(self.flat_grads), off(self.flat_ms), off(self.flat_mg), off(self.flat_mom),
                ctypes.c_size_t(n_model), ctypes.c_size_t(hi - lo), p(self.lr_dev), tail_mult, cfg.rms_decay, cfg.rms_momentum,
                cfg.rms_eps, 1.0, p(self.step_dev) if counters else None, p(self.rng_state) if counters else None,
                ctypes.c_uint64(self._rng_inc if counters else 0))
        return (L.air_step_epilogue, args, "air_step_epilogue", lane)

    # ------------------------------------------------------------------------------------------------------------
    # public API
    # ------------------------------------------------------------------------------------------------------------
    def _sp(self):
        return ctypes.c_void_p(self.stream.cuda_stream)

    def set_learning_rate(self, lr: float):
        self._fill_in(self.lr_dev, float(lr))

    def set_obs(self, obs: torch.Tensor):
        self._copy_in(self.obs, obs)

    def set_noise(self, eps_where, eps_what, u_pres):
        self._copy_in(self.eps_where, eps_where)
        self._copy_in(self.eps_what, eps_what)
        self._copy_in(self.u_pres, u_pres)

    def steps_prior_success_prob(self, global_step=None) -> float:
        cfg = self.cfg
        gs = self.global_step if global_step is None else global_step
        if cfg.nsp_anneal is None:
            return float(cfg.nsp_init)
        return anneal_weight(cfg.nsp_init, cfg.nsp_final, cfg.nsp_anneal, gs, cfg.nsp_steps, cfg.nsp_hold_init,
                             cfg.nsp_steps_div)

    def set_global_step(self, step: int):
        """Host mirror + the device counter the captured graph reads (annealing schedule, model.py:106-124)."""
        self.global_step = int(step)
        self._fill_in(self.step_dev, int(step))

    def sample_noise(self):
        self._run(self._plan_rng, self._sp())

    def forward(self, obs=None, sample_noise=True):
        """T-step unroll + objective terms.  Results live in the engine's buffers (see `outputs()`)."""
        if obs is not None:
            self.set_obs(obs)
        self._run(self._plan_fwd_noise if sample_noise else self._plan_fwd, self._sp())

    def backward(self):
        """Fills flat_grads with d opt_loss / d model vars and d baseline_loss / d baseline vars (model.py:355-367)."""
        self._run(self._plan_bwd, self._sp())

    def optimizer_step(self, grad_scale: float = 1.0):
        plan = self._plan_opt if grad_scale == 1.0 else self._opt_calls_factory(float(grad_scale))
        self._run(plan, self._sp())
        self.global_step += 1

    def _capture_plans(self, plans):
        """Capture a list of plan entries into one hipGraph.  An entry is either a launch plan (list of (fn, args, name)) or
        a callable taking the stream pointer (collectives, stream forks / joins)."""
        L = H.lib()
        sp = self._sp()
        _lib.check(L.air_graph_begin_capture(sp), "air_graph_begin_capture")
        try:
            for pl in plans:
                if callable(pl):
                    pl(sp)
                else:
                    self._run(pl, sp)
        finally:
            exe = ctypes.c_void_p()
            st = L.air_graph_end_capture(sp, ctypes.byref(exe))
        _lib.check(st, "air_graph_end_capture")
        return exe

    def _allreduce_call(self, comm, lo, hi):
        L = H.lib()
        ptr = ctypes.c_void_p(self.flat_grads.data_ptr() + 4 * lo)

        def call(sp):
            st = L.air_allreduce_sum(ptr, ctypes.c_size_t(hi - lo), comm, sp)
            if st != 0:
                raise _lib.AirHipError("air_allreduce_sum failed: %s" % (L.air_comm_last_error() or b"").decode())
        return call

    def _step_plans_for(self, obs):
        """the single-GPU train-step plans [forward, backward, update] with `obs` as the observation buffer"""
        saved = self.obs
        self.obs = obs
        try:
            self._build_plans()
            return self._single_gpu_step_plans()
        finally:
            self.obs = saved

    def _single_gpu_step_plans(self):
        """the plans of one complete single-GPU update, best form first: optimiser riders, plain.  (A two-lane form -- a captured
        graph with a forked side lane for the baseline, the weight gradients and the updates -- was built and measured in round 3:
        0.4287 against 0.2117 ms, a fork/join costs 25-45 us per replay on ROCm 7.2; removed in round 4, the numbers are in
        DESIGN section 3 and profiles/r03_c_bench_c2_b64_two_lane_rejected.json.)"""
        if self._plan_bwd_riders is not None:
            return [self._plan_fwd_train, self._plan_bwd_riders, self._plan_opt_rest]
        return [self._plan_fwd_train, self._plan_bwd, self._plan_opt]

    def attach_dataset(self, data: torch.Tensor, shuffle: bool = True, seed: int = 0, rank: int = 0, world: int = 1):
        """HBM-resident input pipeline (the reference feeds every step through tf.py_func, data.py:121-158): `data` [N, H*W] (or
        [N, H, W]) stays on the device and every train step starts by gathering its own batch -- indices drawn with replacement
        from Philox(seed, device step counter) like np.random.choice, or walking the data in order -- as the first launch of the
        step, inside the captured graph.  `batch_idx` holds the indices of the last batch (for labels).  data=None detaches.
        Data-parallel runs pass `rank` / `world`: every rank then draws from its own Philox stream (the seed is mixed with the
        rank, like the per-rank noise seeds), so the global batch is world * B distinct draws, not B draws seen `world` times.
        For an in-order walk (shuffle=False) under data parallelism, attach each rank's own shard.  Re-capture afterwards."""
        self.release_graphs()
        if data is None:
            self._feeder = None
        else:
            d = data.reshape(data.shape[0], -1)
            if not (d.is_cuda and d.dtype == torch.float32 and d.is_contiguous() and d.shape[1] == self.obs.shape[1]):
                raise ValueError("dataset must be a contiguous float32 device tensor of [N, %d] images" % self.obs.shape[1])
            self._feeder = (d, bool(shuffle))
            if world > 1:
                from .distributed import rank_seed
                seed = rank_seed(seed, rank)
            self.feeder_seed = torch.tensor([int(seed) & (2 ** 63 - 1)], dtype=torch.int64, device=self.device)
            self.batch_idx = torch.zeros(self.B, dtype=torch.int64, device=self.device)
        self._build_plans()

    def set_obs_slot(self, slot: int, obs: torch.Tensor):
        """batch for step `slot` of a multi-step replay (capture(steps_per_replay=K)); slot 0 is the ordinary obs buffer"""
        self._copy_in(self.obs if slot == 0 else self.obs_ring[slot - 1], obs)

    supports_bucketed_backward = True      # capture(split_optimizer=True, split_backward=True) / train_step(allreduce_tail=...)

    def tail_bucket(self):
        """(end, lo): the backward plan's first `end` launches leave elements [lo, n_total) of the flat gradient buffer final -- the
        EARLIEST such cut whose tail holds 30-70 % of the buffer (decoder / baseline / what / glimpse-encoder gradients: 46 % of
        the bytes, final when 40 % of the backward is still to run: the longer the rest of the backward, the more of the tail's
        all-reduce hides under it); None if the plan has no such cut (throughput regime: every weight gradient is formed at the
        end of the backward)."""
        for end, lo, hi in sorted(self._grad_buckets):
            if end < len(self._plan_bwd) and 0.3 * self.n_total <= self.n_total - lo <= 0.7 * self.n_total:
                return (end, lo)
        return None

    def capture(self, split_optimizer: bool = False, comm=None, overlap: bool = False, steps_per_replay: int = 1,
                comm_side=None, split_backward: bool = False):
        """Capture noise + forward + backward (+ gradient all-reduce) + both RMSProp updates into hipGraphs.
        comm=None, split_optimizer=False : single GPU, ONE graph.
        comm=<air_comm handle>           : data parallel, still ONE graph -- the RCCL all-reduce of the flat gradient buffer
                                           is a captured node between the backward and the update (grad_scale = 1/world).
                                           overlap=True additionally forks a captured side stream at the point of the backward
                                           where the tail of the buffer (decoder / baseline / what / glimpse-encoder
                                           gradients, ~half of the bytes) is final and all-reduces that slice there, while the
                                           main stream finishes the backward and all-reduces the head.
        split_optimizer=True             : forward+backward and the update as two graphs, so a host-issued collective
                                           (torch.distributed) can run between them (fallback when RCCL cannot be captured)."""
        self.release_graphs()
        self.stream.synchronize()
        self._steps_per_replay = 1
        self._capture_kwargs = dict(split_optimizer=split_optimizer, comm=comm, overlap=overlap,
                                    steps_per_replay=steps_per_replay, comm_side=comm_side, split_backward=split_backward)
        self._graph_b2, self._tail_lo = None, None
        L = H.lib()
        if comm is not None:
            opt = self._opt_calls_factory(1.0 / self.world_size)
            cut = None
            if overlap and len(self._grad_buckets) >= 2:
                # the latest cut whose tail holds at most ~60 % of the buffer: [lo, n_total) is final after plan index `end`
                for end, lo, hi in self._grad_buckets:
                    if self.n_total - lo <= 0.6 * self.n_total:
                        cut = (end, lo)
            if cut is None:
                plans = [self._plan_fwd_train, self._plan_bwd, self._allreduce_call(comm, 0, self.n_total), opt]
            else:
                end, lo = cut
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(device=self.device)
                    ev = [ctypes.c_void_p(), ctypes.c_void_p()]
                    for e in ev:
                        _lib.check(L.air_event_create(ctypes.byref(e)), "air_event_create")
                    self._side_events = ev
                side = ctypes.c_void_p(self._side_stream.cuda_stream)
                ev_fork, ev_join = self._side_events
                # (the forked stream reduces on a communicator of its own when one is given: two ncclAllReduce calls in
                #  flight on ONE communicator from two streams is not something RCCL promises to order)
                tail_reduce = self._allreduce_call(comm_side if comm_side is not None else comm, lo, self.n_total)

                def fork(sp):
                    _lib.check(L.air_event_record(ev_fork, sp), "air_event_record")
                    _lib.check(L.air_stream_wait_event(side, ev_fork), "air_stream_wait_event")
                    tail_reduce(side)
                    _lib.check(L.air_event_record(ev_join, side), "air_event_record")

                def join(sp):
                    _lib.check(L.air_stream_wait_event(sp, ev_join), "air_stream_wait_event")

                plans = [self._plan_fwd_train, self._plan_bwd[:end], fork, self._plan_bwd[end:],
                         self._allreduce_call(comm, 0, lo), join, opt]
            self._graph = self._capture_plans(plans)
            self._graph_has_opt = True
            return
        self._steps_per_replay = 1
        if steps_per_replay > 1:
            # K consecutive updates per replay (a replay costs ~8 us on top of 1.7 us per node): step j reads its batch from
            # slot j of an observation ring that the caller fills ahead of the replay (set_obs_slot) -- an input queue of depth K.
            # Single GPU only: every step ends with its own update.
            if split_optimizer or self.world_size != 1:
                raise ValueError("steps_per_replay > 1 needs the single-GPU fused step")
            K = int(steps_per_replay)
            if getattr(self, "obs_ring", None) is None or self.obs_ring.shape[0] != K - 1:
                self.obs_ring = torch.empty((K - 1,) + tuple(self.obs.shape), dtype=torch.float32, device=self.device)
                self.obs_ring.copy_(self.obs.unsqueeze(0).expand_as(self.obs_ring))
            plans = []
            for j in range(K):
                plans += self._step_plans_for(self.obs if j == 0 else self.obs_ring[j - 1])
            self._step_plans_for(self.obs)                      # leave the engine's own plans on the ordinary buffer
            self._graph = self._capture_plans(plans)
            self._graph_has_opt = True
            self._steps_per_replay = K
            return
        cut = self.tail_bucket() if (split_optimizer and split_backward) else None
        if not split_optimizer and self.world_size == 1:
            self._graph = self._capture_plans(self._single_gpu_step_plans())
        elif cut is not None:
            # host-issued collective, two gradient buckets: [forward + backward up to the cut] | all-reduce of the tail (async, on the
            # collective library's stream) | [rest of the backward] | all-reduce of the head | [update]
            end, lo = cut
            self._graph = self._capture_plans([self._plan_fwd_train, self._plan_bwd[:end]])
            self._graph_b2 = self._capture_plans([self._plan_bwd[end:]])
            self._tail_lo = lo
        else:
            self._graph = self._capture_plans([self._plan_fwd_train, self._plan_bwd] + ([] if split_optimizer else [self._plan_opt]))
        self._graph_has_opt = not split_optimizer
        if split_optimizer:
            self._graph_opt = self._capture_plans([self._opt_calls_factory(1.0 / self.world_size)])

    # ---- the reference's non-trainable variables (model.py:58,71,307-308, mnist_model.py:24-26) ----------------------------
    KNOBS = ("use_prior", "explore_eps", "step_bias", "transform_var_bias", "output_multiplier")

    def update_config(self, **changes):
        """Change run-time switches the reference holds in non-trainable tf.Variables -- use_prior (toggle_prior), explore_eps,
        step_bias, transform_var_bias, output_multiplier -- on a built engine.  They are launch arguments of the step's kernels,
        so the plans are rebuilt and a captured graph is re-captured with the arguments it was captured with (milliseconds;
        these switches change a handful of times per run, the step itself stays free of extra loads).  Returns True if
        anything changed."""
        import dataclasses
        bad = set(changes) - set(self.KNOBS)
        if bad:
            raise ValueError("not run-time switches of the engine: %s" % sorted(bad))
        norm = lambda k, v: (bool(v) if k == "use_prior" else (None if v is None else float(v)))
        new = {k: norm(k, v) for k, v in changes.items() if norm(k, v) != norm(k, getattr(self.cfg, k))}
        if not new:
            return False
        self.synchronize()
        hook = getattr(self, "_recapture_hook", None)         # a wrapper that owns the captured graph (DataParallelEngine, ipc-rsag)
        recapture = getattr(self, "_capture_kwargs", None) if (self._graph is not None and hook is None) else None
        self.release_graphs()
        self.cfg = dataclasses.replace(self.cfg, **new)
        self._build_plans()
        if hook is not None:
            hook()
        elif recapture is not None:
            self.capture(**recapture)
        return True

    def release_graphs(self):
        L = H.lib()
        for g in (self._graph_opt, getattr(self, "_graph_b2", None), self._graph):
            if g is not None:
                L.air_graph_destroy(g)
        self._graph = self._graph_opt = self._graph_b2 = None

    def stream_context(self):
        """torch stream context of the engine's stream (collectives issued through torch.distributed run inside it)."""
        return torch.cuda.stream(self.stream)

    def train_step(self, obs=None, allreduce=None, allreduce_async=None):
        """One full update: fresh noise, forward, backward, (all-reduce), centred RMSProp x2.
        `allreduce(grads)`: optional callable run on the engine stream between backward and the update (used when the
        collective is NOT part of the captured graph).  `allreduce_async(grads) -> handle with .wait()`: with a graph captured as
        capture(split_optimizer=True, split_backward=True) the tail bucket of the gradients is handed to it as soon as it is
        final, the rest of the backward runs underneath, the head goes through `allreduce`, and the update waits for both."""
        if obs is not None:
            self.set_obs(obs)
        sp = self._sp()
        if self._graph is not None:
            _lib.check(H.lib().air_graph_launch(self._graph, sp), "air_graph_launch")
            if getattr(self, "_steps_per_replay", 1) > 1:
                self.global_step += self._steps_per_replay
                return
            if getattr(self, "_graph_b2", None) is not None:
                lo, handle = self._tail_lo, None
                with torch.cuda.stream(self.stream):
                    if allreduce_async is not None:
                        handle = allreduce_async(self.flat_grads[lo:])
                    elif allreduce is not None:
                        allreduce(self.flat_grads[lo:])
                _lib.check(H.lib().air_graph_launch(self._graph_b2, sp), "air_graph_launch")
                with torch.cuda.stream(self.stream):
                    if allreduce is not None:
                        allreduce(self.flat_grads[:lo])
                    if handle is not None:
                        handle.wait()
                _lib.check(H.lib().air_graph_launch(self._graph_opt, sp), "air_graph_launch")
            elif not self._graph_has_opt:
                if allreduce is not None:
                    with torch.cuda.stream(self.stream):
                        allreduce(self.flat_grads)
                _lib.check(H.lib().air_graph_launch(self._graph_opt, sp), "air_graph_launch")
        else:
            if allreduce is None and self.world_size == 1:
                for pl in self._single_gpu_step_plans():
                    self._run(pl, sp)
                self.global_step += 1
                return
            self._run(self._plan_fwd_train, sp)
            self._run(self._plan_bwd, sp)
            if allreduce is not None:
                with torch.cuda.stream(self.stream):
                    allreduce(self.flat_grads)
            self._run(self._plan_opt if self.world_size == 1 else self._opt_calls_factory(1.0 / self.world_size), sp)
        self.global_step += 1

    def synchronize(self):
        self.stream.synchronize()

    # ---- checkpoint / resume (the reference only ever *saves*, multi_mnist.py:116,145-146) ---------------
    def state_dict(self):
        """Everything needed to resume bit-exactly: flat parameters, the three RMSProp slot buffers, the step counter and
        the Philox state; plus the name -> (offset, shape) map so the flat buffer can be read without this class."""
        if getattr(self, "_slots_sharded", False):
            raise _lib.AirHipError("the RMSProp slots are sharded over the ranks (ipc-rsag): checkpoint through "
                                   "DataParallelEngine.state_dict(), which gathers them first")
        self.synchronize()
        return {"flat_params": self.flat_params.detach().cpu().clone(), "flat_ms": self.flat_ms.cpu().clone(),
                "flat_mg": self.flat_mg.cpu().clone(), "flat_mom": self.flat_mom.cpu().clone(),
                "global_step": int(self.step_dev.item()), "rng_state": self.rng_state.cpu().clone(),
                "learning_rate": float(self.lr_dev.item()),
                **({"ema": self.ema_dev.cpu().clone()} if getattr(self, "ema_dev", None) is not None else {}),
                "param_offsets": dict(self.param_offsets), "param_shapes": {k: tuple(v) for k, v in self.param_shapes.items()}}

    def load_state_dict(self, sd):
        if dict(sd["param_offsets"]) != dict(self.param_offsets):
            raise _lib.AirHipError("checkpoint was written for a different architecture (parameter layout differs)")
        for dst, key in ((self.flat_params, "flat_params"), (self.flat_ms, "flat_ms"), (self.flat_mg, "flat_mg"),
                         (self.flat_mom, "flat_mom"), (self.rng_state, "rng_state")):
            self._copy_in(dst, sd[key])
        self.set_learning_rate(float(sd["learning_rate"]))
        self.set_global_step(int(sd["global_step"]))
        if "ema" in sd and getattr(self, "ema_dev", None) is not None:
            self._copy_in(self.ema_dev, sd["ema"])
        self._sync_param_shadow()
        self.synchronize()

    # ---- read-outs (plumbing; not part of the timed step) ---------------------------------------------------------
    def outputs(self) -> Dict[str, torch.Tensor]:
        """The reference's model attributes (model.py:86-104, 319-343) as tensors, time-major."""
        cfg, T, B = self.cfg, self.T, self.B
        self.synchronize()
        o = {}
        o["what"], o["what_loc"], o["what_scale"] = self.what, self.what_loc, self.what_scale
        o["where"], o["where_loc"], o["where_scale"] = self.where, self.where_loc, self.where_scale
        o["presence_prob"] = self.presence_prob.view(T, B, 1)
        o["presence"] = self.presence.view(T, B, 1)
        o["glimpse_raw"] = self.gd.out[-1].view(T, B, cfg.n_crop)
        o["glimpse"] = (o["presence"] * torch.sigmoid(o["glimpse_raw"])).view(T, B, *cfg.crop_size)
        if self.canvas_steps is not None:
            o["canvas"] = self.canvas_steps.view(T, B, *cfg.img_size) * cfg.output_multiplier
        o["final_canvas"] = self.final_canvas.view(B, *cfg.img_size) * cfg.output_multiplier
        o["final_state"] = (self.h_seq[T], self.c_seq[T])
        o["rec_loss_per_sample"] = self.rec
        o["rec_loss"] = self.rec.mean()
        o["num_steps_posterior"] = self.q_n
        o["kl_num_steps_per_sample"] = self.kl_n
        o["kl_num_steps"] = self.kl_n.mean()
        wts = self._kl_weights                                  # q(n > t) (analytic) or the sampled presences (model.py:157-163)
        o["prior_step_weight"] = wts
        # (a prior left at None: its KL term is not part of the loss -- model.py:174-209 -- and reads as zero here)
        o["kl_what_per_sample"] = (self.kl_what_row * wts).sum(0) if cfg.what_prior is not None else torch.zeros_like(self.kl_n)
        o["kl_where_per_sample"] = ((self.kl_where_row * wts).sum(0) if cfg.where_scale_prior is not None and cfg.where_shift_prior is not None
                                    else torch.zeros_like(self.kl_n))
        o["kl_what"] = o["kl_what_per_sample"].mean()
        o["kl_where"] = o["kl_where_per_sample"].mean()
        pw = 1.0 if cfg.use_prior else 0.0
        o["prior_loss"] = float(cfg.nsp_weight) * o["kl_num_steps"] + o["kl_what"] + o["kl_where"]
        o["loss"] = o["rec_loss"] + pw * o["prior_loss"]
        o["num_step_per_sample"] = self.presence.sum(0)
        o["num_steps_log_prob"] = self.logp
        o["opt_loss"] = o["loss"]
        if cfg.use_reinforce:
            o["baseline"] = self.bl.out[-1]
            o["reinforce_loss"] = self.nvil_out[0]
            o["baseline_loss"] = self.nvil_out[1]
            o["imp_weight_mean"] = self.nvil_out[2]
            o["imp_weight_var"] = self.nvil_out[3]
            o["opt_loss"] = o["loss"] + o["reinforce_loss"]
        if cfg.l2_weight and cfg.l2_weight > 0.0:                  # model.py:346-353; evaluated on the CURRENT parameters (a read-out)
            sq = sum((v * v).sum() for k, v in self.params.items() if v.dim() == 2 and not k.startswith("baseline/"))
            o["l2_loss"] = float(cfg.l2_weight) * sq / 2
            o["opt_loss"] = o["opt_loss"] + o["l2_loss"]
        if getattr(self, "ema_dev", None) is not None:
            o["imp_weight_moving_mean"], o["imp_weight_moving_var"] = self.ema_dev[0], self.ema_dev[1]
        return o

    def named_grads(self) -> Dict[str, torch.Tensor]:
        self.synchronize()
        return dict(self.grads)

    def kernel_launch_count(self) -> Dict[str, int]:
        """launches per train step: of the plans a single-GPU step runs (riders / folded closing update included), of the plain
        plans under data parallelism"""
        if self.world_size == 1:
            f, b, o = self._single_gpu_step_plans()
            return {"forward": len(f), "backward": len(b), "optimizer": len(o)}
        return {"forward": len(self._plan_fwd_train), "backward": len(self._plan_bwd), "optimizer": len(self._plan_opt)}
