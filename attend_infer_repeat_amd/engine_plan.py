"""Plan construction of the fused engine (engine.AIREngine): the train step as fixed lists of C-ABI launches over the engine's
buffers -- forward, hand-derived backward, optimiser, the fusions and riders of each regime -- and the bf16 mirror bookkeeping of the
throughput regime.  Split from engine.py in round 5 (no behaviour change): engine.py keeps memory, capture / replay, the public API,
the feeder and checkpoints."""
import ctypes
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from . import hip as H
from .engine_config import EngineConfig, _mlp_shapes, _Mlp


class PlanMixin:
    """`_build_plans` and its helpers; mixed into AIREngine"""

    def _build_plans(self):
        """The step as a fixed list of C-ABI launches.  It is launch/latency bound at batch 64 (~6 us per dependent
        launch), so independent GEMMs are dispatched together (air_gemm_grouped: the dW / dX pair of a layer, the
        transform / steps heads, decoder + baseline levels) and the tiny ops are fused (step prologue / epilogue,
        presence + num-steps)."""
        L = H.lib()
        cfg, B, T, M = self.cfg, self.B, self.T, self.M
        Hd, A, P, hw = cfg.n_hidden, cfg.n_appearance, cfg.n_pix, cfg.n_crop
        (Hi, Wi), (hc, wc) = cfg.img_size, cfg.crop_size
        p = H._p
        wsp, wsb = p(self.ws), ctypes.c_size_t(self.ws.numel() * 4)
        fwd, bwd, rng = [], [], []
        if cfg.mfma_dtype not in ("f32", "bf16"):
            raise ValueError("mfma_dtype must be 'f32' or 'bf16', got %r" % (cfg.mfma_dtype,))
        prec = 1 if cfg.mfma_dtype == "bf16" else 0
        self._keep = getattr(self, "_keep", [])             # descriptor arrays of every plan ever built stay alive (graphs hold pointers)
        NONE, BIAS, BELU, MDELU, ADDAUX = H.EPI_NONE, H.EPI_BIAS, H.EPI_BIAS_ELU, H.EPI_MUL_DELU, H.EPI_ADD_AUX
        ADDAUX_ELU = H.EPI_ADD_AUX_ELU
        dp = lambda t: (t.data_ptr() if t is not None else None)

        def desc(ta, tb, Mm, Nn, Kk, Aa, lda, Bb, ldb, Cc, ldc, bias=None, epi=NONE, aux=None, ldaux=0, beta=0.0,
                 colsum=None, A2=None, a_bias=None, a_elu=0, a_out=None):
            return _lib.AirGemmDesc(int(ta), int(tb), Mm, Nn, Kk, dp(Aa), lda, dp(Bb), ldb, dp(Cc), ldc, dp(bias), epi,
                                    dp(aux), ldaux, float(beta), dp(colsum), prec, dp(A2), dp(a_bias), int(a_elu),
                                    dp(a_out))

        # Throughput regime (thousands of rows): the weight gradients (K = rows, tiny outputs) leave the dX chain and are
        # formed at the end of the backward in a few launches of their own -- nothing but the optimiser consumes them, and a
        # launch that holds ALL of them has hundreds of 64x64 tiles: enough to fill the chip with the wide-tile kernel (16-byte
        # operand loads, 8 waves split K inside the workgroup) instead of ~11 launches of 16x16 tiles with 16-way K splits.
        defer_dw = M >= int(os.environ.get("AIR_DEFER_DW_MIN_ROWS", "768"))
        deferred_dw = []
        self._defer_dw = defer_dw
        throughput = defer_dw
        # bf16 DATA path (throughput regime of mfma_dtype="bf16"): the dense products read bf16 mirrors of their operands -- a
        # shadow of the flat parameter buffer kept by the optimiser launch, mirrors of the activations / gradients that GEMM
        # epilogues produce, the observation batch converted at the start of the step -- see _apply_bf16_mirrors
        use16 = prec == 1 and throughput and os.environ.get("AIR_BF16_STORAGE", "1") == "1"
        self._use16 = use16
        if use16:
            self._alloc_bf16_mirrors()
        m16 = self._mirror_ptr if use16 else (lambda t: None)

        harvest = [False]      # True: launch() only records its problems, one entry per call (see the fused canvas launch below)

        def launch(plan, descs, allow_splitk=False):
            """one dispatch for all `descs` (a lone long-K problem may use the split-K single-GEMM entry instead)"""
            if harvest[0]:
                arr = (_lib.AirGemmDesc * len(descs))(*descs)
                self._keep.append(arr)
                plan.append((L.air_gemm_grouped, (arr, len(descs)), "air_gemm_grouped"))
                return
            if defer_dw and plan is bwd:
                deferred_dw.extend(d for d in descs if d.ta and not d.tb)
                descs = [d for d in descs if not (d.ta and not d.tb)]
                if not descs:
                    return
            tiles16 = sum(((d.M + 15) // 16) * ((d.N + 15) // 16) for d in descs)

            def wide_ok(d):       # what air_gemm_grouped's wide-tile kernels need of a problem (gemm_kernels.hip)
                strict = tiles16 > 2048 or (d.A % 16 == 0 and d.lda % 4 == 0 and d.K % 4 == 0)
                return (strict and not (d.ta and d.tb) and not d.A2 and d.B % 16 == 0 and d.ldb % 4 == 0
                        and (not (d.ta or d.tb) or d.K % 4 == 0) and d.M >= 4 and d.N >= 4 and d.K >= 4
                        and (not d.ta or d.M % 4 == 0) and (d.tb or d.N % 4 == 0))
            if throughput and len(descs) > 1 and any(wide_ok(d) for d in descs) and not all(wide_ok(d) for d in descs):
                # one odd problem (N = 1, K = 50 ...) would keep the whole group off the wide-tile kernels: it gets its own launch
                launch(plan, [d for d in descs if wide_ok(d)])
                launch(plan, [d for d in descs if not wide_ok(d)])
                return
            if (len(descs) == 1 and allow_splitk and descs[0].K >= 1024 and tiles16 > 256 and not use16
                    and not (throughput and wide_ok(descs[0]))):
                d = descs[0]
                plan.append((L.air_gemm_bf16 if prec else L.air_gemm, (d.ta, d.tb, d.M, d.N, d.K, d.A, d.lda, d.B, d.ldb, d.C, d.ldc, d.bias,
                                          d.epilogue, d.aux, d.ldaux, d.beta, d.colsum, wsp, wsb), "air_gemm"))
                return
            t16 = lambda d: ((d.M + 15) // 16) * ((d.N + 15) // 16)
            groups = [descs]
            if tiles16 > 1536 and len(descs) > 1:
                # large batch: launches are cheap relative to the work, and the library picks ONE tile shape / K-split per
                # launch -- keep the long-K few-tile problems (weight gradients: K = T*B) apart from the many-tile ones
                long_k = [d for d in descs if d.K >= 1024 and t16(d) <= 1024]
                rest = [d for d in descs if not (d.K >= 1024 and t16(d) <= 1024)]
                groups = [g for g in (long_k, rest) if g]
            for grp in groups:
                for i in range(0, len(grp), 8):
                    chunk = grp[i:i + 8]
                    arr = (_lib.AirGemmDesc * len(chunk))(*chunk)
                    self._keep.append(arr)
                    plan.append((L.air_gemm_grouped, (arr, len(chunk)), "air_gemm_grouped"))

        def fwd_desc(m: _Mlp, i, x, ldx):
            k, n = m.shapes[i]
            last = i == m.n - 1
            xin, ld_in = (m.out[i - 1], m.shapes[i - 1][1]) if i > 0 else (x, ldx)
            return desc(0, 0, m.rows, n, k, xin, ld_in, m.w[i], n, m.out[i], n, bias=m.b[i],
                        epi=BIAS if (last and m.last_linear) else BELU)

        def mlp_fwd_multi(plan, chains, splitk_first=False):
            """chains: [(mlp, x, ldx)] advanced level by level, one dispatch per level"""
            depth = max(m.n for m, _, _ in chains)
            for i in range(depth):
                descs = [fwd_desc(m, i, x, ldx) for m, x, ldx in chains if i < m.n]
                if i == 0 and splitk_first:
                    for d in descs:
                        launch(plan, [d], allow_splitk=True)
                else:
                    launch(plan, descs)

        def mlp_bwd_multi(plan, chains, extra_first=(), extra_last=(), first_dx_done=False):
            """chains: dicts(m, x, ldx, g_last, dx_out=None, dx_aux=None).  g_last = gradient wrt the last layer's
            pre-activation.  Per level one dispatch holding every chain's dW (+db) and dX.
            first_dx_done: the dX of every chain's LAST layer already exists (air_attend_bwd_dx); that level then only has
            weight gradients, which ride in the next level's dispatch."""
            depth = max(c["m"].n for c in chains)
            gcur = {id(c["m"]): c["g_last"] for c in chains}
            carry = []
            for s_ in range(depth):
                descs = carry
                carry = []
                for c in chains:
                    m = c["m"]
                    i = m.n - 1 - s_
                    if i < 0:
                        continue
                    k, n = m.shapes[i]
                    g = gcur[id(m)]
                    if i == 0 and "x_parts" in c:           # first layer fed by a concat: one dW problem per part
                        for j, (xp, ldp, k0, kn) in enumerate(c["x_parts"]):
                            descs.append(desc(1, 0, kn, n, m.rows, xp, ldp, g, n, m.dw[0][k0:k0 + kn], n,
                                              colsum=m.db[0] if j == 0 else None))
                        continue
                    xin, ld_in = (m.out[i - 1], m.shapes[i - 1][1]) if i > 0 else (c["x"], c["ldx"])
                    descs.append(desc(1, 0, k, n, m.rows, xin, ld_in, g, n, m.dw[i], n, colsum=m.db[i]))
                    skip_dx = first_dx_done and i == m.n - 1
                    if i > 0:
                        if not skip_dx:
                            descs.append(desc(0, 1, m.rows, k, n, g, n, m.w[i], n, m.g[i - 1], k, epi=MDELU,
                                              aux=m.out[i - 1], ldaux=k))
                        gcur[id(m)] = m.g[i - 1]
                    elif c.get("dx_out") is not None and not skip_dx:
                        aux = c.get("dx_aux")
                        descs.append(desc(0, 1, m.rows, k, n, g, n, m.w[0], n, c["dx_out"], k,
                                          epi=MDELU if aux is not None else NONE, aux=aux, ldaux=k if aux is not None else 0))
                if s_ == 0:
                    descs = list(extra_first) + descs
                if s_ == depth - 1:
                    descs = descs + list(extra_last)
                if first_dx_done and s_ == 0 and depth > 1 and not extra_first:
                    carry = descs                  # only weight gradients at this level: dispatch them with the next one
                    continue
                launch(plan, descs)

        # ---- noise only (used when forward() is asked to keep injected noise: the prologue then draws nothing) -------
        n_norm, n_uni = self.noise_normal.numel(), self.u_pres.numel()
        self._rng_inc = (n_norm + 3) // 4 + (n_uni + 3) // 4
        rng.append((L.air_rng_fill, (p(self.noise_normal), ctypes.c_size_t(n_norm), p(self.u_pres),
                                     ctypes.c_size_t(n_uni), p(self.rng_state)), "air_rng_fill"))

        # ---- forward ------------------------------------------------------------------------------------------------
        anneal = {None: 0, "exp": 1, "linear": 2}[cfg.nsp_anneal]

        def prologue(with_noise):
            return (L.air_step_prologue,
                    (p(self.noise_normal), ctypes.c_size_t(n_norm if with_noise else 0), p(self.u_pres),
                     ctypes.c_size_t(n_uni if with_noise else 0), p(self.rng_state), p(self.step_dev), anneal,
                     float(cfg.nsp_init), float(cfg.nsp_final), float(cfg.nsp_steps), float(cfg.nsp_hold_init),
                     float(cfg.nsp_steps_div), p(self.prior_dev), T, p(self.params["lstm/h0"]),
                     p(self.params["lstm/c0"]), p(self.h_seq[0]), p(self.c_seq[0]), B, Hd), "air_step_prologue")

        # cell.py:125 (hoisted out of the time loop) + the obs columns of the baseline's first layer (modules.py:131-143):
        # both contract over the P pixels of obs
        # The two products over the P pixels of obs have K = P (2500 / 10000) on B/16 x 16 tiles each: 128 workgroups, each
        # pulling its whole K range through ONE CU -- ingest-bound at half of the chip.  In the latency regime K is split in two
        # (4 problems, 256 workgroups, same launch) and the CONSUMERS add the halves where they read them: the next layer's
        # A-operand prologue forms elu(slab0 + slab1 + bias) (and stores it for the backward); the baseline's second half is
        # written straight into the buffer its second stage accumulates into (beta = 1) next to aux = first half.  No
        # cross-workgroup hand-off, no extra launch, fixed summation order.
        E0 = self.enc.shapes[0][1]
        lvl0_tiles = ((B + 15) // 16) * ((E0 + 15) // 16)
        # (the consumer -- the product over the E0 columns of the first hidden layer -- runs on the A-prologue kernel, which the
        #  library only has on 16x16 tiles: its launch must stay below the 1536-tile switch to 32x32 tiles)
        n_after = self.enc.shapes[1][1] if self.enc.n > 1 else 4 * Hd
        split0 = (lvl0_tiles * (2 if cfg.use_reinforce else 1) <= 128 and P >= 2048 and P % 4 == 0 and E0 % 16 == 0
                  and ((B + 15) // 16) * ((n_after + 15) // 16) <= 1536 and os.environ.get("AIR_SPLIT_K0", "1") == "1"
                  # (not on the bf16 data path -- many steps at a small batch, e.g. B = 64, T = 12: the consumer's A-prologue has no
                  #  bf16-mirror form and the first layer's activation would be a mirrored buffer no epilogue writes)
                  and not use16)
        self._split0 = split0
        if split0:
            kh = (P // 2) // 16 * 16                       # both halves start 16-byte aligned (and on a chunk boundary)
            s0, s1 = self._buf("enc_slab0", (B, E0)), self._buf("enc_slab1", (B, E0))
            lvl0 = [desc(0, 0, B, E0, kh, self.obs, P, self.enc.w[0], E0, s0, E0),
                    desc(0, 0, B, E0, P - kh, self.obs[:, kh:], P, self.enc.w[0][kh:], E0, s1, E0)]
            if cfg.use_reinforce:
                n0 = self.bl.shapes[0][1]
                lvl0 += [desc(0, 0, B, n0, kh, self.obs, P, self.bl.w[0][:kh], n0, self.bl_obs, n0, bias=self.bl.b[0], epi=BIAS),
                         desc(0, 0, B, n0, P - kh, self.obs[:, kh:], P, self.bl.w[0][kh:P], n0, self.bl.out[0], n0)]
            launch(fwd, lvl0)
            enc_pro = dict(A=s0, A2=s1, a_bias=self.enc.b[0], a_elu=1, a_out=self.enc.out[0])
        else:
            lvl0 = [fwd_desc(self.enc, 0, self.obs, P)]
            if cfg.use_reinforce:
                n0 = self.bl.shapes[0][1]
                lvl0.append(desc(0, 0, B, n0, P, self.obs, P, self.bl.w[0][:P], n0, self.bl_obs, n0, bias=self.bl.b[0],
                                 epi=BIAS))
            if ((B + 15) // 16) * ((max(d.N for d in lvl0) + 15) // 16) <= 256 or throughput:
                launch(fwd, lvl0)       # few tiles: one launch, 16 waves per tile share the long K; throughput regime: one
                                        # wide-tile launch for both products over obs (-2 % of the batch-1024 fp32 step)
            else:
                for d in lvl0:
                    launch(fwd, [d], allow_splitk=True)
            enc_pro = None

        def after_enc0(k, n, Bmat, ldb, Cc, bias, epi):
            """the product that consumes the first encoder layer's activation [B, E0]"""
            if enc_pro is None:
                return desc(0, 0, B, n, k, self.enc.out[0], E0, Bmat, ldb, Cc, n, bias=bias, epi=epi)
            return desc(0, 0, B, n, k, enc_pro["A"], E0, Bmat, ldb, Cc, n, bias=bias, epi=epi, A2=enc_pro["A2"],
                        a_bias=enc_pro["a_bias"], a_elu=enc_pro["a_elu"], a_out=enc_pro["a_out"])

        if self.enc.n > 1:
            k1, n1 = self.enc.shapes[1]
            launch(fwd, [after_enc0(k1, n1, self.enc.w[1], n1, self.enc.out[1], self.enc.b[1], BELU)])
        for i in range(2, self.enc.n):
            launch(fwd, [fwd_desc(self.enc, i, None, 0)])
        enc_out, E = self.enc.out[-1], self.enc.shapes[-1][1]
        wg, bg = self.params["lstm/w_gates"], self.params["lstm/b_gates"]
        w_x, w_h = wg[:E], wg[E:]
        # Round 5: in the latency regime the hoisted product gx = enc_out . W_x + b has no launch of its own -- the first LSTM step,
        # whose recurrent operand is the one-row initial state, accumulates it next to h0 . W_h and writes gx for the later steps
        # (air_lstm_first_step_fwd: same sums, same order, one dependent launch fewer).  Not when the first encoder layer is the
        # only one (its K-split halves are reduced by the gx product's A-prologue) and not beyond the fused-step tile count.
        # (latency regime only: in the throughput regime the gx product keeps its wide-tile launch)
        fold_gx = (self.enc.n > 1 and not throughput
                   and ((B + 15) // 16) * ((Hd + 15) // 16) <= int(os.environ.get("AIR_FUSE_LSTM_TILES", "512"))
                   and os.environ.get("AIR_FOLD_GX", "1") == "1")
        self._fold_gx = fold_gx
        if fold_gx:
            pass
        elif self.enc.n == 1:
            launch(fwd, [after_enc0(E, 4 * Hd, w_x, 4 * Hd, self.gx, bg, BIAS)])
        else:
            launch(fwd, [desc(0, 0, B, 4 * Hd, E, enc_out, E, w_x, 4 * Hd, self.gx, 4 * Hd, bias=bg, epi=BIAS)])
        # Recurrent product + gate math in ONE launch per step while the chain is latency bound (it is the only truly
        # sequential part of the step); at large batch the 32x32-tile GEMM + a pointwise pass re-reads less (measured:
        # B=1024 0.938 vs 0.954 ms/step), so the pair is kept there.
        fuse_lstm = ((B + 15) // 16) * ((Hd + 15) // 16) <= int(os.environ.get("AIR_FUSE_LSTM_TILES", "512"))
        # forward: beyond 512 tiles the library's wide-tile form of the fused step (16 rows x 64 units x 4 gates per workgroup)
        fuse_lstm_fwd = fuse_lstm or (Hd % 64 == 0 and E % 4 == 0 and os.environ.get("AIR_FUSE_LSTM_WIDE", "1") == "1")
        if not fuse_lstm_fwd:
            self.gates = self._buf("gates", (T, B, 4 * Hd))
        # (the step prologue rides in the first fused step only in the latency regime: the wide-tile form holds 133 KB of LDS
        #  per workgroup, so riding prologue workgroups would wait for a free CU -- 14.4 us against 8.5 + 4.9 for two launches)
        prologue_rides = fuse_lstm
        for t in range(T):                                                                  # cell.py:126-127
            if prologue_rides and t == 0:
                fwd.append(None)        # placeholder: the first step carries the step prologue (filled in per plan below)
                lstm0_index = len(fwd) - 1
                continue
            if fuse_lstm_fwd and use16 and not fuse_lstm and self._lstm16_ok():
                # bf16 data path: W_h from the shadow, h_t from its mirror (h_0 = the tiled initial state has none)
                fwd.append((L.air_lstm_step_fwd_bf16, (p(self.h_seq[t]), m16(self.h_seq[t]), p(self.c_seq[t]), m16(w_h), 4 * Hd,
                                                       p(self.gx), 4 * Hd, p(self.h_seq[t + 1]), m16(self.h_seq[t + 1]),
                                                       p(self.c_seq[t + 1]), p(self.gate_act[t]), B, Hd, 1.0),
                            "air_lstm_step_fwd_bf16"))
                continue
            if fuse_lstm_fwd:
                fwd.append((L.air_lstm_step_fwd, (p(self.h_seq[t]), p(self.c_seq[t]), p(w_h), 4 * Hd, p(self.gx), 4 * Hd,
                                                  p(self.h_seq[t + 1]), p(self.c_seq[t + 1]), p(self.gate_act[t]), B, Hd,
                                                  1.0, prec), "air_lstm_step_fwd"))
                continue
            launch(fwd, [desc(0, 0, B, 4 * Hd, Hd, self.h_seq[t], Hd, w_h, 4 * Hd, self.gates[t], 4 * Hd, epi=ADDAUX,
                              aux=self.gx, ldaux=4 * Hd)])
            fwd.append((L.air_lstm_pointwise_fwd, (p(self.gates[t]), p(self.c_seq[t]), p(self.h_seq[t + 1]),
                                                   p(self.c_seq[t + 1]), p(self.gate_act[t]), B, Hd, 1.0),
                        "air_lstm_pointwise_fwd"))
        h_all = self.h_seq[1:]                                                              # [T,B,Hd] contiguous
        # priors left at None (model.py:174-209: that KL term is simply not added): the kernels still evaluate the rows against a
        # standard normal, but they enter neither the loss nor any gradient (weights 0 / pointers NULL below)
        has_what = cfg.what_prior is not None
        has_where = cfg.where_scale_prior is not None and cfg.where_shift_prior is not None
        analytic = bool(cfg.nsp_analytic)
        # continuous steps (cell.py:150-151): the entry points that draw the presence take no uniform variates and write the probability
        # itself; the canvas write's backward returns d/d presence, which joins d/d presence_prob in the steps-logit backward
        discrete = bool(cfg.discrete_steps)
        u_p = p(self.u_pres) if discrete else None
        self.dpres = None if discrete else self._buf("dpres", (T, B))
        dpres_p = None if discrete else p(self.dpres)
        self._kl_weights = self.step_w if analytic else self.presence       # model.py:157-163
        sp, shp = (cfg.where_scale_prior, cfg.where_shift_prior) if has_where else ((0.0, 1.0), (0.0, 1.0))
        if shp[0] is None:                  # a shift prior without `loc` is centred on the posterior's own mean (model.py:203-207):
            shp = (float("nan"), shp[1])    # the kernels' NaN convention (include/air_hip.h, air_gauss_sample_fwd)
        eps = -1.0 if cfg.explore_eps is None else float(cfg.explore_eps)
        # "attend" fusion: output layers of the transform / steps MLPs + where sampling + presence / num-steps + the glimpse
        # read in ONE launch (three dependent launches otherwise).  Needs a 16-byte addressable image that fits the
        # register-prefetch staging, and (backward) one workgroup per glimpse.
        fuse_attend = (P % 4 == 0) and (P // 4 <= 3 * 1024) and T <= 28 and M <= int(os.environ.get("AIR_FUSE_ATTEND_M", str(1 << 30)))
        if fuse_attend:
            for i in range(max(self.tr.n, self.st.n) - 1):
                launch(fwd, [fwd_desc(m, i, h_all, Hd) for m in (self.tr, self.st) if i < m.n - 1])
            tr_in, tr_k = (self.tr.out[-2], self.tr.shapes[-1][0]) if self.tr.n > 1 else (h_all, Hd)
            st_in, st_k = (self.st.out[-2], self.st.shapes[-1][0]) if self.st.n > 1 else (h_all, Hd)
            fwd.append((L.air_attend_fwd, (p(tr_in), p(self.tr.w[-1]), p(self.tr.b[-1]), tr_k, p(st_in),
                                           p(self.st.w[-1]), p(self.st.b[-1]), st_k, p(self.tr.out[-1]),
                                           p(self.st.out[-1]), p(self.eps_where), cfg.transform_var_bias, sp[0], sp[1],
                                           shp[0], shp[1], p(self.where_loc), p(self.where_scale), p(self.where),
                                           p(self.kl_where_row), u_p, cfg.step_bias, eps, p(self.prior_dev),
                                           p(self.presence_prob), p(self.presence), p(self.q_n), p(self.kl_n),
                                           p(self.logp), p(self.step_w), p(self.obs), p(self.glimpse_in), T, B, Hi, Wi,
                                           hc, wc, prec, float(cfg.guard_eps)), "air_attend_fwd"))                      # cell.py:129-151
        else:
            mlp_fwd_multi(fwd, [(self.tr, h_all, Hd), (self.st, h_all, Hd)])                # cell.py:129,138
            fwd.append((L.air_heads_fwd, (p(self.tr.out[-1]), 8, p(self.eps_where), cfg.transform_var_bias, 1,
                                          sp[0], sp[1], shp[0], shp[1], p(self.where_loc), p(self.where_scale),
                                          p(self.where), p(self.kl_where_row), M, 4,             # cell.py:129-133
                                          p(self.st.out[-1]), u_p, cfg.step_bias, eps, p(self.prior_dev),
                                          p(self.presence_prob), p(self.presence), p(self.q_n), p(self.kl_n), p(self.logp),
                                          p(self.step_w), T, B, float(cfg.guard_eps)), "air_heads_fwd"))               # cell.py:137-151, prior.py
            fwd.append((L.air_st_read_fwd, (p(self.obs), p(self.where), p(self.glimpse_in), M, B, Hi, Wi, hc, wc),
                        "air_st_read_fwd"))                                                 # cell.py:135
        mlp_fwd_multi(fwd, [(self.ge, self.glimpse_in, hw)])                                # cell.py:153
        ge_out, G = self.ge.out[-1], self.ge.shapes[-1][1]
        wp = cfg.what_prior if has_what else (0.0, 1.0)
        # Round 5, latency regime with REINFORCE: the whole `what` head -- the product q = ge_out . W + b, the sampling with its KL
        # terms and the latent columns of the baseline input -- is ONE launch (air_what_head_fwd: a tile holds both halves of its
        # (row, latent dim) pairs) instead of a GEMM launch + air_what_sample_pack.  The KL row of a sample spans several tiles: the
        # tiles leave shares, the backward launch of the same head (air_gauss_sample_bwd*) adds them; a forward() on its own -- an
        # evaluation pass -- adds them with a small launch of its own, which the train step drops.
        what_head = (cfg.use_reinforce and analytic and discrete and not throughput and os.environ.get("AIR_FUSE_WHAT_HEAD", "1") == "1")
        self._what_head = what_head
        self._kl_parts_args = (None, 0, None)
        if what_head:
            n_kl = int(L.air_what_head_parts(A))
            self.kl_what_parts = self._buf("kl_what_parts", (n_kl, M))
            self._kl_parts_args = (p(self.kl_what_parts), n_kl, p(self.kl_what_row))
            fwd.append((L.air_what_head_fwd, (p(ge_out), G, G, p(self.params["what/w"]), p(self.params["what/b"]), p(self.eps_what),
                                              cfg.what_scale_offset, wp[0], wp[1], p(self.q), p(self.what_loc), p(self.what_scale),
                                              p(self.what), p(self.kl_what_parts), A, p(self.where), p(self.presence),
                                              p(self.h_seq[T]), p(self.c_seq[T]), p(self.base_lat), T, B, Hd, Hd,
                                              float(cfg.guard_eps), prec), "air_what_head_fwd"))        # modules.py:20-21, cell.py:154-156
            fwd.append((L.air_sum_leading, (p(self.kl_what_parts), p(self.kl_what_row), n_kl, ctypes.c_size_t(M)),
                        "air_sum_leading:kl_what"))
        else:
            launch(fwd, [desc(0, 0, M, 2 * A, G, ge_out, G, self.params["what/w"], 2 * A, self.q, 2 * A,
                              bias=self.params["what/b"], epi=BIAS)])                       # modules.py:20-21
        if cfg.use_reinforce:                                                               # model.py:218-259
            # sample `what` + assemble the latent columns of the baseline input in one launch
            KL = cfg.baseline_in - P
            if not what_head:
                fwd.append((L.air_what_sample_pack, (p(self.q), 2 * A, p(self.eps_what), cfg.what_scale_offset, wp[0], wp[1],
                                                     p(self.what_loc), p(self.what_scale), p(self.what), p(self.kl_what_row),
                                                     A, p(self.where), p(self.presence), p(self.h_seq[T]), p(self.c_seq[T]),
                                                     p(self.base_lat), T, B, Hd, Hd, float(cfg.guard_eps)), "air_what_sample_pack"))
            n0 = self.bl.shapes[0][1]
            launch(fwd, [desc(0, 0, B, n0, KL, self.base_lat, KL, self.bl.w[0][P:], n0, self.bl.out[0], n0,
                              epi=ADDAUX_ELU if self.bl.n > 1 or not self.bl.last_linear else ADDAUX,
                              aux=self.bl_obs, ldaux=n0, beta=1.0 if split0 else 0.0),
                         fwd_desc(self.gd, 0, self.what, A)])                               # cell.py:158
            depth = max(self.gd.n, self.bl.n)
            for i in range(1, depth):
                launch(fwd, [fwd_desc(m, i, None, 0) for m in (self.gd, self.bl) if i < m.n])
        else:
            fwd.append((L.air_gauss_sample_fwd, (p(self.q), 2 * A, p(self.eps_what), cfg.what_scale_offset, 0, wp[0],
                                                 wp[1], wp[0], wp[1], p(self.what_loc), p(self.what_scale), p(self.what),
                                                 p(self.kl_what_row), M, A, float(cfg.guard_eps)), "air_gauss_sample_fwd"))
            mlp_fwd_multi(fwd, [(self.gd, self.what, A)])
        decoded = self.gd.out[-1]
        NB = self.n_bands
        fwd.append((L.air_canvas_unroll_fwd_banded, (p(decoded), p(self.where), p(self.presence), p(self.obs),
                                                     p(self.canvas_steps), p(self.final_canvas), p(self.rec_parts), NB,
                                                     T, B, Hi, Wi, hc, wc, cfg.output_multiplier, cfg.output_std),
                    "air_canvas_unroll_fwd_banded"))                                        # cell.py:159-165, model.py:319-324
        # NVIL (model.py:218-259): forward() alone finishes with it so that outputs() is complete; a train step evaluates it
        # as one extra workgroup of the canvas backward launch instead (independent work, one launch fewer).  Either way it
        # is the consumer that adds the per-band shares of rec_loss_per_sample (and stores the sum in self.rec).
        nvil_args = (p(self.rec_parts), NB, p(self.rec), p(self.bl.out[-1]), p(self.logp), p(self.nvil_out),
                     p(self.dlogp), p(self.dbase))
        self._nvil_args = nvil_args
        # decay_rate: the two moving averages + the rate + the update switch as one device block (air_nvil's `ema_dev`); evaluation
        # passes (forward() alone) read the averages without moving them
        ema_p = None
        if cfg.decay_rate is not None and cfg.use_reinforce:
            if getattr(self, "ema_dev", None) is None:
                self.ema_dev = torch.tensor([0.0, 1.0, float(cfg.decay_rate), 1.0], dtype=torch.float32, device=self.device)
            self._fill_in(self.ema_dev[2:3], float(cfg.decay_rate))
            ema_p = p(self.ema_dev)
        rec_sum = (L.air_sum_leading, (p(self.rec_parts), p(self.rec), NB, ctypes.c_size_t(B)), "air_sum_leading")
        fwd_tail = [(L.air_nvil_parts, nvil_args + (B, ema_p), "air_nvil_parts")] if cfg.use_reinforce else [rec_sum]
        nvil_direct = None
        if cfg.use_reinforce and not analytic:
            # a non-analytic num-steps prior (model.py:157-163, 339-340): the step weights are the sampled presences and the prior's
            # per-sample value joins the importance weight -- formed by a small launch of its own, NVIL as the plain launch behind it
            self.imp = self._buf("imp", (B,))
            nvil_direct = [(L.air_imp_weight, (p(self.rec_parts), NB, p(self.rec), p(self.kl_n), float(cfg.nsp_weight),
                                               p(self.kl_what_row) if has_what else None, p(self.kl_where_row) if has_where else None,
                                               p(self.presence), T, B, p(self.imp), None, 0.0), "air_imp_weight"),
                           (L.air_nvil, (p(self.imp), p(self.bl.out[-1]), p(self.logp), p(self.nvil_out), p(self.dlogp), p(self.dbase),
                                         B, ema_p), "air_nvil")]
            fwd_tail = list(nvil_direct)
        if ema_p is not None:               # (forward() is an evaluation pass: the update switch is off around its NVIL launch)
            sw = ctypes.c_void_p(self.ema_dev.data_ptr() + 12)
            fwd_tail = [(L.air_fill, (sw, ctypes.c_size_t(1), 0.0), "air_fill")] + fwd_tail + [(L.air_fill, (sw, ctypes.c_size_t(1), 1.0), "air_fill")]

        # ---- backward of opt_loss = mean(rec) + pw*(mean kl_n + mean sum_t w*(kl_what+kl_where)) + reinforce -------
        pw = 1.0 if cfg.use_prior else 0.0
        inv_b = 1.0 / B
        # Latency regime with REINFORCE: the canvas forward and the (recompute-form) backward are ONE launch -- the backward
        # re-forms the canvas on each glimpse's footprint, so it reads nothing the forward writes (air_canvas_unroll_fwd_bwd) -- the
        # train step's forward list then ends before the canvas; NVIL, which needs the forward's reconstruction shares, rides on
        # the next pointwise launch (air_gauss_sample_bwd_nvil) and the baseline's backward, which needs NVIL, rides with the
        # three launches after that (what / glimpse-encoder backward) instead of the decoder's.  35 -> 34 dependent launches.
        # The two-role launch of the latency regime measured slower in the throughput regime (VALU bound chip-wide plus the
        # recomputation: 0.594 against 0.580 ms at batch 1024).
        # n_split: workgroups per backward unit of that launch (disjoint dglimpse rows, dwhere as n_split slabs that air_attend_bwd_dx
        # adds).  Measured (profiles/r04_canvas_split_ab.txt): with 2 the launch is FASTER only while it leaves CUs idle (batch 8:
        # 9.0 against 9.5 us); at batch 64 (192 units + 256 forward workgroups on 256 CUs) every workgroup repeats the unit's
        # staging and tables and the launch is slower (15.0 against 11.3 us; the step 0.2093 against 0.2070 ms): default 1 above
        # 128 glimpses, AIR_CANVAS_SPLIT overrides.
        n_split = int(os.environ.get("AIR_CANVAS_SPLIT", "2" if M <= 128 else "1")) if (fuse_attend and M * 2 + B * NB <= 1024) else 1
        n_split = max(1, min(4, n_split))
        self._canvas_split = n_split
        fuse_canvas = (cfg.use_reinforce and analytic and discrete and (not throughput or os.environ.get("AIR_FUSE_CANVAS_THROUGHPUT", "0") == "1")
                       # (the fused launch's backward re-forms the canvas from ALL T glimpses on each unit's footprint -- T^2 taps:
                       #  measured 0.2095 against 0.2111 ms per step at T = 3 (50x50 / 20x20), 0.3231 against 0.3194 ms at T = 5
                       #  (100x100 / 28x28; tools/runs/r04_x.sh): by default only up to T = 3; "1" / "0" force it on / off)
                       and (os.environ.get("AIR_FUSE_CANVAS", "auto") == "1"
                            or (os.environ.get("AIR_FUSE_CANVAS", "auto") == "auto" and T <= 3))
                       # (what the library's launch takes: both grids at most 4096 workgroups, the LDS of both roles)
                       and L.air_canvas_unroll_fwd_bwd_fits(NB, n_split, T, B, Hi, Wi, hc, wc) == 1)
        bl_chain = dict(m=self.bl, g_last=self.dbase,
                        x_parts=[(self.obs, P, 0, P), (self.base_lat, cfg.baseline_in - P, P, cfg.baseline_in - P)])
        bl_levels = [[], [], []]
        if fuse_canvas:
            tmp = []
            harvest[0] = True
            mlp_bwd_multi(tmp, [bl_chain])                 # the baseline's backward, level by level, to ride later launches
            harvest[0] = False
            lv = [list(e[1][0]) for e in tmp]
            # one level per launch, in order: the `what` backward, the glimpse encoder's first level, its last level
            slots = 1 + min(self.ge.n, 2)
            if len(lv) > slots or any(e[2] != "air_gemm_grouped" for e in tmp):
                fuse_canvas = False                          # (a deeper baseline than there are launches to ride: the plain plan)
            else:
                lv = lv + [[]] * (3 - len(lv))
                bl_levels = [lv[0], lv[1], lv[2]] if self.ge.n >= 2 else [lv[0], lv[1], []]
        self._fuse_canvas = fuse_canvas
        if not fuse_canvas:
            n_split = self._canvas_split = 1
        cu_args = (p(decoded), p(self.where), p(self.presence), p(self.obs), p(self.final_canvas), p(self.gd.g[-1]),
                   p(self.dwhere_w), T, B, Hi, Wi, hc, wc, cfg.output_multiplier, cfg.output_std, inv_b)
        if fuse_canvas:
            bwd.append((L.air_canvas_unroll_fwd_bwd, (p(decoded), p(self.where), p(self.presence), p(self.obs),
                                                      p(self.canvas_steps), p(self.final_canvas), p(self.rec_parts), NB,
                                                      p(self.gd.g[-1]), p(self.dwhere_w), n_split, T, B, Hi, Wi, hc, wc,
                                                      cfg.output_multiplier, cfg.output_std, inv_b),
                        "air_canvas_unroll_fwd_bwd"))
        elif not discrete:
            # continuous steps: the canvas backward also returns d/d presence; a non-analytic prior weighs the KL rows with the presence
            # ITSELF (model.py:162-163), whose gradient -- the weighted rows -- is added behind it by the importance-weight launch
            bwd.append((L.air_canvas_unroll_bwd_dpresence, cu_args[:7] + (dpres_p,) + cu_args[7:], "air_canvas_unroll_bwd_dpresence"))
            if not analytic:
                bwd.append((L.air_imp_weight, (p(self.rec_parts), NB, p(self.rec), p(self.kl_n), float(cfg.nsp_weight),
                                               p(self.kl_what_row) if has_what else None, p(self.kl_where_row) if has_where else None,
                                               p(self.presence), T, B, p(self.imp) if cfg.use_reinforce else None, dpres_p,
                                               (1.0 if cfg.use_prior else 0.0) * inv_b), "air_imp_weight"))
                if cfg.use_reinforce:
                    bwd.append(nvil_direct[1])
            elif cfg.use_reinforce:
                bwd.append((L.air_nvil_parts, nvil_args + (B, ema_p), "air_nvil_parts"))
            else:
                bwd.append(rec_sum)
        elif cfg.use_reinforce and nvil_direct is not None:
            bwd.extend(nvil_direct)
            bwd.append((L.air_canvas_unroll_bwd, cu_args, "air_canvas_unroll_bwd"))
        elif cfg.use_reinforce:
            bwd.append((L.air_canvas_unroll_bwd_nvil, cu_args + nvil_args + (ema_p,), "air_canvas_unroll_bwd_nvil"))
        else:
            bwd.append(rec_sum)          # nobody consumes rec in the step itself; keeps outputs() complete after train_step
            bwd.append((L.air_canvas_unroll_bwd, cu_args, "air_canvas_unroll_bwd"))
        chains = [dict(m=self.gd, x=self.what, ldx=A, g_last=self.gd.g[-1], dx_out=self.d_what)]
        if cfg.use_reinforce and not fuse_canvas:                                           # model.py:253-259, 362-367
            chains.append(bl_chain)
        mlp_bwd_multi(bwd, chains)
        marks = [] if fuse_canvas else [(len(bwd), "glimpse_decoder/0/w")]   # gradients of [glimpse_decoder .. baseline] are final here
        gb_args = (p(self.q), 2 * A, p(self.eps_what), cfg.what_scale_offset, 0, wp[0], wp[1], wp[0], wp[1], p(self.what_loc),
                   p(self.what_scale), p(self.d_what), None, p(self._kl_weights), pw * inv_b if has_what else 0.0, p(self.dq), 2 * A, M, A)
        # Round 5, latency regime: the backward of the `what` head needs no launch of its own -- the decoder's first-layer dX IS its
        # sample gradient, so the thread that finishes d_what[m, a] writes dq[m, a] and dq[m, A + a] in the same epilogue
        # (air_gemm_grouped_gauss_bwd), and NVIL / the sum of the head's KL shares ride behind the tiles of that launch.
        self._fold_gauss_bwd = False
        last = bwd[-1]
        if (not throughput and analytic and discrete and os.environ.get("AIR_FUSE_GAUSS_BWD", "1") == "1" and last[2] == "air_gemm_grouped"
                and sum(((d.M + 15) // 16) * ((d.N + 15) // 16) for d in last[1][0]) <= 1000):
            arr, n_d = last[1]
            which = [i for i in range(n_d) if arr[i].C == self.d_what.data_ptr() and not arr[i].ta and arr[i].N == A
                     and arr[i].epilogue == NONE and arr[i].beta == 0.0 and not arr[i].colsum]
            if len(which) == 1:
                epi = _lib.AirGaussBwdEpi(which[0], dp(self.q), 2 * A, dp(self.eps_what), cfg.what_scale_offset, wp[0], wp[1],
                                          dp(self.what_loc), dp(self.what_scale), dp(self._kl_weights), pw * inv_b if has_what else 0.0, dp(self.dq), 2 * A, A,
                                          float(cfg.guard_eps))
                self._keep.append(epi)
                nv = (nvil_args + (B, ema_p)) if fuse_canvas else (None, 0, None, None, None, None, None, None, 0, None)
                bwd[-1] = (L.air_gemm_grouped_gauss_bwd, (arr, n_d, ctypes.byref(epi)) + nv + self._kl_parts_args + (M,),
                           "air_gemm_grouped_gauss_bwd")
                self._fold_gauss_bwd = True
        if self._fold_gauss_bwd:
            pass
        elif fuse_canvas:
            bwd.append((L.air_gauss_sample_bwd_nvil, gb_args + nvil_args + (B, float(cfg.guard_eps), ema_p) + self._kl_parts_args, "air_gauss_sample_bwd_nvil"))
        else:
            bwd.append((L.air_gauss_sample_bwd, gb_args + (float(cfg.guard_eps),) + self._kl_parts_args, "air_gauss_sample_bwd"))
        launch(bwd, [desc(1, 0, G, 2 * A, M, ge_out, G, self.dq, 2 * A, self.grads["what/w"], 2 * A,
                          colsum=self.grads["what/b"]),
                     desc(0, 1, M, G, 2 * A, self.dq, 2 * A, self.params["what/w"], 2 * A, self.ge.g[-1], G, epi=MDELU,
                          aux=ge_out, ldaux=G)] + bl_levels[0])
        mlp_bwd_multi(bwd, [dict(m=self.ge, x=self.glimpse_in, ldx=hw, g_last=self.ge.g[-1], dx_out=self.d_glimpse_in)],
                      extra_first=bl_levels[1], extra_last=bl_levels[2])
        marks.append((len(bwd), "glimpse_encoder/0/w"))        # + [glimpse_encoder, what] (+ decoder, baseline when fused)
        dlogp_p = p(self.dlogp) if cfg.use_reinforce else None
        if fuse_attend:
            # ... including the dX of the two MLP output layers (K = 8 and 1): their launch disappears from the chain, their
            # dW / bias gradients join the next level's launch
            def last_dx(m, dx_top):
                if m.n > 1:
                    return m.out[-2], m.g[-2], m.shapes[-1][0], m.shapes[-1][0]
                return None, dx_top, Hd, Hd
            tr_y, tr_dx, tr_kk, tr_ld = last_dx(self.tr, self.dH)
            st_y, st_dx, st_kk, st_ld = last_dx(self.st, self.dH_b)
            bwd.append((L.air_attend_bwd_dx, (p(self.obs), p(self.where), p(self.d_glimpse_in), p(self.dwhere_r),
                                              p(self.tr.out[-1]), p(self.eps_where), cfg.transform_var_bias, sp[0], sp[1],
                                              shp[0], shp[1], p(self.where_loc), p(self.where_scale), p(self.dwhere_w), n_split,
                                              p(self._kl_weights), pw * inv_b if has_where else 0.0, p(self.tr.g[-1]),
                                              p(self.presence_prob), p(self.presence), p(self.prior_dev), pw * inv_b * float(cfg.nsp_weight),
                                              p(self.kl_what_row) if has_what else None, p(self.kl_where_row) if has_where else None,
                                              pw * inv_b if analytic else 0.0, dlogp_p, dpres_p,
                                              p(self.st.out[-1]), cfg.step_bias, eps, p(self.st.g[-1]), T, B, Hi, Wi, hc, wc,
                                              p(self.tr.w[-1]), p(tr_y) if tr_y is not None else None, p(tr_dx), tr_kk, tr_ld,
                                              p(self.st.w[-1]), p(st_y) if st_y is not None else None, p(st_dx), st_kk, st_ld,
                                              prec, float(cfg.guard_eps)), "air_attend_bwd_dx"))
        else:
            bwd.append((L.air_st_read_bwd, (p(self.obs), p(self.where), p(self.d_glimpse_in), p(self.dwhere_r), None, M,
                                            B, Hi, Wi, hc, wc), "air_st_read_bwd"))
            bwd.append((L.air_heads_bwd, (p(self.tr.out[-1]), 8, p(self.eps_where), cfg.transform_var_bias, 1,
                                          sp[0], sp[1], shp[0], shp[1], p(self.where_loc), p(self.where_scale),
                                          p(self.dwhere_w), p(self.dwhere_r), p(self._kl_weights), pw * inv_b if has_where else 0.0,
                                          p(self.tr.g[-1]), 8, M, 4,
                                          p(self.presence_prob), p(self.presence), p(self.prior_dev), pw * inv_b * float(cfg.nsp_weight),
                                          p(self.kl_what_row) if has_what else None, p(self.kl_where_row) if has_where else None,
                                          pw * inv_b if analytic else 0.0,
                                          dlogp_p, dpres_p, p(self.st.out[-1]), cfg.step_bias,
                                          eps, p(self.st.g[-1]), T, B, float(cfg.guard_eps)), "air_heads_bwd"))
        mlp_bwd_multi(bwd, [dict(m=self.tr, x=h_all, ldx=Hd, g_last=self.tr.g[-1], dx_out=self.dH),
                            dict(m=self.st, x=h_all, ldx=Hd, g_last=self.st.g[-1], dx_out=self.dH_b)],
                      first_dx_done=fuse_attend)
        marks.append((len(bwd), "transform/0/w"))              # + [transform, steps]
        # BPTT through the T recurrences (dgates_t . W_h^T accumulates into dH[t-1] with beta = 1)
        gw = self.grads["lstm/w_gates"]
        dc_in, dc_out = None, self.dc_a
        dgx = self.dgx if T > 1 else self.dgates[0]            # sum over time of dgates (what the hoisted x.W_x receives)
        rider_hosts = []                    # (index in bwd, entry with an optimiser slice): see _plan_bwd_riders below
        fuse_lstm_bwd = fuse_lstm_fwd            # (the library picks the 16-wave or the wide-tile form of the link by size)
        lstm16 = use16 and fuse_lstm_bwd and not fuse_lstm and self._lstm16_ok()
        # Round 5 (session 3): in the latency regime the entry of the BPTT -- the pointwise backward of step T-1, a launch of its own
        # whose only consumer is the link of step T-2 -- is formed inside that link (air_lstm_step_bwd_entry: every workgroup builds
        # its A operand dgates_{T-1} from the saved activations; one dependent launch fewer).  AIR_LSTM_BWD_ENTRY=0: the two launches.
        entry_fold = (fuse_lstm and fuse_lstm_bwd and not lstm16 and T >= 2 and prec == 0
                      and L.air_lstm_step_bwd_entry_fits(B, Hd) == 1 and os.environ.get("AIR_LSTM_BWD_ENTRY", "1") == "1")
        self._lstm_entry_fold = entry_fold
        for t in reversed(range(T)):
            if entry_fold and t == T - 1:
                # (step T-1's results go where the pointwise launch wrote them: dgates[T-1], dc_a)
                entry_dc = dc_out
                dc_in, dc_out = dc_out, (self.dc_b if dc_out is self.dc_a else self.dc_a)
                continue
            if entry_fold and t == T - 2:
                ent_args = (p(self.gate_act[T - 1]), p(self.c_seq[T - 1]), p(self.c_seq[T]), p(self.dH[T - 1]), p(self.dH_b[T - 1]),
                            p(self.dgates[T - 1]), p(entry_dc), p(w_h), p(self.dH[t]), p(self.dH_b[t]), p(self.gate_act[t]),
                            p(self.c_seq[t]), p(self.c_seq[t + 1]), p(self.dgates[t]), p(dc_out), p(self.dgx), B, Hd)
                rider_hosts.append((len(bwd), L.air_lstm_step_bwd_entry, ent_args, "air_lstm_step_bwd_entry"))
                bwd.append((L.air_lstm_step_bwd_entry, ent_args + (None,), "air_lstm_step_bwd_entry"))
                dc_in, dc_out = dc_out, (self.dc_b if dc_out is self.dc_a else self.dc_a)
                continue
            if lstm16 and t == T - 1:
                bwd.append((L.air_lstm_pointwise_bwd_bf16, (p(self.gate_act[t]), p(self.c_seq[t]), p(self.c_seq[t + 1]),
                                                            p(self.dH[t]), p(self.dH_b[t]), None, p(self.dgates[t]),
                                                            m16(self.dgates[t]), p(dc_out), B, Hd),
                            "air_lstm_pointwise_bwd_bf16"))
                dc_in, dc_out = dc_out, (self.dc_b if dc_out is self.dc_a else self.dc_a)
                continue
            if lstm16:
                bwd.append((L.air_lstm_step_bwd_bf16, (p(self.dgates[t + 1]), m16(self.dgates[t + 1]), m16(w_h), p(self.dH[t]),
                                                       p(self.dH_b[t]), p(dc_in), p(self.gate_act[t]), p(self.c_seq[t]),
                                                       p(self.c_seq[t + 1]),
                                                       p(self.dgates[T - 1]) if t == T - 2 else p(self.dgx),
                                                       p(self.dgates[t]), m16(self.dgates[t]), p(dc_out), p(self.dgx),
                                                       m16(self.dgx), B, Hd), "air_lstm_step_bwd_bf16"))
                dc_in, dc_out = dc_out, (self.dc_b if dc_out is self.dc_a else self.dc_a)
                continue
            if t == T - 1 or not fuse_lstm_bwd:
                pw_args = (p(self.gate_act[t]), p(self.c_seq[t]), p(self.c_seq[t + 1]), p(self.dH[t]), p(self.dH_b[t]),
                           p(dc_in) if dc_in is not None else None, p(self.dgates[t]), p(dc_out), B, Hd)
                if fuse_lstm:
                    rider_hosts.append((len(bwd), L.air_lstm_pointwise_bwd_opt, pw_args, "air_lstm_pointwise_bwd_opt"))
                bwd.append((L.air_lstm_pointwise_bwd, pw_args, "air_lstm_pointwise_bwd"))
                if not fuse_lstm_bwd and t > 0:    # dgates_t . W_h^T accumulates into dH[t-1] (beta = 1)
                    launch(bwd, [desc(0, 1, B, Hd, 4 * Hd, self.dgates[t], 4 * Hd, w_h, 4 * Hd, self.dH[t - 1], Hd,
                                      beta=1.0)])
            else:
                # one BPTT link per launch: dgates_{t+1}.W_h^T + direct dh terms -> gate backward of step t -> running dgx
                link_args = (p(self.dgates[t + 1]), p(w_h), p(self.dH[t]), p(self.dH_b[t]), p(dc_in), p(self.gate_act[t]),
                             p(self.c_seq[t]), p(self.c_seq[t + 1]), p(self.dgates[T - 1]) if t == T - 2 else p(self.dgx),
                             p(self.dgates[t]), p(dc_out), p(self.dgx), B, Hd, prec)
                rider_hosts.append((len(bwd), L.air_lstm_step_bwd_opt, link_args, "air_lstm_step_bwd_opt"))
                bwd.append((L.air_lstm_step_bwd, link_args, "air_lstm_step_bwd"))
            dc_in, dc_out = dc_out, (self.dc_b if dc_out is self.dc_a else self.dc_a)
        if not fuse_lstm_bwd and T > 1:
            bwd.append((L.air_sum_leading, (p(self.dgates), p(self.dgx), T, ctypes.c_size_t(B * 4 * Hd)),
                        "air_sum_leading"))
        launch(bwd, [desc(0, 1, B, Hd, 4 * Hd, self.dgates[0], 4 * Hd, w_h, 4 * Hd, self.dh_init, Hd),   # d h_{-1}
                     desc(0, 1, B, E, 4 * Hd, dgx, 4 * Hd, w_x, 4 * Hd, self.enc.g[-1], E, epi=MDELU, aux=enc_out,
                          ldaux=E)])                                                         # d enc_out (pre-activation)
        # dW_h (+ db_gates) and dW_x have no dependants before the optimiser: they ride along with the last launch of the
        # chain (the input encoder's first-layer dW, another wide throughput-type problem) instead of costing their own
        lstm_dw = [desc(1, 0, Hd, 4 * Hd, M, self.h_seq[:T], Hd, self.dgates, 4 * Hd, gw[E:], 4 * Hd,
                        colsum=self.grads["lstm/b_gates"]),                                  # dW_h, db_gates
                   desc(1, 0, E, 4 * Hd, B, enc_out, E, dgx, 4 * Hd, gw[:E], 4 * Hd)]        # dW_x
        self._lstm_tail = [desc(1, 0, 1, Hd, B, self.ones_b, 1, self.dh_init, Hd, self.grads["lstm/h0"], Hd),   # dh0
                           desc(1, 0, 1, Hd, B, self.ones_b, 1, dc_in, Hd, self.grads["lstm/c0"], Hd)]          # dc0
        # Round 5 (late): in the latency regime the LSTM's weight gradients leave the LAST launch for the one in front of it (they
        # have been ready since the BPTT closed): the closing launch -- 19 us with the folded update of input encoder + LSTM --
        # then holds the input encoder's first layer only, and the LSTM's half of that update runs in the epilogue of a launch
        # that left most of the chip idle.  AIR_LSTM_DW_EARLY=0: the previous placement (A/B).
        self._lstm_dw_early = (not throughput and not use16 and self.enc.n >= 2 and os.environ.get("AIR_LSTM_DW_EARLY", "1") == "1")
        if self._lstm_dw_early:
            mlp_bwd_multi(bwd, [dict(m=self.enc, x=self.obs, ldx=P, g_last=self.enc.g[-1])], extra_first=self._lstm_tail + lstm_dw)
        else:
            mlp_bwd_multi(bwd, [dict(m=self.enc, x=self.obs, ldx=P, g_last=self.enc.g[-1])], extra_first=self._lstm_tail,
                          extra_last=lstm_dw)

        if deferred_dw:
            # wide-tile eligible problems (16-byte loads along M and N: both multiples of 4, aligned) together, longest K first
            # so that the heaviest tiles start first; the rest (M = 50 / 677 / 1, N = 1) in a launch of the ordinary kernel
            def wide_ok(d):
                return d.M % 4 == 0 and d.M >= 4 and d.N % 4 == 0 and d.ldb % 4 == 0 and d.K % 4 == 0 and d.B % 16 == 0
            # a row count that is not a multiple of 4 (50 latent / 677 baseline-input rows): the first M - M % 4 rows go wide,
            # the remaining one to three rows are a problem of their own (the bias gradient stays with the first part)
            parts = []
            for d in deferred_dw:
                m4 = d.M // 4 * 4
                if d.M % 4 and m4 >= 16 and d.N % 4 == 0 and d.ldb % 4 == 0 and d.K % 4 == 0 and d.B % 16 == 0:
                    parts.append(_lib.AirGemmDesc(d.ta, d.tb, m4, d.N, d.K, d.A, d.lda, d.B, d.ldb, d.C, d.ldc, d.bias,
                                                  d.epilogue, d.aux, d.ldaux, d.beta, d.colsum, d.precision, None, None, 0, None))
                    parts.append(_lib.AirGemmDesc(d.ta, d.tb, d.M - m4, d.N, d.K, d.A + 4 * m4, d.lda, d.B, d.ldb,
                                                  d.C + 4 * m4 * d.ldc, d.ldc, d.bias, d.epilogue, d.aux, d.ldaux, d.beta, None,
                                                  d.precision, None, None, 0, None))
                else:
                    parts.append(d)
            wide = sorted((d for d in parts if wide_ok(d)), key=lambda d: (-d.K, -d.M * d.N))
            rest = [d for d in parts if not wide_ok(d)]
            # The library takes up to 24 problems in one launch when at least one is wide-tile eligible: the odd-shaped rest (one to
            # three rows, a single column: eight long-K reductions) rides in the same grid on 16x16 tiles instead of costing a
            # 14.5 us launch of its own.
            def pack(problems, n_wide):
                """launches of up to 24 problems while wide-tile members are among them (the library's mixed form needs at
                least one), of up to 8 otherwise"""
                out, i = [], 0
                while i < len(problems):
                    step = 24 if i < n_wide else 8
                    chunk = problems[i:i + step]
                    if len(chunk) <= 8 and i < n_wide < i + len(chunk):     # (8 or fewer go through the all-one-kind launches)
                        out += [problems[i:n_wide], problems[n_wide:i + len(chunk)]]
                    else:
                        out.append(chunk)
                    i += step
                return out
            if prec == 0:
                # fp32 (MFMA-issue bound tiles): ONE launch for everything -- long-K tiles first, the CUs that finish early keep
                # pulling short-K tiles instead of idling until a launch of their own (batch 1024: 0.636 -> 0.607 ms)
                groups = pack(wide + rest, len(wide))
            else:
                # bf16 operands (L2 / L1 traffic bound tiles): the long-K problems in a launch of 8 with the grid-wide
                # XCD-contiguous tile map, the others + the rest in a second one (0.474 against 0.485 ms for a single launch)
                groups = ([wide[:8]] if wide[:8] else []) + pack(wide[8:] + rest, len(wide[8:]))
            for grp in groups:
                arr = (_lib.AirGemmDesc * len(grp))(*grp)
                self._keep.append(arr)
                bwd.append((L.air_gemm_grouped, (arr, len(grp)), "air_gemm_grouped"))
            marks = []                            # no gradient slice is final before the end of the backward

        # ---- L2 term (model.py:346-353): g += l2 * w on the 2-D model variables, the last launch of the backward (so every protocol
        #      -- single GPU, data parallel: the all-reduced sum of `world` identical terms is scaled back by 1/world -- sees it).
        #      Rare switch (0 in the script): the riders / folded update, which consume gradients before the end, are then off.
        if cfg.l2_weight and cfg.l2_weight > 0.0:
            spans = []
            for k, shape in self.param_shapes.items():
                if len(shape) == 2 and not k.startswith("baseline/"):
                    lo = self.param_offsets[k]
                    if spans and spans[-1][1] == lo:
                        spans[-1][1] = lo + self.param_sizes[k]
                    else:
                        spans.append([lo, lo + self.param_sizes[k]])
            if len(spans) > 32:
                raise _lib.AirHipError("more than 32 separate 2-D model tensors: the L2 launch takes 32 slices")
            lo_arr = (ctypes.c_size_t * len(spans))(*[a for a, _ in spans])
            hi_arr = (ctypes.c_size_t * len(spans))(*[b_ for _, b_ in spans])
            self._keep += [lo_arr, hi_arr]
            bwd.append((L.air_l2_grad_add, (p(self.flat_grads), p(self.flat_params), lo_arr, hi_arr, len(spans), float(cfg.l2_weight)),
                        "air_l2_grad_add"))
            marks, rider_hosts = [], []
        # ---- optimiser: both centred-RMSProp updates + device counters in one launch ---------------------------------
        tail_mult = cfg.baseline_lr_mult if cfg.use_reinforce else 0.0
        pre_fwd = []
        if use16:
            pre_fwd = self._apply_bf16_mirrors([fwd, bwd])
            self._dx_chain_launches = 0
            # (round 6, measured and not adopted: 0.47-0.50 against 0.415-0.425 ms per step at configs[4], profiles/r06_dx_chain_rejected.txt;
            #  AIR_DX_CHAIN=1 switches the pass on)
            if throughput and os.environ.get("AIR_DX_CHAIN", "0") == "1":
                bwd[:] = self._fuse_dx_chains(bwd)
            shadow = p(self.flat_params16)
            self._opt_calls_factory = lambda gscale: [
                (L.air_step_epilogue_shadow, (p(self.flat_params), p(self.flat_grads), p(self.flat_ms), p(self.flat_mg),
                                              p(self.flat_mom), ctypes.c_size_t(self.n_model), ctypes.c_size_t(self.n_total),
                                              p(self.lr_dev), tail_mult, cfg.rms_decay, cfg.rms_momentum, cfg.rms_eps, gscale,
                                              p(self.step_dev), p(self.rng_state), ctypes.c_uint64(self._rng_inc), shadow),
                 "air_step_epilogue_shadow")]
        else:
            self._opt_calls_factory = lambda gscale: [
                (L.air_step_epilogue, (p(self.flat_params), p(self.flat_grads), p(self.flat_ms), p(self.flat_mg),
                                       p(self.flat_mom), ctypes.c_size_t(self.n_model), ctypes.c_size_t(self.n_total),
                                       p(self.lr_dev), tail_mult, cfg.rms_decay, cfg.rms_momentum, cfg.rms_eps, gscale,
                                       p(self.step_dev), p(self.rng_state), ctypes.c_uint64(self._rng_inc)),
                 "air_step_epilogue")]
        if not cfg.rms_centered:
            # tf.train.RMSPropOptimizer(centered=False) (model.py:265 with other opt_kwargs): the generic update kernel on the two
            # segments + the counters; no riders (the BPTT riders and the fold are the centred form)
            if use16:
                raise _lib.AirHipError("the bf16 data path keeps its parameter shadow in the centred update launch: centered=False is not available with it")
            seg = lambda lo, hi, mult, gscale: (L.air_rmsprop, (
                ctypes.c_void_p(self.flat_params.data_ptr() + 4 * lo), ctypes.c_void_p(self.flat_grads.data_ptr() + 4 * lo),
                ctypes.c_void_p(self.flat_ms.data_ptr() + 4 * lo), ctypes.c_void_p(self.flat_mg.data_ptr() + 4 * lo),
                ctypes.c_void_p(self.flat_mom.data_ptr() + 4 * lo), ctypes.c_size_t(hi - lo), p(self.lr_dev), mult, cfg.rms_decay,
                cfg.rms_momentum, cfg.rms_eps, 0, gscale), "air_rmsprop")
            self._opt_calls_factory = lambda gscale: (
                [seg(0, self.n_model, 1.0, gscale)] + ([seg(self.n_model, self.n_total, tail_mult, gscale)] if self.n_total > self.n_model and tail_mult else [])
                + [(L.air_counter_add, (p(self.step_dev), ctypes.c_int64(1)), "air_counter_add"),
                   (L.air_rng_advance, (p(self.rng_state), ctypes.c_uint64(self._rng_inc)), "air_rng_advance")])
            rider_hosts = []
        self._plan_rng = rng
        def fwd_plan(with_noise):
            """the forward list with its prologue: a launch of its own, or -- with the fused LSTM steps -- extra workgroups
            of the first LSTM step, which then reads h0 / c0 with a broadcast row stride"""
            if not prologue_rides:
                return [prologue(with_noise)] + fwd
            pro_tail = (p(self.noise_normal), ctypes.c_size_t(n_norm if with_noise else 0), p(self.u_pres),
                        ctypes.c_size_t(n_uni if with_noise else 0), p(self.rng_state), p(self.step_dev), anneal,
                        float(cfg.nsp_init), float(cfg.nsp_final), float(cfg.nsp_steps), float(cfg.nsp_hold_init),
                        float(cfg.nsp_steps_div), p(self.prior_dev), T, p(self.h_seq[0]), p(self.c_seq[0]))
            if fold_gx:
                lstm0 = (L.air_lstm_first_step_fwd,
                         (p(enc_out), E, E, p(w_x), p(bg), p(self.params["lstm/h0"]), p(self.params["lstm/c0"]), p(w_h), 4 * Hd,
                          p(self.gx), 4 * Hd, p(self.h_seq[1]), p(self.c_seq[1]), p(self.gate_act[0]), B, Hd, 1.0, prec) + pro_tail,
                         "air_lstm_first_step_fwd")
                return fwd[:lstm0_index] + [lstm0] + fwd[lstm0_index + 1:]
            lstm0 = (L.air_lstm_step_fwd_prologue,
                     (p(self.params["lstm/h0"]), p(self.params["lstm/c0"]), p(w_h), 4 * Hd, p(self.gx), 4 * Hd,
                      p(self.h_seq[1]), p(self.c_seq[1]), p(self.gate_act[0]), B, Hd, 1.0, prec,
                      p(self.noise_normal), ctypes.c_size_t(n_norm if with_noise else 0), p(self.u_pres),
                      ctypes.c_size_t(n_uni if with_noise else 0), p(self.rng_state), p(self.step_dev), anneal,
                      float(cfg.nsp_init), float(cfg.nsp_final), float(cfg.nsp_steps), float(cfg.nsp_hold_init),
                      float(cfg.nsp_steps_div), p(self.prior_dev), T, p(self.h_seq[0]), p(self.c_seq[0])),
                     "air_lstm_step_fwd_prologue")
            return fwd[:lstm0_index] + [lstm0] + fwd[lstm0_index + 1:]

        if pre_fwd and not prologue_rides and os.environ.get("AIR_FUSE_PROLOGUE_CVT", "1") == "1" and self.obs.numel() % 4 == 0:
            # bf16 data path (round 5): the conversion of the observation batch rides in the step prologue -- the two independent tiny
            # launches that opened the step are one (air_step_prologue_cvt)
            cvt_tail = (p(self.obs), ctypes.c_void_p(self.obs16.data_ptr()), ctypes.c_size_t(self.obs.numel()))
            plain_fwd_plan = fwd_plan

            def fwd_plan(with_noise):                      # noqa: F811
                pl = plain_fwd_plan(with_noise)
                assert pl[0][2] == "air_step_prologue"
                return [(L.air_step_prologue_cvt, pl[0][1] + cvt_tail, "air_step_prologue_cvt")] + pl[1:]
            pre_fwd = []
        self._plan_fwd_noise = pre_fwd + fwd_plan(True) + fwd_tail    # forward(): complete outputs
        self._plan_fwd = pre_fwd + fwd_plan(False) + fwd_tail
        # train step: NVIL rides in the first backward launch, the `what` KL shares are added by the backward of that head
        self._plan_fwd_train = [e for e in pre_fwd + fwd_plan(True) if e[2] != "air_sum_leading:kl_what"]
        if fuse_canvas:                                               # ... and the canvas forward IS the first backward launch
            assert self._plan_fwd_train[-1][2] == "air_canvas_unroll_fwd_banded"
            self._plan_fwd_train = self._plan_fwd_train[:-1]
        feeder = getattr(self, "_feeder", None)
        if feeder is not None:
            # the batch itself is drawn by the first launch of the train step (attach_dataset): no host work between updates
            data, shuffle = feeder
            gather = (L.air_batch_gather, (p(data), ctypes.c_longlong(data.shape[0]), int(data.shape[1]),
                                           p(self.feeder_seed), p(self.step_dev), int(shuffle), p(self.obs), B,
                                           p(self.batch_idx)), "air_batch_gather")
            # Round 6: when the step opens with the products over the pixels of obs (latency regime: the K-split halves of the input
            # encoder's and the baseline's first layers), the gather is folded into their A-operand load -- row m is read from item
            # idx_m of the dataset and the first column of tiles writes it to `obs` for every later reader -- instead of being a
            # dependent launch of its own (air_gemm_grouped_gather; AIR_FOLD_GATHER=0: the two launches).  Bit-identical either way.
            self._fold_gather = False
            first = self._plan_fwd_train[0] if self._plan_fwd_train else None
            if (first is not None and first[2] == "air_gemm_grouped" and os.environ.get("AIR_FOLD_GATHER", "1") == "1"):
                arr, n_d = first[1]
                obs_lo = self.obs.data_ptr()
                offs = [(int(arr[i].A) - obs_lo) // 4 for i in range(n_d)]
                # the problems that write obs: one per distinct (column offset, K) -- together they must tile [0, P)
                seen, mask = {}, 0
                for i in range(n_d):
                    key = (offs[i], arr[i].K)
                    if key not in seen:
                        seen[key] = i; mask |= 1 << i
                cover = sorted(seen)
                tiled = bool(cover) and cover[0][0] == 0 and all(cover[j][0] + cover[j][1] == cover[j + 1][0] for j in range(len(cover) - 1)) \
                    and cover[-1][0] + cover[-1][1] == P
                bg = _lib.AirBatchGather(data.data_ptr(), int(data.shape[0]), int(data.shape[1]), int(shuffle), B,
                                         self.feeder_seed.data_ptr(), self.step_dev.data_ptr(), obs_lo, self.batch_idx.data_ptr(), mask)
                if tiled and int(data.shape[1]) == P and L.air_gemm_grouped_gather_fits(arr, n_d, ctypes.byref(bg)) == 1:
                    self._keep.append(bg)
                    self._plan_fwd_train = [(L.air_gemm_grouped_gather, (arr, n_d, ctypes.byref(bg)), "air_gemm_grouped_gather")] + self._plan_fwd_train[1:]
                    self._fold_gather = True
            if (not self._fold_gather and first is not None and first[2] == "air_step_prologue_cvt" and os.environ.get("AIR_FOLD_GATHER", "1") == "1"
                    and int(data.shape[1]) == P and P % 4 == 0):
                # bf16 data path (throughput regime): the step opens with the prologue whose extra workgroups convert obs to its bf16
                # mirror -- they read the rows from the dataset instead and write both (air_step_prologue_gather_cvt): no gather launch
                bg = _lib.AirBatchGather(data.data_ptr(), int(data.shape[0]), int(data.shape[1]), int(shuffle), B,
                                         self.feeder_seed.data_ptr(), self.step_dev.data_ptr(), self.obs.data_ptr(), self.batch_idx.data_ptr(), 0)
                self._keep.append(bg)
                pro = first[1]                       # (..., x, x_bf16, n_x): the prologue's own arguments, then the conversion's
                assert int(pro[-1].value) == B * P
                self._plan_fwd_train = [(L.air_step_prologue_gather_cvt, pro[:-3] + (ctypes.byref(bg), pro[-2]), "air_step_prologue_gather_cvt")] \
                    + self._plan_fwd_train[1:]
                self._fold_gather = True
            if not self._fold_gather:
                self._plan_fwd_train = [gather] + self._plan_fwd_train
        self._plan_bwd = bwd
        # data-parallel gradient buckets: (end index in the backward plan, [lo, hi) slice of the flat gradient buffer that
        # is final once the plan has run up to that index); contiguous, from the tail of the buffer to its head
        self._grad_buckets, hi = [], self.n_total
        for idx, first in marks:
            lo = self.param_offsets[first]
            self._grad_buckets.append((idx, lo, hi)); hi = lo
        self._grad_buckets.append((len(bwd), 0, hi))
        self._plan_opt = self._opt_calls_factory(1.0)
        # Single-GPU train step in the latency regime: the centred-RMSProp update of everything whose gradient is final before
        # the BPTT chain (all but the input encoder and the LSTM: half of the 94 MB the update streams) rides as extra
        # workgroups of the BPTT launches -- 64 tiles each, three quarters of the chip idle -- and the closing launch only
        # updates the head of the buffer.  backward() / data-parallel steps (update after the all-reduce) keep the plain plans.
        self._plan_bwd_riders = self._plan_opt_rest = self._fold = None
        if rider_hosts and marks and not self._defer_dw and os.environ.get("AIR_OPT_RIDERS", "1") == "1":
            r_lo, r_hi = self.param_offsets[marks[-1][1]], self.n_total
            if r_lo % 4 == 0 and r_hi % 4 == 0 and self.n_model % 4 == 0 and r_lo > 0 and r_hi > r_lo:
                nh = len(rider_hosts)
                cuts = [r_lo + ((r_hi - r_lo) * i // nh) // 4 * 4 for i in range(nh)] + [r_hi]
                riders = list(bwd)
                self._rider_slices_all = getattr(self, "_rider_slices_all", [])
                self._rider_slices = []
                for (idx, fn, args, name), lo, hi in zip(rider_hosts, cuts[:-1], cuts[1:]):
                    sl = _lib.AirRmspropSlice(dp(self.flat_params), dp(self.flat_grads), dp(self.flat_ms), dp(self.flat_mg),
                                              dp(self.flat_mom), lo, hi, self.n_model, dp(self.lr_dev), tail_mult,
                                              cfg.rms_decay, cfg.rms_momentum, cfg.rms_eps, 1.0)
                    self._rider_slices.append(sl)
                    self._rider_slices_all.append(sl)
                    riders[idx] = (fn, args + (ctypes.byref(sl),), name)
                self._plan_bwd_riders = riders
                self._plan_opt_rest = [
                    (L.air_step_epilogue, (p(self.flat_params), p(self.flat_grads), p(self.flat_ms), p(self.flat_mg),
                                           p(self.flat_mom), ctypes.c_size_t(min(self.n_model, r_lo)), ctypes.c_size_t(r_lo),
                                           p(self.lr_dev), tail_mult, cfg.rms_decay, cfg.rms_momentum, cfg.rms_eps, 1.0,
                                           p(self.step_dev), p(self.rng_state), ctypes.c_uint64(self._rng_inc)),
                     "air_step_epilogue")]
                self._fold_closing_update(riders, r_lo, tail_mult)

    def _fold_closing_update(self, riders, r_lo, tail_mult):
        """Round 5: the closing launch of the single-GPU latency-regime step (air_step_epilogue over the head [0, r_lo) of the flat
        buffers -- input encoder + LSTM, whose gradients the last launches of the backward form) disappears: the weight-gradient
        problems of the LAST backward launch apply centred RMSProp to the elements they finish, in the same epilogue
        (air_gemm_grouped_opt: on one GPU a tile's gradient is final when formed; nothing in that launch reads a parameter), and
        whatever else of the head was final before it rides as extra workgroups of the same launch, one of which advances the
        step counter and the Philox offset.  33 launches instead of 34 at BASELINE configs[1].  Applies when the head is made
        of whole float4 tensors that the last launch either forms or that earlier launches left final; otherwise the plan keeps
        its closing launch.  AIR_OPT_FOLD=0 switches it off (A/B)."""
        self._fold = None
        if os.environ.get("AIR_OPT_FOLD", "1") != "1" or self._use16:
            return
        L, dp = H.lib(), (lambda t: t.data_ptr())
        g0 = self.flat_grads.data_ptr()
        p0, p1 = self.flat_params.data_ptr(), self.flat_params.data_ptr() + 4 * self.n_total
        head = [(self.param_offsets[k], self.param_sizes[k]) for k in self.param_shapes if self.param_offsets[k] < r_lo]
        if any(sz % 4 for _, sz in head) or sum(sz for _, sz in head) != r_lo:
            return                                   # padding inside the head: the closing launch stays
        spans = sorted((self.param_offsets[k], self.param_offsets[k] + self.param_sizes[k]) for k in self.param_shapes)

        def tensors_read(entry):
            """flat-buffer spans of the parameter tensors the problems of a grouped launch read (a dX problem reads its layer's
            weights; weight gradients read activations / gradients only)"""
            arr, n = entry[1]
            out = []
            for i in range(n):
                for x in (arr[i].A, arr[i].B, arr[i].aux, arr[i].bias, arr[i].A2):
                    if p0 <= int(x or 0) < p1:
                        off = (int(x) - p0) // 4
                        out += [sp for sp in spans if sp[0] <= off < sp[1]]
            return out

        def foldable(entry, barred):
            """(mask, covered spans) of the launch's weight-gradient problems whose parameters nothing in `barred` reads"""
            arr, n = entry[1]
            mask, cov = 0, []
            for i in range(n):
                d = arr[i]
                if not (d.ta and not d.tb):
                    continue
                off = (int(d.C) - g0) // 4
                if not (0 <= off < r_lo) or d.ldc != d.N or d.beta != 0.0 or d.epilogue != H.EPI_NONE:
                    continue
                mine = [(off, off + d.M * d.N)]
                if d.colsum:
                    coff = (int(d.colsum) - g0) // 4
                    mine.append((coff, coff + d.N))
                if any(a0 < b1 and b0 < a1 for a0, a1 in mine for b0, b1 in barred):
                    continue
                mask |= 1 << i
                cov += mine
            return mask, cov

        def wide_form(entry):
            # the library declines (AIR_E_UNSUPPORTED) a group its wide-tile dispatch would take -- all weight gradients, K >= 256, more
            # than AIR_GEMM_WIDE_MIN_TILES 16x16 tiles (a long batch at T = 1): such a plan keeps its closing launch
            arr, n = entry[1]
            tiles16 = sum(((arr[i].M + 15) // 16) * ((arr[i].N + 15) // 16) for i in range(n))
            return (tiles16 > int(os.environ.get("AIR_GEMM_WIDE_MIN_TILES", "1000")) and all(arr[i].ta and not arr[i].tb for i in range(n))
                    and min(arr[i].K for i in range(n)) >= 256)

        def shortk_mixed(entry):
            # ... and a group that MIXES a short-K streaming weight gradient (gemm_kernels.hip shortk_eligible: fp32 TN, K <= 64 in whole
            # chunks, N a multiple of 64 up to 256, >= AIR_GEMM_SHORTK_MIN_M rows, no epilogue) with tile problems: such a problem keeps
            # the product of the streaming body in every plan, and the folded launch takes it only when all its problems are of that kind
            if os.environ.get("AIR_GEMM_SHORTK", "1") == "0":
                return False
            min_m = int(os.environ.get("AIR_GEMM_SHORTK_MIN_M", "4096"))
            arr, n = entry[1]
            if any(arr[i].A2 or arr[i].C16 or arr[i].precision != 0 for i in range(n)):
                return False
            el = [bool(arr[i].ta and not arr[i].tb and 16 <= arr[i].K <= 64 and arr[i].K % 16 == 0 and 64 <= arr[i].N <= 256
                       and arr[i].N % 64 == 0 and arr[i].M >= min_m and arr[i].epilogue == 0 and arr[i].beta == 0.0 and not arr[i].bias)
                  for i in range(n)]
            return any(el) and not all(el)

        def disjoint(cov):
            cov = sorted(cov)
            return all(b0 >= a1 for (a0, a1), (b0, b1) in zip(cov, cov[1:]))

        last = riders[-1]
        if last[2] != "air_gemm_grouped" or wide_form(last) or shortk_mixed(last):
            return
        # the closing launch folds what it forms; nothing in it may read a parameter at all (the rider workgroups behind its tiles
        # update the rest of the head, whatever it is)
        if tensors_read(last):
            return
        mask_b, cov_b = foldable(last, [])
        if not mask_b:
            return
        # the launch in front of it (round 5, late: it holds the LSTM's weight gradients): folds the problems whose parameters neither
        # it nor the closing launch reads (its dX problem reads ITS layer's weights -- that layer's update stays with the closing launch)
        prev, mask_a, cov_a = None, 0, []
        if getattr(self, "_lstm_dw_early", False) and len(riders) >= 2 and riders[-2][2] == "air_gemm_grouped" and not wide_form(riders[-2]) and not shortk_mixed(riders[-2]):
            prev = riders[-2]
            mask_a, cov_a = foldable(prev, tensors_read(prev))
            if not mask_a or not disjoint(cov_a + cov_b):
                prev, mask_a, cov_a = None, 0, []
        covered = sorted(cov_a + cov_b)
        if not disjoint(covered):
            return                                   # overlapping outputs: not a layout this fold understands
        ranges, cur = [], 0
        for a0, a1 in covered + [(r_lo, r_lo)]:
            if a0 > cur:
                ranges.append((cur, a0))
            cur = max(cur, a1)
        if len(ranges) > 4 or any(lo % 4 or hi % 4 for lo, hi in ranges):
            return
        cfg = self.cfg

        def make_fold(mask, rngs, counters):
            fold = _lib.AirOptFold()
            fold.p, fold.g, fold.ms, fold.mg, fold.mom = (dp(self.flat_params), g0, dp(self.flat_ms), dp(self.flat_mg), dp(self.flat_mom))
            fold.n_model = self.n_model
            fold.lr_dev = dp(self.lr_dev)
            fold.lr_mult_tail, fold.decay, fold.momentum, fold.eps, fold.grad_scale = (tail_mult, cfg.rms_decay, cfg.rms_momentum, cfg.rms_eps, 1.0)
            fold.fold_mask, fold.n_ranges = mask, len(rngs)
            for j, (lo, hi) in enumerate(rngs):
                fold.range_lo[j], fold.range_hi[j] = lo, hi
            if counters:
                fold.global_step_dev, fold.rng_state_dev, fold.rng_increment = dp(self.step_dev), dp(self.rng_state), self._rng_inc
            self._keep.append(fold)
            return fold

        riders = list(riders)
        self._fold_early = None
        if prev is not None:
            self._fold_early = make_fold(mask_a, [], False)          # (no rider slices, the counters move in the closing launch)
            riders[-2] = (L.air_gemm_grouped_opt, (prev[1][0], prev[1][1], ctypes.byref(self._fold_early)), "air_gemm_grouped_opt")
        self._fold = make_fold(mask_b, ranges, True)
        riders[-1] = (L.air_gemm_grouped_opt, (last[1][0], last[1][1], ctypes.byref(self._fold)), "air_gemm_grouped_opt")
        self._plan_bwd_riders = riders
        self._plan_opt_rest = []

    def _alloc_bf16_mirrors(self):
        """bf16 mirrors (same shape) of every buffer ALL of whose writers keep the mirror up to date -- see _apply_bf16_mirrors"""
        if getattr(self, "_mirror_spans", None) is not None:
            return
        dev, bf = self.device, torch.bfloat16
        self.flat_params16 = torch.zeros(self.n_total, dtype=bf, device=dev)
        self.obs16 = torch.zeros(self.obs.shape, dtype=bf, device=dev)
        self.h_seq16 = torch.zeros(self.h_seq.shape, dtype=bf, device=dev)
        self.dgates16 = torch.zeros(self.dgates.shape, dtype=bf, device=dev)
        self.dgx16 = torch.zeros(self.dgx.shape, dtype=bf, device=dev)
        self._mirror_of = {}
        trusted = [(self.flat_params, self.flat_params16), (self.obs, self.obs16)]
        # the LSTM's products: h_1..h_T (h_0 is the tiled initial state, written by the prologue: no mirror), dgates, running dgx
        lstm16 = self._lstm16_ok()
        if lstm16:
            trusted += [(self.h_seq[1:], self.h_seq16[1:]), (self.dgates, self.dgates16)]
            if self.T > 1:
                trusted.append((self.dgx, self.dgx16))
        gemm_made = []
        for m in (self.enc, self.ge, self.gd, self.bl):
            gemm_made += list(m.out)
        for m in (self.tr, self.st):     # the heads' output layers are written by air_attend_fwd when it is fused (and no product
            gemm_made += list(m.out[:-1])  # reads them as an operand either way): no mirror to trust
        for m in (self.enc, self.ge, self.gd, self.bl):      # (transform / steps: attend_bwd writes part of their gradient chain)
            gemm_made += list(m.g[:-1])
        gemm_made += [self.ge.g[-1], self.enc.g[-1]]
        for t in gemm_made:
            self._mirror_of[t.data_ptr()] = torch.zeros(t.shape, dtype=bf, device=dev)
            trusted.append((t, self._mirror_of[t.data_ptr()]))
        self._gemm_made = gemm_made
        self._mirror_spans = [(t.data_ptr(), t.data_ptr() + 4 * t.numel(), m16.data_ptr()) for t, m16 in trusted]

    def _lstm16_ok(self):
        """the shapes air_lstm_step_*_bf16 take (the library's wide-tile LSTM form)"""
        Hd, E = self.cfg.n_hidden, int(self.cfg.inpt_encoder_hidden[-1])
        # (the same threshold _build_plans uses for `fuse_lstm`: below it the fp32 fused steps run and write no mirror)
        return (((self.B + 15) // 16) * ((Hd + 15) // 16) > int(os.environ.get("AIR_FUSE_LSTM_TILES", "512"))
                and Hd % 64 == 0 and E % 4 == 0
                and os.environ.get("AIR_FUSE_LSTM_WIDE", "1") == "1" and os.environ.get("AIR_BF16_LSTM", "1") == "1")

    def _mirror_ptr(self, t):
        """address of the bf16 mirror of tensor / address `t` (None if it has none)"""
        ptr = t.data_ptr() if torch.is_tensor(t) else (int(t) if t else 0)
        if not ptr:
            return None
        for lo, hi, base16 in self._mirror_spans:
            if lo <= ptr < hi:
                return ctypes.c_void_p(base16 + (ptr - lo) // 2)
        return None

    def _apply_bf16_mirrors(self, plans):
        """bf16 data path: give every GEMM descriptor of `plans` the bf16 mirrors of its operands and of its output.

        Mirrored buffers (same shape, bf16): the flat parameter buffer (`flat_params16`: refreshed by every writer of the
        parameters -- the optimiser launch, load_parameters / init / load_state_dict), the observation batch (`obs16`: a convert
        launch at the start of every forward), the activations / gradients whose ONLY writers are GEMM epilogues -- each
        MLP's layer outputs and the hidden-layer gradients of the chains that GEMMs produce end to end (the library writes the
        mirror of C on every bf16 code path when the descriptor names one) -- and the LSTM's h_1..h_T, dgates and running dgx
        (air_lstm_step_*_bf16 write them).  Whatever another kernel writes (sampled latents, glimpses, the gradients the loss
        kernels hand to the chains, the tiled initial state) has no mirror: those operands are fetched as fp32 and rounded in
        registers, as before.  The values a product sees are identical either way (the mirror
        holds bf16(x), the register path computes bf16(x)); only the bytes moved change.
        Returns the launches that must precede a forward (the conversion of obs)."""
        L, p = H.lib(), H._p
        out_spans = [(t.data_ptr(), t.data_ptr() + 4 * t.numel()) for t in self._gemm_made]

        def mirror(ptr):
            m = self._mirror_ptr(ptr)
            return m.value if m is not None else None

        for plan in plans:
            for e in plan:
                if e is None or e[2] != "air_gemm_grouped":
                    continue
                for d in e[1][0]:
                    d.A16, d.B16 = mirror(d.A), mirror(d.B)
                    c = int(d.C) if d.C else 0
                    d.C16 = mirror(c) if any(lo <= c < hi for lo, hi in out_spans) else None
        self._sync_param_shadow()
        return [(L.air_f32_to_bf16, (p(self.obs), ctypes.c_void_p(self.obs16.data_ptr()), ctypes.c_size_t(self.obs.numel())),
                 "air_f32_to_bf16")]

    def _fuse_dx_chains(self, plan):
        """bf16 data path, throughput regime (round 6): runs of consecutive grouped-GEMM launches whose dX problems feed each other
        -- layer after layer of an MLP's backward, dA_{l-1} = (dA_l . W_l^T) * elu'(out_{l-1}) -- become ONE row-slab launch
        (air_mlp_dx_chain_bf16: a workgroup walks the whole chain for its 16 rows; nothing crosses rows).  A chain starts at a dX
        problem whose input no earlier problem of the run produces and is placed where its first layer was (every later layer only
        needs the layer before it and saved activations, so running it earlier is safe); problems that are not part of a chain of at
        least two layers stay in their grouped launch.  Same values as the per-layer launches up to the order of the fp32
        accumulation (each layer reads bf16 of the previous fp32 result either way)."""
        L = H.lib()
        MDELU, NONE = H.EPI_MUL_DELU, H.EPI_NONE

        def dx_ok(d):
            return (not d.ta and d.tb and d.epilogue in (MDELU, NONE) and d.beta == 0.0 and not d.A2 and not d.colsum and not d.bias
                    and d.B16 and d.ldb == d.K and (d.epilogue == NONE or d.aux) and L.air_mlp_dx_chain_fits(d.K, d.N) == 1)

        out, i = [], 0
        while i < len(plan):
            e = plan[i]
            if e is None or e[2] != "air_gemm_grouped":
                out.append(e); i += 1
                continue
            j = i
            while j < len(plan) and plan[j] is not None and plan[j][2] == "air_gemm_grouped":
                j += 1
            launches = [[x[1][0][q] for q in range(x[1][1])] for x in plan[i:j]]
            chains = []                                      # [first launch index, last launch index, [descs]]
            for a_idx, ds in enumerate(launches):
                for d in ds:
                    if not dx_ok(d):
                        continue
                    host = None
                    for c in chains:
                        t = c[2][-1]
                        if (c[1] < a_idx and len(c[2]) < 4 and int(d.A) == int(t.C) and d.lda == t.ldc and d.K == t.N and d.M == t.M):
                            host = c
                            break
                    if host is not None:
                        host[2].append(d); host[1] = a_idx
                    else:
                        chains.append([a_idx, a_idx, [d]])
            chains = [c for c in chains if len(c[2]) >= 2]
            fused = {id(d) for c in chains for d in c[2]}
            for a_idx, ds in enumerate(launches):
                starts = [c for c in chains if c[0] == a_idx]
                for k0 in range(0, len(starts), 4):
                    grp = starts[k0:k0 + 4]
                    arr = (_lib.AirDxChain * len(grp))()
                    for ci, c in enumerate(grp):
                        d0 = c[2][0]
                        arr[ci].g_in, arr[ci].ld_in, arr[ci].rows, arr[ci].n_layers = d0.A, d0.lda, d0.M, len(c[2])
                        for li, d in enumerate(c[2]):
                            y = arr[ci].layer[li]
                            y.w_bf16, y.aux, y.out, y.out_bf16 = d.B16, (d.aux if d.epilogue == MDELU else None), d.C, d.C16
                            y.n_in, y.n_out, y.ldaux, y.ldout = d.K, d.N, d.ldaux, d.ldc
                    self._keep.append(arr)
                    out.append((L.air_mlp_dx_chain_bf16, (arr, len(grp)), "air_mlp_dx_chain_bf16"))
                    self._dx_chain_launches += 1
                rest = [d for d in ds if id(d) not in fused]
                if rest:
                    arr = (_lib.AirGemmDesc * len(rest))(*rest)
                    self._keep.append(arr)
                    out.append((L.air_gemm_grouped, (arr, len(rest)), "air_gemm_grouped"))
            i = j
        return out

    def _sync_param_shadow(self):
        """bf16 shadow of the parameters after anything but the optimiser launch wrote them"""
        if getattr(self, "flat_params16", None) is not None:
            st = H.lib().air_f32_to_bf16(H._p(self.flat_params), ctypes.c_void_p(self.flat_params16.data_ptr()),
                                         ctypes.c_size_t(self.n_total), self._sp())
            _lib.check(st, "air_f32_to_bf16")
