"""Planned-graph engine: a network pass is a static list of fused HIP stages over pre-allocated
NHWC buffers, issued with ONE host call (pdae_run_ops).

This replaces the per-module Python dispatch of the reference (TimestepSequential, model/module.py:131-140,
and torch autograd) with a plan built once per (network, batch shape, mode):
  * `Plan`     -- buffer pool + op records + shared workspace, compiled to a ctypes array;
  * `Builder`  -- emitters for the fused stages (conv, GroupNorm/AdaGN/SiLU, linear, attention) and
                  block-level forward / hand-derived backward graphs for ResBlock / ResBlockShift
                  (model/module.py:205-384) and AttentionBlock (:387-428).
Only what the path needs is differentiated: the frozen trunk / eps-branch of ShiftUNet emit no backward.
"""
import math
import bisect
import copy
import os
from types import SimpleNamespace as NS

import torch

from . import hip as H

GROUPS = 32       # normalization(channels) = GroupNorm(32, C)  (model/module.py:56-63)
GN_EPS = 1e-5


class Plan:
    def __init__(self, device):
        self.device = torch.device(device)
        self.recs = []
        self.ws_patch = []
        self.pool = {}
        self.ws_bytes = 4096
        self.arr = None
        self.n = 0
        self.ws = None
        self.live = []          # every tensor ever allocated (keeps storage alive)
        self.drop_ops = []      # indices of ops carrying a dropout (seed, offset)
        self.bytes_alloc = 0
        self.init_recs = []     # ops that depend only on frozen parameters (prepared weights), see emit_init
        self.init_arr = None
        self.watch = []         # modules whose _frozen_version gates a re-run of the init ops
        self.seen = None
        self.pinned = set()       # ids of buffers that free() must leave alone: results of the loop-invariant prefix [0, n_const) of a sampling plan
        self.n_const = 0          # ops [0, n_const) depend only on (z, weights): a sampling loop runs them on its first step only (model/graph.py)
        self.dense_grid = set()   # indices of conv records that run a stride-2 convolution in the dense-grid form (4x the algorithmic products: bench.py)
        self._tickets = None    # zeroed uint32 words for the "last block finishes" GroupNorm kernels (one stream runs a plan: shared by all its ops)
        self.wprep_jobs = []    # prepared copies of TRAINABLE conv weights: refreshed by ONE grouped launch at the head of every run (compile)
        self.pre_arr = None
        # second stream (H.OPF_SIDE, emit_side): storage pointers a pending side op reads -> free() parks such buffers until the next join
        self.side_busy = set()
        self.side_parked = []
        self.side_parked_bytes = 0
        self.side_budget = int(os.environ.get("PDAE_SIDE_BUDGET_MB", "6144")) << 20
        self.side_branch_budget = int(os.environ.get("PDAE_SIDE_BRANCH_BUDGET_MB", "24576")) << 20      # parked bytes above which a side() branch joins
        self.side_parked_peak = 0
        self.ws_side_bytes = 0
        self.ws_side = None
        self.ws_patch_side = []
        self.n_side = 0
        self.side_branch_ws = False
        self.side_mode = False    # inside `with plan.side():` every emitted op goes to the second stream (a whole branch of the graph)
        self._spans = []          # sorted (start, end) address ranges of the pool's buffers: which buffer does an op's raw pointer belong to

    def tickets(self, n):
        """>= n + 1 zero-initialised ticket words (pdae_gn_stats_coef / pdae_gn_bwd) with PDAE_GN_TICKETS=1, else None (default).
        The single-launch "last block finalizes" forms are correct (tests) but SLOWER on this path: every block's agent-scope release fence
        writes back the XCD's L2, which is full of the producer's freshly written activations -- measured 80.2 vs 76.5 ms per FFHQ-128 step on
        one box.  Kept as an opt-in for hosts whose tensors are small enough to sit clean in L2."""
        if os.environ.get("PDAE_GN_TICKETS", "0") != "1":
            return None
        # ops of a side-stream branch run beside the main stream's: their ticket words are their own (one shared array raced)
        which = "_tickets_side" if self.side_mode else "_tickets"
        cur = getattr(self, which, None)
        if cur is None or cur.numel() < n + 1:
            cur = torch.zeros(max(n + 1, 257), dtype=torch.int32, device=self.device)
            setattr(self, which, cur)
            self.live.append(cur)
        return cur

    # ---- memory
    def buf(self, *shape, dtype=torch.float32, zero=False):
        n = 1
        for s in shape:
            n *= int(s)
        key = (n, dtype)
        lst = self.pool.get(key)
        if lst:
            t = lst.pop().view(*shape)
        else:
            t = torch.empty(n, dtype=dtype, device=self.device).view(*shape)
            self.live.append(t)
            self.bytes_alloc += n * t.element_size()
            bisect.insort(self._spans, (t.data_ptr(), t.data_ptr() + n * t.element_size()))
        if zero:
            self.emit(H.op_memset(t, t.numel() * t.element_size()))
        return t

    def free(self, *ts):
        for t in ts:
            if t is not None and not isinstance(t, NoFree) and id(t) not in self.pinned:
                if self.side_busy and t.untyped_storage().data_ptr() in self.side_busy:
                    self.side_parked.append(t)                      # a pending side-stream op still reads it: recycled at the next join
                    self.side_parked_bytes += t.numel() * t.element_size()
                    self.side_parked_peak = max(self.side_parked_peak, self.side_parked_bytes)
                    # a whole branch on the second stream parks everything it touches: bounded too (a batch-100 sampling plan kept the
                    # whole shift branch live).  A join inside the branch is legal: the ops behind it fork again from the main stream's position.
                    if self.side_mode and self.side_parked_bytes > self.side_branch_budget:
                        self.join()
                else:
                    self.pool.setdefault((t.numel(), t.dtype), []).append(t)

    # ---- second stream (pdae_hip.h: PDAE_OPF_SIDE / PDAE_OP_JOIN)
    def emit_side(self, op, reads, ws_slot=None, wsb_slot=None, ws_bytes=0):
        """Appends `op` flagged for the executor's second stream.  It starts when everything emitted before it has finished and runs beside
        what is emitted after it, until the next join.  reads: the pooled tensors it reads -- free() parks them until that join, so nothing
        emitted in between can be handed their memory (the write-after-read hazard of a recycled buffer; the plan's buffers are only ever
        rewritten through recycling).  Its workspace is the plan's SECOND workspace.  Nothing emitted before the join may read its outputs."""
        op.flags = H.OPF_SIDE
        self.recs.append(op)
        self.n_side += 1
        for t in reads:
            if t is not None and not isinstance(t, NoFree):
                self.side_busy.add(t.untyped_storage().data_ptr())
        if ws_slot is not None:
            self.ws_side_bytes = max(self.ws_side_bytes, int(ws_bytes))
            self.ws_patch_side.append((len(self.recs) - 1, ws_slot, wsb_slot))
        return len(self.recs) - 1

    def join(self, force=True):
        """The main stream waits for the side ops emitted so far; their parked inputs return to the pool.  force=False: only when the parked
        bytes exceed the budget (PDAE_SIDE_BUDGET_MB)."""
        if not self.side_busy or (not force and self.side_parked_bytes <= self.side_budget):
            return
        self.recs.append(H.op_join())
        self.side_busy.clear()
        parked, self.side_parked, self.side_parked_bytes = self.side_parked, [], 0
        self.free(*parked)

    def need_ws(self, nbytes):
        self.ws_bytes = max(self.ws_bytes, int(nbytes))

    # ---- ops
    def emit(self, op, ws_slot=None, wsb_slot=None):
        """Appends an op record (built eagerly: records hold raw device pointers, the tensors stay alive in
        `self.live` / the parameter store).  ws_slot / wsb_slot: pointer / int slots that receive the shared
        workspace pointer and its size when the plan is compiled."""
        if self.side_mode:
            return self._emit_side_branch(op, ws_slot, wsb_slot)
        self.recs.append(op)
        if ws_slot is not None:
            self.ws_patch.append((len(self.recs) - 1, ws_slot, wsb_slot))
        return len(self.recs) - 1

    def _emit_side_branch(self, op, ws_slot, wsb_slot):
        """An op of a branch that runs on the second stream as a whole (side()): EVERY pool buffer one of its pointers falls into -- inputs and
        outputs -- is parked when freed, until the next join: nothing the other stream allocates in the meantime can alias them."""
        op.flags = H.OPF_SIDE
        self.recs.append(op)
        self.n_side += 1
        for k in range(len(op.p)):
            a = op.p[k]
            if a:
                j = bisect.bisect_right(self._spans, (a, 1 << 62)) - 1
                if j >= 0 and self._spans[j][0] <= a < self._spans[j][1]:
                    self.side_busy.add(self._spans[j][0])
        if ws_slot is not None:
            self.ws_patch_side.append((len(self.recs) - 1, ws_slot, wsb_slot))
            self.side_branch_ws = True                       # (sized like the main workspace at compile: need_ws may still grow)
        return len(self.recs) - 1

    def side(self, on=True):
        """Context: the ops emitted inside run on the executor's second stream, in their order, beside what the main stream is given next.
        For a branch of the graph whose inputs are complete when it starts and whose results are read only behind a join (the caller emits
        it): the shift branch of ShiftUNet beside the frozen trunk's output blocks (model/graph.py)."""
        plan = self

        class _Ctx:
            def __enter__(self_):
                self_.prev = plan.side_mode
                plan.side_mode = bool(on)

            def __exit__(self_, *exc):
                plan.side_mode = self_.prev
        return _Ctx()

    def emit_init(self, op, module):
        """Op that only depends on `module`'s frozen parameters (prepared conv weights): runs once, and again whenever
        module._frozen_version moves (FlatModule.load_state_dict / reset_parameters / frozen_changed)."""
        self.init_recs.append(op)
        if all(m is not module for m in self.watch):
            self.watch.append(module)

    def compile(self):
        self.guard = H.SaturationGuard.get(self.device)       # arms the fp16-window counter of the math-4 kernels on this device
        self.join()                                           # (every pdae_run_ops call joins at its end anyway: this returns the parked buffers)
        self.ws = torch.empty(self.ws_bytes // 4 + 64, dtype=torch.float32, device=self.device)
        if self.ws_patch_side:
            self.ws_side_bytes = max(self.ws_side_bytes, self.ws_bytes if self.side_branch_ws else 0)
            self.ws_side = torch.empty(self.ws_side_bytes // 4 + 64, dtype=torch.float32, device=self.device)
        # the two-stream schedule is PROVEN hazard-free before the plan can run (plancheck.py): no main-stream op between a side op and its join
        # writes what the side op touches or reads what it writes
        if self.n_side and os.environ.get("PDAE_PLAN_CHECK", "1") != "0":
            from .plancheck import check_plan
            self.check = check_plan(self)
        self.arr = H.ops_array(self.recs)
        self.n = len(self.recs)
        self.init_arr = H.ops_array(self.init_recs) if self.init_recs else None
        if self.wprep_jobs:
            jt, ft, tot = H.wprep_group_tables(self.wprep_jobs, self.device)
            self.live.extend([jt, ft])
            self.pre_arr = H.ops_array([H.op_conv_wprep_group(jt, ft, len(self.wprep_jobs), tot)])
        for idx, ws_slot, wsb_slot in self.ws_patch:
            self.arr[idx].p[ws_slot] = self.ws.data_ptr()
            if wsb_slot is not None:
                self.arr[idx].i[wsb_slot] = self.ws_bytes
        for idx, ws_slot, wsb_slot in self.ws_patch_side:
            self.arr[idx].p[ws_slot] = self.ws_side.data_ptr()
            if wsb_slot is not None:
                self.arr[idx].i[wsb_slot] = self.ws_side_bytes
        return self

    def pin(self, *ts):
        for t in ts:
            if t is not None:
                self.pinned.add(id(t))

    def run(self, first=0, last=None, stream=None, prep=True):
        """Runs ops[first:last] on the current (or given) stream.  prep=False: the prepared copies of the trainable weights are current (a
        sampling loop after its first step)."""
        if self.device.type != "cuda":
            raise H.PdaeError("pdae_amd plans only execute on a ROCm device (no CPU fallback)")
        last = self.n if last is None else last
        if self.init_arr is not None:
            cur = tuple(m._frozen_version for m in self.watch)
            if cur != self.seen:
                H.run_ops(self.init_arr, len(self.init_recs), stream)
                self.seen = cur
        if first == 0 and prep and self.pre_arr is not None:    # the trainable weights may have changed since the last run: all prepared copies, one launch
            H.run_ops(self.pre_arr, 1, stream)
        if os.environ.get("PDAE_DEBUG_SYNC"):          # one op at a time, synchronised, index printed first
            import sys
            for k in range(first, last):
                print(f"[pdae] op {k}/{self.n} kind {self.arr[k].kind} i={list(self.arr[k].i)[:17]}", file=sys.stderr, flush=True)
                H.run_ops(self.arr[k], 1, stream)
                torch.cuda.synchronize()
            return
        if last > first:
            import ctypes
            sub = ctypes.cast(ctypes.addressof(self.arr) + first * ctypes.sizeof(H.PdaeOp), ctypes.POINTER(H.PdaeOp * (last - first))).contents
            H.run_ops(sub, last - first, stream)

    def set_dropout(self, seed, step):
        for k, (idx, si, oi) in enumerate(self.drop_ops):
            self.arr[idx].i[si] = int(seed)
            self.arr[idx].i[oi] = int(step)


class NoFree:
    """Marks a persistent buffer handed out where pooled buffers are expected: Plan.free ignores it."""

    def __init__(self, t):
        self.t = t

    def data_ptr(self):
        return self.t.data_ptr()


class Builder:
    """Emits fused stages into a Plan.  `grads` maps parameter name -> gradient tensor (same memory
    layout as the parameter); presence of a name means that parameter is trained by this plan."""

    def __init__(self, plan, params, grads=None, save=False, drop_p=0.0, acc_grads=False, math=None, frozen_of=None):
        self.p = plan
        # optional cheaper arithmetic for the gradient convolutions only (default: same as forward = fp32-grade bf16x6).  bf16x3 keeps
        # ~2^-17 per product -- tighter than the TF32 convolutions of the reference's own cuDNN default -- and is NOT used for any reported number
        bm = os.environ.get("PDAE_BWD_MATH")
        self.bwd_math = H.MATH_NAMES[bm] if bm else None
        # gradient convolutions of the f16x3 mode: fp16 format with a per-tensor power-of-two dY scale from its abs-max (pdae_amax), or ("0")
        # the exact bf16 split
        self.f16_grads = os.environ.get("PDAE_F16_GRADS", "1") != "0"
        self._amax_dy, self._amax_buf, self._amax_until = None, None, -1
        self.fuse_db = os.environ.get("PDAE_FUSE_DB", "1") != "0"      # bias gradients ride in the weight-gradient launch
        self.fuse_skip = os.environ.get("PDAE_FUSE_SKIP", "1") != "0"  # ResBlock skip_connection rides in conv2's K loop (conv_skip)
        self.skip_direct_ratio = float(os.environ.get("PDAE_SKIP_DIRECT_RATIO", "0"))   # > 0: skip / main channel ratio from which a Winograd-eligible conv2 keeps the direct form + fused skip (default: never)
        self.fuse_gn = os.environ.get("PDAE_FUSE_GN", "1") != "0"      # forward-only GN+SiLU+conv3x3 stages run fused (gn_conv)
        # TRAINED in_layers stages (GroupNorm -> SiLU -> conv3x3, no dropout) run the same fused forward and their weight gradient recomputes
        # the activation while it stages X (gn_conv_saved): the activated tensor is never written, saved or re-read
        # Round 6: OFF by default.  With the producer / consumer weight gradient (conv3x3v) the plain launch is 0.858 ms against 1.055 ms for the
        # recomputing one (128^2 256 -> 128), and the step measured 46.5 / 46.7 ms without against 46.9 / 47.0 with it (same box): the form is now a
        # MEMORY switch (-2.5 GiB of plan buffers at B = 32), not the default
        self.fuse_gn_train = os.environ.get("PDAE_FUSE_GN_TRAIN", "0") != "0"
        # ... where it pays: the weight gradient stages (and now maps) X once per 64 output channels, the forward once per 128, while the pass it
        # replaces costs two tensor passes whatever the width -- so the form is taken up to this many output channels (measured)
        self.fuse_gn_train_max_cout = int(os.environ.get("PDAE_FUSE_GN_TRAIN_MAXCOUT", "128"))
        # the two per-(n, c) sums of a GroupNorm backward come out of the epilogue of the data gradient that writes dA (conv_dgrad(gnb=...)) where
        # the kernels build it (Winograd-form 3x3 data gradients, SiLU, no dropout): no reduction pass over (x, dA)
        self.fuse_gn_bwd = os.environ.get("PDAE_FUSE_GN_BWD", "1") != "0"
        # weight gradients on the executor's second stream (Plan.emit_side): beside the GroupNorm-backward / data-gradient chain instead of inside it
        self.side_wgrad = os.environ.get("PDAE_SIDE_WGRAD", "1") != "0"
        # forward 3x3 convolutions leave the GroupNorm partial statistics of their output behind (pdae_conv_stats_arm); the GroupNorm that
        # reads such a tensor takes them instead of a statistics pass over it
        self.fuse_stats = os.environ.get("PDAE_FUSE_GN_STATS", "1") != "0"
        # stride-2 3x3 convolutions (encoder) in the dense-grid form: backward always, forward up to s2_fwd_max output pixels
        self.s2_dense = os.environ.get("PDAE_S2_DENSE", "1") != "0"
        self.s2_fwd_max = int(os.environ.get("PDAE_S2_FWD_M", "8192"))
        self._dyf = None
        self._ystats = {}         # id(tensor) -> (tensor, partial sums, wave-tiles per image)
        self.fuse_attn = os.environ.get("PDAE_FUSE_ATTN", "1") != "0"  # QK^T -> softmax -> PV (and its backward) as one kernel (pdae_attn_fwd / _bwd)
        self.group_wprep = os.environ.get("PDAE_GROUP_WPREP", "1") != "0"   # prepared copies of trainable weights: one grouped launch per run (Plan.pre_arr)
        self.frozen_of = frozen_of      # FlatModule owning `params`: enables the persistent prepared-weight cache for its frozen part
        # prepared copies of FROZEN weights are shared by every plan of the module (the eps-only sampling plan used to hold a second copy
        # of the whole trunk's prepared weights): the buffers live on the module, each plan still registers its own (re)preparation ops
        if frozen_of is not None and not hasattr(frozen_of, "_wp_cache"):
            object.__setattr__(frozen_of, "_wp_cache", {})
        self._frozen_wp = frozen_of._wp_cache if frozen_of is not None else {}
        self._frozen_emitted = set()
        self.acc = int(bool(acc_grads))     # parameter gradients accumulate into (pre-zeroed) buffers
        self.math = H.MATH_NAMES[H.default_math()] if math is None else (H.MATH_NAMES[math] if isinstance(math, str) else int(math))
        self.P = params
        self.Gr = grads or {}
        self._emb = {}            # weight-name prefix -> (y, ctx) of linears already computed by prefetch_emb
        self.save = save          # keep activations for backward (else buffers are recycled)
        self.drop_p = drop_p
        self.drop_layers = 0

    # ------------------------------------------------------------------ fused output statistics
    def _stats_buf(self, c, cs=None):
        """Partial-statistics buffer for the output of forward convolution c (None when it does not run on the 3x3 patch kernels)."""
        if not self.fuse_stats:
            return None, 0
        nbytes, tpi = H.conv_stats_bytes(c, cs)
        if nbytes == 0:
            return None, 0
        part = torch.empty(nbytes // 4, dtype=torch.float32, device=self.p.device)      # lives as long as the plan: any later GroupNorm may read it
        self.p.live.append(part)
        self.p.bytes_alloc += nbytes
        return part, tpi

    def _note_stats(self, y, part, tpi):
        if part is not None:
            self._ystats[id(y)] = (y, part, tpi)

    def stats_pass(self, y, min_elems=1 << 20):
        """One pass that leaves the partial statistics of y [N,H,W,C] in the producers' format (pdae_gn_stats_quads) when y has none -- for a large
        tensor that several GroupNorms read (the stem output: input_blocks.1 and, as the last skip, the final output block of each branch), each of
        which would otherwise run its own statistics pass over it (or over the concat that contains it)."""
        if not self.fuse_stats or self._stats_of(y) is not None or y.numel() < min_elems or self.p.device.type != "cuda":
            return
        N, Hh, W, C = y.shape
        HW = Hh * W
        if (C // GROUPS) % 4 or HW % 128:
            return
        tiles = HW // 128
        part = torch.empty(N * tiles * (C // 4) * 2, dtype=torch.float32, device=self.p.device)
        self.p.live.append(part)
        self.p.bytes_alloc += part.numel() * 4
        self.p.emit(H.op_gn_stats_quads(y, N, HW, C, tiles, part))
        self._note_stats(y, part, tiles)

    def _stats_of(self, x):
        e = self._ystats.get(id(x)) if x is not None else None
        return (e[1], e[2]) if (e is not None and e[0] is x) else None        # identity check: a recycled buffer is a new tensor object

    def _gn_stats_coef(self, x0, C0, x1, C1, N, HW, gamma, beta, ss, zss, mean, rstd, coef):
        """GroupNorm statistics + coefficients of [x0 | x1]: from the partial sums of the producing convolutions when every source has them."""
        pl = self.p
        s0, s1 = self._stats_of(x0), self._stats_of(x1)
        C = C0 + C1
        if s0 is not None and (x1 is None or s1 is not None) and (C // GROUPS) % 4 == 0 and C0 % 4 == 0:
            pl.emit(H.op_gn_coef_from_conv_stats(N, HW, C0, C1, GROUPS, GN_EPS, s0[0], s0[1], s1[0] if s1 else None, s1[1] if s1 else 0,
                                                 gamma, beta, ss, zss, mean, rstd, coef))
            return
        pl.need_ws(H.gn_ws_bytes(N, C))
        pl.emit(H.op_gn_stats_coef(x0, C0, x1, C1, N, HW, GROUPS, GN_EPS, gamma, beta, ss, zss, mean, rstd, coef, None, ticket=pl.tickets(N)), ws_slot=9)

    # ------------------------------------------------------------------ primitives
    def conv(self, x0, x1, wname, k, stride=1, up=False, res=None, res_mode=0, bias=True):
        N, Hh, W, C0 = x0.shape
        C1 = 0 if x1 is None else x1.shape[3]
        w = self.P[wname + ".weight"]
        b = self.P[wname + ".bias"] if bias else None
        c = H.Conv(N, Hh, W, C0, C1, w.shape[0], k=k, stride=stride, up=up, math=self.math)
        assert w.numel() == c.Cout * k * k * c.Cin, (wname, tuple(w.shape), c.Cin)
        c1 = self._dense_grid_desc(c)
        if c1 is not None and res is None and N * c.Ho * c.Wo <= self.s2_fwd_max:
            # small stride-2 layers: the stride-1 convolution on the patch kernel (split-K fills the chip) + the even grid of its output
            # -- the generic implicit GEMM runs them as <= 128 workgroups with a serial K loop of 2304 (130 us for 0.6 GFLOP)
            yf = self.p.buf(N, Hh, W, c.Cout)
            wp = self._wprep(c1, w, 0)
            self.p.dense_grid.add(len(self.p.recs))
            self.p.emit(H.op_conv_fwd(c1, x0, None, w, b, yf, wp=wp))
            if wp is not None:
                self.p.free(wp)
            y = self.p.buf(N, c.Ho, c.Wo, c.Cout)
            self.p.emit(H.op_subsample2(yf, N, Hh, W, c.Cout, y))
            self.p.free(yf)
            return y, NS(c=c, x0=x0, x1=x1, wname=wname, y=y, c1=c1)
        y = self.p.buf(N, c.Ho, c.Wo, c.Cout)
        wp = self._wprep(c, w, 0)
        part, tpi = self._stats_buf(c) if (wp is not None and k == 3) else (None, 0)
        self.p.emit(H.op_conv_fwd(c, x0, x1, w, b, y, res=res, res_mode=res_mode, wp=wp, stats=part))
        self._note_stats(y, part, tpi)
        if wp is not None:
            self.p.free(wp)
        return y, NS(c=c, x0=x0, x1=x1, wname=wname, y=y, c1=c1)

    def _dense_grid_desc(self, c):
        """Stride-1 descriptor of a stride-2 3x3 convolution whose backward (and, for small layers, forward) runs in the dense-grid form:
        the same weights on the stride-1 patch / weight-gradient kernels, dY scattered onto the even grid of a zero tensor
        (pdae_zero_insert2) -- four times the products, on kernels three to six times faster than the generic implicit GEMM with its
        dilated (75 % zero) gather.  None when the layer is not eligible (PDAE_S2_DENSE=0: never)."""
        if not self.s2_dense or c.stride != 2 or c.KH != 3 or c.C1 != 0 or c.up or (c.Hi & 1) or (c.Wi & 1) or (c.Cout & 3):
            return None
        c1 = H.Conv(c.N, c.Hi, c.Wi, c.C0, 0, c.Cout, k=3, stride=1, math=c.math)
        if c1.wprep_bytes(0) == 0 or c1.wprep_bytes(1) == 0:
            return None
        return c1

    def _dense_dy(self, cx, dy):
        """dY of stride-2 conv cx on the stride-1 grid: shared by its weight- and data-gradient launches when they are emitted back to back (the
        copy is valid only while no other record has been emitted since conv_bwd_params returned: pooled buffers are recycled OBJECTS, identity
        of `dy` alone would also match a later tensor), freed by the data gradient or by the next request."""
        if self._dyf is not None and self._dyf[0] is dy and self._dyf[2] == len(self.p.recs):
            return self._dyf[1]
        if self._dyf is not None:
            self.p.free(self._dyf[1])
        c = cx.c
        dyf = self.p.buf(c.N, c.Hi, c.Wi, c.Cout)
        self.p.emit(H.op_zero_insert2(dy, c.N, c.Ho, c.Wo, c.Cout, dyf))
        self._dyf = (dy, dyf, -1)
        return dyf

    def _wprep(self, c, w, transposed, gn=False, f16_grad=False):
        """Fragment-ordered bf16 planes of w for the patch kernel (None when the conv is not eligible).  Refreshed right before
        every use -- the weights change each optimizer step and the copy costs ~10 bytes per parameter, noise next to the
        convolution itself -- so no cache has to be kept coherent with the optimizer."""
        nbytes = c.wprep_bytes(transposed, gn=gn, f16_grad=f16_grad)
        if nbytes == 0:
            return None
        transposed = int(transposed) | (4 if gn else 0) | (16 if f16_grad else 0)          # PDAE_WPREP_* flags from here on
        if self.frozen_of is not None and self.frozen_of.is_frozen_storage(w):
            # frozen weights (the pre-trained trunk of ShiftUNet: never touched by the optimizer / EMA kernels) are prepared once
            # per plan into a persistent buffer; Plan.run refreshes them when the module reports a parameter (re)load
            # (the byte count is part of the key: the prepared LAYOUT -- and with it the size -- follows the form knobs PDAE_W1 / W1_EFF / ROWS8,
            # which may change between two plan builds of one module; a buffer sized for another form must never be reused: ADVICE r5)
            key = (w.data_ptr(), tuple(c.fields()), int(transposed), int(nbytes))
            wp = self._frozen_wp.get(key)
            if wp is None or wp.device != self.p.device or wp.numel() * 4 < nbytes:
                wp = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.p.device)
                self.p.bytes_alloc += wp.numel() * 4
                self._frozen_wp[key] = wp
            if key not in self._frozen_emitted:
                self.p.live.append(wp)
                self.p.emit_init(H.op_conv_wprep(c, w, transposed, wp), self.frozen_of)
                self._frozen_emitted.add(key)
            return NoFree(wp)
        if self.group_wprep and self.p.device.type == "cuda":
            # persistent copy, refreshed at the head of every plan run together with all others (they change once per optimizer step)
            wp = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.p.device)
            self.p.live.append(wp)
            self.p.bytes_alloc += wp.numel() * 4
            self.p.wprep_jobs.append(H.wprep_job(c, w, transposed, wp))
            return NoFree(wp)
        wp = self.p.buf((nbytes + 3) // 4)
        self.p.emit(H.op_conv_wprep(c, w, transposed, wp))
        return wp

    def amax_ok(self, c):
        """True when conv c's gradient kernels take a dY abs-max (fp16 format)."""
        c = self._bwd_desc(c)
        return bool(self.f16_grads and c.math == H.MATH_NAMES["f16x3"] and c.stride == 1 and ((c.KH == 3 and (c.C1 == 0 or getattr(c, "gn_in", False))) or c.KH == 1))

    def dy_amax(self, c, dy):
        """Device scalar max|dy| for the fp16-format gradient kernels of conv c (None: they run the exact bf16 split).  wgrad and dgrad of
        one layer are emitted back to back on the same dy: the scalar is computed once and shared."""
        if not self.f16_grads or c.math != H.MATH_NAMES["f16x3"] or c.stride != 1:
            return None
        if not ((c.KH == 3 and (c.C1 == 0 or getattr(c, "gn_in", False))) or c.KH == 1):
            return None
        if self._amax_dy is dy and self._amax_until == len(self.p.recs):
            return self._amax_buf
        buf = self.p.buf(4)
        self.p.emit(H.op_amax(dy, dy.numel(), buf))
        if self._amax_buf is not None:
            self.p.free(self._amax_buf)
        self._amax_dy, self._amax_buf = dy, buf
        self._amax_until = len(self.p.recs)
        return buf

    def _bwd_desc(self, c):
        """Descriptor of the backward launches of conv c: same geometry, arithmetic mode `bwd_math` when it is set (PDAE_BWD_MATH)."""
        if self.bwd_math is None or self.bwd_math == c.math:
            return c
        b = copy.copy(c)
        b.math = self.bwd_math
        return b

    def conv_bwd_params(self, cx, dy, amax=None):
        """dW, db of a conv stage (only if the parameter is trained by this plan)."""
        c = self._bwd_desc(cx.c)
        gw = self.Gr.get(cx.wname + ".weight")
        gb = self.Gr.get(cx.wname + ".bias")
        dense = getattr(cx, "c1", None) is not None and gw is not None
        if dense:
            c = self._bwd_desc(cx.c1)
            amax = self.dy_amax(c, dy) if amax is None else amax            # max|dY| of the small tensor: the zeros do not change it
            dy_small, dy = dy, self._dense_dy(cx, dy)
            self.p.dense_grid.add(len(self.p.recs))
        if gw is not None:
            wsb = c.wgrad_ws_bytes()
            self.p.need_ws(wsb)
            # the bias gradient rides along: the 3x3 kernel sums dY while staging it, the other paths run the column sum themselves
            ride = self.fuse_db
            am = (amax if (amax is not None and self.amax_ok(c)) else self.dy_amax(c, dy))       # 3x3 and 1x1 weight-gradient kernels: fp16 format with dy_amax
            gn = getattr(cx, "gn", None)                     # (coef, act): x0 / x1 are the RAW sources of a fused-GroupNorm forward (gn_conv_saved)
            wop = H.op_conv_wgrad(c, cx.x0, cx.x1, dy, gw, None, 0, accumulate=self.acc, db=gb if ride else None, dy_amax=am,
                                  gn_coef=gn[0] if gn else None, gn_act=gn[1] if gn else 0)
            if self.side_wgrad:
                # the weight gradient feeds nothing in the backward chain: second stream, beside the GroupNorm-backward / data-gradient ops
                # emitted next (Plan.emit_side).  It reads x (a saved forward tensor, never recycled), dy and the abs-max scalar.
                self.p.emit_side(wop, [cx.x0, cx.x1, dy, am, gn[0] if gn else None], ws_slot=4, wsb_slot=len(c.fields()) + 1, ws_bytes=wsb)
                self.p.join(force=False)
            else:
                self.p.emit(wop, ws_slot=4, wsb_slot=len(c.fields()) + 1)
            if am is not None:
                self._amax_until = len(self.p.recs)
            if ride:
                gb = None
        if gb is not None:
            M = c.N * c.Ho * c.Wo
            self.p.need_ws(H.colsum_ws_bytes(M, c.Cout))
            self.p.emit(H.op_colsum(dy, M, c.Cout, gb, None, acc=self.acc), ws_slot=2)
        if dense:
            self._dyf = (dy_small, dy, len(self.p.recs))                    # conv_dgrad may take it if it is the very next thing emitted

    def conv_dgrad(self, cx, dy, ci_off=0, ci_cnt=None, out=None, accumulate=0, amax=None, gnb=None):
        """gnb: ctx of the GroupNorm whose output this convolution read (same resolution): when the launch can, it also leaves that GroupNorm's
        backward sums (gnb.parts = (buffer, tiles per image), consumed by gn_bwd)."""
        c = self._bwd_desc(cx.c)
        ci_cnt = c.Cin if ci_cnt is None else ci_cnt
        w = self.P[cx.wname + ".weight"]
        dx = out if out is not None else self.p.buf(c.N, c.Hl, c.Wl, ci_cnt)
        whole = ci_off == 0 and ci_cnt == c.Cin
        if getattr(cx, "c1", None) is not None and whole:
            c = self._bwd_desc(cx.c1)
            dyf = self._dense_dy(cx, dy)                                    # first: the shared copy is valid only while nothing else has been emitted
            am = self.dy_amax(c, dy) if amax is None else amax
            wp_t = self._wprep(c, w, 1, f16_grad=am is not None)
            self.p.dense_grid.add(len(self.p.recs))
            self.p.emit(H.op_conv_dgrad(c, dyf, w, dx, accumulate=accumulate, wp_t=wp_t, dy_amax=am))
            self.p.free(wp_t, dyf)
            self._dyf = None
            return dx
        am = (amax if (amax is not None and self.amax_ok(c)) else self.dy_amax(c, dy)) if (whole or c.KH == 1) else None
        wp_t = self._wprep(c, w, 1, f16_grad=am is not None) if (whole or (c.KH == 1 and ci_off % 32 == 0 and ci_cnt % 4 == 0)) else None
        if wp_t is None:
            am = None
        gn_arg = None
        if (gnb is not None and self.fuse_gn_bwd and wp_t is not None and whole and not accumulate and c.KH == 3 and gnb.act == 1 and gnb.mode == 0
                and gnb.drop_p == 0 and gnb.C0 + gnb.C1 == c.Cin):
            cg = c if (gnb.C0, gnb.C1) == (c.C0, c.C1) else H.Conv(c.N, c.Hi, c.Wi, gnb.C0, gnb.C1, c.Cout, k=3, up=c.up, math=c.math)
            nbytes, tiles = H.conv_gnbwd_bytes(cg, f16_grad=am is not None)
            if nbytes:
                part = self.p.buf(nbytes // 4)
                gn_arg = (gnb.x0, gnb.C0, gnb.x1, gnb.C1, gnb.coef, part)
                gnb.parts = (part, tiles)
                c = cg
        self.p.emit(H.op_conv_dgrad(c, dy, w, dx, ci_off=ci_off, ci_cnt=ci_cnt, accumulate=accumulate, wp_t=wp_t, dy_amax=am, gnb=gn_arg))
        if wp_t is not None:
            self.p.free(wp_t)
        return dx

    def linear(self, x, wname, pre_bias=True):
        Nb, K = x.shape
        w, b = self.P[wname + ".weight"], self.P[wname + ".bias"]
        out = w.shape[0]
        assert w.numel() == out * K, (wname, tuple(w.shape), K)
        y = self.p.buf(Nb, out)
        self.p.emit(H.op_gemm(0, 1, Nb, out, K, x, K, w, K, y, out, bias=b))
        return y, NS(x=x, wname=wname, Nb=Nb, K=K, out=out)

    def linear_group(self, jobs):
        """jobs = [(x [Nb,K], weight-name prefix)]: every  y = x W^T + b  in ONE launch (pdae_linear_group).  Returns [(y, ctx)] with the same
        ctx objects linear() produces, so linear_bwd works unchanged.  Falls back to single launches when the shapes do not fit the kernel."""
        if not jobs:
            return []
        Nb, K = jobs[0][0].shape
        fits = K % 8 == 0 and all(x.shape == (Nb, K) for x, _ in jobs) and os.environ.get("PDAE_GROUP_LINEAR", "1") != "0"
        if not fits:
            return [self.linear(x, w) for x, w in jobs]
        out, items = [], []
        for x, wname in jobs:
            w, b = self.P[wname + ".weight"], self.P[wname + ".bias"]
            assert w.numel() == w.shape[0] * K, (wname, tuple(w.shape), K)
            y = self.p.buf(Nb, w.shape[0])
            if Nb <= 32:
                items.append((x, w, b, y))
            else:                                        # sampling batches (100, 128): row slices of 32 as items of the same launch
                for r0 in range(0, Nb, 32):
                    items.append((x[r0:r0 + 32], w, b, y[r0:r0 + 32], min(32, Nb - r0)))
            out.append((y, NS(x=x, wname=wname, Nb=Nb, K=K, out=w.shape[0])))
        it, first, total = H.linear_group_tables(items, self.p.device)
        self.p.live.extend([it, first])                  # the device tables live as long as the plan
        gop = H.op_linear_group(it, first, len(items), total, min(Nb, 32), K)
        gop.rw = ([t for i_ in items for t in i_[:3]], [i_[3] for i_ in items])      # operands live in the device table: plancheck reads them here
        self.p.emit(gop)
        return out

    def prefetch_emb(self, ea, prefixes, eza=None, z_prefixes=()):
        """All emb_layers (input SiLU(emb)) and emb_z_layers (input SiLU(shift_emb)) Linear layers of the listed ResBlocks in one launch;
        resblock() picks its (scale, shift) vectors up from here."""
        jobs = [(ea, p + ".emb_layers.1") for p in prefixes] + [(eza, p + ".emb_z_layers.1") for p in z_prefixes]
        res = self.linear_group(jobs)
        for (x, wname), r in zip(jobs, res):
            self._emb[wname] = r

    def linear_bwd_group(self, jobs):
        """jobs = [(linear ctx, dy, dx or None, dx_acc)]: the backward of several M <= 32 linear layers (dW, db, optional dx) in ONE launch
        (pdae_linear_bwd_group) instead of 2-3 launches each.  At most one job may carry a given dx."""
        jobs = [j for j in jobs if j is not None]
        if not jobs:
            return
        Nb, K = jobs[0][0].Nb, jobs[0][0].K
        dxs = [id(dx) for _, _, dx, _ in jobs if dx is not None]
        fits = (Nb <= 32 and K % 4 == 0 and all(lx.Nb == Nb and lx.K == K for lx, _, _, _ in jobs) and len(dxs) == len(set(dxs))
                and os.environ.get("PDAE_GROUP_LINEAR_BWD", "1") != "0")
        items = []
        for lx, dy, dx, dx_acc in jobs:
            gw, gb = self.Gr.get(lx.wname + ".weight"), self.Gr.get(lx.wname + ".bias")
            if not fits or gw is None or gb is None:
                fits = False
                break
            items.append((lx.x, dy, self.P[lx.wname + ".weight"], gw, gb, dx, self.acc, dx_acc))
        if not fits:
            for lx, dy, dx, dx_acc in jobs:
                self.linear_bwd(lx, dy, dx=dx, dx_acc=dx_acc)
            return
        it, first, total = H.linear_bwd_group_tables(items, Nb, self.p.device)
        self.p.live.extend([it, first])
        gop = H.op_linear_bwd_group(it, first, len(items), total, Nb, K)
        gop.rw = ([t for i_ in items for t in i_[:3]], [t for i_ in items for t in i_[3:6]])
        self.p.emit(gop)

    def linear_bwd(self, lx, dy, dx=None, dx_acc=0):
        """dW = dy^T x, db = colsum(dy), optionally dx (+)= dy W."""
        Nb, K, out = lx.Nb, lx.K, lx.out
        w = self.P[lx.wname + ".weight"]
        gw, gb = self.Gr.get(lx.wname + ".weight"), self.Gr.get(lx.wname + ".bias")
        if gw is not None:
            self.p.emit(H.op_gemm(1, 0, out, K, Nb, dy, out, lx.x, K, gw, K, accumulate=self.acc))
        if gb is not None:
            self.p.need_ws(H.colsum_ws_bytes(Nb, out))
            self.p.emit(H.op_colsum(dy, Nb, out, gb, None, acc=self.acc), ws_slot=2)
        if dx is not None:
            self.p.emit(H.op_gemm(0, 0, Nb, K, out, dy, out, w, K, dx, K, accumulate=dx_acc))

    def silu(self, x):
        y = self.p.buf(*x.shape)
        self.p.emit(H.op_silu(x, y, x.numel()))
        return y

    def _skip_parts(self, c, skip, gn=False):
        """(cs, s0, s1, wps, bias_s, ctx) of a fusable 1x1 skip convolution riding on the 3x3 conv c, or None."""
        if skip is None or not self.fuse_skip:
            return None
        s0, s1, sname = skip
        ws, bs = self.P[sname + ".weight"], self.P[sname + ".bias"]
        cs = H.Conv(c.N, c.Ho, c.Wo, s0.shape[3], 0 if s1 is None else s1.shape[3], c.Cout, k=1, math=self.math)
        # A convolution in the Winograd-along-x form takes no skip chunks: the skip convolution then runs as its own 1x1 launch and enters as the
        # residual.  PDAE_SKIP_DIRECT_RATIO = r pins the direct form + fused skip (PDAE_MATH_DIRECT in the descriptor, BEFORE the main weights are
        # prepared) where the skip input has >= r x the main input's channels; measured on the FFHQ-128 step (one box): never 51.42 ms, r = 2
        # 51.93, r = 1.4 52.07, always 52.07 -- so the default is never.
        pinned = False
        if self.skip_direct_ratio > 0 and c.winograd_form(0, gn=gn) and cs.Cin >= self.skip_direct_ratio * c.Cin:
            c.direct = pinned = True
        nbytes = H.conv_skip_wprep_bytes(c, cs)
        if nbytes == 0:
            if pinned:
                c.direct = False                 # (a caller that keeps using c must not lose the Winograd form for nothing)
            return None
        if self.frozen_of is not None and self.frozen_of.is_frozen_storage(ws):
            key = (ws.data_ptr(), tuple(c.fields()), "skip", int(nbytes))
            wps = self._frozen_wp.get(key)
            if wps is None or wps.device != self.p.device or wps.numel() * 4 < nbytes:
                wps = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.p.device)
                self.p.bytes_alloc += wps.numel() * 4
                self._frozen_wp[key] = wps
            if key not in self._frozen_emitted:
                self.p.live.append(wps)
                self.p.emit_init(H.op_conv_skip_wprep(c, cs, ws, wps), self.frozen_of)
                self._frozen_emitted.add(key)
            wps = NoFree(wps)
        elif self.group_wprep and self.p.device.type == "cuda":
            wps = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.p.device)
            self.p.live.append(wps)
            self.p.bytes_alloc += wps.numel() * 4
            self.p.wprep_jobs.append(H.skip_wprep_job(c, cs, ws, wps))
            wps = NoFree(wps)
        else:
            wps = self.p.buf((nbytes + 3) // 4)
            self.p.emit(H.op_conv_skip_wprep(c, cs, ws, wps))
        return cs, s0, s1, wps, bs, NS(c=cs, x0=s0, x1=s1, wname=sname, y=None)

    def conv_skip(self, x, wname, skip):
        """3x3 conv of x plus the fused 1x1 skip convolution of skip = (s0, s1, name) (pdae_conv2d_fwd_skip); None if not eligible."""
        N, Hh, W, C0 = x.shape
        w, b = self.P[wname + ".weight"], self.P[wname + ".bias"]
        c = H.Conv(N, Hh, W, C0, 0, w.shape[0], k=3, math=self.math)
        if c.wprep_bytes(0) == 0:
            return None
        sp = self._skip_parts(c, skip)
        if sp is None:
            return None
        cs, s0, s1, wps, bs, cs_ctx = sp
        wp = self._wprep(c, w, 0)
        y = self.p.buf(N, c.Ho, c.Wo, c.Cout)
        part, tpi = self._stats_buf(c, cs)
        self.p.emit(H.op_conv_fwd_skip(c, x, None, None, 0, wp, b, cs, s0, s1, wps, bs, y, stats=part))
        self._note_stats(y, part, tpi)
        self.p.free(wp, wps)
        return y, NS(c=c, x0=x, x1=None, wname=wname, y=y), cs_ctx

    def gn_conv(self, x0, x1, gname, wname, ss=None, zss=None, up=False, res=None, res_mode=0, skip=None):
        """GroupNorm [+AdaGN] + SiLU + 3x3 conv with the normalisation applied inside the conv's patch staging (pdae_conv2d_fwd_gn): the
        activated tensor is never written.  Forward-only stages (nothing saved for a backward); returns None when the conv is not
        eligible for the fused kernel."""
        if self.save or not self.fuse_gn:
            return None
        N, Hh, W, C0 = x0.shape
        C1 = 0 if x1 is None else x1.shape[3]
        C = C0 + C1
        w, b = self.P[wname + ".weight"], self.P[wname + ".bias"]
        c = H.Conv(N, Hh, W, C0, C1, w.shape[0], k=3, up=up, math=self.math)
        if c.wprep_bytes(0, gn=True) == 0:
            return None
        sp = self._skip_parts(c, skip, gn=True)
        if skip is not None and sp is None:
            return None
        pl = self.p
        gamma, beta = self.P[gname + ".weight"], self.P[gname + ".bias"]
        mean, rstd, coef = pl.buf(N * GROUPS), pl.buf(N * GROUPS), pl.buf(3, N, C)
        self._gn_stats_coef(x0, C0, x1, C1, N, Hh * W, gamma, beta, ss, zss, mean, rstd, coef)
        wp = self._wprep(c, w, 0, gn=True)
        y = pl.buf(N, c.Ho, c.Wo, c.Cout)
        if sp is not None:
            cs, s0, s1, wps, bs, _ = sp
            part, tpi = self._stats_buf(c, cs)
            pl.emit(H.op_conv_fwd_skip(c, x0, x1, coef, 1, wp, b, cs, s0, s1, wps, bs, y, stats=part))
            pl.free(wps)
        else:
            part, tpi = self._stats_buf(c)
            pl.emit(H.op_conv_fwd_gn(c, x0, x1, coef, 1, wp, b, y, res=res, res_mode=res_mode, stats=part))
        self._note_stats(y, part, tpi)
        pl.free(mean, rstd, coef, wp)
        return y

    def gn_conv_saved(self, x0, x1, gname, wname, up=False):
        """Training form of GroupNorm + SiLU + 3x3 conv (module.py:241-242, 279-284: in_layers; no AdaGN, no dropout): the forward is the fused
        launch of gn_conv, and the weight gradient recomputes act(a (x - mu) + b) on the raw two-source input while it stages X
        (pdae_conv_gn_input_arm), so the activated tensor -- one write in gn_apply, one read each in the forward conv and the weight gradient, and its
        slot among the saved activations -- does not exist.  Returns (y, GroupNorm ctx for gn_bwd, conv ctx) or None when the kernels do not
        take the shape (the materialised form runs)."""
        if not (self.save and self.fuse_gn and self.fuse_gn_train):
            return None
        N, Hh, W, C0 = x0.shape
        C1 = 0 if x1 is None else x1.shape[3]
        C = C0 + C1
        w, b = self.P[wname + ".weight"], self.P[wname + ".bias"]
        c = H.Conv(N, Hh, W, C0, C1, w.shape[0], k=3, up=up, math=self.math)
        c.gn_in = True
        if (wname + ".weight") not in self.Gr or c.Cout > self.fuse_gn_train_max_cout or c.wprep_bytes(0, gn=True) == 0 or not H.conv_wgrad_gn_ok(self._bwd_desc(c)):
            return None
        pl = self.p
        gamma, beta = self.P[gname + ".weight"], self.P[gname + ".bias"]
        mean, rstd, coef = pl.buf(N * GROUPS), pl.buf(N * GROUPS), pl.buf(3, N, C)
        self._gn_stats_coef(x0, C0, x1, C1, N, Hh * W, gamma, beta, None, None, mean, rstd, coef)
        wp = self._wprep(c, w, 0, gn=True)
        y = pl.buf(N, c.Ho, c.Wo, c.Cout)
        part, tpi = self._stats_buf(c)
        pl.emit(H.op_conv_fwd_gn(c, x0, x1, coef, 1, wp, b, y, stats=part))
        self._note_stats(y, part, tpi)
        pl.free(wp)
        g = NS(x0=x0, x1=x1, C0=C0, C1=C1, N=N, H=Hh, W=W, gname=gname, ss=None, zss=None, act=1, mode=0, mean=mean, rstd=rstd, coef=coef, y=None,
               xpool=None, drop_p=0.0, layer=0)
        return y, g, NS(c=c, x0=x0, x1=x1, wname=wname, y=y, c1=None, gn=(coef, 1))

    def gn(self, x0, x1, gname, ss=None, zss=None, act=1, mode=0, want_xpool=False, dropout=False):
        """GroupNorm(32) [+AdaGN] [+SiLU] [+dropout] [+2x2 avg-pool].  Returns ctx with .y (.xpool)."""
        N, Hh, W, C0 = x0.shape
        C1 = 0 if x1 is None else x1.shape[3]
        C = C0 + C1
        gamma, beta = self.P[gname + ".weight"], self.P[gname + ".bias"]
        pl = self.p
        mean, rstd, coef = pl.buf(N * GROUPS), pl.buf(N * GROUPS), pl.buf(3, N, C)
        Ho, Wo = (Hh // 2, W // 2) if mode == 1 else (Hh, W)
        y = pl.buf(N, Ho, Wo, C)
        xpool = pl.buf(N, Ho, Wo, C) if (mode == 1 and want_xpool) else None
        self._gn_stats_coef(x0, C0, x1, C1, N, Hh * W, gamma, beta, ss, zss, mean, rstd, coef)
        dp = self.drop_p if dropout else 0.0
        layer = 0
        if dp > 0:
            self.drop_layers += 1
            layer = self.drop_layers
        idx = pl.emit(H.op_gn_apply(x0, C0, x1, C1, N, Hh, W, coef, act, mode, y, xpool=xpool, drop_p=dp, seed=layer, offset=0))
        if dp > 0:
            pl.drop_ops.append((idx, 7, 8))
        ctx = NS(x0=x0, x1=x1, C0=C0, C1=C1, N=N, H=Hh, W=W, gname=gname, ss=ss, zss=zss, act=act, mode=mode, mean=mean, rstd=rstd,
                 coef=coef, y=y, xpool=xpool, drop_p=dp, layer=layer)
        if not self.save:
            pl.free(mean, rstd, coef)
        return ctx

    def gn_bwd(self, g, dA, bmode, add=None, dx0=None, acc0=0, dx1=None, acc1=0, want_dss=False, want_dzss=False, dx0_amax=None):
        """Backward of a gn() stage.  bmode: 0 same, 1 y was pooled, 2 consumer read y upsampled."""
        pl = self.p
        C = g.C0 + g.C1
        gamma, beta = self.P[g.gname + ".weight"], self.P[g.gname + ".bias"]
        dgamma, dbeta = self.Gr.get(g.gname + ".weight"), self.Gr.get(g.gname + ".bias")
        dss = pl.buf(g.N, 2 * C) if (want_dss and g.ss is not None) else None
        dzss = pl.buf(g.N, 2 * C) if (want_dzss and g.zss is not None) else None
        pl.need_ws(H.gn_ws_bytes(g.N, C))
        parts = getattr(g, "parts", None)                # left by the data gradient that wrote dA (conv_dgrad(gnb=g)); one use
        if parts is not None:
            assert bmode == 0 and g.drop_p == 0
            g.parts = None
        idx = pl.emit(H.op_gn_bwd(g.x0, g.C0, g.x1, g.C1, g.N, g.H, g.W, GROUPS, g.coef, g.rstd, gamma, beta, g.ss, g.zss, dA, g.act,
                                  bmode, None, add=add, dx0=dx0, acc0=acc0, dx1=dx1, acc1=acc1, dgamma=dgamma, dbeta=dbeta, acc_param=self.acc, dss=dss,
                                  dzss=dzss, drop_p=g.drop_p, seed=g.layer, offset=0, dx0_amax=dx0_amax, ticket=None if parts else pl.tickets(g.N),
                                  parts=parts[0] if parts else None, parts_tiles=parts[1] if parts else 0), ws_slot=16)
        if parts is not None:
            pl.free(parts[0])
        if g.drop_p > 0:
            pl.drop_ops.append((idx, 11, 12))
        return dss, dzss

    # ------------------------------------------------------------------ ResBlock / ResBlockShift
    def resblock(self, pre, x0, x1, ea, eza=None, up=False, down=False, dropout=False):
        """model/module.py:278-297 / :361-384.  ea = SiLU(emb), eza = SiLU(shift_emb) (shared by all blocks)."""
        pl = self.p
        has_skip = (pre + ".skip_connection.weight") in self.P
        assert not (has_skip and (up or down)), "channel-changing up/down ResBlocks do not occur on this path"
        g1 = c1 = None
        h1 = None if down else self.gn_conv(x0, x1, pre + ".in_layers.0", pre + ".in_layers.2", up=up)       # fused when forward-only
        if h1 is None and not down:
            r = self.gn_conv_saved(x0, x1, pre + ".in_layers.0", pre + ".in_layers.2", up=up)                # fused in training as well
            if r is not None:
                h1, g1, c1 = r
        if h1 is None:
            g1 = self.gn(x0, x1, pre + ".in_layers.0", act=1, mode=1 if down else 0, want_xpool=down)
            h1, c1 = self.conv(g1.y, None, pre + ".in_layers.2", 3, up=up)
            if not self.save:
                pl.free(g1.y)
        ss, l_ss = self._emb.pop(pre + ".emb_layers.1", None) or self.linear(ea, pre + ".emb_layers.1")
        zss, l_zss = (None, None)
        if eza is not None:
            zss, l_zss = self._emb.pop(pre + ".emb_z_layers.1", None) or self.linear(eza, pre + ".emb_z_layers.1")
        cs = g2 = c2 = out = None
        skip = (x0, x1, pre + ".skip_connection") if has_skip else None
        res, res_mode = None, 0
        if down:
            res, res_mode = g1.xpool, 1
        elif not has_skip:
            res, res_mode = x0, (2 if up else 1)
        plain_drop = dropout and self.drop_p > 0
        # 1) forward-only: GN + SiLU + conv2 (+ skip conv) in one launch
        if not plain_drop:
            out = self.gn_conv(h1, None, pre + ".out_layers.0", pre + ".out_layers.3", ss=ss, zss=zss, res=res, res_mode=res_mode, skip=skip)
            if out is None and has_skip and not self.save:
                # skip not fusable: fused GN with the separately computed skip as residual
                res, cs = self.conv(x0, x1, pre + ".skip_connection", 1)
                res_mode, skip = 1, None
                out = self.gn_conv(h1, None, pre + ".out_layers.0", pre + ".out_layers.3", ss=ss, zss=zss, res=res, res_mode=1)
        if out is None:
            g2 = self.gn(h1, None, pre + ".out_layers.0", ss=ss, zss=zss, act=1, mode=0, dropout=dropout)
            r = self.conv_skip(g2.y, pre + ".out_layers.3", skip) if skip is not None else None
            if r is not None:                         # 2) materialised GN output, skip conv fused into conv2
                out, c2, cs = r
            else:                                     # 3) separate kernels
                if skip is not None:
                    res, cs = self.conv(x0, x1, pre + ".skip_connection", 1)
                    res_mode = 1
                out, c2 = self.conv(g2.y, None, pre + ".out_layers.3", 3, res=res, res_mode=res_mode)
            if not self.save:
                pl.free(g2.y)
        if not self.save:
            pl.free(h1, ss, zss)
        if has_skip and res is not None:
            pl.free(res)                     # the separately computed skip tensor is never needed by backward
        elif down:
            pl.free(g1.xpool)
        return out, NS(pre=pre, g1=g1, c1=c1, l_ss=l_ss, l_zss=l_zss, g2=g2, c2=c2, cs=cs, up=up, down=down, has_skip=has_skip, h1=h1)

    def resblock_bwd(self, r, dout, need_dx0=True, need_dx1=False, d_ea=None, d_eza=None, dout_amax=None, out_amax=False):
        """Returns (dx0, dx1, amax0).  d_ea / d_eza: accumulators [N,E] for d SiLU(emb) / d SiLU(shift_emb) (or None).
        dout_amax: device scalar max|dout| when the producer of dout already knows it (saves the separate pdae_amax pass);
        out_amax: also return max|dx0| -- it falls out of the final GroupNorm-backward apply pass -- for the consumer of dx0."""
        pl = self.p
        g1, g2 = r.g1, r.g2
        C0, C1 = g1.C0, g1.C1
        # conv2
        self.conv_bwd_params(r.c2, dout, amax=dout_amax)
        d_a2 = self.conv_dgrad(r.c2, dout, amax=dout_amax, gnb=g2)
        # channel-changing skip: 1x1 conv over the raw (concat) input
        dx0 = dx1 = None
        if r.has_skip:
            self.conv_bwd_params(r.cs, dout, amax=dout_amax)          # same dout as conv2: its abs-max is already known
            if need_dx0:
                dx0 = self.conv_dgrad(r.cs, dout, ci_off=0, ci_cnt=C0, amax=dout_amax)
            if need_dx1:
                dx1 = self.conv_dgrad(r.cs, dout, ci_off=C0, ci_cnt=C1, amax=dout_amax)
        # AdaGN + SiLU (+dropout)
        dh1 = pl.buf(*r.h1.shape)
        am1 = pl.buf(4) if self.amax_ok(r.c1.c) else None      # max|dh1| falls out of the GroupNorm-backward apply pass
        dss, dzss = self.gn_bwd(g2, d_a2, 0, dx0=dh1, want_dss=True, want_dzss=True, dx0_amax=am1)
        pl.free(d_a2)
        # emb_layers / emb_z_layers Linear backward (dW, db, and the gradient of SiLU(emb) / SiLU(shift_emb)): one launch for the pair
        self.linear_bwd_group([(r.l_ss, dss, d_ea, 1), (r.l_zss, dzss, d_eza, 1) if dzss is not None else None])
        pl.free(dss, dzss)
        # conv1
        self.conv_bwd_params(r.c1, dh1, amax=am1)
        trainable_gn1 = (g1.gname + ".weight") in self.Gr
        if need_dx0 or need_dx1 or trainable_gn1:
            bmode = 1 if r.down else (2 if r.up else 0)
            d_a1 = self.conv_dgrad(r.c1, dh1, amax=am1, gnb=g1 if bmode == 0 else None)
            if need_dx0 and dx0 is None:
                dx0 = pl.buf(g1.N, g1.H, g1.W, C0)
            if need_dx1 and dx1 is None:
                dx1 = pl.buf(g1.N, g1.H, g1.W, C1)
            am0 = pl.buf(4) if (out_amax and need_dx0 and self.f16_grads) else None
            self.gn_bwd(g1, d_a1, bmode, add=None if r.has_skip else dout, dx0=dx0 if need_dx0 else None, acc0=int(r.has_skip),
                        dx1=dx1 if need_dx1 else None, acc1=int(r.has_skip), dx0_amax=am0)
            pl.free(d_a1)
        else:
            am0 = None
        pl.free(dh1, am1)
        return dx0, dx1, am0

    # ------------------------------------------------------------------ AttentionBlock
    def attention(self, pre, x, heads, new_order):
        """model/module.py:422-428 with QKVAttentionLegacy (:431-457) or QKVAttention (:460-488)."""
        pl = self.p
        N, Hh, W, C = x.shape
        T = Hh * W
        ch = C // heads
        gx = self.gn(x, None, pre + ".norm", act=0)
        qkv, cq = self.conv(gx.y, None, pre + ".qkv", 1)
        if not self.save:
            pl.free(gx.y)
        if new_order:       # [q(all heads) | k | v]
            oq, ok, ov, hs = 0, C, 2 * C, ch
        else:               # per head [q k v]
            oq, ok, ov, hs = 0, ch, 2 * ch, 3 * ch
        scale2 = 1.0 / math.sqrt(ch)          # (ch^-1/4)^2 : q and k are each scaled by ch^-1/4 (module.py:451-453)
        o = pl.buf(N, Hh, W, C)
        fused = self.fuse_attn and H.attn_fused_ok(T, C, heads)
        Pm = lse = None
        if fused:
            # QK^T -> softmax -> PV in one launch, probabilities never written (pdae_attn_fwd); only the row log-sum-exp is kept for backward
            lse = pl.buf(N * heads, T) if self.save else None
            pl.emit(H.op_attn_fwd(qkv, N, T, C, heads, new_order, o, lse))
        else:
            Pm = pl.buf(N * heads, T, T)
            qp, kp, vp = qkv.data_ptr() + 4 * oq, qkv.data_ptr() + 4 * ok, qkv.data_ptr() + 4 * ov
            pl.emit(H.op_gemm(0, 1, T, T, ch, qp, 3 * C, kp, 3 * C, Pm, T, alpha=scale2, batch_outer=N, batch_inner=heads,
                              sA=(T * 3 * C, hs), sB=(T * 3 * C, hs), sC=(heads * T * T, T * T)))
            pl.emit(H.op_softmax(Pm, N * heads * T, T))
            pl.emit(H.op_gemm(0, 0, T, ch, T, Pm, T, vp, 3 * C, o, C, batch_outer=N, batch_inner=heads,
                              sA=(heads * T * T, T * T), sB=(T * 3 * C, hs), sC=(T * C, ch)))
        out, cp = self.conv(o, None, pre + ".proj_out", 1, res=x, res_mode=1)
        if not self.save:
            pl.free(qkv, Pm, o)
        return out, NS(pre=pre, gx=gx, cq=cq, cp=cp, qkv=qkv, Pm=Pm, lse=lse, o=o, N=N, T=T, C=C, heads=heads, ch=ch, offs=(oq, ok, ov, hs), scale2=scale2,
                       shape=(N, Hh, W, C), new_order=bool(new_order))

    def attention_bwd(self, a, dout, need_dx=True, dout_amax=None, out_amax=False):
        """Returns (dx, amax of dx or None); dout_amax / out_amax as in resblock_bwd."""
        pl = self.p
        N, T, C, heads, ch = a.N, a.T, a.C, a.heads, a.ch
        oq, ok, ov, hs = a.offs
        self.conv_bwd_params(a.cp, dout, amax=dout_amax)
        d_o = self.conv_dgrad(a.cp, dout, amax=dout_amax)          # [N,H,W,C]
        dqkv = pl.buf(*a.qkv.shape)
        if a.lse is not None:                                      # fused backward: probabilities recomputed from q, k and the saved log-sum-exp
            wsd = pl.buf(N * heads, T)
            pl.emit(H.op_attn_bwd(a.qkv, a.o, a.lse, d_o, N, T, C, heads, a.new_order, dqkv, wsd))
            pl.free(d_o, wsd)
            return self._attention_bwd_tail(a, dout, dqkv, need_dx, out_amax)
        dP = pl.buf(N * heads, T, T)
        qp, kp, vp = a.qkv.data_ptr() + 4 * oq, a.qkv.data_ptr() + 4 * ok, a.qkv.data_ptr() + 4 * ov
        dqp, dkp, dvp = dqkv.data_ptr() + 4 * oq, dqkv.data_ptr() + 4 * ok, dqkv.data_ptr() + 4 * ov
        bA = (heads * T * T, T * T)
        bQ = (T * 3 * C, hs)
        bO = (T * C, ch)
        # dV[s,c] = sum_t P[t,s] dO[t,c]
        pl.emit(H.op_gemm(1, 0, T, ch, T, a.Pm, T, d_o, C, dvp, 3 * C, batch_outer=N, batch_inner=heads, sA=bA, sB=bO, sC=bQ))
        # dP[t,s] = sum_c dO[t,c] v[s,c]
        pl.emit(H.op_gemm(0, 1, T, T, ch, d_o, C, vp, 3 * C, dP, T, batch_outer=N, batch_inner=heads, sA=bO, sB=bQ, sC=bA))
        pl.emit(H.op_softmax_bwd(a.Pm, dP, N * heads * T, T))
        # dQ = scale2 * dS K ; dK = scale2 * dS^T Q
        pl.emit(H.op_gemm(0, 0, T, ch, T, dP, T, kp, 3 * C, dqp, 3 * C, alpha=a.scale2, batch_outer=N, batch_inner=heads, sA=bA, sB=bQ, sC=bQ))
        pl.emit(H.op_gemm(1, 0, T, ch, T, dP, T, qp, 3 * C, dkp, 3 * C, alpha=a.scale2, batch_outer=N, batch_inner=heads, sA=bA, sB=bQ, sC=bQ))
        pl.free(d_o, dP)
        return self._attention_bwd_tail(a, dout, dqkv, need_dx, out_amax)

    def _attention_bwd_tail(self, a, dout, dqkv, need_dx, out_amax):
        pl = self.p
        self.conv_bwd_params(a.cq, dqkv)
        dx = am = None
        trainable_norm = (a.pre + ".norm.weight") in self.Gr
        if need_dx or trainable_norm:
            d_xn = self.conv_dgrad(a.cq, dqkv)
            if need_dx:
                dx = pl.buf(*a.shape)
                am = pl.buf(4) if (out_amax and self.f16_grads) else None
            self.gn_bwd(a.gx, d_xn, 0, add=dout, dx0=dx, dx0_amax=am)
            pl.free(d_xn)
        pl.free(dqkv)
        return dx, am
