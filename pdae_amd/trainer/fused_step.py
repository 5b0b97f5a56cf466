"""One optimisation step as a single planned graph (three variants share one driver).

FusedRLStep replaces the body of RepresentationLearningTrainer.train (trainer/train_representation_learning.py:81-124):
    zero_grad -> encoder fwd -> randint/randn -> q_sample -> ShiftUNet fwd -> weighted L2 -> backward
    (DDP all-reduce of the trainable gradients) -> Adam -> EMA
with one static op list over pre-allocated buffers:
  * the frozen trunk / eps-branch keep no activations and emit no backward;
  * loss and d(loss)/d(shift) come out of one kernel; the loss stays on the device (the reference's two
    `.item()` syncs per micro-batch, :107-108, are gone -- read `last_loss` when you want the value);
  * gradients live in flat buffers (decoder-trainable, encoder); the data-parallel exchange is an all-reduce(sum)
    over contiguous ranges of them, launched as soon as the backward has finished each range (reverse execution
    order) so RCCL overlaps with the rest of the backward; 1/world is folded into Adam;
  * Adam (torch.optim.Adam / AdamW semantics) and the EMA update are one kernel per flat buffer;
  * `num_iterations` micro-batches accumulate into the flat gradients before one optimizer step, EMA every `ema_every`
    steps (runner_config of every reference trainer: train_*.py main loops);
  * fp16-window guard (hip.SaturationGuard): while the device counter is non-zero the optimizer kernel applies nothing;
    `handle_saturation()` (host sync, call it at the logging cadence) rewinds the step counter by the discarded steps and
    rebuilds the plan in the range-free "bf16x6" arithmetic.
FusedRegularStep (config #1) and FusedLatentStep (config #5) are the same driver around other graphs.
"""
import math
import os
import random
import sys
import time

import torch
import torch.distributed as dist

from .. import hip as H
from ..engine import Plan, Builder
from ..model import graph as G


class _FusedStep:
    """Driver shared by the three fused steps.  A subclass provides `_build(math)` (emits forward + backward into self.plan, sets
    self.n_bwd, self.loss and self._marks) and `_load(*inputs)` (copies one micro-batch into the plan's input buffers)."""

    def _init_driver(self, nets, emas, lr, betas, eps, weight_decay, decoupled, ema_decay, ema_every, num_iterations, process_group,
                     bucket_mb, math, native_comm=None):
        self.flat_nets = list(nets)
        self.emas = list(emas)
        self.lr, self.b1, self.b2, self.adam_eps, self.wd, self.decoupled = lr, betas[0], betas[1], eps, weight_decay, int(decoupled)
        self.ema_decay, self.ema_every, self.num_iterations = ema_decay, int(ema_every), int(num_iterations)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.bucket_mb = bucket_mb
        self.step_count = 0
        self.micro = 0
        for n in self.flat_nets:
            n.grads()                                   # allocate the flat gradient buffers
        self.m = [torch.zeros_like(n.flat_train) for n in self.flat_nets]
        self.v = [torch.zeros_like(n.flat_train) for n in self.flat_nets]
        self.comm_events = None                         # set by enable_comm_timing()
        # gradient exchange: torch.distributed (ProcessGroupNCCL = RCCL) by default; native_comm / PDAE_NATIVE_RCCL=1 selects the library's own
        # RCCL communicator driven with HIP streams and events (pdae_amd/comm.py, include/pdae_hip.h: pdae_allreduce_bucket)
        if native_comm is None:
            native_comm = os.environ.get("PDAE_NATIVE_RCCL", "0") == "1"
        self.ncomm = None
        if native_comm and self.flat_nets[0].device.type == "cuda":
            from ..comm import NativeComm
            self.ncomm = NativeComm(self.flat_nets[0].device, group=process_group) if self.world > 1 else NativeComm(self.flat_nets[0].device, 0, 1)
        self.rebuild(math)

    # ------------------------------------------------------------------ plan (re)construction
    def rebuild(self, math=None):
        """(Re)builds the op list in arithmetic `math` (None: hip.default_math()); optimizer state and step count are kept."""
        self.math_name = math if isinstance(math, str) else (H.default_math() if math is None else
                                                             {v: k for k, v in H.MATH_NAMES.items()}[int(math)])
        self.plan = Plan(self.flat_nets[0].device)
        self._marks = []
        self._build(self.math_name)
        p = self.plan
        # the optimizer reads every gradient: the side-stream weight gradients are joined IN the op list (the step used to rely on the implicit join
        # at the end of the backward's pdae_run_ops call: plancheck.py found the unordered pair when it walked the list as a whole)
        p.join()
        self.n_bwd = len(p.recs)
        guard = H.SaturationGuard.get(p.device)
        self.guard = guard
        self.adam_idx = []
        for k, (net, ema) in enumerate(zip(self.flat_nets, self.emas)):
            idx = p.emit(H.op_adam_ema(net.flat_train, net.flat_grad, self.m[k], self.v[k], ema.flat_train if ema is not None else None,
                                       net.flat_train.numel(), self.lr, self.b1, self.b2, self.adam_eps, self.wd, self.decoupled, self.lr, 1.0, 1.0,
                                       self.ema_decay, guard=guard.ptr() if guard is not None else None, count_skip=int(k == 0)))
            self.adam_idx.append((idx, ema.flat_train.data_ptr() if ema is not None else 0))
        p.compile()
        self.buckets = self._make_buckets(self._marks, self.bucket_mb)

    # ------------------------------------------------------------------ DDP buckets
    def _make_buckets(self, marks, bucket_mb):
        """[(op_index_after_which_ready, flat_grad_tensor_slice)] in backward order.  marks = [(op index, net, prefix)]: every
        parameter gradient of `net` under `prefix` is final once the ops before `op index` have run."""
        limit = int(bucket_mb * (1 << 20)) // 4
        buckets = []
        seen = []
        for net in self.flat_nets:
            ranges = []
            for op_idx, mnet, prefix in marks:
                if mnet is not net:
                    continue
                offs = [(o, n) for k, (tr, o, n) in net._offs.items() if tr and k.startswith(prefix)]
                if offs:
                    ranges.append((op_idx, min(o for o, _ in offs), max(o + n for o, n in offs)))
            if not ranges:                               # no marks for this network: one bucket after the whole backward
                seen.append((self.n_bwd, net.flat_grad))
                continue
            cur_hi = net.flat_grad.numel()
            pend_lo = cur_hi
            # a network smaller than two buckets (the encoder: 15.6 MB, more than half of it in the final Linear whose gradient is final
            # first) still gets an early bucket: half of its bytes is the threshold
            limit_net = min(limit, max(net.flat_grad.numel() // 2, 1))
            for op_idx, lo, hi in ranges:                # backward order: descending offsets
                pend_lo = min(pend_lo, lo)
                if cur_hi - pend_lo >= limit_net:
                    buckets.append((op_idx, net.flat_grad[pend_lo:cur_hi]))
                    cur_hi = pend_lo
            if cur_hi > 0:
                buckets.append((ranges[-1][0], net.flat_grad[0:cur_hi]))
        buckets.extend(seen)
        buckets.sort(key=lambda b: b[0])
        return buckets

    # ------------------------------------------------------------------ one micro-batch / one step
    def _patch_adam(self):
        step = self.step_count + 1
        bc1, bc2 = 1.0 - self.b1 ** step, 1.0 - self.b2 ** step
        use_ema = (step % self.ema_every) == 0
        for idx, ema_ptr in self.adam_idx:
            op = self.plan.arr[idx]
            op.f[5] = self.lr / bc1
            op.f[6] = 1.0 / math.sqrt(bc2)
            op.f[7] = 1.0 / self.world
            op.p[4] = ema_ptr if (use_ema and ema_ptr) else None

    def _run_micro(self):
        p = self.plan
        if self.micro == 0 and self.num_iterations > 1:
            for n in self.flat_nets:
                n.flat_grad.zero_()
        if p.drop_ops:
            p.set_dropout(random.getrandbits(31), self.step_count * self.num_iterations + self.micro)   # host RNG: no device sync
        last = self.micro == self.num_iterations - 1
        if last and (self.world > 1 or self.ncomm is not None):
            try:
                self.backward_with_allreduce(p.run)
            except (RuntimeError, H.PdaeError):
                if self.num_iterations != 1 or not getattr(self, "_comm_fallback", False):
                    raise
                # the retry must not start on one rank while its peers still sit in (or have yet to time out of) their bucketed
                # collectives -- the fall-back's whole-buffer all-reduces would pair with collectives of another size.  Every rank that abandoned
                # the bucketed exchange of THIS step meets the others in the rendezvous store first; a rank that never arrives (its step
                # succeeded?  a sub-group?) makes the others give up and raise
                if not self._fallback_rendezvous():
                    raise
                self.comm_retries = getattr(self, "comm_retries", 0) + 1
                self.backward_with_allreduce(p.run)          # fall-back path: the whole forward + backward again, then one all-reduce per buffer
        else:
            p.run(0, self.n_bwd)
        self.micro += 1
        if last:
            self._patch_adam()
            p.run(self.n_bwd, p.n)
            self.step_count += 1
            self.micro = 0
        return self.loss

    def _fallback_rendezvous(self):
        """Store-based meeting point of the ranks that abandoned the bucketed exchange of the current step (world 1: nothing to agree on).  Returns
        False when the others do not show up within the collective timeout + 60 s (the caller then raises instead of retrying alone)."""
        if not dist.is_initialized():
            return True
        n_ranks = dist.get_world_size(self.pg)
        if n_ranks <= 1:
            return True
        import time
        try:
            store = dist.distributed_c10d._get_default_store()
            #  the key names the process group and counts the retries of this step: a second failure in the same step, or a step of a
            # sub-group, must not meet the counter of an earlier rendezvous
            try:
                gid = "-".join(str(r) for r in dist.get_process_group_ranks(self.pg)) if self.pg is not None else "world"
            except Exception:                     # noqa: BLE001
                gid = "world"
            key = f"pdae_amd/comm_fallback/{gid}/step{self.step_count}/try{getattr(self, 'comm_retries', 0)}"
            n = store.add(key, 1)
            deadline = time.time() + float(os.environ.get("PDAE_COMM_TIMEOUT_S", "300")) + 60.0
            while n < n_ranks:
                if time.time() > deadline:
                    return False
                time.sleep(0.02)
                n = store.add(key, 0)
            return True
        except Exception as e:                        # noqa: BLE001  (no store: no agreement, no retry)
            print(f"[pdae_amd] fallback rendezvous unavailable ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
            return False

    def backward_with_allreduce(self, run):
        """Issues the forward+backward op list in segments; as soon as a bucket's gradients are final its all-reduce(sum) is
        launched asynchronously (RCCL stream), overlapping with the rest of the backward.  `run(first, last)` issues plan ops.
        The saturation word travels with the last bucket (MAX) so that every rank takes the same skip decision.
        A collective that raises (RCCL init / enqueue error) is reported with the rank, the exchange switches -- for good -- to ONE all-reduce
        per flat gradient buffer after the backward (same process group), and the exception is RE-RAISED: this step's gradients are partly
        reduced.  `_run_micro` catches it and re-runs the micro-batch through that fall-back when the plan overwrites its gradients
        (num_iterations == 1: same batch, same dropout seed, identical gradients); an accumulating plan cannot be repaired and the error
        reaches the trainer.  Peers of a rank whose enqueue failed leave their outstanding collectives through the process-group timeout
        (bench.py: PDAE_COMM_TIMEOUT_S) and take the same route; nothing here can unblock them earlier."""
        if getattr(self, "_comm_fallback", False):
            run(0, self.n_bwd)
            for n in self.flat_nets:
                dist.all_reduce(n.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
            if self.guard is not None and self.math_name == "f16x3":
                dist.all_reduce(self.guard.t[0:1], op=dist.ReduceOp.MAX, group=self.pg)
            return
        try:
            self._bucketed(run)
        except (RuntimeError, H.PdaeError) as e:            # ProcessGroupNCCL / pdae_allreduce_bucket errors surface as RuntimeError / PdaeError
            rank = dist.get_rank(self.pg) if dist.is_initialized() else 0
            print(f"[pdae_amd] rank {rank}: bucketed gradient exchange failed ({type(e).__name__}: {e}); falling back to one all-reduce per "
                  "gradient buffer after the backward", file=sys.stderr, flush=True)
            self._comm_fallback, self.ncomm = True, None
            if self.flat_nets[0].device.type == "cuda":
                torch.cuda.synchronize()
            raise                                             # this step's gradients are in an unknown state: the caller decides (bench: retry)

    def _bucketed(self, run):
        works, cur = [], 0
        ev = self.comm_events
        nc = self.ncomm
        mark = (lambda: None) if ev is None else ev["mark"]
        if ev is not None:
            ev["t0"] = mark()
            ev["enqueue"], ev["complete"] = [], []
        for op_idx, view in self.buckets:
            run(cur, op_idx)
            cur = op_idx
            if ev is not None:
                ev["enqueue"].append(mark())             # the bucket's gradients are final at this point of the compute stream
            if nc is not None:
                nc.all_reduce(view)                      # side stream, behind an event of the compute stream
            else:
                works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        run(cur, self.n_bwd)
        if ev is not None:
            ev["bwd_done"] = mark()
        if self.guard is not None and self.math_name == "f16x3":
            if nc is not None:
                nc.all_reduce(self.guard.t[0:1], op="max")
            else:
                works.append(dist.all_reduce(self.guard.t[0:1], op=dist.ReduceOp.MAX, group=self.pg, async_op=True))
        if nc is not None:
            nc.wait()
        for k, w in enumerate(works):
            w.wait()
            if ev is not None and k < len(self.buckets):
                ev["complete"].append(mark())            # the compute stream has passed bucket k's completion (in bucket order: an upper bound)
        if ev is not None:
            ev["comm_done"] = mark()

    def enable_comm_timing(self):
        """Time marks around one step's backward / collective tail (bench.py): t0, per-bucket `enqueue` (gradients final on the compute stream) and
        `complete` (the compute stream has waited for the bucket's all-reduce), bwd_done, comm_done.  CUDA events on a ROCm device, host clocks in the
        CPU dry run.  comm_timing_ms() converts them after a synchronisation."""
        if self.flat_nets[0].device.type == "cuda":
            def mark():
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                return e
        else:
            mark = time.perf_counter
        self.comm_events = {"mark": mark}
        return self.comm_events

    def comm_timing_ms(self):
        """{exposed: comm_done - bwd_done, bucket_enqueue_to_complete: [...], bucket_complete_after_bwd: [...]} in ms from the marks of the last step."""
        ev = self.comm_events
        if ev is None or "comm_done" not in ev:
            return None
        if self.flat_nets[0].device.type == "cuda":
            torch.cuda.synchronize()
            d = lambda a, b: float(a.elapsed_time(b))
        else:
            d = lambda a, b: (b - a) * 1e3
        n = min(len(ev["enqueue"]), len(ev["complete"]))
        return {"exposed": d(ev["bwd_done"], ev["comm_done"]), "backward": d(ev["t0"], ev["bwd_done"]),
                "bucket_enqueue_to_complete": [round(d(ev["enqueue"][k], ev["complete"][k]), 3) for k in range(n)],
                "bucket_complete_after_bwd": [round(max(0.0, d(ev["bwd_done"], ev["complete"][k])), 3) for k in range(n)]}

    @property
    def last_loss(self):
        return float(self.loss.item())

    # ------------------------------------------------------------------ fp16-window guard
    def saturation(self):
        """(saturated launches, optimizer steps discarded) since the last reset -- host sync."""
        return self.guard.read() if self.guard is not None else (0, 0)

    def handle_saturation(self, log=True):
        """Polls the guard; if it fired: rewinds the step counter by the discarded steps, switches the process default to "bf16x6",
        rebuilds this plan and resets the counter.  Returns the number of optimizer steps that were discarded -- exactly the amount
        `step_count` was rewound by, so a trainer that subtracts it from its own counter stays in lock-step with the optimizer (the guard
        can fire inside a micro-batch whose optimizer step has not run yet: events > 0, nothing discarded).  `recoveries` counts the firings."""
        events, skipped = self.saturation()
        if events == 0:
            return 0
        self.step_count -= skipped
        self.recoveries = getattr(self, "recoveries", 0) + 1
        if log:
            print(f"[pdae_amd] fp16 window exceeded in {events} convolution launch(es): {skipped} optimizer step(s) discarded, "
                  "continuing in bf16x6 arithmetic", file=sys.stderr, flush=True)
        H.set_default_math("bf16x6")
        for n in self.flat_nets:
            n.invalidate_plans()
        self.rebuild("bf16x6")
        self.guard.reset()
        return skipped


class FusedRLStep(_FusedStep):
    def __init__(self, gaussian_diffusion, encoder, decoder, ema_encoder, ema_decoder, batch, height, width, lr=1e-4, betas=(0.9, 0.999),
                 eps=1e-8, weight_decay=0.0, decoupled=False, ema_decay=0.9999, ema_every=1, num_iterations=1, process_group=None,
                 bucket_mb=48, math=None, native_comm=None):
        self.gd, self.enc, self.dec, self.ema_enc, self.ema_dec = gaussian_diffusion, encoder, decoder, ema_encoder, ema_decoder
        self.N, self.Hh, self.W = batch, height, width
        self._init_driver([decoder, encoder], [ema_decoder, ema_encoder], lr, betas, eps, weight_decay, decoupled, ema_decay, ema_every,
                          num_iterations, process_group, bucket_mb, math, native_comm)

    def _build(self, math):
        gd, encoder, decoder = self.gd, self.enc, self.dec
        cfg = decoder.cfg
        N, Hh, W, Cimg = self.N, self.Hh, self.W, cfg["input_channel"]
        per = Hh * W * Cimg
        acc = self.num_iterations > 1
        drop = float(cfg["dropout"]) if decoder._shift_train else 0.0
        p = self.plan
        self.x0 = p.buf(N, Hh, W, Cimg)
        self.noise = p.buf(N, Hh, W, Cimg)
        self.t = p.buf(N, dtype=torch.int64)
        self.loss = p.buf(1)
        Be = Builder(p, encoder.P, encoder.grads(), save=True, acc_grads=acc, math=math)
        Bd = Builder(p, decoder.P, decoder.grads(), save=False, drop_p=drop, acc_grads=acc, math=math, frozen_of=decoder)
        # ---- forward
        # the encoder pass feeds only the shift branch (through z): it runs on the executor's second stream beside the frozen trunk's input blocks
        # (PDAE_SIDE_ENC=0: in front of them on the caller's stream, as through round 5); the join at the end of unet_forward covers it
        side_enc = os.environ.get("PDAE_SIDE_ENC", "1") != "0" and os.environ.get("PDAE_SIDE_SHIFT", "1") != "0"
        with p.side(side_enc):
            z, ex = G.encoder_forward(Be, encoder.NAME, self.x0)
        x_t = p.buf(N, Hh, W, Cimg)
        p.emit(H.op_q_sample(self.x0, self.noise, self.t, gd.sqrt_alphas_cumprod, gd.sqrt_one_minus_alphas_cumprod, N, per, x_t))
        fx = G.unet_forward(Bd, cfg, x_t, self.t, decoder.freqs, z=z, shift=True, train_shift=True, dropout=drop > 0, z_side=side_enc)
        d_shift = p.buf(N, Hh, W, Cimg)
        p.emit(H.op_loss(self.noise, fx.eps, fx.shift, self.t, gd.shift_coef, gd.weight, N, per, self.loss, None, dg=d_shift,
                         scale=1.0 / self.num_iterations), ws_slot=9)
        self.n_fwd = len(p.recs)
        self.z, self.eps, self.shift = z, fx.eps, fx.shift
        self.fx = fx                                      # graph context (saved activations of the shift branch): inspection / tests
        # ---- backward, with the op index at which each parameter block's gradients are final
        dz = G.shift_backward(Bd, fx, d_shift, mark=lambda prefix: self._marks.append((len(p.recs), decoder, prefix)))
        G.encoder_backward(Be, ex, dz, mark=lambda prefix: self._marks.append((len(p.recs), encoder, prefix)))

    def load_batch(self, x_0, t=None, noise=None):
        """x_0 / noise: (N,C,H,W) tensors of any strides.  t, noise are drawn like the reference
        (torch.randint then torch.randn_like, gaussian_diffusion.py:240-241) unless injected."""
        self.x0.copy_(x_0.permute(0, 2, 3, 1))
        if t is None:
            t = torch.randint(0, self.gd.timesteps, (self.N,), device=self.x0.device, dtype=torch.long)
        self.t.copy_(t)
        if noise is None:
            self.noise.normal_()
        else:
            self.noise.copy_(noise.permute(0, 2, 3, 1))

    def step(self, x_0, t=None, noise=None):
        """Runs one micro-batch; every `num_iterations`-th call also reduces gradients and applies Adam + EMA.
        Returns the (device-resident) loss of this micro-batch."""
        self.load_batch(x_0, t, noise)
        return self._run_micro()


# ----------------------------------------------------------------------------------------------------------
# torch.optim.Adam-compatible optimizer state (checkpoint key 'optimizer', train_representation_learning.py:214-226)
# ----------------------------------------------------------------------------------------------------------
def _param_order(groups):
    """[(net, name)] in the reference's param-group order."""
    order = []
    for net, prefixes in groups:
        for k, p in net.P.items():
            if p.requires_grad and (prefixes is None or k.startswith(prefixes)):
                order.append((net, k))
    return order


def export_adam_state(step_obj, groups, lr, betas, eps, weight_decay):
    nets = {id(n): i for i, n in enumerate(step_obj.flat_nets)}
    state, idx, param_groups = {}, 0, []
    for net, prefixes in groups:
        ids = []
        for n_, k in _param_order([(net, prefixes)]):
            tr, o, n = net._offs[k]
            fi = nets[id(net)]
            state[idx] = {"step": torch.tensor(float(step_obj.step_count)),
                          "exp_avg": net._view(step_obj.m[fi], o, net._shapes[k]).clone(),
                          "exp_avg_sq": net._view(step_obj.v[fi], o, net._shapes[k]).clone()}
            ids.append(idx)
            idx += 1
        # torch.optim.AdamW state dicts carry the same keys as Adam's: `decoupled_weight_decay` (torch >= 2.6 naming) tells them apart
        param_groups.append({"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False,
                             "decoupled_weight_decay": bool(step_obj.decoupled), "params": ids})
    return {"state": state, "param_groups": param_groups}


def load_adam_state(step_obj, groups, sd):
    nets = {id(n): i for i, n in enumerate(step_obj.flat_nets)}
    idx = 0
    for net, prefixes in groups:
        for n_, k in _param_order([(net, prefixes)]):
            st = sd["state"].get(idx)
            if st is not None:
                tr, o, n = net._offs[k]
                fi = nets[id(net)]
                net._view(step_obj.m[fi], o, net._shapes[k]).copy_(st["exp_avg"])
                net._view(step_obj.v[fi], o, net._shapes[k]).copy_(st["exp_avg_sq"])
                step_obj.step_count = int(float(st["step"]))
            idx += 1


class FusedRegularStep(_FusedStep):
    """One optimisation step of a plain DDPM UNet (config #1, trainer/train_regular_diffusion.py:59-141 +
    gaussian_diffusion.py:199-211): q_sample -> UNet fwd -> L2 -> full backward -> all-reduce -> Adam -> EMA, one plan."""

    def __init__(self, gaussian_diffusion, net, ema_net, batch, height, width, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 decoupled=False, ema_decay=0.9999, ema_every=1, num_iterations=1, process_group=None, bucket_mb=48, math=None):
        self.gd, self.net, self.ema = gaussian_diffusion, net, ema_net
        self.N, self.Hh, self.W = batch, height, width
        self._init_driver([net], [ema_net], lr, betas, eps, weight_decay, decoupled, ema_decay, ema_every, num_iterations, process_group,
                          bucket_mb, math)

    def _build(self, math):
        gd, net = self.gd, self.net
        cfg = net.cfg
        N, Hh, W, Cimg = self.N, self.Hh, self.W, cfg["input_channel"]
        per = Hh * W * Cimg
        p = self.plan
        self.x0, self.noise = p.buf(N, Hh, W, Cimg), p.buf(N, Hh, W, Cimg)
        self.t = p.buf(N, dtype=torch.int64)
        self.cond = p.buf(N, dtype=torch.int64) if cfg.get("num_class") is not None else None
        self.loss = p.buf(1)
        drop = float(cfg["dropout"])
        B = Builder(p, net.P, net.grads(), save=True, drop_p=drop, acc_grads=self.num_iterations > 1, math=math)
        x_t = p.buf(N, Hh, W, Cimg)
        p.emit(H.op_q_sample(self.x0, self.noise, self.t, gd.sqrt_alphas_cumprod, gd.sqrt_one_minus_alphas_cumprod, N, per, x_t))
        fx = G.unet_forward(B, cfg, x_t, self.t, net.freqs, cond=self.cond, dropout=drop > 0)
        self.eps = fx.eps
        d_eps = p.buf(*fx.eps.shape)
        p.emit(H.op_loss(self.noise, fx.eps, None, None, None, None, N, fx.eps.numel() // N, self.loss, None, deps=d_eps,
                         scale=1.0 / self.num_iterations), ws_slot=9)
        G.unet_backward(B, fx, d_eps)

    def step(self, x_0, condition=None, t=None, noise=None):
        self.x0.copy_(x_0.permute(0, 2, 3, 1))
        self.t.copy_(torch.randint(0, self.gd.timesteps, (self.N,), device=self.x0.device, dtype=torch.long) if t is None else t)
        if noise is None:
            self.noise.normal_()
        else:
            self.noise.copy_(noise.permute(0, 2, 3, 1))
        if self.cond is not None:
            self.cond.copy_(condition)
        return self._run_micro()


class FusedLatentStep(_FusedStep):
    """One optimisation step of the latent DPM (config #5, trainer/train_latent_diffusion.py:95-178 +
    gaussian_diffusion.py:373-398): q_sample on the latent schedule (constant beta 0.008) -> MLPSkipNet fwd -> L1 ->
    backward -> all-reduce -> Adam / AdamW -> EMA, one plan.  `z_0` is the already normalised latent."""

    def __init__(self, gaussian_diffusion, net, ema_net, batch, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, decoupled=True,
                 ema_decay=0.9999, ema_every=1, num_iterations=1, process_group=None, bucket_mb=48, math=None):
        self.gd, self.net, self.ema = gaussian_diffusion, net, ema_net
        self.N, self.timesteps = batch, gaussian_diffusion.latent_diffusion_config["timesteps"]
        self._init_driver([net], [ema_net], lr, betas, eps, weight_decay, decoupled, ema_decay, ema_every, num_iterations, process_group,
                          bucket_mb, math)

    def _build(self, math):
        net, batch = self.net, self.N
        lcfg = self.gd.latent_diffusion_config
        ic = net.cfg["input_channel"]
        p = self.plan
        self.z0, self.noise = p.buf(batch, ic), p.buf(batch, ic)
        self.t = p.buf(batch, dtype=torch.int64)
        self.loss = p.buf(1)
        B = Builder(p, net.P, net.grads(), save=True, acc_grads=self.num_iterations > 1, math=math)
        z_t = p.buf(batch, ic)
        p.emit(H.op_q_sample(self.z0, self.noise, self.t, lcfg["sqrt_alphas_cumprod"], lcfg["sqrt_one_minus_alphas_cumprod"], batch, ic, z_t))
        fx = net._emit_forward(B, z_t, self.t)
        d_out = p.buf(batch, ic)
        p.emit(H.op_loss(self.noise, fx.out, None, None, None, None, batch, ic, self.loss, None, deps=d_out, l1=1,
                         scale=1.0 / self.num_iterations), ws_slot=9)
        net._emit_backward(B, fx, d_out)

    def step(self, z_0, t=None, noise=None):
        self.z0.copy_(z_0)
        self.t.copy_(torch.randint(0, self.timesteps, (self.N,), device=self.z0.device, dtype=torch.long) if t is None else t)
        if noise is None:
            self.noise.normal_()
        else:
            self.noise.copy_(noise)
        return self._run_micro()
