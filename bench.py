#!/usr/bin/env python
"""Benchmark of the PDAE hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one complete representation-learning optimisation step (BASELINE.json metric, config
ffhq_representation_learning): FFHQEncoder forward, q_sample, ShiftUNet forward (frozen trunk + eps branch +
trainable shift branch), weighted L2 loss, backward through the shift branch and the encoder, (all-reduce of the
82.4 M trainable gradients when N>1), Adam and EMA -- fp32, per-GPU batch 32, 3x128x128 synthetic images already
resident in HBM, randomly initialised weights of the FFHQ-128 architecture ([ASSUMED] Diff-AE-compatible
hyper-parameters).  Nothing is skipped inside the timed region.

Rank 0 prints ONE JSON line; `value` is whole-job images/s.  Extra objects:
  roofline     -- the dominant kernel: the 3x3 forward / data-gradient patch kernels (conv3x3y: Winograd F(2,3) along x, persistent, on
                  chip-filling layers -- priced at the 2 MFMAs per product it issues; conv3x3r: direct persistent form for launches with fused
                  skip chunks; conv3x3p on the small ones), split-fp16 MFMA: algorithmic FLOPs of their launches in one step / their summed
                  durations, measured here with HIP events on the launch stream (per-op event pairs, one extra un-timed step); `wgrad` and
                  `family` carry the same figures for the 3x3 weight-gradient kernel and for every conv / GEMM launch together;
  step_mfma_roofline_frac / step_hbm_roofline_frac -- the whole step against the MFMA floor (step FLOPs x 3 products / 2.5 PFLOP/s) and
                  against the HBM floor SURVEY 8(d) quotes north_star's target on (B x 2.3 GB + 3.0 GB at 8 TB/s);
  other_legs   -- BASELINE.md section 3 report items: the same step with bf16 operands (enable_amp), and config #2 (CelebA-64, bf16);
  cpu_baseline -- the CPU oracle (torch fp32, all host threads) running the same step on a bounded sample; ddim100.cpu_baseline the
                  oracle's decoder forward (what a DDIM step costs on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIG = "config/ffhq_representation_learning.yml"      # BASELINE.json configs[2]; its trained_ddpm_config names the UNet hyper-parameters
# (pre-trained-dpms/ffhq128/config.yml is a download in the reference: the shipped stand-in holds the Diff-AE-compatible values implied by the
# checkpoint key names -- [ASSUMED], SURVEY.md section 0.2.  Nothing below hard-codes them.)


def load_workload():
    """(training config, denoise_fn_config of the pre-trained DPM) from the shipped YAMLs."""
    from pdae_amd.utils import load_yaml
    c = load_yaml(os.path.join(ROOT, CONFIG))
    ddpm = load_yaml(os.path.join(ROOT, c["trained_ddpm_config"]))["denoise_fn_config"]
    return c, ddpm


PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 MFMA
MFMA_PER_PRODUCT = {"f32": 1, "bf16": 1, "bf16x3": 3, "bf16x6": 6, "f16x3": 3}
TRAIN_GFLOP_PER_IMG = 481.4         # fwd 258.4 + 2 x 110.7 (shift-branch bwd) + 3 x 0.549 (encoder)
FWD_GFLOP_PER_IMG = 258.4
HBM_GB_PER_IMG, HBM_GB_PER_STEP = 2.3, 3.0          # algorithmic HBM bytes of the step = B x 2.3 GB activations + 3.0 GB parameters / optimizer state
PEAK_HBM_TBS = 8.0


def randomize(net, seed):
    """random-init weights of the architecture; the reference's zero-initialised layers are randomised too so no
    kernel sees all-zero operands (zero data raises clocks: guide 5.4 rule 25)."""
    g = torch.Generator(device=net.device).manual_seed(seed)
    with torch.no_grad():
        for k, p in net.P.items():
            if p.dim() > 1 and float(p.abs().max()) == 0.0:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g, device=net.device) / fan_in ** 0.5)


def op_flops(op):
    """Algorithmic FLOPs of one igemm-family op record (0 for everything else)."""
    from pdae_amd import hip as H
    i = op.i
    if op.kind in (H.OP_CONV_FWD, H.OP_CONV_WGRAD, H.OP_CONV_FWD_GN):
        N, Ho, Wo, Cout, KH, KW, cin = i[0], i[5], i[6], i[7], i[8], i[9], i[3] + i[4]
        return 2.0 * N * Ho * Wo * Cout * KH * KW * cin
    if op.kind == H.OP_CONV_FWD_SKIP:                 # 3x3 conv + the 1x1 skip convolution riding in its K loop
        return 2.0 * i[0] * i[5] * i[6] * i[7] * (9 * (i[3] + i[4]) + i[15] + i[16])
    if op.kind == H.OP_CONV_DGRAD:
        N, Hi, Wi, Cout, KH, KW, up = i[0], i[1], i[2], i[7], i[8], i[9], i[12]
        s = 2 if up else 1
        return 2.0 * N * (Hi * s) * (Wi * s) * i[15] * KH * KW * Cout
    if op.kind == H.OP_GEMM:
        return 2.0 * i[2] * i[3] * i[4] * i[14] * i[15]
    return 0.0


def profile_plan(plan, first, last):
    """Per-op durations (ms) of plan ops [first, last) with HIP events on the launch stream."""
    from pdae_amd import hip as H
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(last - first + 1)]
    torch.cuda.synchronize()
    side = H.get_knob("PDAE_SIDE_STREAM")
    H.set_knob("PDAE_SIDE_STREAM", 0)        # one op at a time: the ops flagged for the second stream are timed on this one (no fork / join in the figure)
    try:
        for k in range(first, last):
            evs[k - first].record()
            H.run_ops(plan.arr[k], 1)
        evs[-1].record()
        torch.cuda.synchronize()
    finally:
        H.set_knob("PDAE_SIDE_STREAM", side)
    return [evs[j].elapsed_time(evs[j + 1]) for j in range(last - first)]


def cpu_baseline(batch, steps, warm=3):
    """The CPU oracle (oracle/pdae_oracle.py, validated against the reference by tests/test_oracle_golden.py)
    running the same train step: forward + autograd backward + Adam + EMA on all host threads; `warm` untimed + `steps` timed steps, median
    (SURVEY 8d protocol: 3 + 5)."""
    from oracle import pdae_oracle as O
    torch.manual_seed(0)
    cfg = dict(load_workload()[1], dropout=0.0)
    enc_sd = O.synth_state_dict(O.encoder_param_shapes("FFHQEncoder", 512), 1)
    dec_sd = O.synth_state_dict(O.unet_param_shapes(cfg, shift=True, latent_dim=512), 2)
    s = O.Schedules()
    train = {"e:" + k: v for k, v in enc_sd.items()}
    train.update({"d:" + k: v for k, v in dec_sd.items() if O.shift_unet_trainable(k)})
    m = {k: torch.zeros_like(v) for k, v in train.items()}
    v2 = {k: torch.zeros_like(v) for k, v in train.items()}
    ema = {k: v.clone() for k, v in train.items()}
    x0 = torch.rand(batch, 3, 128, 128) * 2 - 1
    times = []
    for step in range(steps + warm):
        t0 = time.perf_counter()
        for p in train.values():
            p.requires_grad_(True)
            p.grad = None
        t = torch.randint(0, 1000, (batch,))
        noise = torch.randn_like(x0)
        loss = O.rl_loss(s, enc_sd, "FFHQEncoder", dec_sd, cfg, x0, t, noise)
        loss.backward()
        with torch.no_grad():
            for k, p in train.items():
                pn, m[k], v2[k] = O.adam_step(p.detach(), p.grad, m[k], v2[k], step + 1, 1e-4)
                p.requires_grad_(False)
                p.copy_(pn)
                ema[k] = O.ema_update(ema[k], p, 0.9999)
        times.append(time.perf_counter() - t0)
    timed = sorted(times[warm:])
    best = timed[len(timed) // 2]
    return batch / best, best


def cpu_decoder_forward(batch, reps=3):
    """Seconds per ShiftUNet forward of the CPU oracle at `batch` (one DDIM step = one such forward + an elementwise update)."""
    from oracle import pdae_oracle as O
    cfg = dict(load_workload()[1], dropout=0.0)
    sd = O.synth_state_dict(O.unet_param_shapes(cfg, shift=True, latent_dim=512), 2)
    x, t, z = torch.randn(batch, 3, 128, 128), torch.randint(0, 1000, (batch,)), torch.randn(batch, 512)
    ts = []
    with torch.no_grad():
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            O.shift_unet_forward(sd, cfg, x, t, z)
            ts.append(time.perf_counter() - t0)
    return sorted(ts[1:])[len(ts[1:]) // 2]


def train_leg(config_file, math, steps, dev, batch=None):
    """images/s of the fused representation-learning step of another shipped config / arithmetic (BASELINE.md section 3 report items)."""
    import copy
    from pdae_amd.model.representation_learning import decoder as decoder_module, encoder as encoder_module
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    from pdae_amd.trainer.fused_step import FusedRLStep
    from pdae_amd.utils import load_yaml
    c = load_yaml(os.path.join(ROOT, config_file))
    ddpm = load_yaml(os.path.join(ROOT, c["trained_ddpm_config"]))["denoise_fn_config"]
    size, B = int(c["train_dataset_config"]["image_size"]), batch or int(c["dataloader_config"]["train"]["batch_size"])
    enc = getattr(encoder_module, c["encoder_config"]["model"])(device=dev, **c["encoder_config"])
    dec = getattr(decoder_module, c["decoder_config"]["model"])(device=dev, latent_dim=c["decoder_config"]["latent_dim"], **ddpm)
    randomize(enc, 11); randomize(dec, 12)
    enc.train(); dec.set_train_mode()
    oc, rc = c["optimizer_config"], c["runner_config"]
    st = FusedRLStep(GaussianDiffusion(c["diffusion_config"], dev), enc, dec, copy.deepcopy(enc), copy.deepcopy(dec), B, size, size, lr=float(oc["lr"]),
                     betas=eval(oc["adam_betas"]), eps=float(oc["adam_eps"]), weight_decay=float(oc["weight_decay"]), ema_decay=float(rc["ema_decay"]),
                     ema_every=int(rc["ema_every"]), num_iterations=int(rc["num_iterations"]), math=math)
    x0 = torch.rand(B, 3, size, size, device=dev) * 2 - 1
    for _ in range(2):
        st.step(x0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step(x0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"config": config_file, "math": math or "f16x3", "per_gpu_batch": B, "image": f"3x{size}x{size}", "ms_per_step": round(dt * 1e3, 3),
            "images_per_sec": round(B / dt, 2), "final_loss": round(st.last_loss, 6), "steps": steps}


def host_cores():
    """Usable host cores: min(affinity mask, cgroup CPU quota) -- the GPU box advertises 256 logical CPUs under a 16-CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


_T0 = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU (the reference's launch is
    torchrun --nproc_per_node, scripts/dist_train_representation_learning.sh:10-14).  Rank 0's JSON line is the only stdout."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


DRY_CFG = dict(input_channel=3, base_channel=32, channel_multiplier=[1, 2, 2], num_residual_blocks_of_a_block=1, attention_resolutions=[4],
               num_heads=1, head_channel=32, use_new_attention_order=False, dropout=0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: dataloader_config.train.batch_size of the YAML = 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ddim-batch", type=int, default=100, help="batch of the DDIM-100 measurement (sampler/autoencoding_eval.py:125 uses 100)")
    ap.add_argument("--no-ddim", action="store_true")
    ap.add_argument("--autoencode", action="store_true", help="also time the evaluator's whole protocol at the DDIM batch: ddim1000 encode + ddim100 decode "
                    "(sampler/autoencoding_eval.py:74-78; 1099 decoder passes, ~90 s) -> ddim100.autoencode_1099_steps_seconds")
    ap.add_argument("--no-legs", action="store_true", help="skip the other_legs object (F128 with bf16 operands, CelebA-64 bf16)")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--bucket-mb", type=float, default=48, help="gradient bucket size of the data-parallel all-reduce (default 48 MB: the timed region starts "
                                                              "right behind the --warmup steps); 0 = sweep 16/48/96 MB first (12 extra untimed steps per "
                                                              "rank) and keep the fastest")
    ap.add_argument("--native-rccl", action="store_true", help="gradient exchange through the library's own RCCL communicator on HIP streams "
                                                              "(pdae_allreduce_bucket) instead of torch.distributed.all_reduce")
    ap.add_argument("--dry", action="store_true", help="plumbing check without a GPU: gloo backend, a small network on CPU tensors, kernels "
                                                      "replaced by a no-op recorder (exercises launcher, process group, buckets, JSON)")
    ap.add_argument("--math", default=None, choices=["f32", "bf16x6", "bf16x3", "bf16", "f16x3"],
                    help="conv arithmetic on fp32 tensors (default f16x3 = two-fp16-plane split with power-of-two scales, fp32 grade; "
                         "bf16x6 = exact 3-bf16-plane split; f32 = f32 MFMA)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.math:
        os.environ["PDAE_CONV_MATH"] = args.math
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["NCCL_DEBUG"] = os.environ.get("PDAE_NCCL_DEBUG", "WARN")      # RCCL's version banner goes to stdout: keep it to the one JSON line
        # knobs for the CU contention between RCCL's copy kernels and the power-capped MFMA kernels (recorded in `comm`):
        #   PDAE_RCCL_CHANNELS = n   -> NCCL_MIN_NCHANNELS = NCCL_MAX_NCHANNELS = n (each channel is one workgroup per peer direction)
        if os.environ.get("PDAE_RCCL_CHANNELS"):
            os.environ["NCCL_MIN_NCHANNELS"] = os.environ["NCCL_MAX_NCHANNELS"] = os.environ["PDAE_RCCL_CHANNELS"]
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("PDAE_COMM_TIMEOUT_S", "300")))      # a wedged collective errors out instead of sitting in the driver's timeout
        if dry:
            dist.init_process_group(backend="gloo", timeout=tmo)
        else:
            dist.init_process_group(backend="nccl", device_id=dev, timeout=tmo)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    def sync():
        if not dry:
            torch.cuda.synchronize()

    import copy
    from pdae_amd.model.representation_learning import decoder as decoder_module, encoder as encoder_module
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    from pdae_amd.trainer.fused_step import FusedRLStep
    from pdae_amd.utils import set_seed

    cfg, ddpm_cfg = load_workload()
    if dry:
        ddpm_cfg = dict(DRY_CFG)
    size = 64 if dry else int(cfg["train_dataset_config"]["image_size"])
    oc, rc = cfg["optimizer_config"], cfg["runner_config"]
    log("building networks")
    set_seed(0)                                    # identical weights on every rank (trainer/base_trainer.py:27-28)
    enc_name = "CELEBA64Encoder" if dry else cfg["encoder_config"]["model"]
    enc = getattr(encoder_module, enc_name)(device=dev, **cfg["encoder_config"])
    dec = getattr(decoder_module, cfg["decoder_config"]["model"])(device=dev, latent_dim=cfg["decoder_config"]["latent_dim"], **ddpm_cfg)
    if not dry:
        randomize(enc, 11)
        randomize(dec, 12)
    enc.train()
    dec.set_train_mode()
    ema_enc, ema_dec = copy.deepcopy(enc), copy.deepcopy(dec)
    gd = GaussianDiffusion(cfg["diffusion_config"], dev)
    B = args.batch or (2 if dry else int(cfg["dataloader_config"]["train"]["batch_size"]))
    log("building the fused step plan")
    st = FusedRLStep(gd, enc, dec, ema_enc, ema_dec, B, size, size, lr=float(oc["lr"]), betas=eval(oc["adam_betas"]), eps=float(oc["adam_eps"]),
                     weight_decay=float(oc["weight_decay"]), ema_decay=float(rc["ema_decay"]), ema_every=int(rc["ema_every"]),
                     num_iterations=int(rc["num_iterations"]), bucket_mb=args.bucket_mb or 48, native_comm=bool(args.native_rccl) and not dry)
    if dry:
        st.plan.run = lambda first=0, last=None, stream=None: None          # no kernels: launcher / process-group / bucket plumbing only
    log(f"plan: {st.plan.n} ops, {st.plan.bytes_alloc / 2**30:.1f} GiB of activations/workspaces")
    set_seed(rank)                                 # per-rank data / noise streams (base_trainer.py:50-52)
    x0 = torch.rand(B, 3, size, size, device=dev) * 2 - 1

    # ---- data-parallel exchange: is RCCL really spanning `world` ranks; bucket size (swept, untimed); exposed vs isolated all-reduce time
    comm = None
    if world > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        comm = {"backend": dist.get_backend(), "exchange": "pdae_allreduce_bucket (own RCCL communicator, side stream)" if st.ncomm is not None else "torch.distributed.all_reduce (async)",
                "rccl_ranks": int(ones.item()), "grad_bytes_per_step": int(4 * (dec.flat_grad.numel() + enc.flat_grad.numel()))}
        comm["rccl_channels"] = os.environ.get("PDAE_RCCL_CHANNELS", "default")
        sweep = {}
        for mb in ([] if args.bucket_mb else [16, 48, 96]):
            st.buckets = st._make_buckets(st._marks, mb)
            try:
                st.step(x0); sync(); dist.barrier()
            except (RuntimeError, Exception) as e:              # noqa: BLE001 -- bucketed exchange failed: FusedRLStep has switched to the post-backward fallback
                log(f"rank {rank}: bucketed exchange failed during the sweep ({type(e).__name__}); continuing with the fallback all-reduce")
                comm["fallback"] = f"{type(e).__name__}: {str(e)[:200]}"
                st.step(x0); sync(); dist.barrier()
            t1 = time.perf_counter()
            for _ in range(3):
                st.step(x0)
            sync(); dist.barrier()
            sweep[str(mb)] = round((time.perf_counter() - t1) / 3 * 1e3, 3)
        if sweep:
            tt = torch.tensor([sweep[k] for k in sweep], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sweep = {k: round(float(v), 3) for k, v in zip(sweep, tt.tolist())}
            best = min(sweep, key=sweep.get)
            log(f"bucket sweep (ms/step): {sweep} -> {best} MB")
        else:
            best = args.bucket_mb
        st.buckets = st._make_buckets(st._marks, float(best))
        comm.update(bucket_mb=float(best), bucket_sweep_ms_per_step=sweep or None, buckets=len(st.buckets),
                    bucket_bytes=[int(v.numel() * 4) for _, v in st.buckets])

    for _ in range(args.warmup):
        try:
            st.step(x0)
        except (RuntimeError, Exception) as e:                  # noqa: BLE001 -- a failed bucketed exchange: the step has switched to the post-backward fallback
            if comm is None or not getattr(st, "_comm_fallback", False):
                raise
            log(f"rank {rank}: bucketed exchange failed in the warm-up ({type(e).__name__}); continuing with the fallback all-reduce")
            comm["fallback"] = f"{type(e).__name__}: {str(e)[:200]}"
            st.step(x0)
        sync()
        log("warm-up step done")
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.step(x0)
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        # every rank's own clock around the same K steps (both ends behind a barrier): reported per rank, on the fall-back path too; value uses the MAX
        per_rank = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([dt], device=dev, dtype=torch.float64))
        per_rank = [float(v.item()) for v in per_rank]
        dt = max(per_rank)
        comm["per_rank_ms_per_step"] = [round(v / args.steps * 1e3, 3) for v in per_rank]
        comm["exchange_path"] = "fallback: one all-reduce per gradient buffer after the backward" if getattr(st, "_comm_fallback", False) else "bucketed, overlapped with the backward"
        comm["comm_retries"] = int(getattr(st, "comm_retries", 0))
        comm["adam_grad_scale"] = float(st.plan.arr[st.adam_idx[0][0]].f[7])      # the mean over ranks = sum x (1 / world), folded into the Adam kernel
    loss_val = 0.0 if dry else st.last_loss
    log(f"timed region done: {dt / args.steps * 1e3:.1f} ms/step")
    ms_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    sat = st.saturation()

    if world > 1:                                  # after the timed region: collective tail of one step (per bucket), and the all-reduce alone
        st.enable_comm_timing()
        st.step(x0); sync()
        tm = st.comm_timing_ms()
        st.comm_events = None
        if tm is not None:                           # (None on the fall-back path: there are no buckets then)
            nb = len(tm["bucket_enqueue_to_complete"])
            tt = torch.tensor([tm["exposed"], tm["backward"]] + tm["bucket_enqueue_to_complete"] + tm["bucket_complete_after_bwd"], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            v = [float(x) for x in tt.tolist()]
            exposed = v[0]
            comm.update(allreduce_exposed_ms_per_step=round(exposed, 3), backward_ms=round(v[1], 3),
                        bucket_enqueue_to_complete_ms=[round(x, 3) for x in v[2:2 + nb]],          # all-reduce of bucket k seen from the compute stream (max over ranks)
                        bucket_complete_after_backward_ms=[round(x, 3) for x in v[2 + nb:2 + 2 * nb]])  # how much of it was still running when the backward had ended
            if not dry:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                dist.barrier(); sync()
                e0.record()
                for _, bv in st.buckets:
                    dist.all_reduce(bv, op=dist.ReduceOp.SUM)
                e1.record(); sync()
                alone = e0.elapsed_time(e1)
            else:
                dist.barrier()
                ta = time.perf_counter()
                for _, bv in st.buckets:
                    dist.all_reduce(bv, op=dist.ReduceOp.SUM)
                alone = (time.perf_counter() - ta) * 1e3
            tt = torch.tensor([alone], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            alone = float(tt.item())
            comm.update(allreduce_ms_per_step=round(alone, 3), overlap_frac=round(max(0.0, 1.0 - exposed / max(alone, 1e-9)), 4),
                        allreduce_bus_gbps=round(2 * (world - 1) / world * comm["grad_bytes_per_step"] / max(alone, 1e-9) / 1e6, 1))

    from pdae_amd import hip as H
    if comm is not None and world > 1:
        # A/B of the LDS-displacement remedy: the persistent conv3x3y launches with 256 - k workgroups
        # (knob PDAE_Y_GRID_TRIM) leave k CUs to the collective's channels.  Outside the timed region, after it: 4 steps per setting, MAX over ranks;
        # the headline number above is the default (k = 0).  A driver run on a real multi-GPU node thereby measures what this builder never could.
        ab = {}
        try:
            for k in ((0, 8) if dry else (0, 8, 16, 32)):      # (the dry run only exercises the control flow: no kernels)
                H.set_knob("PDAE_Y_GRID_TRIM", k)
                st.step(x0); sync(); dist.barrier()
                t0 = time.perf_counter()
                for _ in range(4):
                    st.step(x0)
                sync(); dist.barrier()
                tt = torch.tensor([(time.perf_counter() - t0) / 4 * 1e3], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ab[str(k)] = round(float(tt.item()), 3)
        except Exception as e:                       # noqa: BLE001  (an A/B aid behind the timed region must never cost the run its JSON line)
            comm["y_grid_trim_error"] = f"{type(e).__name__}: {str(e)[:200]}"
        finally:
            H.set_knob("PDAE_Y_GRID_TRIM", 0)
        comm["y_grid_trim_ms_per_step"] = ab
    if comm is not None:
        comm["y_grid_trim"] = int(H.get_knob("PDAE_Y_GRID_TRIM"))

    wl = "config/ffhq_representation_learning.yml: PDAE representation learning, FFHQ-128 [ASSUMED denoise_fn_config], encoder FFHQEncoder + ShiftUNet, " \
         f"Adam lr {oc['lr']}, EMA {rc['ema_decay']}, dropout {ddpm_cfg['dropout']}"
    out = {"metric": "train_images_per_sec_ffhq128_representation_learning", "value": round(value, 3), "unit": "images/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": ("DRY RUN (no kernels, small network, gloo): " if dry else "") + wl,
                      "per_gpu_batch": B, "global_batch": B * world, "image": f"3x{size}x{size}", "parallelism": f"dp{world}",
                      "params_total": int(sum(p.numel() for p in dec.P.values()) + sum(p.numel() for p in enc.P.values())),
                      "params_trainable": int(sum(p.numel() for p in dec.P.values() if p.requires_grad) + sum(p.numel() for p in enc.P.values()))},
           "images_per_sec_per_gpu": round(value / world, 3), "final_loss": round(loss_val, 6),
           "fp16_window_events": sat[0], "optimizer_steps_discarded": sat[1],
           "step_tflops_algorithmic": round(TRAIN_GFLOP_PER_IMG * B / ms_step, 3)}
    if comm is not None:
        out["comm"] = comm
    if dry:
        out["dry"] = True
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    if rank == 0:
        # ---- roofline of the dominant kernel family, live, with HIP events (one extra, un-timed step)
        st.load_batch(x0)
        durs = profile_plan(st.plan, 0, st.n_bwd)
        # algorithmic FLOPs: a stride-2 convolution run in the dense-grid form (engine.Builder._dense_grid_desc) executes 4x its products
        fl = [op_flops(st.plan.arr[k]) * (0.25 if k in getattr(st.plan, "dense_grid", ()) else 1.0) for k in range(st.n_bwd)]
        ig_ms = sum(d for d, f in zip(durs, fl) if f > 0)
        ig_fl = sum(fl)
        n_ig = sum(1 for f in fl if f > 0)
        # the single heaviest launch (conv3x3 on the 128x128 grid)
        kbig = max(range(st.n_bwd), key=lambda k: fl[k])
        from pdae_amd import hip as H
        math = os.environ.get("PDAE_CONV_MATH", H.DEFAULT_MATH)
        npm = MFMA_PER_PRODUCT[math]

        def op_npm(op):                               # MFMAs issued per algorithmic product by the kernel that runs this op
            if math == "f16x3":                       # two fp16 planes in the forward 3x3 patch kernel only; everything else bf16x6
                fwd3 = op.kind in (H.OP_CONV_FWD_GN, H.OP_CONV_FWD_SKIP) or (op.kind == H.OP_CONV_FWD and bool(op.p[6]) and op.i[8] == 3)
                grad3 = (op.kind == H.OP_CONV_DGRAD and bool(op.p[4])) or (op.kind == H.OP_CONV_WGRAD and bool(op.p[6]))     # dy_amax given
                if (fwd3 and op.kind != H.OP_CONV_FWD_SKIP) or op.kind == H.OP_CONV_DGRAD and grad3:
                    # Winograd F(2, 3)-along-x form (conv3x3y.hip): 4 transform positions per 2 outputs x 3 taps = two thirds of the MFMAs
                    i = op.i
                    c = H.Conv(i[0], i[1], i[2], i[3], i[4], i[7], k=i[8], stride=i[10], pad=i[11], up=bool(i[12]), math=i[13])
                    if c.winograd_form(int(op.kind == H.OP_CONV_DGRAD), gn=op.kind == H.OP_CONV_FWD_GN, f16_grad=op.kind == H.OP_CONV_DGRAD):
                        return 2
                return 3 if (fwd3 or grad3) else 6
            return npm
        # peak for the arithmetic actually executed: f32 MFMA 157.3 TF, or the dense bf16 MFMA peak divided by the number of
        # bf16 MFMAs issued per algorithmic product (6 for the exact 3-plane split)
        peak = PEAK_F32_MFMA_TFLOPS if math == "f32" else PEAK_BF16_MFMA_TFLOPS / npm
        out["dtype"] = {"f32": "f32", "bf16": "bf16",
                        "f16x3": "f32 as split-operand MFMA (3x3 convs fwd/dgrad/wgrad: 2 fp16 planes x3 products, power-of-two scaled; 1x1 / generic: 3 bf16 planes x6), fp32 accumulate",
                        }.get(math, f"f32 as {math} split-operand MFMA, fp32 accumulate")
        # dominant kernel: conv3x3p_kernel (3x3 forward + data gradient on the patch path = ops that carry prepared weights)
        def is_patch(op):
            fwd = op.kind in (H.OP_CONV_FWD, H.OP_CONV_FWD_GN, H.OP_CONV_FWD_SKIP)
            wl = op.i[6] if fwd else op.i[2] * (2 if op.i[12] else 1)
            if op.i[8] != 3 or wl == 8:               # 8-pixel-wide layers run the image-pair instantiation: a different kernel symbol
                return False
            return op.kind in (H.OP_CONV_FWD_GN, H.OP_CONV_FWD_SKIP) or (op.kind == H.OP_CONV_FWD and bool(op.p[6])) or (op.kind == H.OP_CONV_DGRAD and bool(op.p[3]))

        def patch_bytes(op):                          # algorithmic HBM bytes of one launch: input + output (+ residual) once, fp32
            i = op.i
            N, Hi, Wi, Cin, Ho, Wo, Cout, up = i[0], i[1], i[2], i[3] + i[4], i[5], i[6], i[7], i[12]
            if op.kind == H.OP_CONV_FWD_SKIP:
                return 4.0 * N * (Hi * Wi * Cin + Ho * Wo * (Cout + i[15] + i[16])) + 6.0 * Cout * (9 * Cin + i[15] + i[16])
            if op.kind in (H.OP_CONV_FWD, H.OP_CONV_FWD_GN):
                has_res = bool(op.p[4]) if op.kind == H.OP_CONV_FWD else bool(op.p[5])
                return 4.0 * N * (Hi * Wi * Cin + Ho * Wo * Cout * (2 if has_res else 1)) + 6.0 * Cout * 9 * Cin
            s_ = 2 if up else 1
            # data gradient; with the GroupNorm-backward sums in its epilogue (op_conv_dgrad(gnb=...): i[20] == 1) the launch also reads the
            # GroupNorm's raw input x = [x0 | x1] once -- as many floats as the dX it writes (the figure did not count it)
            gb = 1 if (op.kind == H.OP_CONV_DGRAD and i[20] == 1) else 0
            return 4.0 * N * (Ho * Wo * Cout + Hi * s_ * Wi * s_ * Cin * (1 + gb)) + 6.0 * Cout * 9 * Cin

        dense = getattr(st.plan, "dense_grid", set())    # stride-2 convs run as stride-1 launches on a 75 %-zero grid: not part of the kernel's roofline set
        pk = [k for k in range(st.n_bwd) if is_patch(st.plan.arr[k]) and k not in dense]
        kname = ("conv3x3y_kernel + conv3x3r_kernel + conv3x3p_kernel (3x3 conv forward + data gradient, LDS-patch kernels: Winograd F(2,3)-along-x persistent form on "
                 "chip-filling layers, direct persistent / deferred-epilogue form for launches with fused skip chunks, two-waves-per-SIMD form on the small layers)")
        if not pk:                                    # f32 mode: every convolution runs on the generic f32-MFMA implicit GEMM
            pk = [k for k in range(st.n_bwd) if fl[k] > 0 and st.plan.arr[k].kind != H.OP_GEMM]
            kname = "igemm_kernel (generic implicit-GEMM convolution, f32 MFMA)"
        p_ms, p_fl = sum(durs[k] for k in pk), sum(fl[k] for k in pk)
        if math != "f32":                             # peak of the executed mix: dense low-precision MFMA peak / average MFMAs per product
            peak = PEAK_BF16_MFMA_TFLOPS * p_fl / max(sum(fl[k] * op_npm(st.plan.arr[k]) for k in pk), 1.0)
        fam_peak = peak if math == "f32" else PEAK_BF16_MFMA_TFLOPS * ig_fl / max(sum(f * (op_npm(st.plan.arr[k]) if st.plan.arr[k].kind != H.OP_GEMM else 16)
                                                                                      for k, f in enumerate(fl) if f > 0), 1.0)
        p_by = sum(patch_bytes(st.plan.arr[k]) for k in pk)
        traffic, traffic_src = None, None
        try:                                          # PMC counters cannot be read from inside the process: committed rocprofv3 --pmc passes
            prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
            pmc_file = next(f for f in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json") if os.path.exists(os.path.join(prof, f)))
            pmc = json.load(open(os.path.join(prof, pmc_file)))
            kk = [v for k_, v in pmc["kernels"].items()
                  if k_.startswith("void conv3x3y_kernel<") or k_.startswith("void conv3x3r_kernel<") or (k_.startswith("void conv3x3p_kernel<") and ", 8, false" in k_)]   # every non-pair instantiation
            wk_pmc = [v for k_, v in pmc["kernels"].items() if k_.startswith("void conv3x3v_kernel<") or (k_.startswith("void conv3x3w_kernel<") and "false" in k_)]
            if kk and pmc.get("math", "bf16x6") == math:
                traffic = round(sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in kk) / sum(v["dispatches"] for v in kk))
                traffic_src = f"profiles/{pmc_file}: committed rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE, gfx950 x2 fetch correction) of this command -- PMC counters cannot be read inside the timed process"
        except (OSError, ValueError, KeyError, StopIteration):
            pass
        # whole step against both floors (north_star quotes its target on the HBM one; the step is compute-bound: DESIGN.md section 5)
        step_fl = TRAIN_GFLOP_PER_IMG * B * 1e9
        out["step_mfma_roofline_frac"] = round(step_fl * (npm if math != "f32" else 1) / ((PEAK_BF16_MFMA_TFLOPS if math != "f32" else PEAK_F32_MFMA_TFLOPS) * 1e12) / (ms_step * 1e-3), 4)
        out["step_hbm_roofline_frac"] = round((HBM_GB_PER_IMG * B + HBM_GB_PER_STEP) / (PEAK_HBM_TBS * 1e3) / (ms_step * 1e-3), 4)
        out["step_roofline_note"] = (f"floors at B={B}: MFMA {step_fl * npm / PEAK_BF16_MFMA_TFLOPS / 1e9:.1f} ms ({TRAIN_GFLOP_PER_IMG * B / 1e3:.2f} TFLOP x {npm} MFMA products / 2.5 PFLOP/s), "
                                     f"HBM {(HBM_GB_PER_IMG * B + HBM_GB_PER_STEP) / PEAK_HBM_TBS:.1f} ms ({HBM_GB_PER_IMG * B + HBM_GB_PER_STEP:.1f} GB / 8 TB/s)")
        # 3x3 weight-gradient kernel: same figures (algorithmic bytes: X and dY read once, dW written once)
        wk = [k for k in range(st.n_bwd) if st.plan.arr[k].kind == H.OP_CONV_WGRAD and st.plan.arr[k].i[8] == 3 and bool(st.plan.arr[k].p[6]) and k not in dense]
        if wk:
            w_ms, w_fl = sum(durs[k] for k in wk), sum(fl[k] for k in wk)
            w_by = sum(4.0 * st.plan.arr[k].i[0] * (st.plan.arr[k].i[1] * st.plan.arr[k].i[2] * (st.plan.arr[k].i[3] + st.plan.arr[k].i[4]) + st.plan.arr[k].i[5] * st.plan.arr[k].i[6] * st.plan.arr[k].i[7])
                       + 4.0 * st.plan.arr[k].i[7] * 9 * (st.plan.arr[k].i[3] + st.plan.arr[k].i[4]) for k in wk)
            w_tr = None
            try:
                w_tr = round(sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in wk_pmc) / sum(v["dispatches"] for v in wk_pmc)) if wk_pmc else None
            except NameError:
                pass
            out["roofline_wgrad"] = {"bound": "mfma", "kernel": "conv3x3v_kernel (3x3 weight gradient, producer / consumer waves, transposing LDS reads; conv3x3w_kernel on the 8-pixel-wide and narrow layers)", "achieved": round(w_fl / w_ms / 1e9, 2),
                                     "peak": round(PEAK_BF16_MFMA_TFLOPS / 3, 1), "unit": "TFLOP/s", "frac": round(w_fl / w_ms / 1e9 / (PEAK_BF16_MFMA_TFLOPS / 3), 4),
                                     "launches_per_step": len(wk), "avg_launch_ms": round(w_ms / len(wk), 4), "kernel_ms_per_step": round(w_ms, 3),
                                     "algorithmic_bytes_per_launch": round(w_by / len(wk)), "traffic": w_tr}
        # the HBM-bound convolution kernel: 1x1 forward launches on the prepared-weight path (conv1x1_kernel), algorithmic bytes = every operand once
        ck = [k for k in range(st.n_bwd) if st.plan.arr[k].kind == H.OP_CONV_FWD and st.plan.arr[k].i[8] == 1 and bool(st.plan.arr[k].p[6])]
        if ck:
            def point_bytes(op):
                i = op.i
                return 4.0 * i[0] * (i[1] * i[2] * (i[3] + i[4]) + i[5] * i[6] * i[7] * (2 if bool(op.p[4]) else 1)) + 4.0 * i[7] * (i[3] + i[4])
            c_ms, c_by = sum(durs[k] for k in ck), sum(point_bytes(st.plan.arr[k]) for k in ck)
            kb = max(ck, key=lambda k: point_bytes(st.plan.arr[k]))
            c_tr = None
            try:
                cv = [v for k_, v in pmc["kernels"].items() if k_.startswith("void conv1x1_kernel<")]
                c_tr = round(sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in cv) / sum(v["dispatches"] for v in cv)) if cv else None
            except (NameError, KeyError, ZeroDivisionError):
                pass
            out["roofline_1x1"] = {"bound": "hbm", "kernel": "conv1x1_kernel, forward launches (ResBlock skip convolutions, attention qkv / proj_out)", "achieved": round(c_by / c_ms / 1e6, 1),
                                   "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s", "frac": round(c_by / c_ms / 1e6 / (PEAK_HBM_TBS * 1e3), 4), "launches_per_step": len(ck),
                                   "avg_launch_ms": round(c_ms / len(ck), 4), "kernel_ms_per_step": round(c_ms, 3), "algorithmic_bytes_per_launch": round(c_by / len(ck)),
                                   "traffic": c_tr, "traffic_note": "PMC average over ALL conv1x1_kernel dispatches (forward and data gradient)",
                                   "largest_launch": {"bytes": round(point_bytes(st.plan.arr[kb])), "ms": round(durs[kb], 4),
                                                      "gb_per_s": round(point_bytes(st.plan.arr[kb]) / durs[kb] / 1e6, 1)}}
        out["roofline"] = {"bound": "mfma", "kernel": kname,
                           "math": math, "achieved": round(p_fl / p_ms / 1e9, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                           "frac": round(p_fl / p_ms / 1e9 / peak, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)",
                           "traffic_source": traffic_src, "traffic_measured_in_this_run": False, "algorithmic_bytes_per_launch": round(p_by / max(len(pk), 1)),
                           "peak_note": "algorithmic TFLOP/s; peak = dense 16-bit MFMA peak (2500) / average MFMAs issued per algorithmic product of these launches (direct f16x3: 3; Winograd-along-x f16x3: 2)",
                           "launches_per_step": len(pk), "avg_launch_ms": round(p_ms / max(len(pk), 1), 4),
                           "algorithmic_gflop_per_launch": round(p_fl / 1e9 / max(len(pk), 1), 2), "kernel_ms_per_step": round(p_ms, 3),
                           "frac_of_f32_mfma_peak": round(p_fl / p_ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4),
                           "heaviest_launch": {"gflop": round(fl[kbig] / 1e9, 2), "ms": round(durs[kbig], 4),
                                               "tflops": round(fl[kbig] / durs[kbig] / 1e9, 2)},
                           "family": {"kernels": "conv3x3y + conv3x3r + conv3x3p + conv3x3v + conv3x3w + conv1x1 + igemm(_bf) (every conv fwd/dgrad/wgrad and dense GEMM)",
                                      "achieved": round(ig_fl / ig_ms / 1e9, 2), "frac": round(ig_fl / ig_ms / 1e9 / fam_peak, 4), "peak": round(fam_peak, 1),
                                      "launches_per_step": n_ig, "algorithmic_gflop_per_step": round(ig_fl / 1e9, 1),
                                      "ms_per_step": round(ig_ms, 3), "all_ops_ms_per_step": round(sum(durs), 3)}}
        st.micro = 0
        log("per-op profile done")
        # ---- DDIM-100 sampling throughput (second half of the BASELINE metric): 100 ShiftUNet forwards + fused updates
        if not args.no_ddim:
            dec.set_eval_mode()
            out["ddim100"] = {}
            with torch.no_grad():
                for Bd in sorted({args.ddim_batch or B, 128} if args.ddim_batch == 100 else {args.ddim_batch or B}):
                    z = torch.randn(Bd, 512, device=dev)
                    xT = torch.randn(Bd, 3, size, size, device=dev)
                    gd.representation_learning_ddim_sample("ddim10", None, dec, None, xT, z)       # builds the inference plan
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    gd.representation_learning_ddim_sample("ddim100", None, dec, None, xT, z)
                    torch.cuda.synchronize()
                    dd = time.perf_counter() - t1
                    rec = {"samples_per_sec": round(Bd / dd, 3), "batch": Bd, "seconds": round(dd, 3),
                           "tflops_algorithmic": round(FWD_GFLOP_PER_IMG * Bd * 100 / dd / 1e3, 2)}
                    if Bd == (args.ddim_batch or B):
                        out["ddim100"].update(rec)                     # headline: the evaluator's batch (100)
                    else:
                        out["ddim100"][f"batch_{Bd}"] = rec
                    dec.invalidate_plans()
            # the evaluator's protocol (sampler/autoencoding_eval.py:74-78: encoder -> 999-step DDIM inversion -> 100-step decode) in every line:
            # 50 steps of the ddim1000 inversion are TIMED at the evaluator's batch and the 1099 decoder passes are EXTRAPOLATED from them and
            # from the ddim100 decode measured above (labelled so; --autoencode times the whole protocol, ~90 s)
            with torch.no_grad():
                Bd = args.ddim_batch or B
                xa = torch.rand(Bd, 3, size, size, device=dev) * 2 - 1
                z = torch.randn(Bd, 512, device=dev)
                d1000 = gd._ddim("ddim1000")
                d1000._planned_loop(dec, xa, range(0, 2), True, z=z)                                 # plan
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                d1000._planned_loop(dec, xa, range(0, 50), True, z=z)
                torch.cuda.synchronize()
                enc_step = (time.perf_counter() - t1) / 50
                out["ddim100"]["ddim1000_encode_ms_per_step"] = round(enc_step * 1e3, 3)
                out["ddim100"]["ddim1000_encode_steps_timed"] = 50
                out["ddim100"]["autoencode_1099_steps_seconds_extrapolated"] = round(999 * enc_step + out["ddim100"]["seconds"], 2)
                out["ddim100"]["autoencode_images_per_sec_extrapolated"] = round(Bd / (999 * enc_step + out["ddim100"]["seconds"]), 3)
                dec.invalidate_plans()
            if args.autoencode:                       # the evaluator's protocol on one batch: encoder -> 999-step DDIM inversion -> 100-step decode
                with torch.no_grad():
                    Bd = args.ddim_batch or B
                    xa = torch.rand(Bd, 3, size, size, device=dev) * 2 - 1
                    enc.eval()
                    gd.representation_learning_autoencoding("ddim10", "ddim10", enc, dec, xa)      # plans
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    gd.representation_learning_autoencoding("ddim1000", "ddim100", enc, dec, xa)
                    torch.cuda.synchronize()
                    out["ddim100"]["autoencode_1099_steps_seconds"] = round(time.perf_counter() - t1, 2)
                    out["ddim100"]["autoencode_batch"] = Bd
                    dec.invalidate_plans()
            log("ddim100 done")
            if world == 1 and not args.no_cpu_baseline:          # what the same denoising step costs on the host cores (oracle decoder forward)
                torch.set_num_threads(host_cores())
                sec = cpu_decoder_forward(2)
                out["ddim100"]["cpu_baseline"] = {"value": round(2 / (100 * sec), 5), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                                                  "sample": f"oracle ShiftUNet forward at batch 2, median of 3 ({sec:.2f} s) x 100 steps"}
        if not args.no_legs:
            # BASELINE.md section 3 report items: bf16 operands (optimizer_config.enable_amp) on the headline config, and config #2 (CelebA-64, bf16)
            del st
            dec.invalidate_plans(); enc.invalidate_plans()
            torch.cuda.empty_cache()
            out["other_legs"] = [train_leg(CONFIG, "bf16", 6, dev), train_leg("config/celeba64_representation_learning.yml", "bf16", 10, dev),
                                 train_leg("config/celeba64_representation_learning.yml", None, 10, dev)]
            log("other legs done")
        if world == 1 and not args.no_cpu_baseline:
            torch.set_num_threads(host_cores())
            log(f"cpu baseline on {host_cores()} host cores")
            ips, sec = cpu_baseline(args.cpu_batch, 5, warm=3)
            out["cpu_baseline"] = {"value": round(ips, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"same FFHQ-128 train step (fwd+bwd+Adam+EMA, dropout off), batch {args.cpu_batch}, "
                                             f"median of 5 timed steps after 3 warm-up ({sec:.1f} s/step)"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
