"""CPU: bench.py and the trainer entry points must at least parse their command lines here (no GPU): a broken argparse block would
otherwise only show up on the GPU box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _help(args):
    r = subprocess.run([sys.executable] + args + ["--help"], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_bench_cli_parses():
    out = _help(["bench.py"])
    for flag in ("--gpus", "--steps", "--warmup", "--math", "--ddim-batch", "--no-cpu-baseline", "--bucket-mb", "--dry"):
        assert flag in out


def test_trainer_clis_parse():
    for mod in ("pdae_amd.trainer.train_representation_learning", "pdae_amd.trainer.train_regular_diffusion", "pdae_amd.trainer.train_latent_diffusion"):
        out = _help(["-m", mod])
        assert "--config_path" in out and "--run_path" in out


def test_bench_flop_model_matches_survey():
    """The algorithmic FLOP constants the throughput numbers are derived from (481.4 GFLOP/img train, 258.4 forward)."""
    sys.path.insert(0, ROOT)
    import bench
    assert abs(bench.TRAIN_GFLOP_PER_IMG - 481.4) < 1.0 and abs(bench.FWD_GFLOP_PER_IMG - 258.4) < 1.0
    assert bench.MFMA_PER_PRODUCT["f16x3"] == 3 and bench.MFMA_PER_PRODUCT["bf16x6"] == 6


def test_bench_self_launches_two_ranks_dry():
    """`python bench.py --gpus 2` with no launcher around it (how the driver may call it): bench.py re-execs itself under torch.distributed.run,
    one rank per GPU; --dry swaps RCCL for gloo and the kernels for a recorder so that the whole path -- launcher, process group, bucketed
    all-reduce, bucket sweep, max-over-ranks timing, ONE JSON line from rank 0 -- runs in this container."""
    import json
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry", "--steps", "2", "--warmup", "1"], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry"] is True and out["steps"] == 2 and out["scaling"] == "weak"
    c = out["comm"]
    assert c["rccl_ranks"] == 2 and c["buckets"] >= 2 and sum(c["bucket_bytes"]) == c["grad_bytes_per_step"]
    # default: a fixed 48 MB bucket and NO sweep (an 8-rank driver run reaches its timed region right behind the warm-up steps);
    # every rank's own clock is reported, and which exchange path ran
    assert c["bucket_mb"] == 48.0 and c["bucket_sweep_ms_per_step"] is None
    assert len(c["per_rank_ms_per_step"]) == 2 and all(v > 0 for v in c["per_rank_ms_per_step"]) and c["exchange_path"].startswith("bucketed")
    assert abs(out["ms_per_step"] - max(c["per_rank_ms_per_step"])) < 1e-2
    assert out["config"]["global_batch"] == 2 * out["config"]["per_gpu_batch"]
    # round 6: the A/B of the conv3x3y grid trim behind the timed region (PDAE_Y_GRID_TRIM; control flow only in a dry run), knob restored to 0
    assert set(c["y_grid_trim_ms_per_step"]) == {"0", "8"} and c["y_grid_trim"] == 0 and "y_grid_trim_error" not in c
    # --bucket-mb 0: the sweep over 16 / 48 / 96 MB runs first
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry", "--steps", "1", "--warmup", "1", "--bucket-mb", "0"], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    c = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["comm"]
    assert set(c["bucket_sweep_ms_per_step"]) == {"16", "48", "96"} and c["bucket_mb"] in (16.0, 48.0, 96.0)


def test_bench_self_launches_eight_ranks_dry():
    """the driver's 8-GPU run is the first time RCCL sees more than one rank, so everything around it runs here at the REAL world size:
    eight gloo ranks, the bucket boundaries of the dry network, the 1 / 8 fold into Adam (rank 0 checks the op record), max-over-ranks timing, and the
    per-bucket report (enqueue -> complete as the compute stream sees it, what was still running when the backward had ended, overlap fraction)."""
    import json
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--dry", "--steps", "2", "--warmup", "1"], cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["dry"] is True and out["scaling"] == "weak" and out["config"]["global_batch"] == 8 * out["config"]["per_gpu_batch"]
    c = out["comm"]
    assert c["rccl_ranks"] == 8 and len(c["per_rank_ms_per_step"]) == 8 and abs(out["ms_per_step"] - max(c["per_rank_ms_per_step"])) < 1e-2
    assert c["buckets"] >= 2 and sum(c["bucket_bytes"]) == c["grad_bytes_per_step"] and c["exchange_path"].startswith("bucketed")
    assert len(c["bucket_enqueue_to_complete_ms"]) == c["buckets"] == len(c["bucket_complete_after_backward_ms"])
    assert all(v >= 0 for v in c["bucket_enqueue_to_complete_ms"]) and 0.0 <= c["overlap_frac"] <= 1.0
    assert c["allreduce_ms_per_step"] > 0 and c["allreduce_exposed_ms_per_step"] >= 0 and c["adam_grad_scale"] == 0.125
    # the timed region starts right behind the warm-up steps: no sweep, no extra steps in front of it (stderr timeline)
    assert r.stderr.count("warm-up step done") == 8 * 1


def test_bench_workload_comes_from_the_yaml_files():
    sys.path.insert(0, ROOT)
    import bench
    cfg, ddpm = bench.load_workload()
    assert cfg["dataloader_config"]["train"]["batch_size"] == 32 and cfg["train_dataset_config"]["image_size"] == 128
    assert ddpm["base_channel"] == 128 and ddpm["channel_multiplier"] == [1, 1, 2, 3, 4]
    assert "FFHQ128 = dict" not in open(os.path.join(ROOT, "bench.py")).read()
