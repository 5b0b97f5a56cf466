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
    for flag in ("--gpus", "--steps", "--warmup", "--math", "--ddim-batch", "--no-cpu-baseline"):
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
