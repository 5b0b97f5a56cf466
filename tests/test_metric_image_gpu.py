"""GPU: the fused evaluator (pdae_ssim_mse) and the device input pipeline (pdae_image_prepare + DeviceImagePipeline).

SSIM / MSE: against the vectors the reference emitted (tests/golden/misc.npz) and the CPU oracle, 1e-5 absolute on SSIM (fp32 window sums).
Resize / flip / normalise: BIT-EXACT against Pillow itself (the reference's transforms.Resize runs Pillow's 8-bit BILINEAR resampler) and
against the oracle's restatement; byte work has no tolerance."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import load_golden, T, rel_err
from oracle import pdae_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_ssim_mse_vs_reference_vectors_and_oracle():
    from pdae_amd.metric import calculate_ssim, calculate_mse, ssim_mse
    g = load_golden("misc")
    a, b = T(g["m_a"]).to(DEV), T(g["m_b"]).to(DEV)
    assert np.allclose(calculate_ssim(a, b).cpu().numpy(), g["ssim"], atol=1e-5, rtol=0)
    assert np.allclose(calculate_mse(a, b).cpu().numpy(), g["mse"], rtol=1e-5, atol=0)
    # evaluator sizes and ragged tiles, correlated images (SSIM near 1, where the 3-decimal protocol lives), NHWC memory, fused de-normalisation
    for (N, C, Hh, W) in [(3, 3, 128, 128), (2, 1, 40, 56), (1, 3, 16, 16), (2, 3, 33, 65)]:
        gen = torch.Generator().manual_seed(N * 1000 + W)
        x = torch.rand(N, C, Hh, W, generator=gen) * 2 - 1
        y = (x + 0.05 * torch.randn(N, C, Hh, W, generator=gen)).clamp(-1, 1)
        xn, yn = (x + 1) / 2, (y + 1) / 2
        ref_s, ref_m = O.ssim(xn, yn), O.mse(xn, yn)
        xd = x.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)          # what the DDIM loop returns: NHWC memory, NCHW shape
        s, m = ssim_mse(xd, y.to(DEV), denormalize=True)
        assert np.allclose(s.cpu().numpy(), ref_s.numpy(), atol=2e-5, rtol=0), (N, C, Hh, W)
        assert np.allclose(m.cpu().numpy(), ref_m.numpy(), rtol=1e-4, atol=1e-10)
        s2 = calculate_ssim(xn.to(DEV), yn.to(DEV))
        assert np.allclose(s2.cpu().numpy(), ref_s.numpy(), atol=2e-5, rtol=0)
    ident = torch.rand(2, 3, 32, 32, device=DEV)
    s, m = ssim_mse(ident, ident)
    assert float((s - 1).abs().max()) < 1e-6 and float(m.abs().max()) == 0.0


def _prepare(images, size, crop=None, flips=None, nhwc_out=False):
    import ctypes
    from pdae_amd import hip as H
    from pdae_amd.dataset.resample import bilinear_coefficients
    B, Hs, Ws, C = images.shape
    cy, cx, ch, cw = crop if crop else (0, 0, Hs, Ws)
    kx, bx = bilinear_coefficients(cw, size)
    ky, by = bilinear_coefficients(ch, size)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    src, kxd, bxd, kyd, byd = d(images), d(kx), d(bx), d(ky), d(by)
    fl = d(np.asarray(flips, dtype=np.uint8)) if flips is not None else None
    buf = torch.empty(B, size, size, C, device=DEV) if nhwc_out else torch.empty(B, C, size, size, device=DEV)
    x0 = buf.permute(0, 3, 1, 2) if nhwc_out else buf
    gts = torch.empty(B, size, size, C, dtype=torch.uint8, device=DEV)
    L = H.lib()
    ws = torch.empty(int(L.pdae_image_prepare_workspace_bytes(B, ch, size, C)) + 16, dtype=torch.uint8, device=DEV)
    rc = L.pdae_image_prepare(src.data_ptr(), B, Hs, Ws, C, cy, cx, ch, cw, size, kxd.data_ptr(), bxd.data_ptr(), kx.shape[1], kyd.data_ptr(), byd.data_ptr(),
                              ky.shape[1], fl.data_ptr() if fl is not None else None, x0.data_ptr(), (ctypes.c_int64 * 4)(*x0.stride()), gts.data_ptr(),
                              ws.data_ptr(), ctypes.c_void_p(H.current_stream_ptr()))
    assert rc == 0, L.pdae_last_error()
    torch.cuda.synchronize()
    return x0, gts


@pytest.mark.parametrize("case", [(4, 256, 256, 3, 128, None), (3, 218, 178, 3, 64, (57, 25, 128, 128)), (2, 28, 28, 1, 32, None), (2, 100, 75, 3, 32, None),
                                  (1, 128, 128, 3, 128, None)])
def test_image_prepare_bit_exact_vs_pillow_and_oracle(case):
    from PIL import Image
    B, Hs, Ws, C, S, crop = case
    rng = np.random.default_rng(B * 100 + S)
    imgs = rng.integers(0, 256, (B, Hs, Ws, C), dtype=np.uint8)
    flips = [int(v) for v in rng.integers(0, 2, B)]
    x0, gts = _prepare(imgs, S, crop, flips, nhwc_out=(B % 2 == 0))
    ref_x, ref_g = O.image_batch(imgs, S, crop, flips)
    assert np.array_equal(gts.cpu().numpy(), ref_g)
    assert torch.equal(x0.cpu(), ref_x)                      # (v/255 - 0.5)/0.5 in float32 on both sides
    for b in range(B):                                       # and Pillow itself: crop -> resize(BILINEAR) -> flip
        pil = Image.fromarray(imgs[b] if C == 3 else imgs[b, :, :, 0])
        if crop:
            pil = pil.crop((crop[1], crop[0], crop[1] + crop[3], crop[0] + crop[2]))
        r = np.asarray(pil.resize((S, S), Image.BILINEAR)).reshape(S, S, C)
        assert np.array_equal(gts[b].cpu().numpy(), r[:, ::-1] if flips[b] else r), b


def test_device_image_pipeline_contract_and_trainer_consumes_it(tmp_path):
    """{"idx","x_0","gts"} batches (dataset/ffhq.py:55-74) out of .npy shards through the double-buffered H2D path; the fused training step takes them."""
    import copy
    from pdae_amd import dataset as D
    from pdae_amd.diffusion.gaussian_diffusion import GaussianDiffusion
    from pdae_amd.model.shift_unet import ShiftUNet
    from pdae_amd.model.representation_learning.encoder import CELEBA64Encoder
    from pdae_amd.trainer.fused_step import FusedRLStep
    from tests.golden import make_fixtures_cfg as C
    rng = np.random.default_rng(1)
    shards = [rng.integers(0, 256, (n, 96, 96, 3), dtype=np.uint8) for n in (5, 7)]
    for k, s in enumerate(shards):
        np.save(os.path.join(tmp_path, f"part{k:02d}.npy"), s)
    allimg = np.concatenate(shards)
    ds = D.build({"name": "FFHQ", "data_path": str(tmp_path), "image_size": 64, "image_channel": 3, "augmentation": True}, device=DEV, seed=3)
    assert len(ds) == 12
    seen = []
    for it in range(5):                                      # 4 batches per epoch of 12 (drop_last at B=3): the 5th comes from epoch 1, slots are reused
        b = ds.batch(3, DEV)
        assert set(b) == {"idx", "x_0", "gts"} and b["x_0"].shape == (3, 3, 64, 64) and b["gts"].shape == (3, 64, 64, 3) and b["gts"].dtype == torch.uint8
        torch.cuda.synchronize()
        ids = b["idx"].tolist()
        seen.append(ids)
        for j, i in enumerate(ids):
            r = O.resize_u8(allimg[i], 64)
            got = b["gts"][j].cpu().numpy()
            flipped = np.array_equal(got, r[:, ::-1])
            assert flipped or np.array_equal(got, r), (it, j, i)
            ref_x = (torch.from_numpy((r[:, ::-1] if flipped else r).copy()).float().div(255).permute(2, 0, 1) - 0.5) / 0.5
            assert torch.equal(b["x_0"][j].cpu(), ref_x)
    assert sorted(sum(seen[:4], [])) == list(range(12))     # one epoch = a permutation
    assert seen[4] != seen[0]                                # reshuffled
    # the training step consumes a pipeline batch directly
    dev = torch.device(DEV)
    cfg = dict(C.CFG_SHIFT_64, dropout=0.0)
    enc, dec = CELEBA64Encoder(device=dev, latent_dim=512), ShiftUNet(device=dev, latent_dim=512, **cfg)
    with torch.no_grad():
        for p in dec.parameters():
            if float(p.abs().max()) == 0.0:
                p.normal_(0, 0.05)
    enc.train(); dec.set_train_mode()
    st = FusedRLStep(GaussianDiffusion({"timesteps": 1000, "betas_type": "linear"}, dev), enc, dec, copy.deepcopy(enc), copy.deepcopy(dec), 3, 64, 64)
    w0 = dec.flat_train.clone()
    loss = float(st.step(ds.batch(3, DEV)["x_0"]).item())
    assert np.isfinite(loss) and 0 < loss < 10 and not torch.equal(w0, dec.flat_train)
