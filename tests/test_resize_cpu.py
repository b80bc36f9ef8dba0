"""CPU: the image-resize restatement (oracle/pil_resize.py) against Pillow itself (the third-party library the reference's
processor calls; present in this image) and against golden vectors from the reference's own processor (G12); the product's
host-side coefficient tables (theia_amd/preprocess.py) against the restatement."""
import os

import numpy as np
import pytest
import torch

from oracle import pil_resize as R
from oracle import theia_oracle as O

SIZES = [(50, 37), (300, 200), (224, 301), (301, 224), (448, 448), (17, 400), (225, 223), (1000, 750), (7, 9)]


@pytest.mark.parametrize("resample", [R.BILINEAR, R.BICUBIC])
def test_restatement_matches_pillow(resample):
    Image = pytest.importorskip("PIL.Image")
    pil_mode = {R.BILINEAR: Image.BILINEAR, R.BICUBIC: Image.BICUBIC}[resample]
    rng = np.random.default_rng(5)
    for h, w in SIZES:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for oh, ow in ((224, 224), (256, 256), (h, 224), (224, w), (16, 16)):
            ref = np.asarray(Image.fromarray(img).resize((ow, oh), resample=pil_mode))
            assert np.array_equal(R.resize_u8(img, oh, ow, resample), ref), (h, w, oh, ow)


def test_restatement_matches_reference_processor_g12(golden_dir):
    g = np.load(os.path.join(golden_dir, "g12_resize_processor.npz"))
    rows = g["rows"]
    lut = O.preprocess_lut(True, True)
    for name in ("up_hwc", "down_chw", "xonly_hwc", "yonly_chw"):
        x = g[f"{name}_img"]
        hwc = x if name.endswith("hwc") else np.transpose(x, (0, 2, 3, 1))
        out = np.stack([R.resize_u8(np.ascontiguousarray(im), 224, 224, R.BILINEAR) for im in hwc])
        assert np.array_equal(out[:, rows], g[f"{name}_resized_rows"]), name
        assert int(out.astype(np.int64).sum()) == int(g[f"{name}_resized_sum"])
        # the processor's rescale + normalise on top of the resized image == the 3x256 table the GPU ingest uses
        pv = np.stack([lut[c][out[..., c]] for c in range(3)], -1)
        assert np.array_equal(pv[:, rows[::4]], g[f"{name}_pv_rows"]), name


@pytest.mark.parametrize("resample", [R.BILINEAR, R.BICUBIC])
def test_product_tables_match_restatement(resample):
    from theia_amd.preprocess import resample_tables
    for i in list(range(1, 40)) + [100, 223, 224, 225, 256, 300, 448, 750, 1000, 1920]:
        for o in (224, 256, 16, i):
            b0, k0, ks0 = R.precompute_coeffs(i, o, resample)
            b1, k1, ks1 = resample_tables(i, o, resample)
            assert ks0 == ks1 and np.array_equal(b0, b1) and np.array_equal(k0, k1), (i, o)


def test_deit_hub_processor_pipeline_g14(golden_dir):
    """resize 256 bicubic + center-crop 224 + rescale/normalise (the facebook/deit-*-patch16-224 preprocessor_config) as the
    reference's processor class computes it, against the restatement (resize -> crop -> table)."""
    g = np.load(os.path.join(golden_dir, "g14_deit_processor.npz"))
    rows = g["rows"]
    lut = O.preprocess_lut(True, True)
    for name, img in (("in224", O.synth_images(2, 0).numpy()), ("in300x260", g["in300x260_img"])):
        out = np.stack([R.resize_u8(np.ascontiguousarray(im), 256, 256, R.BICUBIC)[16:240, 16:240] for im in img])
        assert np.array_equal(out[:, rows], g[f"{name}_u8_rows"]), name
        assert int(out.astype(np.int64).sum()) == int(g[f"{name}_u8_sum"])
        pv = np.stack([lut[c][out[..., c]] for c in range(3)], -1)
        assert np.array_equal(pv[:, rows[::4]], g[f"{name}_pv_rows"]), name
