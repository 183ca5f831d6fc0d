"""Training input pipeline (dataloader.py:100-216, 243-283): the oracle restatement against the
reference's own outputs (tests/golden/dataloader.npz), and the HIP kernels / host mirror against the oracle."""
import random

import numpy as np
import pytest
import torch

from oracle import resunet_oracle as O


def _case():
    return O.make_dataloader_case(77)


def test_oracle_zscore_bbox_and_full_reader(golden):
    g = golden("dataloader")
    image, label = _case()
    norm, mean, std = O.zscore_positive(image)
    np.testing.assert_allclose(norm[:, ::3, ::3, ::3].astype(np.float32), g["image_norm_sub"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(O.label_bbox(label, tuple(g["patch"])), g["bbox"])
    data, tgt = O.full_volume_item(image[:, :38, :41, :36], label[:38, :41, :36])
    assert tuple(data.shape) == tuple(g["full_shape"])
    np.testing.assert_allclose(data[:, ::2, ::2, ::2], g["full_data_sub"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(np.packbits(tgt.astype(np.uint8)), g["full_target_bits"])


@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_oracle_augment_matches_reference(golden, k):
    """same seeds for `random` and `numpy.random` -> same draws (order of dataloader.py:141-199) -> same patch"""
    g = golden("dataloader")
    image, label = _case()
    norm, _, _ = O.zscore_positive(image)
    patch = tuple(int(v) for v in g["patch"])
    seed = int(g["seed%d" % k])
    random.seed(seed)
    np.random.seed(seed)
    p = O.draw_augment_params(g["bbox"], patch)
    data, tgt = O.augment_patch(norm, label, p["crop_lo"], patch, p["scale"], p["flips"], p["transpose"], p["gain"], p["bias"])
    np.testing.assert_allclose(data, g["data%d" % k], rtol=0, atol=2e-6)
    np.testing.assert_allclose(tgt, g["target%d" % k], rtol=0, atol=1e-6)


def test_zoom_closed_form_is_scipy():
    """the closed form the kernels implement == scipy.ndimage.affine_transform(order=1, mode='reflect') (third-party, 1.15.3 here)"""
    import warnings
    from scipy.ndimage import affine_transform
    v = np.random.default_rng(3).standard_normal((2, 7, 9, 8))
    for scale in ((0.7, 1.0, 1.3), (1.29, 0.71, 0.95)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = affine_transform(v, (1,) + scale, order=1, mode="reflect")
        np.testing.assert_allclose(O.zoom_linear_reflect(v, scale), ref, rtol=0, atol=1e-13)


# ---------------------------------------------------------------------- HIP
@pytest.mark.gpu
def test_hip_zscore_stats():
    from brats2019_amd import dataloader as DL
    image, _ = _case()
    _, mean, std = O.zscore_positive(image)
    m, s = DL.zscore_stats(torch.from_numpy(image).cuda())
    np.testing.assert_allclose(m, mean, rtol=1e-12)
    np.testing.assert_allclose(s, std, rtol=1e-8)          # the reference divides every element by the count before summing


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_hip_augment_matches_reference(golden, k):
    """device pipeline (z-score on the fly, crop, zoom, flips, transpose, intensity, targets) vs the reference's patches;
    fp32 on the device vs float64 in the reference: tolerance 2e-5 on z-scored intensities, 2e-6 on the soft targets"""
    from brats2019_amd import dataloader as DL
    g = golden("dataloader")
    image, label = _case()
    patch = tuple(int(v) for v in g["patch"])
    vol = DL.DeviceCase(image, label, patch)
    np.testing.assert_array_equal(vol.bbox, g["bbox"])
    seed = int(g["seed%d" % k])
    random.seed(seed)
    np.random.seed(seed)
    p = DL.draw_augment_params(vol.bbox, patch)
    data, tgt = DL.augment_patch(vol, p)
    assert data.is_cuda and tuple(data.shape) == (4,) + patch and tuple(tgt.shape) == (3,) + patch
    np.testing.assert_allclose(data.cpu().numpy(), g["data%d" % k], rtol=0, atol=2e-5)
    np.testing.assert_allclose(tgt.cpu().numpy(), g["target%d" % k], rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_hip_augment_vs_oracle_rectangular_patch():
    """non-cubic patch, every flip / transpose combination, against the oracle on the same explicit parameters"""
    from brats2019_amd import dataloader as DL
    image, label = _case()
    norm, _, _ = O.zscore_positive(image)
    patch = (16, 16, 24)
    vol = DL.DeviceCase(image, label, patch)
    r = np.random.default_rng(1)
    for flags in range(16):
        p = dict(crop_lo=np.array([r.integers(0, 48 - 16), r.integers(0, 56 - 16), r.integers(0, 40 - 24)]),
                 scale=r.uniform(0.7, 1.3, 3), flips=[bool(flags & 1), bool(flags & 2), bool(flags & 4)], transpose=bool(flags & 8),
                 gain=r.uniform(0.9, 1.1, 4), bias=r.uniform(-0.2, 0.2, 4))
        data, tgt = DL.augment_patch(vol, p)
        d0, t0 = O.augment_patch(norm, label, p["crop_lo"], patch, p["scale"], p["flips"], p["transpose"], p["gain"], p["bias"])
        np.testing.assert_allclose(data.cpu().numpy(), d0, rtol=0, atol=2e-5)
        np.testing.assert_allclose(tgt.cpu().numpy(), t0, rtol=0, atol=2e-6)


@pytest.mark.gpu
def test_simple_reader_protocol(golden):
    """SimpleReader keeps the reference's item protocol ([data], [target]) and its draw order"""
    from brats2019_amd import dataloader as DL
    g = golden("dataloader")
    image, label = _case()
    patch = tuple(int(v) for v in g["patch"])
    rd = DL.SimpleReader([(image, label)], patch, images_in_epoch=8, patches_from_single_image=100)
    assert len(rd) == 8
    random.seed(int(g["seed0"]))
    np.random.seed(int(g["seed0"]))
    d, t = rd[0]
    assert isinstance(d, list) and isinstance(t, list)
    np.testing.assert_allclose(d[0].cpu().numpy(), g["data0"], rtol=0, atol=2e-5)
    fr = DL.FullReader([(image[:, :38, :41, :36], label[:38, :41, :36])])
    d, t = fr[0]
    np.testing.assert_allclose(d[0].cpu().numpy()[:, ::2, ::2, ::2], g["full_data_sub"], rtol=0, atol=2e-5)
    np.testing.assert_array_equal(np.packbits(t[0].cpu().numpy().astype(np.uint8)), g["full_target_bits"])


def test_remap_labels_contract():
    """raw BraTS labels {0,1,2,4} -> {0,1,2,3} (loader_helper.py:30); anything else is refused, not silently dropped"""
    from brats2019_amd import dataloader as DL
    raw = np.array([[0, 1], [2, 4]], np.int16)
    np.testing.assert_array_equal(DL.remap_labels(raw), np.array([[0, 1], [2, 3]], np.uint8))
    np.testing.assert_array_equal(DL.remap_labels(np.array([0.0, 3.0, 4.0], np.float32)), np.array([0, 3, 3], np.uint8))
    with pytest.raises(ValueError):
        DL.remap_labels(np.array([0, 5]))
    with pytest.raises(ValueError):
        DL.remap_labels(np.array([0.5]))


@pytest.mark.gpu
def test_hip_raw_brats_labels_give_the_reference_targets(golden):
    """a case handed over with RAW labels {0,1,2,4} must produce the targets the reference builds after its reader's 4 -> 3 step"""
    from brats2019_amd import dataloader as DL
    g = golden("dataloader")
    image, label = _case()
    raw = label.copy()
    raw[label == 3] = 4
    assert (raw == 4).any()
    patch = tuple(int(v) for v in g["patch"])
    vol = DL.DeviceCase(image, raw, patch)
    random.seed(int(g["seed0"]))
    np.random.seed(int(g["seed0"]))
    p = DL.draw_augment_params(vol.bbox, patch)
    _data, tgt = DL.augment_patch(vol, p)
    np.testing.assert_allclose(tgt.cpu().numpy(), g["target0"], rtol=0, atol=2e-6)
