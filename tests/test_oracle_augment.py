"""KATs for the augmentation oracle (oracle/augment.py; torchvision 0.12.0
tensor ops restated -- torchvision itself is not installed: parity unpinned)."""
import colorsys
import math

import torch

from oracle import augment as oa


def _img(seed=0, H=24, W=32):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(3, H, W, generator=g)


def test_colour_ops_known_values():
    img = _img()
    assert torch.equal(oa.adjust_brightness(img, 1.0), img)
    assert torch.allclose(oa.adjust_brightness(img, 0.5), img * 0.5)
    assert torch.allclose(oa.adjust_saturation(img, 1.0), img)
    grey = oa.adjust_saturation(img, 0.0)
    assert torch.allclose(grey[0], grey[1]) and torch.allclose(grey[1], grey[2])
    flat = oa.adjust_contrast(img, 0.0)
    assert float(flat.std()) < 1e-6
    assert abs(float(flat.mean()) - float(oa.rgb_to_grayscale(img).mean())) < 1e-6
    # hue: against colorsys, pixel by pixel
    out = oa.adjust_hue(img, 0.05)
    assert torch.allclose(oa.adjust_hue(img, 0.0), img, atol=1e-6)
    for (y, x) in [(0, 0), (3, 7), (23, 31), (10, 10)]:
        r, g, b = [float(v) for v in img[:, y, x]]
        h, s, v = colorsys.rgb_to_hsv(r, g, b)
        want = colorsys.hsv_to_rgb((h + 0.05) % 1.0, s, v)
        for c in range(3):
            assert abs(float(out[c, y, x]) - want[c]) < 2e-6
    # grey pixels have no hue
    gp = torch.full((3, 2, 2), 0.4)
    assert torch.allclose(oa.adjust_hue(gp, 0.03), gp)


def test_jitter_order_matters_and_is_respected():
    img = _img(1)
    a = oa.color_jitter(img, [0, 1, 2, 3], 1.2, 0.8, 1.3, 0.04)
    b = oa.color_jitter(img, [3, 2, 1, 0], 1.2, 0.8, 1.3, 0.04)
    manual = oa.adjust_hue(oa.adjust_saturation(oa.adjust_contrast(
        oa.adjust_brightness(img, 1.2), 0.8), 1.3), 0.04)
    assert torch.equal(a, manual)
    assert float((a - b).abs().max()) > 1e-3


def test_rotation_conventions():
    H, W = 24, 32
    img = _img(2, H, W)
    lab = torch.arange(H * W).reshape(1, H, W) % 40
    assert torch.allclose(oa.rotate(img, 0.0, "bilinear"), img, atol=1e-6)
    assert torch.equal(oa.rotate(lab, 0.0, "nearest"), lab)
    # 180 degrees about the image centre = flip both axes
    r = oa.rotate(img, 180.0, "bilinear")
    assert torch.allclose(r, img.flip(-1).flip(-2), atol=1e-5)
    # positive angle = counter-clockwise: a bright dot right of the centre
    # moves UP (smaller row index)
    dot = torch.zeros(3, 41, 41)
    dot[:, 20, 30] = 1.0
    r = oa.rotate(dot, 90.0, "bilinear")
    yy, xx = divmod(int(r[0].argmax()), 41)
    assert (yy, xx) == (10, 20)
    # corners fall outside after a 10 degree turn: image 0, label "unknown"
    out_i, out_l = oa.data_aug(img, lab[0] - 1, [0, 1, 2, 3], 1, 1, 1, 0, 10.0,
                               False, output_size=(H, W))
    assert float(out_i[:, 0, 0].abs().max()) == 0.0 and int(out_l[0, 0]) == -1
    assert int(out_l[H // 2, W // 2]) == int(lab[0, H // 2, W // 2]) - 1


def test_flip_and_crop():
    H, W = 24, 32
    img = _img(3, H, W)
    lab = (torch.arange(H * W).reshape(H, W) % 40) - 1
    i0, l0 = oa.data_aug(img, lab, [0, 1, 2, 3], 1, 1, 1, 0, 0.0, False,
                         output_size=(H, W))
    i1, l1 = oa.data_aug(img, lab, [0, 1, 2, 3], 1, 1, 1, 0, 0.0, True,
                         output_size=(H, W))
    assert torch.allclose(i0, img, atol=1e-6) and torch.equal(l0, lab)
    assert torch.equal(i1, i0.flip(-1)) and torch.equal(l1, l0.flip(-1))
    i2, l2 = oa.data_aug(img, lab, [0, 1, 2, 3], 1, 1, 1, 0, 0.0, False,
                         crop_ij=(2, 5), output_size=(16, 20))
    assert torch.allclose(i2, img[:, 2:18, 5:25], atol=1e-6)
    assert torch.equal(l2, lab[2:18, 5:25])
