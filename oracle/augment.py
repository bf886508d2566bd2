"""CPU oracle of the rendered-image augmentation -- TEST
INFRASTRUCTURE, NOT PRODUCT.

reference nr4seg/lightning/joint_train_lightning_net.py:259-302 (``data_aug``)
with the transforms configured at :89-101: ColorJitter(0.3, 0.3, 0.3, 0.05),
rotate +-10 deg (bilinear for the image, nearest for label+1, fill 0),
RandomCrop / CenterCrop to 240x320, horizontal flip with p = 0.5.

The arithmetic lives in torchvision (``requirements.txt:184``:
torchvision==0.12.0), which is not installed here: PARITY UNPINNED.  This file
restates the published tensor code paths of torchvision 0.12.0
(``transforms/functional_tensor.py``: ``_blend``, ``rgb_to_grayscale``,
``adjust_brightness/contrast/saturation/hue``, ``_rgb2hsv``, ``_hsv2rgb``,
``_gen_affine_grid``, ``_apply_grid_transform``; ``transforms/functional.py``:
``_get_inverse_affine_matrix``, ``rotate``; ``transforms/transforms.py``:
``ColorJitter.forward``) on the torch primitives they call
(``torch.nn.functional.grid_sample`` included), anchored on the reference's
call site above.  The random draws (op order, four factors, angle, flip) are
inputs, so the HIP kernel can be compared value for value.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0, 1.0)


def rgb_to_grayscale(img):
    r, g, b = img.unbind(dim=-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)


def adjust_brightness(img, f):
    return _blend(img, torch.zeros_like(img), f)


def adjust_contrast(img, f):
    mean = torch.mean(rgb_to_grayscale(img), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, f)


def adjust_saturation(img, f):
    return _blend(img, rgb_to_grayscale(img), f)


def _rgb2hsv(img):
    r, g, b = img.unbind(dim=-3)
    maxc = torch.max(img, dim=-3).values
    minc = torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    cr_divisor = torch.where(eqc, ones, cr)
    rc = (maxc - r) / cr_divisor
    gc = (maxc - g) / cr_divisor
    bc = (maxc - b) / cr_divisor
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = hr + hg + hb
    h = torch.fmod((h / 6.0 + 1.0), 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(dim=-3)
    i = torch.floor(h * 6.0)
    f = (h * 6.0) - i
    i = i.to(dtype=torch.int32)
    p = torch.clamp((v * (1.0 - s)), 0.0, 1.0)
    q = torch.clamp((v * (1.0 - s * f)), 0.0, 1.0)
    t = torch.clamp((v * (1.0 - (s * (1.0 - f)))), 0.0, 1.0)
    i = i % 6
    mask = i.unsqueeze(dim=-3) == torch.arange(6).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(dtype=img.dtype), a4)


def adjust_hue(img, hue_factor):
    hsv = _rgb2hsv(img)
    h, s, v = hsv.unbind(dim=-3)
    h = (h + hue_factor) % 1.0
    return _hsv2rgb(torch.stack((h, s, v), dim=-3))


def color_jitter(img, order, brightness, contrast, saturation, hue):
    """ColorJitter.forward with its draws given: ``order`` is the permutation
    of (0 brightness, 1 contrast, 2 saturation, 3 hue)."""
    for fn_id in order:
        if fn_id == 0:
            img = adjust_brightness(img, brightness)
        elif fn_id == 1:
            img = adjust_contrast(img, contrast)
        elif fn_id == 2:
            img = adjust_saturation(img, saturation)
        elif fn_id == 3:
            img = adjust_hue(img, hue)
    return img


def _inverse_rotation_matrix(angle_deg):
    # rotate(): _get_inverse_affine_matrix([0, 0], -angle, [0, 0], 1.0, [0, 0])
    rot = math.radians(-angle_deg)
    a, b, c, d = math.cos(rot), -math.sin(rot), math.sin(rot), math.cos(rot)
    return [d, -b, 0.0, -c, a, 0.0]


def _gen_affine_grid(theta, w, h, ow, oh):
    d = 0.5
    base = torch.empty(1, oh, ow, 3, dtype=theta.dtype)
    base[..., 0].copy_(torch.linspace(-ow * 0.5 + d, ow * 0.5 + d - 1, steps=ow))
    base[..., 1].copy_(torch.linspace(-oh * 0.5 + d, oh * 0.5 + d - 1,
                                      steps=oh).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h],
                                                    dtype=theta.dtype)
    return base.view(1, oh * ow, 3).bmm(rescaled).view(1, oh, ow, 2)


def rotate(img, angle_deg, mode):
    """tvf.rotate(img [C,H,W], angle, mode, expand=False, center=None,
    fill=0); integer images (the label) go through float and back."""
    out_dtype = img.dtype
    x = img if img.is_floating_point() else img.to(torch.float32)
    x = x.unsqueeze(0)
    h, w = x.shape[-2:]
    theta = torch.tensor(_inverse_rotation_matrix(angle_deg),
                         dtype=x.dtype).reshape(1, 2, 3)
    grid = _gen_affine_grid(theta, w, h, w, h)
    x = torch.cat((x, torch.ones(1, 1, h, w, dtype=x.dtype)), dim=1)
    x = F.grid_sample(x, grid, mode=mode, padding_mode="zeros",
                      align_corners=False)
    mask = x[:, -1:, :, :].expand(-1, x.shape[1] - 1, -1, -1)
    x = x[:, :-1, :, :]
    fill = torch.zeros_like(x)
    if mode == "nearest":
        x = torch.where(mask < 0.5, fill, x)
    else:
        x = x * mask + (1.0 - mask) * fill
    x = x.squeeze(0)
    if not out_dtype.is_floating_point:
        x = torch.round(x).to(out_dtype)
    return x


def data_aug(img, label, order, brightness, contrast, saturation, hue,
             angle_deg, flip, crop_ij=(0, 0), output_size=(240, 320)):
    """reference data_aug (:259-302) with every random draw given.
    img [3,H,W] float in [0,1], label [H,W] int64 (-1 unknown)."""
    lab = label[None, :, :] + 1
    img = color_jitter(img, order, brightness, contrast, saturation, hue)
    img = rotate(img, angle_deg, "bilinear")
    lab = rotate(lab, angle_deg, "nearest")
    i, j = crop_ij
    th, tw = output_size
    img = img[..., i:i + th, j:j + tw]
    lab = lab[..., i:i + th, j:j + tw]
    if flip:
        img = img.flip(-1)
        lab = lab.flip(-1)
    # CenterCrop(output_size) of an image that already has that size
    return img, (lab - 1)[0]
