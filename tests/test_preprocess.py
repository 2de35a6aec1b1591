"""loadImages.m:57-68,103-215: EXIF orientation and grey -> RGB (CPU; the torch form of the same helpers is checked against
the numpy form, on the device too when there is one)."""
import os
import tempfile

import numpy as np
import pytest
import torch

import apsamd
from importlib import import_module

ip = import_module(apsamd.__name__ + ".imageProcessing")


def _matlab(img, o):
    """The reference's switch written with MATLAB's own definitions: rot90(A, k) = k quarter turns counter-clockwise."""
    def rot90(a, k):
        k %= 4
        for _ in range(k):
            a = np.transpose(a, (1, 0) + tuple(range(2, a.ndim)))[::-1]  # one counter-clockwise quarter turn: A.' then flipud
        return a
    return {1: lambda a: a, 2: lambda a: a[:, ::-1], 3: lambda a: rot90(a, 2), 4: lambda a: a[::-1],
            5: lambda a: rot90(a[:, ::-1], -1), 6: lambda a: rot90(a, -1), 7: lambda a: rot90(a[:, ::-1], 1),
            8: lambda a: rot90(a, 1)}[o](img)


@pytest.mark.parametrize("o", range(1, 9))
def test_orientation_codes(o):
    rng = np.random.default_rng(o)
    img = rng.integers(0, 255, (5, 7, 3), dtype=np.uint8)
    want = _matlab(img, o)
    got = ip.applyOrientation(img, o)
    assert got.shape == want.shape and np.array_equal(got, want)
    t = ip.applyOrientation(torch.from_numpy(img), o)
    assert np.array_equal(t.numpy(), want) and t.is_contiguous()
    # what the code means: a pixel known to be top-left in the upright image
    if o == 6:  # stored rotated 90 deg counter-clockwise; upright = one clockwise turn: old bottom-left -> top-left
        assert np.array_equal(got[0, 0], img[-1, 0])
    if o == 3:
        assert np.array_equal(got[0, 0], img[-1, -1])
    assert ip.applyOrientation(img, 9) is img and ip.applyOrientation(img, None) is img


def test_grey_to_rgb_and_file_round_trip():
    g = np.arange(12, dtype=np.uint8).reshape(3, 4)
    c = ip.convertToRGB(g)
    assert c.shape == (3, 4, 3) and np.array_equal(c[:, :, 2], g)
    assert ip.convertToRGB(c) is c
    assert tuple(ip.convertToRGB(torch.from_numpy(g)).shape) == (3, 4, 3)
    from PIL import Image

    rng = np.random.default_rng(0)
    img = rng.integers(0, 255, (6, 9, 3), dtype=np.uint8)
    with tempfile.TemporaryDirectory() as d:
        for o in (1, 6, 8, 2):
            p = os.path.join(d, "o%d.png" % o)
            im = Image.fromarray(img)
            ex = im.getexif()
            ex[274] = o
            im.save(p, exif=ex)
            got = ip.imreadAutoRotate(p)
            assert np.array_equal(got, _matlab(img, o)), o
        p = os.path.join(d, "plain.png")
        Image.fromarray(img).save(p)
        assert np.array_equal(ip.imreadAutoRotate(p), img)


@pytest.mark.gpu
def test_orientation_on_the_device(gpu):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 255, (33, 20, 3), dtype=np.uint8)
    d = torch.from_numpy(img).cuda()
    for o in range(1, 9):
        out = ip.applyOrientation(d, o)
        assert out.is_cuda and np.array_equal(out.cpu().numpy(), _matlab(img, o))


def test_crop_and_save_panorama_file_names_follow_the_reference():
    """cropNsavePanorama.m:136-208: '<proj>[_cropped|_annotated]_<transformationType>_<myImg>_<ii>_<dataset>.png', base files
    first, then the cropped ones (cropPanorama), then the annotated ones (both show flags); argument checks of :51-66."""
    import pytest

    ip = import_module(apsamd.__name__ + ".imageProcessing")
    inp = {"transformationType": "projective", "cropPanorama": 1, "showPanoramaImgsNums": True, "showCropBoundingBox": True}
    store = [{"spherical": [1, 2, 3]}, {"planar": [1, 2, 3], "stereographic": [1, 2, 3], "cylindrical": []}]
    names = [n for n, _, _, _ in ip.panorama_file_names(inp, store, 2, ["a", "Grand Canyon"])]
    assert names == ["spherical_projective_2_1_Grand Canyon.png", "spherical_cropped_projective_2_1_Grand Canyon.png",
                     "spherical_annotated_projective_2_1_Grand Canyon.png", "planar_projective_2_2_Grand Canyon.png",
                     "stereographic_projective_2_2_Grand Canyon.png", "planar_cropped_projective_2_2_Grand Canyon.png",
                     "stereographic_cropped_projective_2_2_Grand Canyon.png", "planar_annotated_projective_2_2_Grand Canyon.png",
                     "stereographic_annotated_projective_2_2_Grand Canyon.png"]
    inp2 = dict(inp, cropPanorama=0, showCropBoundingBox=False)
    assert [n for n, _, _, _ in ip.panorama_file_names(inp2, store[:1], 1, ["set"])] == ["spherical_projective_1_1_set.png"]
    with pytest.raises(ValueError, match="InvalidDatasetIndex"):
        ip.cropNsavePanorama(inp, store, 3, ["a", "b"])
    with pytest.raises(ValueError, match="InvalidDatasetNameType"):
        ip.cropNsavePanorama(inp, store, 1, [5])
    with pytest.raises(ValueError, match="MissingTransformationType"):
        ip.cropNsavePanorama({"imageWrite": True, "imageSaveFolder": "/tmp"}, store, 1, ["a"])
