"""Input pipeline: image_tool augmentations and ImageBatchIter prefetching.
Modelled on what the reference documents for python/singa/image_tool.py and
python/singa/data.py (it ships no unit tests for them); images are synthetic."""
import random

import numpy as np
import pytest
from PIL import Image

from singa_amd import data, device, image_tool, tensor


def _img(w, h, seed=0):
    rng = np.random.RandomState(seed)
    return Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8), "RGB")


def test_crop_positions():
    im = _img(40, 30)
    a = np.asarray(im)
    assert np.array_equal(np.asarray(image_tool.crop(im, (10, 8), "left_top")),
                          a[:8, :10])
    assert np.array_equal(
        np.asarray(image_tool.crop(im, (10, 8), "right_bottom")), a[22:, 30:])
    assert np.array_equal(np.asarray(image_tool.crop(im, (10, 8), "center")),
                          a[11:19, 15:25])
    with pytest.raises(Exception):
        image_tool.crop(im, (50, 8), "center")
    with pytest.raises(Exception):
        image_tool.crop(im, (10, 8), "nowhere")


def test_resize_and_square_crops():
    im = _img(40, 30)
    assert image_tool.resize(im, 15).size == (20, 15)
    assert image_tool.scale(_img(30, 60), 15).size == (15, 30)
    assert image_tool.resize_by_hw(im, (12, 20)).size == (20, 12)
    for pos in ("left", "center", "right"):
        assert image_tool.crop_and_resize(im, (16, 16), pos).size == (16, 16)
    tall = _img(30, 50)
    for pos in ("top", "middle", "bottom"):
        assert image_tool.crop_and_resize(tall, (16, 16), pos).size == (16, 16)


def test_flips_and_color():
    im = _img(8, 6)
    a = np.asarray(im)
    assert np.array_equal(np.asarray(image_tool.flip(im)), a[:, ::-1])
    assert np.array_equal(np.asarray(image_tool.flip_down(im)), a[::-1])
    random.seed(3)
    c = np.asarray(image_tool.color_cast(im, 20)).astype(int)
    d = c - a.astype(int)
    assert np.abs(d).max() <= 20
    # each channel is shifted by one constant unless it saturates
    for ch in range(3):
        vals = np.unique(d[..., ch][(c[..., ch] > 0) & (c[..., ch] < 255)])
        assert len(vals) <= 1
    e = image_tool.enhance(im, 0.2)
    assert e.size == im.size


def test_tool_chaining_and_num_case():
    tool = image_tool.ImageTool().set([_img(64, 48)])
    out = tool.resize_by_list([40]).crop5((32, 32), 5).flip(2).get()
    assert len(out) == 10 and all(o.size == (32, 32) for o in out)
    assert tool.num_augmentation() == 10
    tool.set([_img(64, 48)])
    assert len(tool.crop8((32, 32), 8, inplace=False)) == 8
    assert len(tool.crop8((32, 32), 3, inplace=False)) == 3
    assert len(tool.crop3((32, 32), 2, inplace=False)) == 2
    assert len(tool.rotate_by_list([10, 20, 30], 2, inplace=False)) == 2
    assert tool.rotate_by_range((0, 5), inplace=False)[0].size == (64, 48)
    assert tool.resize_by_hw_range(((20, 24), (30, 34)),
                                   inplace=False)[0].size[0] in range(30, 34)
    assert tool.random_crop((20, 20), inplace=False)[0].size == (20, 20)
    assert tool.random_crop_resize((24, 24), inplace=False)[0].size == (24, 24)
    assert tool.color_cast(inplace=False)[0].size == (64, 48)
    assert tool.enhance(inplace=False)[0].size == (64, 48)
    assert tool.scale_by_range((30, 31), inplace=False)[0].size == (40, 30)
    with pytest.raises(Exception):
        tool.crop5((32, 32), 6)
    with pytest.raises(Exception):
        tool.crop3((32, 30), 1)
    with pytest.raises(Exception):
        tool.flip(3)
    assert tool.get()[0].size == (64, 48)  # inplace=False left it untouched


def _make_dataset(tmp_path, n, label=True, delim=" "):
    lines = []
    for i in range(n):
        _img(20 + i, 18, seed=i).save(tmp_path / ("im%d.png" % i))
        lines.append("im%d.png%s%s" % (i, delim, i % 3 if label else "cat%d" % i))
    lst = tmp_path / "list.txt"
    lst.write_text("\n".join(lines) + "\n")
    return str(lst)


def _transform(path):
    return image_tool.ImageTool().load(path).resize_by_list([16]) \
        .crop5((16, 16), 5).get()[:1]


def test_image_batch_iter_labels(tmp_path):
    lst = _make_dataset(tmp_path, 7)
    it = data.ImageBatchIter(lst, 3, _transform, shuffle=False,
                             image_folder=str(tmp_path), capacity=2)
    assert it.num_samples == 7
    it.start()
    try:
        ys = []
        for _ in range(5):
            x, y = next(it)
            assert x.shape == (3, 3, 16, 16) and x.dtype == np.float32
            assert y.dtype == np.int32 and y.shape == (3,)
            ys.extend(y.tolist())
        # sequential order with wrap-around after 7 images
        assert ys == [i % 7 % 3 for i in range(15)]
    finally:
        it.end()
    assert it.p is None


def test_image_batch_iter_meta_strings(tmp_path):
    lst = _make_dataset(tmp_path, 4, label=False, delim=",")
    it = data.ImageBatchIter(lst, 2, _transform, shuffle=True, delimiter=",",
                             image_folder=str(tmp_path), capacity=1,
                             num_workers=1)
    it.start()
    try:
        x, y = it.next()
        assert x.shape == (2, 3, 16, 16)
        assert isinstance(y, list) and all(v.startswith("cat") for v in y)
    finally:
        it.end()


def test_device_prefetcher(dev):
    batches = [(np.full((2, 3, 4, 4), i, np.float32),
                np.array([i, i + 1], np.int32)) for i in range(3)]
    got = list(data.DevicePrefetcher(iter(batches), dev))
    assert len(got) == 3
    for i, (x, y) in enumerate(got):
        assert x.device is dev and x.shape == (2, 3, 4, 4)
        np.testing.assert_array_equal(tensor.to_numpy(x), batches[i][0])
        np.testing.assert_array_equal(tensor.to_numpy(y), batches[i][1])
