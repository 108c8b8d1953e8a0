"""PIL-based image augmentation for host-side input pipelines.

Same public surface as the reference's python/singa/image_tool.py (module
functions :41-210, ``ImageTool`` :214-629): every ``ImageTool`` method maps the
held list of PIL images to a new list and either stores it (``inplace=True``,
returns ``self`` so calls chain) or returns it.  Methods with a countable set of
variants take ``num_case``: 1 samples one variant per image (training), the
maximum enumerates all of them (test-time augmentation).

The structure here is table driven: each augmentation is a function
``img -> [imgs]`` and ``ImageTool._apply`` does the fan-out and the
inplace/return handling once.  Pixel work that the reference does with Python
triple loops (colour cast) is vectorised with numpy.
"""
import math
import random

import numpy as np
from PIL import Image, ImageEnhance

_BILINEAR = Image.BILINEAR

# corner/centre anchors expressed as fractions of the free space (x, y)
_CORNERS = {
    "left_top": (0.0, 0.0),
    "left_bottom": (0.0, 1.0),
    "right_top": (1.0, 0.0),
    "right_bottom": (1.0, 1.0),
    "center": (0.5, 0.5),
}
_CORNER_ORDER = ["left_top", "left_bottom", "right_top", "right_bottom",
                 "center"]
# square-crop anchors: (axis the square slides along, fraction of free space)
_SQUARES = {
    "left": (0, 0.0), "center": (0, 0.5), "right": (0, 1.0),
    "top": (1, 0.0), "middle": (1, 0.5), "bottom": (1, 1.0),
}


def load_img(path, grayscale=False):
    """Open an image as 'L' (grayscale) or 'RGB' (always 3 channels)."""
    return Image.open(path).convert("L" if grayscale else "RGB")


def crop(img, patch, position):
    """Cut a (width, height) patch anchored at one of left_top, left_bottom,
    right_top, right_bottom, center."""
    w, h = img.size
    pw, ph = patch
    if w < pw or h < ph:
        raise Exception("img size (%d, %d) is smaller than patch (%d, %d)" %
                        (w, h, pw, ph))
    if position not in _CORNERS:
        raise Exception("position is wrong")
    fx, fy = _CORNERS[position]
    # centre uses floor division of the free space; corners are exact
    left = (w - pw) // 2 if fx == 0.5 else int(fx * (w - pw))
    top = (h - ph) // 2 if fy == 0.5 else int(fy * (h - ph))
    return img.crop((left, top, left + pw, top + ph))


def crop_and_resize(img, patch, position):
    """Cut the largest square at left/center/right (landscape) or
    top/middle/bottom (portrait) and resize it to ``patch``."""
    if position not in _SQUARES:
        raise Exception("position is wrong")
    axis, frac = _SQUARES[position]
    w, h = img.size
    side = h if axis == 0 else w          # square edge = the other dimension
    free = (w if axis == 0 else h) - side
    start = free // 2 if frac == 0.5 else int(frac * free)
    if axis == 0:
        box = (start, 0, start + side, side)
    else:
        box = (0, start, side, start + side)
    return img.crop(box).resize(patch, _BILINEAR)


def resize(img, small_size):
    """Scale so that the shorter side equals ``small_size``."""
    w, h = img.size
    if w < h:
        new = (small_size, int(small_size * h / w))
    else:
        new = (int(small_size * w / h), small_size)
    return img.resize(new, _BILINEAR)


def scale(img, small_size):
    """Alias of :func:`resize`."""
    return resize(img, small_size)


def resize_by_hw(img, size):
    """Resize to exactly (height, width)."""
    return img.resize((size[1], size[0]))


def color_cast(img, offset):
    """Shift each RGB channel, with probability 1/2, by a value drawn from
    [-offset, offset]; results saturate to [0, 255]."""
    shift = np.zeros(3, dtype=np.int16)
    for c in range(3):
        if random.randint(0, 1):
            shift[c] = random.randint(-offset, offset)
    x = np.asarray(img.convert("RGB"), dtype=np.int16) + shift
    return Image.fromarray(np.clip(x, 0, 255).astype(np.uint8), "RGB")


def enhance(img, scale):
    """Apply, each with probability 1/2, Color / Contrast / Brightness /
    Sharpness enhancement by a factor drawn from [1-scale, 1+scale]."""
    for enhancer in (ImageEnhance.Color, ImageEnhance.Contrast,
                     ImageEnhance.Brightness, ImageEnhance.Sharpness):
        factor = random.uniform(1 - scale, 1 + scale) \
            if random.randint(0, 1) else 1.0
        if factor != 1.0:
            img = enhancer(img).enhance(factor)
    return img


def flip(img):
    return img.transpose(Image.FLIP_LEFT_RIGHT)


def flip_down(img):
    return img.transpose(Image.FLIP_TOP_BOTTOM)


def get_list_sample(l, sample_size):
    """``sample_size`` items of ``l`` chosen at random, original order kept."""
    return [l[i] for i in sorted(random.sample(range(len(l)), sample_size))]


def _pick(options, num_case, what):
    options = list(options)
    if num_case < 1 or num_case > len(options):
        raise Exception("num_case must be in [1,%d] (%s)" %
                        (len(options), what))
    if num_case == len(options):
        return options
    return get_list_sample(options, num_case)


class ImageTool(object):
    """Chainable augmentation over a list of PIL images."""

    def __init__(self):
        self.imgs = []

    # ---- container ---------------------------------------------------------
    def load(self, path, grayscale=False):
        self.imgs = [load_img(path, grayscale)]
        return self

    def set(self, imgs):
        self.imgs = imgs
        return self

    def append(self, img):
        self.imgs.append(img)
        return self

    def get(self):
        return self.imgs

    def num_augmentation(self):
        """Number of images currently held (one per augmentation variant)."""
        return len(self.imgs)

    def _apply(self, fn, inplace):
        out = []
        for img in self.imgs:
            out.extend(fn(img))
        if inplace:
            self.imgs = out
            return self
        return out

    # ---- resize / rotate ---------------------------------------------------
    def scale_by_range(self, rng, inplace=True):
        return self.resize_by_range(rng, inplace)

    def scale_by_list(self, size_list, num_case=1, inplace=True):
        return self.resize_by_list(size_list, num_case, inplace)

    def resize_by_range(self, rng, inplace=True):
        """Shorter side resized to one size from [rng[0], rng[1])."""
        return self.resize_by_list(list(range(rng[0], rng[1])), 1, inplace)

    def resize_by_list(self, size_list, num_case=1, inplace=True):
        size_list = list(size_list)
        _pick(size_list, num_case, "size_list")  # validate once up front
        return self._apply(
            lambda im: [resize(im, s)
                        for s in _pick(size_list, num_case, "size_list")],
            inplace)

    def resize_by_hw_range(self, rng, inplace=True):
        """rng = ((h_begin, h_end), (w_begin, w_end)), ends excluded; the two
        ranges are zipped so they must have the same length."""
        (h0, h1), (w0, w1) = rng
        if h1 - h0 != w1 - w0:
            raise Exception("num of widths and heights must be the same!")
        return self.resize_by_hw_list(
            list(zip(range(h0, h1), range(w0, w1))), 1, inplace)

    def resize_by_hw_list(self, size_list, num_case=1, inplace=True):
        size_list = list(size_list)
        _pick(size_list, num_case, "size_list")
        return self._apply(
            lambda im: [resize_by_hw(im, s)
                        for s in _pick(size_list, num_case, "size_list")],
            inplace)

    def rotate_by_range(self, rng, inplace=True):
        return self.rotate_by_list(list(range(rng[0], rng[1])), 1, inplace)

    def rotate_by_list(self, angle_list, num_case=1, inplace=True):
        angle_list = list(angle_list)
        _pick(angle_list, num_case, "angle_list")
        return self._apply(
            lambda im: [im.rotate(a)
                        for a in _pick(angle_list, num_case, "angle_list")],
            inplace)

    # ---- crops -------------------------------------------------------------
    def crop5(self, patch, num_case=1, inplace=True):
        """Four corners and the centre; ``num_case`` of them per image."""
        _pick(_CORNER_ORDER, num_case, "crop5")
        return self._apply(
            lambda im: [crop(im, patch, p)
                        for p in _pick(_CORNER_ORDER, num_case, "crop5")],
            inplace)

    def crop3(self, patch, num_case=1, inplace=True):
        """Largest square at three positions along the longer side, resized to
        the (square) patch."""
        if patch[0] != patch[1]:
            raise Exception("patch must be a square")
        _pick(range(3), num_case, "crop3")

        def fn(im):
            names = ["left", "center", "right"] if im.size[0] > im.size[1] \
                else ["top", "middle", "bottom"]
            return [crop_and_resize(im, patch, p)
                    for p in _pick(names, num_case, "crop3")]

        return self._apply(fn, inplace)

    def crop8(self, patch, num_case=1, inplace=True):
        """Union of crop5 and crop3: ``num_case`` of the 8 variants."""
        chosen = _pick(range(8), num_case, "crop8")
        n5 = sum(1 for s in chosen if s < 5)
        n3 = num_case - n5
        out = []
        if n5:
            out += self.crop5(patch, n5, False)
        if n3:
            out += self.crop3(patch, n3, False)
        if inplace:
            self.imgs = out
            return self
        return out

    def random_crop(self, patch, inplace=True):
        """A (width, height) patch at a uniformly random offset."""
        def fn(im):
            w, h = im.size
            assert w >= patch[0] and h >= patch[1], \
                "img size (%d, %d), patch size (%d, %d)" % (w, h, patch[0],
                                                            patch[1])
            x = random.randint(0, w - patch[0])
            y = random.randint(0, h - patch[1])
            return [im.crop((x, y, x + patch[0], y + patch[1]))]

        return self._apply(fn, inplace)

    def random_crop_resize(self, patch, inplace=True):
        """Inception-style crop: area fraction in [0.08, 1], aspect ratio in
        [3/4, 4/3], up to 10 attempts, then resized to ``patch``; falls back to
        resizing the whole image."""
        def fn(im):
            w, h = im.size
            for _ in range(10):
                area = random.uniform(0.08, 1.0) * w * h
                ratio = random.uniform(3.0 / 4, 4.0 / 3)
                cw = int(round(math.sqrt(area * ratio)))
                ch = int(round(math.sqrt(area / ratio)))
                if w > cw and h > ch:
                    x = random.randint(0, w - cw)
                    y = random.randint(0, h - ch)
                    return [im.crop((x, y, x + cw, y + ch)).resize(
                        patch, _BILINEAR)]
            return [im.resize(patch, _BILINEAR)]

        return self._apply(fn, inplace)

    # ---- flips / colour ----------------------------------------------------
    def _flip_with(self, op, num_case, inplace):
        if num_case not in (1, 2):
            raise Exception("num_case must be in [1,2]")

        def fn(im):
            if num_case == 2:
                return [op(im), im]
            return [op(im) if random.randint(0, 1) else im]

        return self._apply(fn, inplace)

    def flip(self, num_case=1, inplace=True):
        """Left-right mirror: at random (1) or mirrored + original (2)."""
        return self._flip_with(flip, num_case, inplace)

    def flip_down(self, num_case=1, inplace=True):
        """Top-bottom mirror: at random (1) or mirrored + original (2)."""
        return self._flip_with(flip_down, num_case, inplace)

    def color_cast(self, offset=20, inplace=True):
        if offset < 0 or offset > 255:
            raise Exception("offset must be >0 and <255")
        return self._apply(lambda im: [color_cast(im, offset)], inplace)

    def enhance(self, scale=0.2, inplace=True):
        return self._apply(lambda im: [enhance(im, scale)], inplace)


if __name__ == "__main__":
    import sys
    tool = ImageTool()
    imgs = tool.load(sys.argv[1] if len(sys.argv) > 1 else "input.png") \
        .resize_by_list([112]).crop5((96, 96), 5).enhance().flip().get()
    for idx, im in enumerate(imgs):
        im.save("%d.png" % idx)
