"""C++ API (device / memory pool / tensor / layers / FeedForwardNet /
optimizers / updater / IO / endpoint): runs the self-checking native test
program, on the host here and additionally on HipGPU under the gpu marker
(where every layer is compared against the host implementation)."""
import os
import subprocess

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
EXE = os.path.join(ROOT, "singa_amd", "bin", "singa_cpp_test")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        from singa_amd import build
        build.build_cpp(verbose=False)
    return EXE


def _run(args, timeout):
    r = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=timeout)
    tail = "\n".join(r.stdout.splitlines()[-40:])
    assert r.returncode == 0, tail
    assert " 0 failed" in r.stdout, tail
    return r.stdout


def test_cpp_api_host(exe):
    out = _run([exe], 600)
    for name in ("TensorMath/cpu", "LayersGradient/cpu", "FeedForwardNet/cpu",
                 "FeedForwardNet/cpu-graph", "LocalUpdater", "Rnn/cpu", "Io",
                 "EndPoint"):
        assert "[  OK  ] " + name in out


@pytest.mark.gpu
def test_cpp_api_gpu(exe):
    out = _run([exe, "--gpu"], 600)
    for name in ("TensorMath/gpu", "LayersGpuVsCpu", "Rnn/cpu+gpu",
                 "FeedForwardNet/gpu",
                 "FeedForwardNet/gpu-graph"):
        assert "[  OK  ] " + name in out


def _example(name):
    path = os.path.join(ROOT, "singa_amd", "bin", name)
    if not os.path.exists(path):
        from singa_amd import build
        build.build_cpp(verbose=False)
    return path


def test_cpp_example_vgg_parallel_host():
    """Two host replicas of a narrow VGG (batch norm in every block) train
    through the LocalUpdater; the program exits 0 when the test accuracy of
    replica 0 is clearly above chance, which needs its running statistics to
    survive the parameter hand-back."""
    r = subprocess.run([_example("singa_vgg_parallel"), "-cpu", "-width", "4",
                        "-epoch", "1", "-replicas", "2", "-dropout", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "test accuracy" in r.stdout


def test_cpp_example_alexnet_record_pipeline(tmp_path):
    """Synthetic images -> JPEG records -> threaded decode / crop / mirror with
    one batch of prefetch -> a narrow AlexNet; exits 0 when the validation
    accuracy beats chance."""
    r = subprocess.run([_example("singa_alexnet"), "-synthetic", "-cpu",
                        "-data", str(tmp_path)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "val loss" in r.stdout
    assert os.path.exists(str(tmp_path / "train2.bin"))
    assert os.path.exists(str(tmp_path / "mean.bin"))


def _jpeg_tool():
    exe = os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "singa_amd", "bin", "singa_cpp_test")
    if not os.path.exists(exe):
        pytest.skip("C++ test program not built")
    return exe


def _test_image(h=61, w=83):
    import numpy as np
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([255 * xx / (w - 1), 255 * yy / (h - 1),
                    128 + 100 * np.sin(0.15 * (xx + yy))], axis=-1)
    return np.clip(np.round(img), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("subsampling,tol", [(0, 3), (1, 6), (2, 6)])
def test_jpeg_decoder_against_pil(tmp_path, subsampling, tol):
    """Files written by an independent encoder (PIL / libjpeg: optimised or
    standard Huffman tables, 4:4:4 and 4:2:0 sampling) decode to the same
    pixels as PIL's decoder, up to IDCT rounding (4:4:4) and the chroma
    upsampling filter's rounding (4:2:2, 4:2:0)."""
    import subprocess
    import numpy as np
    from PIL import Image
    exe = _jpeg_tool()
    img = _test_image()
    for extra in ({}, {"optimize": True}):
        jpg = str(tmp_path / "a.jpg")
        Image.fromarray(img).save(jpg, quality=90, subsampling=subsampling,
                                  **extra)
        ppm = str(tmp_path / "a.ppm")
        subprocess.run([exe, "--jpeg-decode", jpg, ppm], check=True,
                       timeout=60)
        mine = np.asarray(Image.open(ppm).convert("RGB")).astype(np.int32)
        ref = np.asarray(Image.open(jpg).convert("RGB")).astype(np.int32)
        assert mine.shape == ref.shape
        assert np.abs(mine - ref).max() <= tol
        assert np.abs(mine - ref).mean() < 1.5


def test_jpeg_decoder_restart_markers_and_grey(tmp_path):
    import subprocess
    import numpy as np
    from PIL import Image
    exe = _jpeg_tool()
    grey = _test_image()[:, :, 2]
    jpg = str(tmp_path / "g.jpg")
    Image.fromarray(grey).save(jpg, quality=85, restart_marker_blocks=3)
    ppm = str(tmp_path / "g.pgm")
    subprocess.run([exe, "--jpeg-decode", jpg, ppm], check=True, timeout=60)
    mine = np.asarray(Image.open(ppm)).astype(np.int32)
    ref = np.asarray(Image.open(jpg)).astype(np.int32)
    assert mine.shape == ref.shape and np.abs(mine - ref).max() <= 2


def _varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7f) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def test_image_records_of_the_reference_format(tmp_path):
    """The reference stores an image as an ImageRecord message (shape = 1,
    label = 2, pixel = 3: a JPEG file, src/proto/io.proto:55-59) whose colour
    planes are the tensor's channels taken as blue, green, red
    (src/io/jpg_encoder.cc:38-48).  A record assembled here the way that
    encoder does -- PIL writes the file -- decodes to the tensor, and a record
    written by the encoder under test holds the mirrored planes."""
    import io
    import subprocess
    import numpy as np
    from PIL import Image
    exe = _jpeg_tool()
    img = _test_image()                       # H x W x 3, the tensor
    jpg = io.BytesIO()
    Image.fromarray(img[:, :, ::-1].copy()).save(jpg, "JPEG", quality=100,
                                                  subsampling=0)
    rec = b"".join(b"\x08" + _varint(d) for d in img.shape)
    rec += b"\x10" + _varint(41)
    rec += b"\x1a" + _varint(len(jpg.getvalue())) + jpg.getvalue()
    path = str(tmp_path / "rec.bin")
    open(path, "wb").write(rec)
    ppm = str(tmp_path / "rec.ppm")
    subprocess.run([exe, "--jpeg-decode", path, ppm], check=True, timeout=60)
    assert b"# label 41" in open(ppm, "rb").read(64)
    mine = np.asarray(Image.open(ppm).convert("RGB")).astype(np.int32)
    assert mine.shape == img.shape
    assert np.abs(mine - img.astype(np.int32)).max() <= 4
    # the other direction
    src = str(tmp_path / "src.ppm")
    Image.fromarray(img).save(src)
    out = str(tmp_path / "out.bin")
    subprocess.run([exe, "--jpeg-record", src, out], check=True, timeout=60)
    blob = open(out, "rb").read()
    want_head = b"".join(b"\x08" + _varint(d) for d in img.shape) + b"\x1a"
    assert blob.startswith(want_head)
    at = len(want_head)
    n = shift = 0
    while True:
        b = blob[at]
        at += 1
        n |= (b & 0x7f) << shift
        shift += 7
        if not b & 0x80:
            break
    assert at + n == len(blob) and blob[at:at + 2] == b"\xff\xd8"
    planes = np.asarray(Image.open(io.BytesIO(blob[at:])).convert("RGB"))
    assert np.abs(planes[:, :, ::-1].astype(np.int32) - img).max() <= 4


@pytest.mark.parametrize("quality", [100, 75])
def test_jpeg_encoder_read_by_pil(tmp_path, quality):
    """Files written by the encoder are valid baseline JPEGs for an
    independent decoder and close to the source image."""
    import subprocess
    import numpy as np
    from PIL import Image
    exe = _jpeg_tool()
    img = _test_image()
    ppm = str(tmp_path / "b.ppm")
    Image.fromarray(img).save(ppm)
    jpg = str(tmp_path / "b.jpg")
    subprocess.run([exe, "--jpeg-encode", ppm, jpg, str(quality)], check=True,
                   timeout=60)
    back = Image.open(jpg)
    assert back.format == "JPEG" and back.size == (img.shape[1], img.shape[0])
    err = np.abs(np.asarray(back.convert("RGB")).astype(np.int32) -
                 img.astype(np.int32))
    assert err.max() <= (4 if quality == 100 else 40)
    assert err.mean() < (1.0 if quality == 100 else 4.0)
