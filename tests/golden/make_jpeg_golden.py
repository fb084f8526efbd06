"""Golden vectors for the JPEG -> grey path of the tools (csrc/host/io_jpeg.h).

Writes small JPEG files with Pillow (libjpeg-turbo) in the variants a camera or an export tool produces, and decodes each with
libjpeg's own grayscale output (Image.draft('L', ...) = JCS_GRAYSCALE, the luminance plane that cv::imread(...,
IMREAD_GRAYSCALE) returns).  Run in the build container: python tests/golden/make_jpeg_golden.py
Outputs: tests/golden/jpeg_<name>.jpg and tests/golden/jpeg_golden.npz (expected grey images)."""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def scene(w, h, seed):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    r = 128 + 90 * np.sin(xx / 5.0) * np.cos(yy / 7.0) + rng.normal(0, 12, (h, w))
    g = 110 + 80 * np.cos((xx + 2 * yy) / 9.0) + rng.normal(0, 12, (h, w))
    b = 140 + 100 * np.sin((xx - yy) / 4.0) + rng.normal(0, 25, (h, w))
    img = np.stack([r, g, b], -1)
    img[h // 3: h // 3 + 6, :, :] = 255; img[:, w // 2: w // 2 + 3, :] = 0          # saturated edges: exercises the clamp
    return Image.fromarray(img.clip(0, 255).astype(np.uint8), "RGB")


VARIANTS = {
    "420_q90": dict(size=(67, 45), quality=90, subsampling=2),
    "422_q75_opt": dict(size=(130, 51), quality=75, subsampling=1, optimize=True),
    "444_q98": dict(size=(40, 40), quality=98, subsampling=0),
    "420_q30": dict(size=(97, 83), quality=30, subsampling=2),
    "grey_q85": dict(size=(53, 70), quality=85, grey=True),
    "420_restart": dict(size=(75, 60), quality=80, subsampling=2, restart_marker_blocks=3),
    # progressive (SOF2): libjpeg's standard scan script = spectral selection + successive approximation, i.e. DC / AC first and
    # refinement scans with EOB runs; interleaved DC scan for colour files
    "progressive": dict(size=(48, 32), quality=80, progressive=True, seed=9),
    "prog_444_q95": dict(size=(61, 47), quality=95, subsampling=0, progressive=True),
    "prog_420_q40_opt": dict(size=(131, 99), quality=40, subsampling=2, progressive=True, optimize=True),
    "prog_grey_q90": dict(size=(77, 58), quality=90, progressive=True, grey=True),
    "prog_422_restart": dict(size=(90, 70), quality=85, subsampling=1, progressive=True, restart_marker_blocks=2),
}

if __name__ == "__main__":
    expected = {}
    for i, (name, v) in enumerate(VARIANTS.items()):
        v = dict(v)
        w, h = v.pop("size")
        img = scene(w, h, v.pop("seed", i))
        if v.pop("grey", False):
            img = img.convert("L")
        path = os.path.join(HERE, "jpeg_%s.jpg" % name)
        img.save(path, "JPEG", **v)
        im = Image.open(path)
        im.draft("L", im.size)
        assert im.mode == "L", im.mode
        expected[name] = np.array(im)
        assert expected[name].shape == (h, w)
    # EXIF orientation 1..8: cv::imread applies it (ExifTransform); expected = the luminance plane put upright, by Pillow
    from PIL import ImageOps
    for o in range(1, 9):
        path = os.path.join(HERE, "jpeg_exif_%d.jpg" % o)
        exif = Image.Exif(); exif[0x0112] = o
        scene(37, 29, 20 + o).save(path, "JPEG", quality=85, subsampling=2, exif=exif.tobytes())
        im = Image.open(path)
        im.draft("L", im.size)
        assert im.mode == "L" and im.getexif().get(0x0112) == o
        expected["exif_%d" % o] = np.array(ImageOps.exif_transpose(im))
        assert expected["exif_%d" % o].shape == ((29, 37) if o < 5 else (37, 29))
    np.savez_compressed(os.path.join(HERE, "jpeg_golden.npz"), **expected)
    print({k: v.shape for k, v in expected.items()})
