"""Fixture of the reference's own two-frame alignment test (src/opt/test/test_alignment.cc TestPairAlignment,
test_alignment_util.cc): the data files it reads -- test_data/{identical_images,small_offset}.txt and the four PNGs under
test_data/images -- decoded with Pillow into raw arrays (colour u8, depth u16), the text files parsed into their numbers.  Data only; run in the build container:
    python tests/golden/make_alignment_golden.py   ->  tests/golden/alignment_test_data.npz"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/test_data"

if __name__ == "__main__":
    out = {}
    for n in ("a_image", "a_depth", "b_image", "b_depth"):
        out[n] = np.array(Image.open(os.path.join(SRC, "images", n + ".png")))
    for n in ("identical_images", "small_offset"):
        # calibration w h fx fy cx cy depth_factor | a_image / a_depth / b_image / b_depth paths | a_t_b (3x4) | average_scene_depth
        tok = open(os.path.join(SRC, n + ".txt")).read().split()
        assert tok[0] == "calibration" and tok[8] == "a_image" and tok[14] == "b_depth" and tok[16] == "a_t_b" and tok[29] == "average_scene_depth"
        out[n + "_calibration"] = np.array(tok[1:8], np.float64)
        out[n + "_files"] = np.array([os.path.splitext(os.path.basename(tok[i]))[0] for i in (9, 11, 13, 15)])     # a_image a_depth b_image b_depth
        out[n + "_a_t_b"] = np.array(tok[17:29], np.float64).reshape(3, 4)
        out[n + "_average_scene_depth"] = np.array(float(tok[30]))
    np.savez_compressed(os.path.join(HERE, "alignment_test_data.npz"), **out)
    print({k: (v.shape, v.dtype) for k, v in out.items()})
