#!/usr/bin/env python3
"""Does host-side PNG encoding scale with threads on this box?  dataset_io.encode_png (numpy + zlib, both release the GIL) and Pillow's
Image.save on N threads, 64 images of 800x800x3 noise-like content each.  Prints images per second per thread count."""
import io
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from signerf_amd.dataset_io import encode_png  # noqa: E402

rng = np.random.default_rng(0)
base = (np.add.outer(np.arange(800), np.arange(800))[..., None] // 4 + rng.integers(0, 24, (800, 800, 3))).astype(np.uint8)
imgs = [np.roll(base, k, axis=0).copy() for k in range(64)]


def pil(u8):
    from PIL import Image

    b = io.BytesIO()
    Image.fromarray(u8).save(b, format="PNG")
    return b.getvalue()


print("os.cpu_count", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)))
for name, fn in (("encode_png level 6", lambda u: encode_png(u, 6)), ("encode_png level 1", lambda u: encode_png(u, 1)), ("Pillow level 6", pil)):
    row = []
    for n in (1, 4, 16, 32, 64):
        with ThreadPoolExecutor(n) as ex:
            t = time.perf_counter()
            list(ex.map(fn, imgs))
            row.append(f"{n} threads {len(imgs) / (time.perf_counter() - t):7.1f}/s")
    print(f"{name:20s}", " | ".join(row))
