"""Golden vectors for the score-image front end: outputs of Pillow itself (the library the reference calls in
src/data/preprocessing.py:44-52) on seeded synthetic pixels.  Run in the build container:  python tests/golden/gen_image_golden.py
-> tests/golden/f11_image.npz.  Pillow version recorded in the file."""
import os

import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def synth(rng, h, w, c):
    """Staff-like synthetic page: white background, dark horizontal lines, random blobs and noise (exercises clipping)."""
    a = np.full((h, w, c), 255, dtype=np.uint8)
    for y in range(3, h, 7):
        a[y, :, :] = rng.integers(0, 40)
    n = rng.integers(5, 30)
    for _ in range(n):
        y, x = rng.integers(0, h), rng.integers(0, w)
        a[y:y + rng.integers(1, 6), x:x + rng.integers(1, 9), :] = rng.integers(0, 256, size=c)
    noise = rng.integers(0, 256, size=(h, w, c))
    mask = rng.random((h, w, 1)) < 0.15
    return np.where(mask, noise, a).astype(np.uint8)


def main():
    rng = np.random.default_rng(20260213)
    out = {"pillow_version": np.array(PIL.__version__)}
    cases = [  # (name, h, w, channels, img_height)
        ("down_rgb", 111, 317, 3, 48), ("down_l", 90, 403, 1, 64), ("up_rgb", 37, 91, 3, 96), ("same_rgba", 48, 110, 4, 48),
        ("none_rgb", 30, 57, 3, None), ("down_big", 160, 699, 3, 64), ("tiny", 5, 9, 1, 64), ("thin", 1, 40, 3, 16),
    ]
    for name, h, w, c, H in cases:
        px = synth(rng, h, w, c)
        img = Image.fromarray(px[..., 0] if c == 1 else px, mode={1: "L", 3: "RGB", 4: "RGBA"}[c])
        x = img.convert("L")
        gray = np.asarray(x).copy()
        if H is not None:
            x = x.resize((int(H * x.size[0] / x.size[1]), H))
        out[f"{name}_pixels"] = px
        out[f"{name}_height"] = np.array(-1 if H is None else H)
        out[f"{name}_gray"] = gray
        out[f"{name}_out"] = np.asarray(x).copy()
    np.savez_compressed(os.path.join(HERE, "f11_image.npz"), **out)
    print("wrote f11_image.npz", {k: v.shape for k, v in out.items() if k.endswith("_out")})


if __name__ == "__main__":
    main()
