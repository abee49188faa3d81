"""Derives small statistical fixtures from the three renders the reference ships
(/root/reference/samples/sample{0,1,2}.png — produced by the real Rust binary, README.md:11-13).

They are the ONLY outputs of the reference that exist (it has no tests and cannot be built
here), so they pin the oracle: tests/test_oracle_vs_samples.py compares oracle renders with
them.  Stored per image (derived data, not the images): 8x8 box-filtered
  * mean / std of the 8-bit pixels,
  * mean of the LINEARISED pixels (inverse sRGB OETF, inverse ACES fit: the inverse of the
    reference's output stage src/output.rs:42-49 + src/tonemapping/aces.rs:5-33), so that a
    render at 1/8 resolution — whose pixels are exact box filters of radiance — can be compared
    without the Jensen bias of averaging tonemapped values,
  * the fraction of pixels in the block that are near the 8-bit clamp (unreliable inverse).

Run in the build container (needs /root/reference):  python tests/golden/make_sample_fixtures.py
"""
import os

import numpy as np
from PIL import Image

REF = "/root/reference/samples"
OUT = os.path.dirname(os.path.abspath(__file__))

ACES_IN = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
ACES_OUT = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])


def srgb_to_linear(s):
    return np.where(s < 0.0031308 * 12.92, s / 12.92, ((s + 0.055) / 1.055) ** 2.4)


def inverse_aces(o):
    """o: (..., 3) display-linear values in [0, 1] -> scene-linear radiance."""
    f = o @ np.linalg.inv(ACES_OUT).T
    # f = (c (c + 0.0245786) - 0.000090537) / (c (0.983729 c + 0.4329510) + 0.238081): solve for c >= 0
    A = 1.0 - 0.983729 * f
    B = 0.0245786 - 0.4329510 * f
    Cq = -(0.000090537 + 0.238081 * f)
    disc = np.maximum(B * B - 4 * A * Cq, 0.0)
    c = (-B + np.sqrt(disc)) / (2 * A)
    return c @ np.linalg.inv(ACES_IN).T


def linearise(img8):
    s = (img8.astype(np.float64) + 0.5) / 255.999   # centre of the quantisation bin of `(x*255.999) as u8`
    return inverse_aces(srgb_to_linear(s))


def blocks(a, k=8):
    h, w, c = a.shape
    return a[: h // k * k, : w // k * k].reshape(h // k, k, w // k, k, c).astype(np.float64)


def main():
    out = {}
    for name in ("sample0", "sample1", "sample2"):
        img = np.array(Image.open(os.path.join(REF, name + ".png")).convert("RGB"))
        b8 = blocks(img)
        lin = blocks(linearise(img))
        clipped = blocks(((img >= 250) | (img <= 1)).any(axis=2, keepdims=True).astype(np.float64))
        out[name + "_shape"] = np.array(img.shape)
        out[name + "_mean"] = b8.mean(axis=(1, 3)).astype(np.float32)
        out[name + "_std"] = b8.std(axis=(1, 3)).astype(np.float32)
        out[name + "_linear_mean"] = lin.mean(axis=(1, 3)).astype(np.float32)
        out[name + "_clipped_frac"] = clipped.mean(axis=(1, 3))[..., 0].astype(np.float32)
        out[name + "_global_mean"] = img.reshape(-1, 3).mean(axis=0)
        print(name, img.shape, "global mean", out[name + "_global_mean"])
    np.savez_compressed(os.path.join(OUT, "reference_samples_ds8.npz"), **out)


if __name__ == "__main__":
    main()
