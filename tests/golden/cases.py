"""Case lists shared by the golden generators and the tests (pure data)."""

SEED = 0x5EED0000

# (kind, h, w) images used for the LBP goldens; gray = G channel of the synthetic frame
LBP_IMAGES = [
    ("noise", 24, 31), ("tie", 24, 31), ("ramp", 24, 31), ("flat", 24, 31), ("natural", 24, 31), ("step", 24, 31),
    ("noise", 40, 40), ("tie", 40, 40), ("natural", 40, 40),
    ("noise", 64, 64), ("step", 64, 64), ("ramp", 64, 64),
    ("natural", 80, 80), ("tie", 80, 80),
    ("natural", 256, 256),
    ("noise", 1, 1), ("noise", 1, 9), ("noise", 7, 1), ("noise", 2, 3), ("tie", 3, 70),
]
LBP_PR = [(8, 1.0), (16, 2.0), (24, 3.0)]

# Adversarial LBP image: pixels drawn (by the synthetic generator's hash) from an alphabet whose differences
# k hit the good rational approximations of 1 + sqrt(2) (5/2, 12/5, 29/12, 70/29, 169/70).  With the
# (8,1) bilinear weights 0.70711^2, 0.70711*0.29289, 0.29289^2 that makes many diagonal samples land within
# 3e-3 .. 2e-4 gray levels of the centre without being exact ties: the cases a sign pre-decision with an
# error margin must not get wrong.
LATTICE_ALPHABET = [100, 95, 105, 88, 112, 71, 129, 30, 170, 199, 1]
LATTICE_SHAPE = (192, 200)

# (kind, h, w) for the FFT statistic goldens (numpy 2.2.6)
FFT_IMAGES = [
    ("noise", 24, 31), ("natural", 40, 40), ("noise", 64, 64), ("natural", 80, 80), ("step", 90, 120),
    ("natural", 256, 256), ("noise", 17, 23), ("noise", 1, 1), ("noise", 1, 16), ("noise", 12, 1),
    ("ramp", 48, 36), ("flat", 16, 16), ("natural", 45, 80), ("noise", 2, 2), ("tie", 14, 22),
]

# frame sets for the whole-path goldens: (name, kind, n_frames, h, w)
SCORE_SETS = [
    ("noise64", "noise", 10, 64, 64), ("natural64", "natural", 10, 64, 64), ("step64", "step", 10, 64, 64),
    ("tie64", "tie", 3, 64, 64), ("ramp64", "ramp", 2, 64, 64), ("flat64", "flat", 1, 64, 64),
    ("natural_12frames", "natural", 12, 48, 80), ("empty", "noise", 0, 64, 64),
    ("natural_odd", "natural", 4, 45, 77),
]

# Flat patches of hashed gray levels at large coordinates: (h, w, cell).  Every patch interior is an exact tie
# of all four diagonal (8,1) samples, whose fp64 outcome depends on the gray level and on how row + offset and
# column + offset round at that coordinate (the rounding changes at 512, 1024, 2048, 4096).  Goldens keep the
# histogram and a digest of the code image.
PATCHWORK_CASES = [(1100, 2100, 12), (300, 8192, 40), (8192, 150, 9), (130, 260, 64)]
