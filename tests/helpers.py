"""shared helpers for the tests (fixtures from tests/golden, toy index construction)"""
import json
import math
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
S = np.float32(math.sqrt(0.5))  # std::f32::consts::FRAC_1_SQRT_2 == 0.70710677f32
EMPTY = 0xFFFFFFFFFFFFFFFF
FMAX = float(np.float32(3.4028234663852886e38))


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def sub(x):
    if isinstance(x, list):
        return [sub(v) for v in x]
    if x == "S":
        return float(S)
    if x == "E":
        return EMPTY
    if x == "M":
        return FMAX
    return x


def toy_vectors(broken=False):
    t = load("toy_index.json")
    data = sub(t["vectors"]["data"])
    if broken:
        data = data + [sub(t["vectors"]["extra_broken"]["data"])]
    return np.array(data, dtype=np.float32)
