#!/usr/bin/env python3
"""Extract the reference's own SAD known-answer fixture (build container only).

riscv/programs/benchmarks/sad/dataset1.h holds two 64x64 uint8 arrays and the
expected sum of absolute differences (344807, checked by sad.c:57).  This is a
data file of the reference's tests; it is stored as data (tests/golden/sad64.npz).
"""
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
text = open("/root/reference/riscv/programs/benchmarks/sad/dataset1.h").read()
arrays = re.findall(r"(\w+)\[DATA_SIZE\]\s*=\s*\{(.*?)\};", text, flags=re.S)
vals = {name: np.array([int(v) for v in re.findall(r"-?\d+", body)]) for name, body in arrays}
a, b, want = vals["input_data1"], vals["input_data2"], int(vals["verify_data"][0])
assert a.size == 4096 and b.size == 4096
assert int(np.abs(a - b).sum()) == want == 344807
np.savez_compressed(os.path.join(HERE, "sad64.npz"), a=a.astype(np.uint8).reshape(64, 64), b=b.astype(np.uint8).reshape(64, 64),
                    sad=np.array([want], np.int64))
print("sad64.npz written; SAD =", want)
