#!/usr/bin/env python3
"""One line per bench JSON: the box's streams and the configs[1] legs as fractions of 8 TB/s and of the box's own stream.
    python tools/bench_brief.py gpurun_out/box.json [...]"""
import json, sys
for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().split("\n")[-1])
    r, a = d["roofline"], d["also"]
    sb, f = r["same_box"], a["dct32_fwd_inv_fused"]
    print("streams copy %.2f read %.2f write %.2f TB/s | forward %.3e blocks/s %.3f (%.3f of copy) | inverse %.3f (%.3f) | fused %.3f (%.3f), reconstruction only %.3f (%.3f) | SATD %.3e %.3f (%.3f of read)" % (
        sb["copy_TBps"], sb["read_TBps"], sb["write_TBps"], d["value"], r["frac"], r["frac_of_same_box_copy"],
        a["dct32_inv"]["roofline"]["frac"], a["dct32_inv"]["roofline"]["frac_of_same_box_copy"], f["hbm_frac"], f["frac_of_same_box_copy"],
        f["reconstruction_only"]["hbm_frac"], f["reconstruction_only"]["frac_of_same_box_copy"],
        a["satd8x8"]["value"], a["satd8x8"]["roofline"]["frac"], a["satd8x8"]["roofline"]["frac_of_same_box_read"]))
