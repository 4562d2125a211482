#!/usr/bin/env python3
"""Developer probe (CPU only): instruction classes per kernel of one .hip unit, whole kernel and per basic block.
    tools/isa_count.py x266_amd/csrc/dct32_kernels.hip [name-substring] [--blocks]
Compiles the device side to assembly with the product's flags and counts VALU / MFMA / DS / VMEM / SALU / s_waitcnt.
With --blocks every basic block (label to label) of the matching kernels is listed, so a steady-state loop can be read off."""
import collections, os, re, subprocess, sys, tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only"]


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "ds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return None


def main():
    src = sys.argv[1]
    want = [a for a in sys.argv[2:] if not a.startswith("--")]
    blocks = "--blocks" in sys.argv
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, src], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    parts = re.split(r"\n(_Z[^\n:]*):[^\n]*\n", txt)
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split(".Lfunc_end")[0]
        if want and not any(w in name for w in want): continue
        total = collections.Counter()
        cur_label, cur = "entry", collections.Counter()
        per_block = []
        for line in body.split("\n"):
            line = line.strip()
            m = re.match(r"(\.LBB[0-9_]+):", line)
            if m:
                per_block.append((cur_label, cur)); cur_label, cur = m.group(1), collections.Counter()
                continue
            m = re.match(r"([a-z_0-9]+)", line)
            if not m or line.startswith((".", ";")): continue
            k = classify(m.group(1))
            if k: total[k] += 1; cur[k] += 1
        per_block.append((cur_label, cur))
        meta = re.search(re.escape(name) + r".*?\.vgpr_count:\s*(\d+)", txt, re.S)
        print("%s\n   total %s  vgprs %s" % (name, dict(total), meta.group(1) if meta else "?"))
        if blocks:
            for lab, c in per_block:
                if sum(c.values()) > 8: print("   %-12s %s" % (lab, dict(c)))


if __name__ == "__main__":
    main()
