"""The documents' numbers and lists are generated from (or checked against) the code and the committed measurements, so that
they cannot drift: DESIGN.md section 0 and profiles/MEASURED.md = tools/design_table.py over the committed bench record; the option
keys the header documents = the keys the library accepts; the kernel count DESIGN states = the kernels in the built library."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


MEASURED = os.path.join(ROOT, "profiles", "MEASURED.md")


def test_measured_table_is_the_generated_one():
    import design_table
    s = open(MEASURED).read()
    region = s[s.index(design_table.BEGIN) + len(design_table.BEGIN):s.index(design_table.END)]
    src = re.search(r"generated from `([^`]+)`", region).group(1)
    assert region.split("\n\n", 1)[1].strip() == design_table.table(os.path.join(ROOT, src)).strip(), \
        "profiles/MEASURED.md is stale: python tools/design_table.py %s --write" % src


def test_sq_counter_table_is_the_generated_one():
    import design_table
    s = open(MEASURED).read()
    region = s[s.index(design_table.SQ_BEGIN) + len(design_table.SQ_BEGIN):s.index(design_table.SQ_END)]
    src = re.search(r"generated from `([^`]+)`", region).group(1)
    assert region.split("\n\n", 1)[1].strip() == design_table.sq_table(os.path.join(ROOT, src)).strip(), \
        "profiles/MEASURED.md's SQ table is stale: python tools/design_table.py profiles/<round>_bench.json --write"


def test_design_summary_is_generated_and_design_stays_short():
    """DESIGN.md section 0 = the ten-row summary generated from the committed bench record; the whole document stays readable:
    at most 25 KB and 120 columns (52 KB of 200-400-character lines had become a functional risk)"""
    import design_table
    s = open(os.path.join(ROOT, "DESIGN.md")).read()
    region = s[s.index(design_table.SUM_BEGIN) + len(design_table.SUM_BEGIN):s.index(design_table.SUM_END)]
    src = re.search(r"generated from `([^`]+)`", region).group(1)
    assert region.split("\n\n", 1)[1].strip() == design_table.summary(os.path.join(ROOT, src)).strip(), \
        "DESIGN.md section 0 is stale: python tools/design_table.py %s --write" % src
    assert len(s.encode()) <= 25 * 1024, len(s.encode())
    long_lines = [i + 1 for i, l in enumerate(s.splitlines()) if len(l) > 120]
    assert not long_lines, long_lines[:10]


def _abi_option_keys():
    src = open(os.path.join(ROOT, "x266_amd", "csrc", "x266hip_abi.hip")).read()
    body = src[src.index("kOptions[]"):]
    body = body[:body.index("};")]
    return re.findall(r'\{\s*"([a-z0-9_]+)"', body)


def test_header_documents_exactly_the_option_keys_the_library_accepts():
    keys = _abi_option_keys()
    assert 0 < len(keys) <= 13
    hdr = open(os.path.join(ROOT, "include", "x266hip.h")).read()
    block = hdr[hdr.index("Launch options"):re.search(r"int\s+xHipSetOption", hdr).start()]
    documented = set(re.findall(r"\b((?:dct32|satd|tile|me|adaptive)_[a-z0-9_]+|autotune)\b", block))
    assert set(keys) <= documented, sorted(set(keys) - documented)
    assert documented <= set(keys), "header documents keys the library refuses: %s" % sorted(documented - set(keys))


def _kernels_in_library(lib):
    """Kernel descriptors (.kd symbols) over every gfx950 code object of the library's fat binary (one bundle per source file)."""
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin/"
    if not all(os.path.exists(llvm + t) for t in ("llvm-objcopy", "llvm-objdump", "clang-offload-bundler")):
        return None
    names = set()
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        if subprocess.run([llvm + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(td, "copy.so")]).returncode != 0:
            return None
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for i, st in enumerate(starts):
            part = os.path.join(td, "b%d.bin" % i)
            open(part, "wb").write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(td, "b%d.co" % i)
            r = subprocess.run([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + part, "--output=" + co],
                               capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            sym = subprocess.run([llvm + "llvm-objdump", "-t", co], capture_output=True, text=True).stdout
            names |= {l.split()[-1] for l in sym.splitlines() if l.rstrip().endswith(".kd")}
    return names


def test_design_states_the_kernel_count_of_the_built_library():
    lib = os.path.join(ROOT, "x266_amd", "libx266hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    names = _kernels_in_library(lib)
    if not names:
        pytest.skip("could not read the library's device code with the llvm tools of this image")
    s = open(os.path.join(ROOT, "DESIGN.md")).read()
    stated = [int(x) for x in re.findall(r"(\d+)\*{0,2} kernels\b", s)]
    assert len(names) in stated, "DESIGN.md states %s kernels, the build has %d" % (stated, len(names))
