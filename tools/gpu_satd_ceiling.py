#!/usr/bin/env python3
"""Developer probe (round 4): the SATD batch kernel against THIS box's read-only stream, same process, same
clocks, alternating rounds.  Shapes of the kernel body ("diag_satd_shape": 0 load-then-score, 1 register prefetch,
2 LDS-DMA ping-pong, 3 sequential chains at 64 VGPRs) x launch shapes; HIP events around every launch."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, x266_amd
cd = x266_amd.Codec(0)
n = 1 << 24
d = torch.empty(n * 64, dtype=torch.int16, device="cuda")
out = torch.empty(n, dtype=torch.int32, device="cuda")
z = torch.empty(n * 64, dtype=torch.int16, device="cuda")
cd.fill_residual_dev(d.data_ptr(), d.numel(), 0x267)
torch.cuda.synchronize()
N = 60
ev = [cd.event_create() for _ in range(N + 1)]
BYTES = n * 128


def timed(fn, warm=40):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    for i in range(N):
        cd.event_record(ev[i])
        fn()
    cd.event_record(ev[N])
    t = [cd.event_elapsed_ms(ev[i], ev[i + 1]) for i in range(N)]
    return sum(t) / N, statistics.median(t), min(t)


def line(label, t, nbytes):
    print("%-58s mean %.4f ms %.3f TB/s | median %.4f %.3f TB/s | min %.4f" % (label, t[0], nbytes / t[0] / 1e9, t[1], nbytes / t[1] / 1e9, t[2]), flush=True)


# ragged sizes through every shape, against shape 0 (which tests/test_gpu_parity.py pins to the oracle)
import numpy as np
rs = np.random.RandomState(4)
for nb in (1, 31, 32, 33, 63, 1000, 4099, 65537):
    blk = rs.randint(-32768, 32768, size=(nb, 64)).astype(np.int16)
    want = None
    for shape in (0, 1, 2, 3):
        for gpw in (1, 2, 3, 8):
            cd.set_option("diag_satd_shape", shape)
            cd.set_option("satd_groups_per_wave", gpw)
            cd.set_option("adaptive_per_wave", 0)
            got = cd.satd8x8(blk)
            if want is None:
                want = got
            assert np.array_equal(got, want), (nb, shape, gpw)
cd.set_option("adaptive_per_wave", 1)
print("# every shape equals shape 0 on ragged full-range batches", flush=True)

ref = None
configs = []
for shape in (0, 1, 2, 3):
    for tpb in (64, 128, 256):
        for gpw in ((2, 4, 8) if shape in (1, 2) else (1, 2, 4)):
            for lds in ((8192, 10240, 16384) if shape == 2 else (4096, 6144, 8192)):
                configs.append((shape, tpb, gpw, lds))
quick = os.environ.get("QUICK") == "1"
if quick:
    configs = [c for c in configs if c[1] == 128 and c[3] in (6144, 8192)]
for rnd in range(2):
    print("# round %d" % rnd)
    line("read-only nt stream (xHipMemCeilingDev kind 1)", timed(lambda: cd.mem_ceiling_dev(1, d.data_ptr(), z.data_ptr(), BYTES)), BYTES)
    line("nt copy (xHipMemCeilingDev kind 0)", timed(lambda: cd.mem_ceiling_dev(0, d.data_ptr(), z.data_ptr(), BYTES)), 2 * BYTES)
    for shape, tpb, gpw, lds in configs:
        cd.set_option("diag_satd_shape", shape)
        cd.set_option("satd_wg_threads", tpb)
        cd.set_option("satd_groups_per_wave", gpw)
        cd.set_option("satd_lds_bytes_per_wave", lds)
        out.zero_()
        t = timed(lambda: cd.satd8x8_dev(d.data_ptr(), out.data_ptr(), n), warm=25)
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        line("satd shape %d tpb %3d gpw %d lds %5d %s" % (shape, tpb, gpw, lds, "" if same else "MISMATCH"), t, n * 132)
    line("read-only nt stream (again)", timed(lambda: cd.mem_ceiling_dev(1, d.data_ptr(), z.data_ptr(), BYTES)), BYTES)
