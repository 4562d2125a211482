#!/usr/bin/env python3
"""Developer probe (round 4): where can a process read the shader clock the chip actually runs at while a kernel is executing?"""
import glob, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, x266_amd
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if not k.startswith("_") and k in ("name", "pci_bus_id", "pci_device_id", "pci_domain_id", "clock_rate", "multi_processor_count", "gcnArchName", "uuid")})
for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        print(dev, os.path.realpath(dev), open(dev + "/pp_dpm_sclk").read().replace("\n", " | "))
    except OSError as e:
        print(dev, "pp_dpm_sclk:", e)
    for f in ("gpu_busy_percent", "hwmon/hwmon*/freq1_input"):
        for q in glob.glob(dev + "/" + f):
            try:
                print("   ", q, open(q).read().strip())
            except OSError as e:
                print("   ", q, e)
try:
    print(subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=60).stdout[-1500:])
except Exception as e:
    print("rocm-smi:", e)
cd = x266_amd.Codec(0)
w, h, rng = 3840, 2160, 64
cur = torch.randint(0, 256, (h, w), device="cuda", dtype=torch.int32).to(torch.uint8)
refp = torch.randint(0, 256, (h + 2 * rng, w + 2 * rng), device="cuda", dtype=torch.int32).to(torch.uint8)
best = torch.empty((h // 8) * (w // 8) * 2, dtype=torch.int32, device="cuda")
org = refp.data_ptr() + rng * refp.stride(0) + rng
stop = False
samples = []
def sampler():
    files = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
    while not stop:
        row = []
        for f in files:
            try:
                t = open(f).read()
                row.append((f.split("/")[4], [l for l in t.splitlines() if "*" in l] or t.strip()))
            except OSError:
                pass
        samples.append(row)
        time.sleep(0.05)
th = threading.Thread(target=sampler); th.start()
t0 = time.time()
while time.time() - t0 < 1.5:
    for _ in range(20):
        cd.satd_search_dev(cur.data_ptr(), cur.stride(0), org, refp.stride(0), w, h, rng, best.data_ptr())
    torch.cuda.synchronize()
stop = True; th.join()
for row in samples[::6]:
    print(row)
