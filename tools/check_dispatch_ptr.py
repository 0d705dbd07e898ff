#!/usr/bin/env python3
"""Lists the kernels of orb_slam2_ros2_amd/csrc that ask for the DISPATCH POINTER (none should).

A kernel that keeps a private array the compiler cannot scalarise gets it moved to LDS, addressed by the flat thread number; for that number
the compiler reads the workgroup's y / z sizes from the dispatch packet, and the packet lives in the queue's ring buffer in HOST memory: every
wave of every launch then begins with a scalar load across PCIe that no cache holds (r6: k_brief, 13.6 us per wave of a pair's launch).  The
same goes for blockDim / gridDim.  Run after touching a kernel:  python tools/check_dispatch_ptr.py   (exit code 1 if any kernel is listed)"""
import glob, os, re, subprocess, sys, tempfile
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "orb_slam2_ros2_amd", "csrc")
flags = "-std=c++17 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-function -I../../include -S --cuda-device-only".split()
bad = []
def one(src, tmp):
    out = os.path.join(tmp, os.path.basename(src) + ".s")
    extra = ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"] if src.endswith("k_fast.hip") else []
    r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, *extra, os.path.basename(src), "-o", out], cwd=root, capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out):
        return src, None, r.stderr[-300:]
    s = open(out).read()
    hits = [m.group(1) for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S)
            if "user_sgpr_dispatch_ptr 1" in m.group(2) or "user_sgpr_queue_ptr 1" in m.group(2)]
    return src, hits, ""


from concurrent.futures import ThreadPoolExecutor
with tempfile.TemporaryDirectory() as tmp, ThreadPoolExecutor(max_workers=8) as ex:
    for src, hits, err in ex.map(lambda f: one(f, tmp), sorted(glob.glob(os.path.join(root, "*.hip")))):
        if hits is None:
            print("cannot compile", src, err); sys.exit(2)
        bad += [(os.path.basename(src), k) for k in hits]
for f, k in bad:
    print(f, k)
print(f"{len(bad)} kernel(s) read the dispatch packet" if bad else "no kernel reads the dispatch packet")
sys.exit(1 if bad else 0)
