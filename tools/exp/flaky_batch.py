"""Stress for run-to-run / slot-to-slot identity of the batched path with REUSED device memory: contexts of different geometry are
created, used and destroyed in one process (like the test suite does), so a fresh context's buffers start from the previous
context's bytes; every slot that holds the same image must then hold the same features and stereo results."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from orb_slam2_ros2_amd import synth, _lib as lib
FX, BF = 718.856, 386.1448
base = [synth.stereo_pair(f) for f in range(40, 44)]
small = synth.stereo_pair(3, 640, 480)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dl = torch.from_numpy(np.stack([base[i % 4][0] for i in range(B)])).cuda()
dr = torch.from_numpy(np.stack([base[i % 4][1] for i in range(B)])).cuda()
bad = 0
for it in range(int(sys.argv[1])):
    junk = torch.randint(0, 255, (3 << 30,), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
    c2 = lib.Context(640, 480, n_features=1000 + 37 * it, max_images=8 + it % 5)
    c2.extract(small[0]); c2.close()
    ctx = lib.Context(1241, 376, max_images=2 * B)
    for rep in range(2):
        ctx.stereo_batch_device(dl.data_ptr(), dr.data_ptr(), 1241, 1241 * 376, B, FX, BF)
        ctx.sync()
        ref = {}
        for s in range(2 * B):
            k, d = ctx.fetch_features(s)
            key = (s % 2, (s // 2) % 4)
            if key not in ref:
                ref[key] = (k, d)
            else:
                rk, rd = ref[key]
                if len(k) != len(rk) or not np.array_equal(k, rk) or not np.array_equal(d, rd):
                    bad += 1
                    n = min(len(k), len(rk))
                    diff = [i for i in range(n) if k[i] != rk[i] or not np.array_equal(d[i], rd[i])]
                    print("iter", it, rep, "slot", s, "len", len(k), len(rk), "first diffs", diff[:5], "octaves",
                          [int(k["octave"][i]) for i in diff[:5]], "n diff", len(diff), flush=True)
                    if diff:
                        i = diff[0]
                        print("   got", k[i], "ref", rk[i])
    ctx.close()
print("bad", bad)
