"""probe: does the radix sort run faster per key when keys + ping-pong buffer fit the 256 MB memory-side cache?
python3 tools/sort_size_probe.py   (random 57-bit keys of the configs[2] layout, 4 passes over the group bits, n = 2 M .. 38 M)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fastf_amd as F
from fastf_amd import synth

bt, ft, _, _ = synth.make_lists(50000, 36601, seed=77)
lists = F.Lists(bt, ft, 0.5, 926)
eng = F.Engine.from_lists(lists, rate_depth=0.5, seed=926, umi_max_bases=12)
dev = torch.device("cuda", 0)
bits = eng.key_bits
print("key bits", bits, "skip", eng.skip_bits if hasattr(eng, "skip_bits") else "?")
g = torch.Generator(device=dev); g.manual_seed(1)
for n in (2 << 20, 4 << 20, 8 << 20, 16 << 20, 38 << 20):
    eng.reserve(n, n)
    keys0 = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device=dev, generator=g) >> (62 - bits)
    keys = keys0.clone(); tmp = torch.empty_like(keys)
    d_n = torch.tensor([n], dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    for rep in range(3):
        keys.copy_(keys0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.dev_sort(keys.data_ptr(), tmp.data_ptr(), d_n.data_ptr(), n, stream=s, skip_low=True)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    passes = 4
    print("n = %9d (%4d MB + %4d MB): %.3f ms for %d passes = %.1f ns per key and pass-pair, %.0f GB/s (16 B moved + 8 B counted per key and pass)"
          % (n, n * 8 >> 20, n * 8 >> 20, ms, passes, ms * 1e6 / n / passes, n * 24 * passes / ms / 1e6))
eng.close()
