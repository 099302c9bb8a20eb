"""Times ties_sum on fp32 deltas (the instantiations of the count pass with four fp32 streams): python tools/time_ties_f32.py [n_deltas] [elements]"""
import json
import sys
import pathlib
import torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import sd_mecha_amd as amd

n_deltas = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 27
g = torch.Generator(device="cuda").manual_seed(1)
deltas = [torch.randn(n, device="cuda", generator=g) * 0.01 for _ in range(n_deltas)]
for _ in range(2):
    amd.ties_sum(*deltas, k=0.2)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
a.record()
for _ in range(reps):
    amd.ties_sum(*deltas, k=0.2)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
print(json.dumps({"n_deltas": n_deltas, "elements": n, "ms_per_step": round(ms, 3), "GBs_read_twice_plus_write": round((2 * n_deltas + 1) * 4 * n / ms / 1e6, 1)}))
