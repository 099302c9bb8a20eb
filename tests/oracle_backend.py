"""A compute backend for the CPU tests of the key scheduler: same interface as merging.HipBackend, arithmetic by
the CPU oracle.  TEST DOUBLE ONLY -- it lives under tests/ because only tests may use the oracle; the product has no
CPU backend (merging.HipBackend raises without a ROCm device)."""
import numpy as np
import torch

from oracle import oracle as O

_NAME = {torch.float32: "float32", torch.float16: "float16", torch.bfloat16: "bfloat16"}


def _bits(t: torch.Tensor) -> np.ndarray:
    if t.dtype == torch.float32:
        return t.numpy()
    return t.view(torch.int16).numpy().view(np.uint16)


class OracleBackend:
    def __init__(self):
        self.device = torch.device("cpu")
        self.methods = None
        O.build()

    def stream(self):
        return None

    def workspace_bytes(self, plan, layout):
        return 0

    def run_fused(self, plan, arenas, out, layout, segs_dev, segs_host, flags, workspace):
        in_name, out_name = _NAME[arenas[0].dtype], _NAME[out.dtype]
        a_bits = [_bits(a) for a in arenas]
        o_bits = _bits(out)
        fl = flags.numpy()
        for i, (off, n) in enumerate(zip(layout.offsets, layout.numels)):
            f32 = [O.to_f32(b[off:off + n], in_name) for b in a_bits]
            if not all(O.all_finite(x) for x in f32):
                fl[i] |= 1
                continue
            if plan.kind == "wsum":
                res = O.weighted_sum(f32[0], f32[1], plan.params["alpha"])
            elif plan.kind == "navg":
                res = O.n_average(f32)
            else:
                res = O.add_difference_ties(f32[0], f32[1:], k=plan.params["k"], alpha=plan.params["alpha"], vote_sgn=plan.params["vote_sgn"])
            if not O.all_finite(res):
                fl[i] |= 2
            o_bits[off:off + n] = O.from_f32(res, out_name) if out_name != "float32" else res

    def run_lora(self, entries, base, out, flags):
        """base + sum of LoRA deltas per key (lycoris.py:226-263 + linear.py:64-76) with the oracle's arithmetic."""
        in_name, out_name = _NAME[base.dtype], _NAME[out.dtype]
        b_bits, o_bits, fl = _bits(base), _bits(out), flags.numpy()
        for index, off, rows, cols, nets in entries:
            n = rows * cols
            acc = O.to_f32(b_bits[off:off + n], in_name).reshape(rows, cols)
            ok = O.all_finite(acc)
            for up, down, alpha, weight in nets:
                up32 = O.to_f32(_bits(up.reshape(up.shape[0], -1)), _NAME[up.dtype]).reshape(up.shape[0], -1)
                down32 = O.to_f32(_bits(down.reshape(down.shape[0], -1)), _NAME[down.dtype]).reshape(down.shape[0], -1)
                ok = ok and O.all_finite(up32) and O.all_finite(down32)
                a = None if alpha is None else float(torch.as_tensor(alpha).float())
                acc = O.add_difference(acc, O.lora_delta(up32, down32, a), weight)
            if not ok:
                fl[index] |= 1
                continue
            if not O.all_finite(acc):
                fl[index] |= 2
            o_bits[off:off + n] = O.from_f32(acc, out_name).reshape(-1) if out_name != "float32" else acc.reshape(-1)
