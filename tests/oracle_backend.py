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
