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


class NonFinite(Exception):
    pass


def _np(t) -> np.ndarray:
    return t.detach().cpu().float().contiguous().numpy()


def _scalar(x) -> float:
    return float(x.item()) if isinstance(x, torch.Tensor) else float(x)


class OracleMethods:
    """What merging._KeyRun needs from `backend.methods` (cast of a loaded value, LoRA delta), by the oracle."""

    @staticmethod
    def cast(t, dtype=torch.float32, *, key=None, **_):
        from sd_mecha_amd import NonFiniteStateDictKeyError
        x = O.to_f32(_bits(t.contiguous().reshape(-1)), _NAME[t.dtype]) if t.dtype != torch.float32 else t.reshape(-1).numpy()
        if not O.all_finite(x):
            raise NonFiniteStateDictKeyError(key)
        return torch.from_numpy(np.array(x, dtype=np.float32)).reshape(t.shape)

    @staticmethod
    def lora_delta(up, down, alpha=None, target_shape=None, **_):
        a = None if alpha is None else float(torch.as_tensor(alpha).float())
        up32, down32 = _np(up).reshape(up.shape[0], -1), _np(down).reshape(down.shape[0], -1)
        return torch.from_numpy(O.lora_delta(up32, down32, a)).reshape(target_shape)


def _resolve(x, key):
    return x[key] if not isinstance(x, torch.Tensor) else x


def _t(a: np.ndarray, like: torch.Tensor) -> torch.Tensor:
    return torch.from_numpy(np.asarray(a, dtype=np.float32).copy()).reshape(like.shape)


def _bodies():
    """Method bodies by the oracle, on fp32 CPU tensors, for the identifiers the CPU tests of the per-key path use.  They
    index their StateDict-typed parameters the way the reference's bodies do (linear.py:16-23, :35-39, :72-76)."""
    import math

    def weighted_sum(a, b, alpha=0.5, *, key=None, **_):
        al = _scalar(alpha)
        if math.isclose(al, 0.0):
            return _resolve(a, key)
        if math.isclose(al, 1.0):
            return _resolve(b, key)
        a, b = _resolve(a, key), _resolve(b, key)
        return _t(O.weighted_sum(_np(a).reshape(-1), _np(b).reshape(-1), al), a)

    def n_average(*models, key=None, **_):
        ms = [_resolve(m, key) for m in models]
        return _t(O.n_average([_np(m).reshape(-1) for m in ms]), ms[0])

    def add_difference(a, b, alpha=1.0, *, key=None, **_):
        al = _scalar(alpha)
        if math.isclose(al, 0.0):
            return _resolve(a, key)
        b = _resolve(b, key)
        a = _resolve(a, key)
        return _t(O.add_difference(_np(a).reshape(-1), _np(b).reshape(-1), al), a)

    def subtract(a, b, **_):
        return _t(O.subtract(_np(a).reshape(-1), _np(b).reshape(-1)), a)

    def ties_sum(*models, k=1.0, vote_sgn=False, **_):
        if not models:
            return 0
        if models[0].numel() == 0:
            return models[0]
        return _t(O.ties_sum([_np(m).reshape(-1) for m in models], k=float(k), vote_sgn=bool(vote_sgn)), models[0])

    def scale(a, alpha=1.0, *, key=None, **_):
        al = _scalar(alpha)
        if math.isclose(al, 0.0):
            return 0.0
        a = _resolve(a, key)
        return _t(O.scale(_np(a).reshape(-1), al), a)

    return {"scale": scale, "weighted_sum": weighted_sum, "n_average": n_average, "add_difference": add_difference, "subtract": subtract,
            "ties_sum": ties_sum}


class OracleBackend:
    def __init__(self):
        self.device = torch.device("cpu")
        self.methods = OracleMethods
        self.bodies = _bodies()
        O.build()

    def stream(self):
        return None

    def workspace_bytes(self, plan, layout, shapes=None):
        return 0

    def run_fused(self, plan, arenas, out, layout, segs_dev, segs_host, flags, workspace, shapes=None):
        in_name, out_name = _NAME[arenas[0].dtype], _NAME[out.dtype]
        a_bits = [_bits(a) for a in arenas]
        o_bits = _bits(out)
        fl = flags.numpy()
        for i, (off, n) in enumerate(zip(layout.offsets, layout.numels)):
            f32 = [O.to_f32(b[off:off + n], in_name) for b in a_bits]
            used = f32
            if plan.kind == "wsum" and O.isclose(plan.params["alpha"], 0.0):        # the kernel reads only `a` (linear.py:18-19)
                used = f32[:1]
            elif plan.kind == "wsum" and O.isclose(plan.params["alpha"], 1.0):
                used = f32[1:2]
            if not all(O.all_finite(x) for x in used):
                fl[i] |= 1
                continue
            if plan.kind == "wsum":
                res = O.weighted_sum(f32[0], f32[1], plan.params["alpha"])
            elif plan.kind == "navg":
                res = O.n_average(f32)
            elif plan.kind == "slerp":
                res = O.slerp(f32[0], f32[1], plan.params["alpha"])
            elif plan.kind == "addperp":
                res = O.add_perpendicular(f32[0], f32[1], f32[2], plan.params["alpha"])
            elif plan.kind == "addsub":
                res = O.add_difference(f32[0], O.subtract(f32[1], f32[2]), plan.params["alpha"])
            elif plan.kind == "adddiff":
                res = O.add_difference(f32[0], f32[1], plan.params["alpha"])
            elif plan.kind == "dare":
                q = plan.params
                shp = shapes[i] if q["model_stock"] else (n,)
                ties = O.ties_sum_with_dropout([O.subtract(m, f32[0]).reshape(shp) for m in f32[1:]], probability=q["probability"], rescale=q["rescale"],
                                               k=q["k"], vote_sgn=q["vote_sgn"], apply_stock=not q["model_stock"], cos_eps=q["cos_eps"], seed=q["seed"])
                res = O.add_difference(f32[0], np.ascontiguousarray(ties).reshape(-1), q["alpha"])
            elif plan.kind == "rotate":
                res = O.rotate(f32[0].reshape(shapes[i]), f32[1].reshape(shapes[i]), alignment=float(plan.params["alignment"]),
                               alpha=plan.params["alpha"], centralization=plan.params["centralization"])
                res = np.ascontiguousarray(res, dtype=np.float32).reshape(-1)
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
