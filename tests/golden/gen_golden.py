#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the build container (it imports /root/reference, which never
travels to the GPU box).  It calls the reference's own method bodies
(``method.__wrapped__`` -- the way upstream tests do, tests/merge_methods/
test_ties.py:29) and ``sd_mecha.merge`` on small seeded inputs, and stores
inputs + expected outputs as data in ``.npz`` files.  No reference source text
is stored.

    python tests/golden/gen_golden.py        # rewrites tests/golden/*.npz

Bootstrap: ``fuzzywuzzy`` is not installed and is used by
the reference only to decorate error messages, so a two-line stand-in module is
registered before the import.
"""
import pathlib
import sys
import tempfile
import types

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent

fw = types.ModuleType("fuzzywuzzy")
fwp = types.ModuleType("fuzzywuzzy.process")
fwp.extractOne = lambda q, choices, **k: (next(iter(choices), None), 0)
fw.process = fwp
sys.modules["fuzzywuzzy"] = fw
sys.modules["fuzzywuzzy.process"] = fwp
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402
import sd_mecha  # noqa: E402
from sd_mecha.extensions.builtin.merge_methods import linear, ties, svd  # noqa: E402
from sd_mecha.extensions.builtin import lycoris  # noqa: E402

TDT = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}
SHAPES = [(), (0,), (7,), (33, 7), (64, 96), (8, 4, 3, 3), (1, 1)]


def bits(t: torch.Tensor) -> np.ndarray:
    """Bit pattern of a torch tensor as a numpy array (uint16 for half types, uint32 view for fp32)."""
    t = t.detach().contiguous().cpu()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    if t.dtype == torch.float32:
        return t.numpy().view(np.uint32).copy()
    raise TypeError(t.dtype)


def rand_half(rng, shape, dtype, std=0.02) -> torch.Tensor:
    x = torch.from_numpy(rng.normal(0.0, std, size=shape).astype(np.float32))
    return x.to(TDT[dtype])


def sd(t):
    """StateDict-typed parameters accept a plain dict plus key=."""
    return {"k": t}


def gen_streaming(out):
    rng = np.random.default_rng(20260424)
    case = 0
    for shape in SHAPES:
        for dtype in ("bfloat16", "float16"):
            a, b, c = (rand_half(rng, shape, dtype) for _ in range(3))
            af, bf, cf = a.float(), b.float(), c.float()
            pre = f"s{case}"
            out[pre + "_a"], out[pre + "_b"], out[pre + "_c"] = bits(a), bits(b), bits(c)
            out[pre + "_dtype"] = np.array(dtype)
            out[pre + "_shape"] = np.array(shape, dtype=np.int64)
            for alpha in (0.5, 0.3, 0.75, 0.0, 1.0):
                al = torch.tensor(alpha, dtype=torch.float32)
                out[f"{pre}_wsum_{alpha}"] = bits(linear.weighted_sum.__wrapped__(sd(af), sd(bf), al, key="k"))
                out[f"{pre}_adddiff_{alpha}"] = bits(linear.add_difference.__wrapped__(sd(af), sd(bf), al, key="k"))
            out[pre + "_sub"] = bits(linear.subtract.__wrapped__(af, bf))
            out[pre + "_scale_0.7"] = bits(linear.scale.__wrapped__(sd(af), torch.tensor(0.7), key="k"))
            for n in (2, 3, 8):
                ms = [a, b, c] + [rand_half(rng, shape, dtype) for _ in range(n - 3)] if n > 3 else [a, b, c][:n]
                if n > 3:
                    for j, m in enumerate(ms[3:]):
                        out[f"{pre}_navg{n}_extra{j}"] = bits(m)
                out[f"{pre}_navg{n}"] = bits(linear.n_average.__wrapped__(*[sd(m.float()) for m in ms], key="k"))
            if len(shape) > 0 and int(np.prod(shape)) > 1:
                for alpha in (0.5, 0.3):
                    out[f"{pre}_slerp_{alpha}"] = bits(linear.slerp.__wrapped__(af, bf, torch.tensor(alpha)))
                out[pre + "_perp"] = bits(linear.perpendicular_component.__wrapped__(af, bf))
                # add_perpendicular graph (wrappers.py:42-64) from primitives
                ad, bd = linear.subtract.__wrapped__(af, cf), linear.subtract.__wrapped__(bf, cf)
                perp = linear.perpendicular_component.__wrapped__(ad, bd)
                out[pre + "_addperp_1.0"] = bits(linear.add_difference.__wrapped__(sd(af), sd(perp), torch.tensor(1.0), key="k"))
            # output casts (root cast node, merging.py:157-158)
            r = linear.weighted_sum.__wrapped__(sd(af), sd(bf), torch.tensor(0.3), key="k")
            out[pre + "_wsum_0.3_as_bf16"] = bits(r.to(torch.bfloat16))
            out[pre + "_wsum_0.3_as_f16"] = bits(r.to(torch.float16))
            case += 1
    out["n_cases"] = np.array(case)


def gen_alpha_tensor(out):
    """weighted_sum / add_difference with a FULL alpha tensor (linear.py:16-23, :71-76: alpha.numel() > 1): one weight per
    element, per row (broadcast) and per column, weights on both sides of torch.lerp's |w| < 0.5 test."""
    rng = np.random.default_rng(20261110)
    case = 0
    for shape, ashape in [((33, 7), (33, 7)), ((64, 96), (64, 1)), ((64, 96), (96,)), ((8, 4, 3, 3), (8, 1, 1, 1)), ((8, 4, 3, 3), (8, 4, 3, 3)), ((7,), (7,))]:
        for dtype in ("bfloat16", "float16"):
            a, b = rand_half(rng, shape, dtype), rand_half(rng, shape, dtype)
            alpha = torch.from_numpy(rng.uniform(-0.25, 1.25, size=ashape).astype(np.float32))
            pre = f"a{case}"
            out[pre + "_a"], out[pre + "_b"], out[pre + "_alpha"] = bits(a), bits(b), bits(alpha)
            out[pre + "_dtype"] = np.array(dtype)
            out[pre + "_shape"] = np.array(shape, dtype=np.int64)
            out[pre + "_alpha_shape"] = np.array(ashape, dtype=np.int64)
            out[pre + "_wsum"] = bits(linear.weighted_sum.__wrapped__(sd(a.float()), sd(b.float()), alpha, key="k"))
            out[pre + "_adddiff"] = bits(linear.add_difference.__wrapped__(sd(a.float()), sd(b.float()), alpha, key="k"))
            case += 1
    out["n_cases"] = np.array(case)


def gen_ties(out):
    rng = np.random.default_rng(1708)
    case = 0
    combos = [(shape, n, "bfloat16") for shape in [(7,), (33, 7), (64, 48), (8, 4, 3, 3), (1, 1)] for n in (2, 3, 7)]
    combos += [((33, 7), 3, "float16"), ((64, 48), 7, "float16"), ((129, 65), 3, "bfloat16")]
    for shape, n_models, dtype in combos:
        base = rand_half(rng, shape, dtype)
        models = [rand_half(rng, shape, dtype) for _ in range(n_models)]
        pre = f"t{case}"
        out[pre + "_base"] = bits(base)
        out[pre + "_models"] = np.stack([bits(m) for m in models])
        out[pre + "_dtype"] = np.array(dtype)
        deltas = [linear.subtract.__wrapped__(m.float(), base.float()) for m in models]
        for k in (0.2, 0.5, 1.0):
            numel = deltas[0].numel()
            kth = max(int((1 - k) * numel), 1)
            thr = [d.flatten().abs().float().kthvalue(kth)[0] for d in deltas]
            out[f"{pre}_k{k}_kth"] = np.array(kth)
            out[f"{pre}_k{k}_thr"] = np.stack([bits(t) for t in thr])
            keep = np.stack([(d.abs() >= t).numpy() for d, t in zip(deltas, thr)])
            out[f"{pre}_k{k}_keep"] = np.packbits(keep.reshape(n_models, -1), axis=1)
            for vote in ((False, True) if k == 0.2 else (False,)):
                filt, counts = ties.ties_sum_deltas(*deltas, k=k, vote_sgn=vote)
                stacked = torch.stack([ties.filter_top_k(d, k) for d in deltas])
                fsign = torch.sign(torch.sum(stacked if vote else torch.sign(stacked), dim=0))
                res = ties.ties_sum.__wrapped__(*deltas, k=k, vote_sgn=vote)
                out[f"{pre}_k{k}_v{int(vote)}_out"] = bits(res)
                out[f"{pre}_k{k}_v{int(vote)}_sign"] = fsign.numpy().astype(np.int8)
                out[f"{pre}_k{k}_v{int(vote)}_count"] = counts.numpy().astype(np.uint8)
                if not vote:
                    merged = linear.add_difference.__wrapped__(sd(base.float()), sd(res), torch.tensor(1.0), key="k")
                    out[f"{pre}_k{k}_merged_{'bf16' if dtype == 'bfloat16' else 'f16'}"] = bits(merged.to(TDT[dtype]))
        case += 1
    out["n_cases"] = np.array(case)
    # upstream known-answer vectors are re-stated in tests/test_oracle_golden.py (they are data in upstream tests)


def gen_dare(out):
    rng = np.random.default_rng(114514)
    case = 0
    for shape in [(16,), (33, 7), (48, 32)]:
        for n_models in (2, 3):
            deltas = [torch.from_numpy(rng.normal(0, 0.02, size=shape).astype(np.float32)) for _ in range(n_models)]
            for seed in (0, 114514, -1):                 # -1 stands for seed=None: the generator is never seeded (ties.py:29-31)
                for p in (0.4, 0.9):
                    if seed < 0 and (p != 0.4 or n_models != 2):
                        continue
                    pre = f"d{case}"
                    out[pre + "_deltas"] = np.stack([bits(d) for d in deltas])
                    out[pre + "_seed"] = np.array(seed)
                    out[pre + "_p"] = np.array(p)
                    g = torch.Generator("cpu")
                    if seed >= 0:
                        g.manual_seed(seed)
                    masks = [ties.find_della_dropout(d, torch.tensor(p), 0.0, g) for d in deltas]
                    out[pre + "_masks"] = np.packbits(np.stack([m.numpy().astype(bool).reshape(-1) for m in masks]), axis=1)
                    for apply_stock in (True, False):
                        for rescale in (True, False):
                            res = ties.ties_sum_with_dropout.__wrapped__(
                                *deltas, probability=torch.tensor(p), della_eps=0.0, rescale=rescale, k=0.5,
                                vote_sgn=False, apply_stock=apply_stock, cos_eps=1e-6, apply_median=False,
                                eps=1e-6, maxiter=100, ftol=1e-20, seed=seed if seed >= 0 else None)
                            out[f"{pre}_stock{int(apply_stock)}_rescale{int(rescale)}"] = bits(res)
                    case += 1
    out["n_cases"] = np.array(case)


def gen_stock_dropout(out):
    """model_stock (ties.py:137-159) and dropout (ties.py:202-238) through the reference's own bodies."""
    rng = np.random.default_rng(2718)
    case = 0
    for shape in [(16,), (33, 7), (48, 32), (8, 4, 3, 3)]:
        for n_models in (2, 3, 5):
            deltas = [torch.from_numpy(rng.normal(0, 0.02, size=shape).astype(np.float32)) for _ in range(n_models)]
            pre = f"c{case}"
            out[pre + "_deltas"] = np.stack([bits(d) for d in deltas])
            out[pre + "_stock"] = bits(ties.model_stock.__wrapped__(*deltas, cos_eps=1e-6))
            for seed, p, rescale in ((7, 0.9, 1.0), (11, 0.4, 0.5), (3, 0.0, 1.0)):
                res = ties.dropout.__wrapped__(*deltas, probability=p, rescale=rescale, overlap=1.0, overlap_emphasis=0.0, seed=seed)
                out[f"{pre}_dropout_s{seed}"] = bits(res)
            case += 1
    out["n_cases"] = np.array(case)
    out["dropout_settings"] = np.array([[7, 0.9, 1.0], [11, 0.4, 0.5], [3, 0.0, 1.0]])


def gen_rotate(out):
    rng = np.random.default_rng(360)
    case = 0
    for shape in [(64, 48), (32, 32), (24, 64), (8, 4, 3, 3), (48, 16, 1, 1), (12,)]:
        a = torch.from_numpy(rng.normal(0, 0.02, size=shape).astype(np.float32)).to(torch.bfloat16).float()
        b = torch.from_numpy(rng.normal(0, 0.02, size=shape).astype(np.float32)).to(torch.bfloat16).float()
        pre = f"r{case}"
        out[pre + "_a"], out[pre + "_b"] = bits(a), bits(b)
        for alpha in (0.0, 0.5):
            for cz in (1.0, 0.0):
                res = svd.rotate.__wrapped__(sd(a), sd(b), alignment=1.0, alpha=torch.tensor(alpha), centralization=cz, key="k")
                out[f"{pre}_out_alpha{alpha}_cz{cz}"] = bits(res)
        # the other whole-number alignments (svd.py:84-85, :198-215, :177-183) and the factor cache (svd.py:55-79):
        # the second call reads the transform the first one saved in fp16
        for alignment, alpha in ((0.0, 0.5), (-1.0, 0.25), (-1.0, 0.0)):
            for cz in (1.0, 0.0):
                res = svd.rotate.__wrapped__(sd(a), sd(b), alignment=alignment, alpha=torch.tensor(alpha), centralization=cz, key="k")
                out[f"{pre}_out_align{alignment}_alpha{alpha}_cz{cz}"] = bits(res)
        import collections
        cache = collections.defaultdict(dict)
        first = svd.rotate.__wrapped__(sd(a), sd(b), alignment=1.0, alpha=torch.tensor(0.0), centralization=1.0, key="k", cache=cache)
        second = svd.rotate.__wrapped__(sd(a), sd(b), alignment=1.0, alpha=torch.tensor(0.5), centralization=1.0, key="k", cache=cache)
        out[f"{pre}_out_cached_alpha0.5"] = bits(second)
        if len(shape) >= 2:
            is_conv = len(shape) == 4 and int(np.prod(shape[-2:])) != 1
            s2 = (int(np.prod(shape[:2])), int(np.prod(shape[2:]))) if is_conv else (shape[0], int(np.prod(shape[1:])))
            a2, b2 = a.reshape(s2), b.reshape(s2)
            a2 = a2 - a2.mean(0)
            b2 = b2 - b2.mean(0)
            tr = svd.orthogonal_procrustes(a2, b2)
            eye = torch.eye(s2[1])
            out[pre + "_R"] = bits(tr(eye, 1.0))        # the p x p matrix the transform applies (I @ R)
        case += 1
    out["n_cases"] = np.array(case)


def gen_truncate_rank(out):
    """svd.truncate_rank (svd.py:89-143) with use_approximate_basis=False: the Eckart-Young rank-r matrix, on full-rank Gaussian
    keys (every singular-value gap well above fp32 resolution) in bf16 / fp32, 2-D, conv-shaped (flatten(start_dim=1)) and 1-D;
    plus the early-outs (rank >= max rank -> a, rank 0 -> zeros)."""
    rng = np.random.default_rng(143)
    case = 0
    for shape, dtype, ranks in [((64, 48), "float32", (1, 8, 47)), ((24, 64), "bfloat16", (4, 23)), ((96, 96), "float32", (16, 64)),
                                ((8, 4, 3, 3), "float32", (3,)), ((300, 272), "float32", (8, 100)), ((12,), "float32", (4,))]:
        a = rand_half(rng, shape, dtype) if dtype != "float32" else torch.from_numpy(rng.normal(0, 0.02, size=shape).astype(np.float32))
        pre = f"t{case}"
        out[pre + "_a"] = bits(a)
        out[pre + "_dtype"] = np.array(dtype)
        out[pre + "_shape"] = np.array(shape)
        out[pre + "_ranks"] = np.array(ranks)
        for r in ranks:
            res = svd.truncate_rank.__wrapped__(a.float(), rank=r, use_approximate_basis=False)
            out[f"{pre}_out_rank{r}"] = bits(res.float())
        if len(shape) >= 2:
            full = min(shape[0], int(np.prod(shape[1:])))
            assert torch.equal(svd.truncate_rank.__wrapped__(a.float(), rank=full, use_approximate_basis=False), a.float())
            assert not svd.truncate_rank.__wrapped__(a.float(), rank=0, use_approximate_basis=False).any()
        case += 1
    out["n_cases"] = np.array(case)


def gen_lora(out):
    rng = np.random.default_rng(16)
    cases = [((64, 8), (8, 48), (64, 48)), ((32, 8, 1, 1), (8, 16, 3, 3), (32, 16, 3, 3)), ((160, 8), (8, 192), (160, 192))]
    for i, (us, ds, ts) in enumerate(cases):
        up = rand_half(rng, us, "float16", std=0.05)
        down = rand_half(rng, ds, "float16", std=0.05)
        alpha = torch.tensor(4.0, dtype=torch.float16)
        state = {"lora_x.lora_up.weight": up.float(), "lora_x.lora_down.weight": down.float(), "lora_x.alpha": alpha.float()}
        helper = lycoris.StateDictKeyHelper(state, "lora_x")
        pre = f"l{i}"
        out[pre + "_up"], out[pre + "_down"] = bits(up), bits(down)
        out[pre + "_target_shape"] = np.array(ts, dtype=np.int64)
        out[pre + "_delta_alpha4"] = bits(lycoris.LoraAlgorithm.compose_alpha(helper, torch.Size(ts)))
        out[pre + "_delta_noalpha"] = bits(lycoris.LoraAlgorithm.compose(helper, torch.Size(ts)))
        base = rand_half(rng, ts, "bfloat16")
        out[pre + "_base"] = bits(base)
        delta = lycoris.LoraAlgorithm.compose_alpha(helper, torch.Size(ts))
        merged = linear.add_difference.__wrapped__(sd(base.float()), sd(delta), torch.tensor(0.75), key="k")
        out[pre + "_merged_0.75_bf16"] = bits(merged.to(torch.bfloat16))
    out["n_cases"] = np.array(len(cases))


def gen_e2e(out):
    """The whole chain through sd_mecha.merge (cast nodes, omit_non_finite, root cast) on a 4-key toy checkpoint set."""
    import safetensors.torch
    from tqdm import tqdm

    class quiet(tqdm):
        def __init__(self, *a, **k):
            k["disable"] = True
            super().__init__(*a, **k)

    rng = np.random.default_rng(4)
    shapes = {"w.linear": (128, 96), "w.conv": (32, 16, 3, 3), "w.bias": (128,), "w.scalar": ()}
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        paths = []
        for m in range(4):
            state = {k: rand_half(rng, s, "bfloat16") for k, s in shapes.items()}
            if m == 3:
                state["w.bias"][5] = float("inf")      # non-finite input: the key must be skipped where this model is read
            p = td / f"m{m}.safetensors"
            safetensors.torch.save_file(state, str(p))
            paths.append(p)
            for k, t in state.items():
                out[f"e_m{m}_{k}"] = bits(t)
        models = [sd_mecha.model(p) for p in paths]
        kw = dict(threads=0, output_dtype=torch.bfloat16, tqdm=quiet)

        def store(name, res):
            res = {k: v for k, v in res.items() if k != "__metadata__"}
            out[f"e_{name}_keys"] = np.array(sorted(res.keys()))
            for k, v in res.items():
                out[f"e_{name}_{k}"] = bits(v)

        store("wsum", sd_mecha.merge(linear.weighted_sum(models[0], models[1], alpha=0.5), **kw))
        store("wsum_f16", sd_mecha.merge(linear.weighted_sum(models[0], models[1], alpha=0.3), **{**kw, "output_dtype": torch.float16}))
        store("navg3", sd_mecha.merge(linear.n_average(models[0], models[1], models[2]), **kw))
        store("navg4", sd_mecha.merge(linear.n_average(*models), **kw))      # m3 has a non-finite bias
        deltas = [linear.subtract(m, models[0]) for m in models[1:3]]
        store("ties", sd_mecha.merge(linear.add_difference(models[0], ties.ties_sum(*deltas, k=0.2), alpha=1.0), **kw))
        deltas4 = [linear.subtract(m, models[0]) for m in models[1:4]]
        store("ties4_fallback", sd_mecha.merge(
            linear.add_difference(models[0], ties.ties_sum(*deltas4, k=0.2), alpha=1.0) | models[0], **kw))
    out["shapes_keys"] = np.array(list(shapes.keys()))


def gen_e2e_plans(out):
    """Round 3: the graph shapes the scheduler gained batched plans for, through the real sd_mecha.merge on the SAME four toy
    checkpoints as gen_e2e (read back from e2e.npz): slerp (linear.py:43-61), the add_perpendicular graph (wrappers.py:42-64),
    add_difference(a, b - c) (wrappers.py:9-39 with `c`), add_difference(a, delta model), and add_ties_with_dare-shaped graphs
    (wrappers.py:231-273; ties.py:10-50) with apply_stock True (no Model-Stock factor, ties.py:90) and False (factor applied)."""
    import safetensors.torch
    from tqdm import tqdm

    class quiet(tqdm):
        def __init__(self, *a, **k):
            k["disable"] = True
            super().__init__(*a, **k)

    g = np.load(HERE / "e2e.npz")
    shapes = {"w.linear": (128, 96), "w.conv": (32, 16, 3, 3), "w.bias": (128,), "w.scalar": ()}
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        paths = []
        for m in range(4):
            state = {k: torch.from_numpy(g[f"e_m{m}_{k}"].view(np.int16).copy()).view(torch.bfloat16).reshape(s) for k, s in shapes.items()}
            p = td / f"m{m}.safetensors"
            safetensors.torch.save_file(state, str(p))
            paths.append(p)
        # a fifth checkpoint with an all-zero bias: slerp and perpendicular_component take their NaN branches on that key
        state = {k: torch.from_numpy(g[f"e_m1_{k}"].view(np.int16).copy()).view(torch.bfloat16).reshape(s).clone() for k, s in shapes.items()}
        state["w.bias"].zero_()
        safetensors.torch.save_file(state, str(td / "z.safetensors"))
        out["p_z_w.bias"] = bits(state["w.bias"])
        ms = [sd_mecha.model(p) for p in paths]
        z = sd_mecha.model(td / "z.safetensors")
        kw = dict(threads=0, output_dtype=torch.bfloat16, tqdm=quiet)

        def store(name, res):
            res = {k: v for k, v in res.items() if k != "__metadata__"}
            out[f"p_{name}_keys"] = np.array(sorted(res.keys()))
            for k, v in res.items():
                out[f"p_{name}_{k}"] = bits(v)

        def addperp(a, b, c, alpha):
            return linear.add_difference(a, linear.perpendicular_component(linear.subtract(a, c), linear.subtract(b, c)), alpha=alpha)

        store("slerp", sd_mecha.merge(linear.slerp(ms[0], ms[1], alpha=0.3), **kw))
        store("slerp_zero", sd_mecha.merge(linear.slerp(z, ms[2], alpha=0.6), **kw))                # |a| = 0 on w.bias: NaN -> lerp
        store("slerp_nonfinite", sd_mecha.merge(linear.slerp(ms[0], ms[3], alpha=0.5) | ms[0], **kw))  # m3 has an inf: key from the fallback
        store("addperp", sd_mecha.merge(addperp(ms[0], ms[1], ms[2], 0.7), **kw))
        store("addperp_zero", sd_mecha.merge(addperp(z, ms[2], ms[1], 1.0), **kw))                   # a - c = 0 on w.bias: NaN -> zeros
        store("addsub", sd_mecha.merge(linear.add_difference(ms[0], linear.subtract(ms[1], ms[2]), alpha=0.6) | ms[0], **kw))
        store("adddiff", sd_mecha.merge(linear.add_difference(ms[0], sd_mecha.model(paths[1], merge_space="delta"), alpha=0.25), **kw))
        deltas = [linear.subtract(m, ms[0]) for m in ms[1:3]]
        for name, extra in (("dare_nostock", dict(apply_stock=True, seed=114514, probability=0.4, k=0.5)),
                            ("dare_nostock_norescale", dict(apply_stock=True, seed=7, probability=0.9, k=1.0, rescale=False)),
                            ("dare_default_seed", dict(apply_stock=True, probability=0.5, k=0.2))):
            store(name, sd_mecha.merge(linear.add_difference(ms[0], ties.ties_sum_with_dropout(*deltas, **extra), alpha=1.0), **kw))
        # apply_stock False (the wrapper's default): the Model-Stock factor is applied; 0-d keys raise in CosineSimilarity(dim=-1)
        three = {k: s for k, s in shapes.items() if len(s) >= 1}
        for m in range(3):
            st = {k: torch.from_numpy(g[f"e_m{m}_{k}"].view(np.int16).copy()).view(torch.bfloat16).reshape(s) for k, s in three.items()}
            safetensors.torch.save_file(st, str(td / f"s{m}.safetensors"))
        ss = [sd_mecha.model(td / f"s{m}.safetensors") for m in range(3)]
        sdeltas = [linear.subtract(m, ss[0]) for m in ss[1:]]
        store("dare_stock", sd_mecha.merge(linear.add_difference(ss[0], ties.ties_sum_with_dropout(*sdeltas, seed=3, probability=0.4, k=0.5), alpha=0.8), **kw))


def gen_e2e_mixed(out):
    """Executor semantics through the real sd_mecha.merge on checkpoints shaped like real ones: an int64
    position_ids key, an fp32 key and an EMPTY int64 key among bf16 keys (the reference casts every key to fp32 and the result
    to the output dtype, merging.py:157-158, :673-681); a lazily skipped non-finite input (merging.py:579-589); a node with two
    consumers (merge_context.py:164-205); a user method that reads the four extra kwargs (merge_methods.py:330-337); and a
    `|` default that holds keys its left side lacks, under a known config (logistics.py:15-19)."""
    import safetensors.torch
    from tqdm import tqdm
    from sd_mecha.extensions.merge_methods import merge_method, Parameter, Return
    from torch import Tensor

    class quiet(tqdm):
        def __init__(self, *a, **k):
            k["disable"] = True
            super().__init__(*a, **k)

    seen = {}

    @merge_method(identifier="golden_kwargs_sum")
    def golden_kwargs_sum(a: Parameter(Tensor), b: Parameter(Tensor), alpha: Parameter(Tensor) = 0.5, **kwargs) -> Return(Tensor):
        (key,), inputs = relation = kwargs["key_relation"]
        assert key == kwargs["key"] and set(inputs) == {"a", "b", "alpha"} and inputs["a"] == (key,)
        assert kwargs["cache"] is None and isinstance(kwargs["output_reused"], bool)
        seen[key] = (kwargs["output_reused"], relation.meta, tuple(alpha.shape), str(alpha.device), alpha.dtype)
        return (1 - alpha) * a + alpha * b

    rng = np.random.default_rng(77)
    shapes = {"w.linear": (96, 64), "w.conv": (16, 8, 3, 3), "w.bias": (96,)}
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        paths = []
        for m in range(3):
            state = {k: rand_half(rng, s, "bfloat16") for k, s in shapes.items()}
            state["w.f32"] = torch.from_numpy(rng.normal(0, 0.02, size=(40, 24)).astype(np.float32))
            state["w.position_ids"] = torch.arange(77, dtype=torch.int64).reshape(1, 77) + m
            state["w.empty"] = torch.zeros((0,), dtype=torch.int64)
            if m == 1:
                state["w.bias"][3] = float("nan")
            p = td / f"m{m}.safetensors"
            safetensors.torch.save_file(state, str(p))
            paths.append(p)
            for k, t in state.items():
                out[f"x_m{m}_{k}"] = bits(t) if t.is_floating_point() else t.numpy()
        out["x_keys"] = np.array(list(state.keys()))
        models = [sd_mecha.model(p) for p in paths]
        kw = dict(threads=0, output_dtype=torch.bfloat16, tqdm=quiet)

        def store(name, res):
            res = {k: v for k, v in res.items() if k != "__metadata__"}
            out[f"x_{name}_keys"] = np.array(sorted(res.keys()))
            for k, v in res.items():
                assert v.dtype == torch.bfloat16, (name, k, v.dtype)
                out[f"x_{name}_{k}"] = bits(v)
                out[f"x_{name}_{k}_shape"] = np.array(v.shape, dtype=np.int64)

        store("wsum", sd_mecha.merge(linear.weighted_sum(models[0], models[2], alpha=0.3), **kw))
        store("wsum_nan_b", sd_mecha.merge(linear.weighted_sum(models[0], models[1], alpha=0.3), **kw))     # w.bias skipped
        store("wsum_lazy0", sd_mecha.merge(linear.weighted_sum(models[0], models[1], alpha=0.0), **kw))     # b never loaded: w.bias kept
        store("wsum_lazy1", sd_mecha.merge(linear.weighted_sum(models[1], models[0], alpha=1.0), **kw))     # a never loaded
        deltas = [linear.subtract(m, models[0]) for m in models[1:]]
        store("ties_fb", sd_mecha.merge(linear.add_difference(models[0], ties.ties_sum(*deltas, k=0.5), alpha=1.0) | models[0], **kw))
        d = linear.subtract(models[2], models[0])                              # one delta, two consumers
        store("shared", sd_mecha.merge(linear.add_difference(linear.add_difference(models[0], d, alpha=0.5), d, alpha=0.25), **kw))
        store("kwargs", sd_mecha.merge(golden_kwargs_sum(models[0], models[2], alpha=0.25), **kw))
        out["x_kwargs_seen"] = np.array(json_dumps({k: [v[0], v[1], list(v[2]), v[3], str(v[4])] for k, v in seen.items()}))

        # a `|` default with more keys than its left side, under a known config
        kshapes = {"unet.out.2.weight": (4, 320, 3, 3), "unet.out.2.bias": (4,), "unet.input_blocks.0.0.bias": (320,),
                   "te1.text_model.embeddings.position_embedding.weight": (77, 768)}
        full = {k: rand_half(rng, s, "bfloat16") for k, s in kshapes.items()}
        part_a = {k: rand_half(rng, kshapes[k], "bfloat16") for k in list(kshapes)[:2]}
        part_b = {k: rand_half(rng, kshapes[k], "bfloat16") for k in list(kshapes)[:3]}
        for name, sdict in (("full", full), ("part_a", part_a), ("part_b", part_b)):
            safetensors.torch.save_file(sdict, str(td / f"{name}.safetensors"))
            out[f"u_{name}_keys"] = np.array(list(sdict.keys()))
            for k, t in sdict.items():
                out[f"u_{name}_{k}"] = bits(t)
        out["u_shapes"] = np.array(json_dumps({k: list(v) for k, v in kshapes.items()}))
        fm, pa, pb = (sd_mecha.model(td / f"{n}.safetensors", config="sdxl-kohya") for n in ("full", "part_a", "part_b"))
        res = {k: v for k, v in sd_mecha.merge(linear.weighted_sum(pa, pb, alpha=0.5) | fm, **kw).items() if k != "__metadata__"}
        out["u_union_keys"] = np.array(sorted(res.keys()))
        for k, v in res.items():
            out[f"u_union_{k}"] = bits(v)
        res = {k: v for k, v in sd_mecha.merge(linear.weighted_sum(pa, pb, alpha=0.5), fallback_model=fm, **kw).items() if k != "__metadata__"}
        out["u_fallback_model_keys"] = np.array(sorted(res.keys()))
        for k, v in res.items():
            out[f"u_fallback_model_{k}"] = bits(v)


def json_dumps(x):
    import json
    return json.dumps(x)


def gen_mecha_text(_out):
    """The `.mecha` text the reference embeds in its outputs (serialization.py:184-238 through merging.py:260, :287-288) for
    the benchmark recipes, as a JSON fixture {name: text}; the directory of the model files is replaced by <DIR>."""
    import json
    import safetensors.torch
    from tqdm import tqdm

    class quiet(tqdm):
        def __init__(self, *a, **k):
            k["disable"] = True
            super().__init__(*a, **k)

    g = torch.Generator().manual_seed(5)
    texts = {}
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        shapes = {"w.linear": (12, 8), "w.bias": (12,)}
        for m in range(3):
            safetensors.torch.save_file({k: torch.randn(s, generator=g).to(torch.bfloat16) for k, s in shapes.items()}, str(td / f"m{m}.safetensors"))
        kshapes = {"unet.out.2.weight": (4, 320, 3, 3), "unet.out.2.bias": (4,)}
        for name in ("ka", "kb"):
            safetensors.torch.save_file({k: torch.randn(s, generator=g).to(torch.bfloat16) for k, s in kshapes.items()}, str(td / f"{name}.safetensors"))
        safetensors.torch.save_file({"lora_unet_out_2.lora_up.weight": (torch.randn(4, 2, generator=g) * 0.05).half(),
                                     "lora_unet_out_2.lora_down.weight": (torch.randn(2, 320, 3, 3, generator=g) * 0.05).half(),
                                     "lora_unet_out_2.alpha": torch.tensor(1.0).half()}, str(td / "lora0.safetensors"))
        ms = [sd_mecha.model(td / f"m{m}.safetensors") for m in range(3)]
        ka, kb = (sd_mecha.model(td / f"{n}.safetensors", config="sdxl-kohya") for n in ("ka", "kb"))
        lora_delta = sd_mecha.convert(sd_mecha.model(td / "lora0.safetensors"), ka)     # convert_'sdxl-kohya_kohya_lora'_to_base (lycoris.py:52)
        kw = dict(threads=0, output_dtype=torch.bfloat16, tqdm=quiet)
        d = [linear.subtract(m, ms[0]) for m in ms[1:]]
        recipes = {
            "wsum": (linear.weighted_sum(ms[0], ms[1], alpha=0.5), {}),
            "wsum_default": (linear.weighted_sum(ms[0], ms[1]), {}),
            "navg": (linear.n_average(*ms), {}),
            "ties": (linear.add_difference(ms[0], ties.ties_sum(*d, k=0.2), alpha=1.0) | ms[0], {}),
            "ties_vote_fallback_model": (linear.add_difference(ms[0], ties.ties_sum(*d, k=0.2, vote_sgn=True), alpha=0.7), {"fallback_model": ms[0]}),
            "rotate": (svd.rotate(ms[0], ms[1], alignment=1.0, alpha=0.0), {}),
            "operators": (ms[0] + (ms[1] - ms[0]) * 0.5, {}),
            "slerp": (linear.slerp(ms[0], ms[1], alpha=0.3), {}),
            "known_config": (linear.weighted_sum(ka, kb, alpha=0.25), {}),
            "lora": (linear.add_difference(ka, lora_delta, alpha=0.75) | ka, {}),
        }
        for name, (r, extra) in recipes.items():
            res = sd_mecha.merge(r, **kw, **extra)
            texts[name] = res["__metadata__"]["mecha_recipe"].replace(str(td), "<DIR>")
    (HERE / "mecha_text.json").write_text(json.dumps(texts, indent=1))
    print("mecha_text", len(texts), "recipes")


def gen_e2e_alias(out):
    """Row (f)2 through the real sd_mecha.merge: two FLUX-like files with tiny tensors under REAL key names -- one under the
    config's canonical names, one under the alias names a diffuser-only file uses, both with a key the config does not know --:
    the config is detected from the names (graph_finalization.py:1024-1058), the alias-named tensors are read
    (merging.py:394-403) and the output carries canonical names."""
    import safetensors.torch
    from tqdm import tqdm
    from sd_mecha.extensions import model_configs

    class quiet(tqdm):
        def __init__(self, *a, **k):
            k["disable"] = True
            super().__init__(*a, **k)

    cfg = model_configs.resolve("flux-flux")
    canon = list(cfg.components()["diffuser"].keys())[:60]
    aliases = [cfg.keys()[k].aliases[0] for k in canon]
    rng = np.random.default_rng(33)
    a = {k: rand_half(rng, (8, 16), "bfloat16") for k in canon}
    b = {al: rand_half(rng, (8, 16), "bfloat16") for al in aliases}
    a["junk.weight"] = torch.ones(3, dtype=torch.bfloat16)
    b["junk.weight"] = torch.zeros(3, dtype=torch.bfloat16)
    out["canon"], out["aliases"] = np.array(canon), np.array(aliases)
    out["a"] = np.stack([bits(a[k]) for k in canon])
    out["b"] = np.stack([bits(b[al]) for al in aliases])
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        safetensors.torch.save_file(a, str(td / "a.safetensors"))
        safetensors.torch.save_file(b, str(td / "b.safetensors"))
        res = sd_mecha.merge(linear.weighted_sum(sd_mecha.model(td / "a.safetensors"), sd_mecha.model(td / "b.safetensors"), alpha=0.3),
                             threads=0, output_dtype=torch.bfloat16, tqdm=quiet, check_extra_keys=False)
    text = res.pop("__metadata__")["mecha_recipe"]
    out["config_line"] = np.array([ln for ln in text.splitlines() if ln.startswith("model ")][0].split("model_config=")[1].split(" ")[0])
    out["out_keys"] = np.array(sorted(res))
    out["out"] = np.stack([bits(res[k]) for k in canon])


def gen_e2e_lora(out):
    """sd_mecha.convert(model(lora_file), base) + add_difference through sd_mecha.merge (lycoris.py:52-85) on a
    toy checkpoint with REAL sdxl-kohya key names and shapes (LycorisToBase takes the target shape from the config).
    The wrapper sd_mecha.add_difference raises on merge-node inputs in this revision, so the recipes are
    built from the registered primitive and explicit `|` fallbacks."""
    import safetensors.torch
    from tqdm import tqdm

    class quiet(tqdm):
        def __init__(self, *a, **k):
            k["disable"] = True
            super().__init__(*a, **k)

    g = torch.Generator().manual_seed(31)
    shapes = {"unet.input_blocks.0.0.weight": (320, 4, 3, 3), "unet.out.2.weight": (4, 320, 3, 3),
              "te1.text_model.embeddings.position_embedding.weight": (77, 768), "unet.out.2.bias": (4,), "unet.input_blocks.0.0.bias": (320,)}
    base = {k: (torch.randn(s, generator=g) * 0.02).to(torch.bfloat16) for k, s in shapes.items()}

    def lora_for(keys, rank, with_alpha):
        sd = {}
        for k in keys:
            s = shapes[k]
            stem = "lora_" + "_".join(k.split(".")[:-1])
            sd[stem + ".lora_up.weight"] = (torch.randn(s[0], rank, generator=g) * 0.05).half()
            sd[stem + ".lora_down.weight"] = (torch.randn((rank,) + tuple(s[1:]), generator=g) * 0.05).half()
            if with_alpha:
                sd[stem + ".alpha"] = torch.tensor(rank / 2).half()
        return sd

    two_d = [k for k, s in shapes.items() if len(s) >= 2]
    loras = [lora_for(two_d, 8, True), lora_for(two_d[:2], 4, False)]        # the second file has no entry for the te1 key
    with tempfile.TemporaryDirectory() as td:
        td = pathlib.Path(td)
        safetensors.torch.save_file(base, str(td / "base.safetensors"))
        for i, sd in enumerate(loras):
            safetensors.torch.save_file(sd, str(td / f"l{i}.safetensors"))
        for k, t in base.items():
            out[f"base_{k}"] = bits(t)
        for i, sd in enumerate(loras):
            out[f"l{i}_keys"] = np.array(sorted(sd.keys()))
            for k, t in sd.items():
                out[f"l{i}_{k}"] = bits(t)
        b = sd_mecha.model(td / "base.safetensors", config="sdxl-kohya")
        d = [sd_mecha.convert(sd_mecha.model(td / f"l{i}.safetensors"), b) for i in range(2)]
        kw = dict(threads=0, output_dtype=torch.bfloat16, tqdm=quiet)
        x1 = linear.add_difference(b, d[0], alpha=0.75) | b
        recipes = {
            "chain": linear.add_difference(x1, d[1], alpha=1.0) | x1,            # a missing network keeps the previous value
            "outer_base": linear.add_difference(x1, d[1], alpha=1.0) | b,       # ... or falls all the way back to the base
            "strict": linear.add_difference(linear.add_difference(b, d[0], alpha=0.75), d[1], alpha=0.5),   # no fallback: keys are omitted
        }
        for name, r in recipes.items():
            res = {k: v for k, v in sd_mecha.merge(r, **kw).items() if k != "__metadata__"}
            out[f"{name}_keys"] = np.array(sorted(res.keys()))
            for k, v in res.items():
                out[f"{name}_{k}"] = bits(v)
    out["shapes_keys"] = np.array(list(shapes.keys()))


def gen_configs(out):
    """Config auto-detection (graph_finalization.py:1024-1058): for several key sets, the reference's candidate statistics
    of the four configs this repo ships and the config it would pick among them."""
    from sd_mecha import graph_finalization as gf
    from sd_mecha.extensions import model_configs
    shipped = ["sd1-ldm", "sdxl-sgm", "sdxl-kohya", "flux-flux"]
    rng = np.random.default_rng(11)

    def keys_of(ident, components=None, alias=False):
        cfg = model_configs.resolve(ident)
        ks = []
        for cname, comp in cfg.components().items():
            if components is not None and cname not in components:
                continue
            for k, m in comp.keys().items():
                ks.append(m.aliases[0] if alias and m.aliases else k)
        return ks

    cases = {
        "sdxl_full": keys_of("sdxl-sgm"),
        "sdxl_kohya_full": keys_of("sdxl-kohya"),
        "sd1_no_ema": keys_of("sd1-ldm", ("clip_l", "vae", "diffuser", "scheduler")),
        "sd1_aliases": keys_of("sd1-ldm", alias=True),
        "flux_diffuser_aliases": keys_of("flux-flux", ("diffuser",), alias=True),
        "flux_full": keys_of("flux-flux"),
        "sdxl_plus_junk": keys_of("sdxl-sgm") + [f"junk.{i}" for i in range(400)],
        "sdxl_vae_only": keys_of("sdxl-sgm", ("vae",)),
        "unknown": [f"layer.{i}.weight" for i in range(50)],
        "sdxl_tenth": [k for k in keys_of("sdxl-sgm") if rng.random() < 0.1] + [f"junk.{i}" for i in range(100)],
    }
    out["case_names"] = np.array(list(cases))
    out["shipped"] = np.array(shipped)
    for name, ks in cases.items():
        out[f"{name}_keys"] = np.array(ks)
        stats = gf.generate_configs_stats(ks)
        cand = [c for c in shipped if c in stats]
        out[f"{name}_candidates"] = np.array(cand, dtype="U32")
        out[f"{name}_stats"] = np.array([[len(stats[c].intersection), stats[c].state_dict_misses, stats[c].config_misses] for c in cand], dtype=np.float64).reshape(len(cand), 3)
        best = min(((c, stats[c]) for c in cand), key=lambda kv: kv[1].state_dict_misses)[0] if cand else ""
        out[f"{name}_best"] = np.array(best)


def main():
    torch.set_num_threads(1)      # the goldens must not depend on torch's thread-chunked reduction tails
    only = set(sys.argv[1:])
    for name, fn in [("streaming", gen_streaming), ("alpha_tensor", gen_alpha_tensor), ("ties", gen_ties), ("dare", gen_dare), ("stock_dropout", gen_stock_dropout),
                     ("rotate", gen_rotate), ("truncate_rank", gen_truncate_rank), ("lora", gen_lora), ("e2e", gen_e2e), ("e2e_plans", gen_e2e_plans), ("e2e_mixed", gen_e2e_mixed), ("e2e_alias", gen_e2e_alias), ("e2e_lora", gen_e2e_lora), ("mecha_text", gen_mecha_text),
                     ("configs", gen_configs)]:
        if only and name not in only:
            continue
        out = {}
        fn(out)
        if not out:
            continue
        np.savez_compressed(HERE / f"{name}.npz", **out)
        print(name, len(out), "arrays", (HERE / f"{name}.npz").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
