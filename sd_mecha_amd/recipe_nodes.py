"""Minimal recipe graph hosting the key scheduler.

The reference's recipe layer (recipe_nodes.py, merge_methods.py, serialization.py) "stays as is" per the north
star; this is the build-owned counterpart the GPU box needs: three node kinds + a visitor-free evaluator, the
``@merge_method`` wrapper (identifier registry, ``__wrapped__``), the operator sugar of recipe_nodes.py:54-80
and the ``.mecha`` text of serialization.py:184-238 for ``__metadata__.mecha_recipe``.
"""
from __future__ import annotations

import inspect
import pathlib
from typing import Any, Callable, Dict, List, Optional, Tuple


class RecipeNode:
    def __add__(self, other):
        return _registry["add_difference"](self, other)

    def __sub__(self, other):
        return _registry["subtract"](self, other)

    def __mul__(self, other):
        return _registry["scale"](self, other)

    def __rmul__(self, other):
        return _registry["scale"](self, other)

    def __or__(self, other):
        return _registry["fallback"](self, other)

    def __hash__(self):
        return id(self)

    def __eq__(self, other):
        return self is other


class ModelRecipeNode(RecipeNode):
    """A checkpoint on disk (helpers.py:15-43 ``sd_mecha.model(path)``)."""

    def __init__(self, path, config=None):
        self.path = pathlib.Path(path)
        self.config = config                  # model config identifier, or None: inferred from the key names at merge time

    def __repr__(self):
        return f"model({str(self.path)!r})"


class LiteralRecipeNode(RecipeNode):
    """A python scalar / str / bool / None, or an in-memory dict of tensors."""

    def __init__(self, value):
        self.value = value

    def __repr__(self):
        return f"literal({self.value!r})"


class MergeRecipeNode(RecipeNode):
    def __init__(self, merge_method: "MergeMethod", args: Tuple[RecipeNode, ...], kwargs: Dict[str, RecipeNode]):
        self.merge_method = merge_method
        self.args = args
        self.kwargs = kwargs

    def __repr__(self):
        return f"{self.merge_method.identifier}(...)"


def value_to_node(v) -> RecipeNode:
    """merge_methods.py:856-934, reduced: nodes pass through, paths become models, the rest literals."""
    if isinstance(v, RecipeNode):
        return v
    if isinstance(v, pathlib.Path) or (isinstance(v, str) and v.endswith(".safetensors")):
        return ModelRecipeNode(v)
    return LiteralRecipeNode(v)


def model(path, config=None) -> ModelRecipeNode:
    """helpers.py:15-43: ``config`` is a model config identifier ("sdxl-sgm", ...) or None to infer it."""
    return ModelRecipeNode(path, config)


def literal(value) -> LiteralRecipeNode:
    return LiteralRecipeNode(value)


class MergeMethod:
    """merge_methods.py:182-445, reduced to what the hot path needs: an identifier, the raw body
    (``__wrapped__``), and ``__call__`` that builds a recipe node when given nodes and runs the body when
    given tensors."""

    def __init__(self, fn: Callable, identifier: Optional[str] = None, cache_factory: Optional[Callable] = None):
        self.__wrapped__ = fn
        self.cache_factory = cache_factory
        self.identifier = identifier or fn.__name__
        self.signature = inspect.signature(fn)
        self.accepts_kwargs = any(p.kind == inspect.Parameter.VAR_KEYWORD for p in self.signature.parameters.values())
        self.__name__ = self.identifier
        self.__doc__ = fn.__doc__

    def get_identifier(self) -> str:
        return self.identifier

    def __call__(self, *args, **kwargs):
        if any(isinstance(a, RecipeNode) for a in args) or any(isinstance(a, RecipeNode) for a in kwargs.values()):
            return MergeRecipeNode(self, tuple(value_to_node(a) for a in args), {k: value_to_node(v) for k, v in kwargs.items()})
        return self.__wrapped__(*args, **kwargs)

    def create_cache(self):
        """merge_methods.py:271: a fresh cache object for one node of this method, or None if the method keeps none."""
        return self.cache_factory() if self.cache_factory is not None else None

    def merge_key(self, args, kwargs, key: str, cache=None):
        """merge_methods.py:304-337: extra kwargs only if the body declares **kwargs or names them."""
        extra = {}
        if self.accepts_kwargs or "key" in self.signature.parameters:
            extra["key"] = key
        if self.accepts_kwargs or "cache" in self.signature.parameters:
            extra["cache"] = cache
        return self.__wrapped__(*args, **kwargs, **extra)


_registry: Dict[str, MergeMethod] = {}


def merge_method(fn: Optional[Callable] = None, *, identifier: Optional[str] = None, register: bool = True,
                 cache_factory: Optional[Callable] = None):
    """``@merge_method`` (merge_methods.py:665-711): wrap a per-key function and register it by identifier.
    ``cache_factory`` as in the reference (svd.py:10: ``lambda: defaultdict(dict)``)."""
    def wrap(f):
        mm = MergeMethod(f, identifier, cache_factory)
        if register:
            if mm.identifier in _registry:
                raise ValueError(f"merge method {mm.identifier!r} is already registered")
            _registry[mm.identifier] = mm
        return mm
    return wrap(fn) if fn is not None else wrap


def resolve(identifier: str) -> MergeMethod:
    """merge_methods.py:795-800."""
    try:
        return _registry[identifier]
    except KeyError:
        raise ValueError(f"unknown merge method: {identifier}") from None


def get_all() -> List[MergeMethod]:
    return list(_registry.values())


# ------------------------------------------------------------------ .mecha text (serialization.py:143-238)
MECHA_FORMAT_VERSION = "0.1.0"


def serialize(recipe: RecipeNode) -> str:
    instructions: List[str] = []

    def add(line: str) -> str:
        if line in instructions:
            return f"&{instructions.index(line)}"
        instructions.append(line)
        return f"&{len(instructions) - 1}"

    def value(v) -> str:
        if isinstance(v, str):
            return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
        if isinstance(v, bool):
            return "true" if v else "false"
        if v is None:
            return "null"
        if isinstance(v, (int, float)):
            return str(v)
        if isinstance(v, dict):
            return add("dict " + " ".join(f"{k}={value(x)}" for k, x in v.items()))
        if isinstance(v, RecipeNode):
            return node(v)
        raise TypeError(f"type {type(v)} cannot be serialized: {v}")

    done: Dict[int, str] = {}                   # recipes are DAGs (`x | x`-style sharing): each node is visited once

    def node(n: RecipeNode) -> str:
        if isinstance(n, ModelRecipeNode):
            return add(f"model {value(str(n.path))} model_config=null merge_space=\"weight\"")
        if isinstance(n, LiteralRecipeNode):
            return value(n.value)
        if isinstance(n, MergeRecipeNode):
            if id(n) not in done:
                parts = ["merge", value(n.merge_method.identifier)] + [value(a) for a in n.args] + [f"{k}={value(v)}" for k, v in n.kwargs.items()]
                done[id(n)] = add(" ".join(parts))
            return done[id(n)]
        raise TypeError(type(n))

    node(recipe)
    return "\n".join([f"version {MECHA_FORMAT_VERSION}"] + instructions)
