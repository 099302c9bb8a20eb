"""Minimal recipe graph hosting the key scheduler.

The reference's recipe layer (recipe_nodes.py, merge_methods.py, serialization.py) "stays as is" per the north
star; this is the build-owned counterpart the GPU box needs: three node kinds + a visitor-free evaluator, the
``@merge_method`` wrapper (identifier registry, ``__wrapped__``), the operator sugar of recipe_nodes.py:54-80
and the ``.mecha`` text of serialization.py:184-238 for ``__metadata__.mecha_recipe``.
"""
from __future__ import annotations

import inspect
import pathlib
import re
from typing import Any, Callable, Dict, List, Mapping, Optional, Sequence, Tuple


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

    def __init__(self, path, config=None, merge_space="weight"):
        self.path = pathlib.Path(path)
        self.config = config                  # model config identifier, or None: inferred from the key names at merge time
        self.merge_space = merge_space        # "weight" | "delta" (helpers.py:15-43): recorded in the recipe text only

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


def model(path, config=None, merge_space="weight") -> ModelRecipeNode:
    """helpers.py:15-43: ``config`` is a model config identifier ("sdxl-sgm", ...) or None to infer it; ``merge_space`` says
    whether the file holds weights or deltas (a delta checkpoint is what add_difference takes as `b`)."""
    return ModelRecipeNode(path, config, merge_space)


def literal(value) -> LiteralRecipeNode:
    return LiteralRecipeNode(value)


# ------------------------------------------------------------------ parameter annotations (merge_methods.py:41-144)
class StateDict(Mapping):
    """``StateDict[T]`` (merge_methods.py:25-38): the annotation of a parameter the executor hands over LAZILY -- a mapping
    that evaluates its recipe node for a key when the body indexes it (merging.py:566-608), so a body that returns
    ``a[key]`` never loads ``b``."""

    def __class_getitem__(cls, item):
        return _StateDictAlias(item)


class _StateDictAlias:
    def __init__(self, item):
        self.item = item

    def __repr__(self):
        return f"StateDict[{getattr(self.item, '__name__', self.item)}]"


class ParameterType:
    interface: Any = None
    merge_space: Any = None
    model_config: Any = None


class ReturnType(ParameterType):
    pass


def Parameter(interface, merge_space=None, model_config=None):
    """``Parameter(interface, merge_space, model_config)`` (merge_methods.py:53-101).  Only ``interface`` changes what the
    executor passes: ``StateDict[...]`` = lazy mapping, ``Tensor`` = tensor (python scalars become 0-d fp32 CPU tensors,
    merging.py:632-633), ``float`` / ``int`` / ``bool`` / ``str`` = python values (merging.py:626-631).  Merge spaces and
    configs belong to the graph layer (out of scope) and are kept only as attributes."""
    return type("Parameter", (ParameterType,), {"interface": interface, "merge_space": merge_space, "model_config": model_config})


def Return(interface, merge_space=None, model_config=None):
    """merge_methods.py:109-144."""
    return type("Return", (ReturnType,), {"interface": interface, "merge_space": merge_space, "model_config": model_config})


class RealizedKeyRelation:
    """keys_map.py:18-26: ``outputs`` (the keys one call produces), ``inputs`` (per parameter, the keys it may read),
    ``meta``.  Iterates as (outputs, inputs) like the reference, so ``(key,), _ = kwargs["key_relation"]`` works."""
    __slots__ = ("outputs", "inputs", "meta")

    def __init__(self, outputs, inputs, meta=None):
        self.outputs, self.inputs, self.meta = tuple(outputs), inputs, meta

    def __iter__(self):
        yield self.outputs
        yield self.inputs

    def __repr__(self):
        return f"RealizedKeyRelation(outputs={self.outputs}, inputs={dict(self.inputs)}, meta={self.meta})"


_IMPLEMENTATION_KEYWORDS = frozenset({"key", "cache", "out_dtype", "flags", "check_finite", "return_thresholds", "return_sweeps", "keep_bits"})


def _interface_of(annotation):
    if isinstance(annotation, type) and issubclass(annotation, ParameterType):
        return annotation.interface
    return annotation


class MergeMethod:
    """merge_methods.py:182-445, reduced to what the hot path needs: an identifier, the raw body
    (``__wrapped__``: a function, or a class with ``__call__`` that is instantiated once per recipe node,
    merge_context.py:94), ``__call__`` that builds a recipe node when given nodes and runs the body when given tensors,
    and ``merge_key``."""

    def __init__(self, fn: Callable, identifier: Optional[str] = None, cache_factory: Optional[Callable] = None,
                 lazy: Sequence[str] = (), reuse_outputs: bool = True):
        self.__wrapped__ = fn
        self.wrapped_is_class = inspect.isclass(fn)
        self.cache_factory = cache_factory
        self.reuse_outputs = reuse_outputs
        self.identifier = identifier or fn.__name__
        sig = inspect.signature(fn.__call__ if self.wrapped_is_class else fn)
        if self.wrapped_is_class and not isinstance(inspect.getattr_static(fn, "__call__"), (classmethod, staticmethod)):
            sig = sig.replace(parameters=list(sig.parameters.values())[1:])          # drop self (merge_methods.py:214-217)
        self.signature = sig
        self.accepts_kwargs = any(p.kind == inspect.Parameter.VAR_KEYWORD for p in sig.parameters.values())
        # The signature recipes are built and serialized with = the reference method's parameters.  This package's own
        # shims also carry implementation keywords (output dtype, flag words, ...) that are no part of a recipe.
        annotated = any(isinstance(p.annotation, type) and issubclass(p.annotation, ParameterType) for p in sig.parameters.values())
        self.public_signature = sig.replace(parameters=[
            p for p in sig.parameters.values()
            if p.kind != p.VAR_KEYWORD and (annotated or not (p.kind == p.KEYWORD_ONLY and p.name in _IMPLEMENTATION_KEYWORDS))])
        for reserved in ("key", "key_relation", "cache", "output_reused"):             # merge_methods.py:262-264
            p = sig.parameters.get(reserved)
            if p is not None and p.annotation is not p.empty and _interface_of(p.annotation) is not p.annotation:
                raise TypeError(f"A merge method cannot define a parameter named {reserved}.")
        # how each parameter is handed over (merging.py:469-477)
        self.param_interfaces: Dict[str, Any] = {}
        self.lazy_params = set(lazy)
        for p in sig.parameters.values():
            if p.kind == p.VAR_KEYWORD:
                continue
            iface = _interface_of(p.annotation) if p.annotation is not p.empty else None
            self.param_interfaces[p.name] = iface
            if isinstance(iface, _StateDictAlias) or iface is StateDict:
                self.lazy_params.add(p.name)
        self.__name__ = self.identifier
        self.__doc__ = fn.__doc__

    def get_identifier(self) -> str:
        return self.identifier

    def __call__(self, *args, **kwargs):
        if any(isinstance(a, RecipeNode) for a in args) or any(isinstance(a, RecipeNode) for a in kwargs.values()):
            # merge_methods.py:339-358: bind to the signature and apply the defaults, so that a recipe (and its .mecha text)
            # names every hyper-parameter; positional-or-keyword parameters travel as args, keyword-only ones as kwargs
            bound = self.public_signature.bind(*args, **kwargs)
            bound.apply_defaults()
            return MergeRecipeNode(self, tuple(value_to_node(a) for a in bound.args), {k: value_to_node(v) for k, v in bound.kwargs.items()})
        if self.wrapped_is_class:
            return self.__wrapped__()(*args, **kwargs)
        return self.__wrapped__(*args, **kwargs)

    def create_cache(self):
        """merge_methods.py:271: a fresh cache object for one node of this method, or None if the method keeps none."""
        return self.cache_factory() if self.cache_factory is not None else None

    def instantiate(self):
        """merge_methods.py:266-269."""
        return self.__wrapped__() if self.wrapped_is_class else None

    def param_names(self, n_args: int) -> List[str]:
        """Parameter name of each positional argument (a ``*varargs`` parameter repeats)."""
        names, var = [], None
        for p in self.public_signature.parameters.values():
            if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD):
                names.append(p.name)
            elif p.kind == p.VAR_POSITIONAL:
                var = p.name
        if len(names) < n_args:
            if var is None:
                raise TypeError(f"{self.identifier}: takes {len(names)} positional arguments but {n_args} were given")
            names += [var] * (n_args - len(names))
        return names[:n_args]

    def merge_key(self, args, kwargs, key: str, key_relation=None, cache=None, context=None, output_reused: bool = True,
                  body: Optional[Callable] = None):
        """merge_methods.py:304-337: the body is called as fn(*args, **kwargs); a body that declares ``**kwargs`` also
        receives ``key``, ``key_relation``, ``cache`` and ``output_reused``.  (Extension kept from round 1: a body without
        ``**kwargs`` that NAMES ``key`` or ``cache`` as a keyword parameter receives that one.)"""
        kwargs = dict(kwargs)
        if self.accepts_kwargs:
            kwargs.update(key=key, key_relation=key_relation, cache=cache, output_reused=output_reused)
        else:
            if "key" in self.signature.parameters:
                kwargs["key"] = key
            if "cache" in self.signature.parameters:
                kwargs["cache"] = cache
        if body is not None:                         # a backend's own body for this identifier (tests: the CPU double)
            return body(*args, **kwargs)
        if self.wrapped_is_class:
            if context is None:
                raise AssertionError(f"class merge method {self.identifier} received self=None")
            return context(*args, **kwargs)
        return self.__wrapped__(*args, **kwargs)


_registry: Dict[str, MergeMethod] = {}


def merge_method(fn: Optional[Callable] = None, *, identifier: Optional[str] = None, register: bool = True,
                 cache_factory: Optional[Callable] = None, lazy: Sequence[str] = (), reuse_outputs: bool = True):
    """``@merge_method`` (merge_methods.py:665-711): wrap a per-key function (or a class with ``__call__``) and register it
    by identifier.  ``cache_factory`` as in the reference (svd.py:10: ``lambda: defaultdict(dict)``).  ``lazy``: names of
    parameters to hand over as lazy ``StateDict`` mappings when the body carries no ``Parameter(StateDict[...])``
    annotations (this package's own shims)."""
    def wrap(f):
        mm = MergeMethod(f, identifier, cache_factory, lazy, reuse_outputs)
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


# ------------------------------------------------------------------ .mecha text (serialization.py:26-238)
MECHA_FORMAT_VERSION = "0.1.0"
_CONSTANTS = {"null": None, "true": True, "false": False}


def serialize(recipe: RecipeNode, model_configs: Optional[Mapping[int, str]] = None) -> str:
    """The reference's `.mecha` text (serialization.py:143-238, SerializerVisitor :184-238), line for line: python scalars as
    the three lines the reference's graph gives them (``dict key=V`` / ``literal &d model_config="singleton-mecha"
    merge_space="param"`` / ``merge "convert_singleton" &l``, merge_methods.py:905-919), models with the config the merge
    solved for them (``model_configs``: {id(node): identifier}; default: the node's own config or ``"structural"``), merge
    lines with positional and keyword arguments as bound by the method's signature, defaults included.  Equal lines are
    emitted once and referenced by ``&n`` (:233-238)."""
    instructions: List[str] = []

    def add(line: str) -> str:
        try:
            return f"&{instructions.index(line)}"
        except ValueError:
            instructions.append(line)
            return f"&{len(instructions) - 1}"

    def value(v) -> str:
        if isinstance(v, str):
            return '"' + v.replace("\\", "\\\\").replace('"', '\\"') + '"'
        if isinstance(v, Mapping):
            return add("dict " + " ".join(f"{k}={value(x)}" for k, x in v.items()))
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            return str(v)
        if v is None or isinstance(v, bool):
            return {None: "null", True: "true", False: "false"}[v]
        if isinstance(v, RecipeNode):
            return node(v)
        raise TypeError(f"type {type(v)} cannot be serialized: {v}")

    done: Dict[int, str] = {}                   # recipes are DAGs: each node is visited once
    aux_configs: Dict[int, str] = {}            # LoRA files: the config their conversion method is registered for (lycoris.py:115-117)

    def node(n: RecipeNode) -> str:
        if id(n) in done:
            return done[id(n)]
        if isinstance(n, ModelRecipeNode):
            config = (model_configs or {}).get(id(n)) or n.config or aux_configs.get(id(n)) or "structural"
            ref = add(f"model {value(str(n.path))} model_config={value(config)} merge_space={value(n.merge_space)}")
        elif isinstance(n, LiteralRecipeNode):
            if isinstance(n.value, Mapping):
                ref = value(n.value)
            else:                               # a scalar hyper-parameter: singleton dict -> literal -> convert_singleton
                lit = add(f"literal {value({'key': n.value})} model_config=\"singleton-mecha\" merge_space=\"param\"")
                ref = add(f"merge \"convert_singleton\" {lit}")
        elif isinstance(n, MergeRecipeNode):
            if getattr(n.merge_method, "lora_config", None) and n.args:
                aux_configs[id(n.args[0])] = n.merge_method.lora_config
            parts = ["merge", value(n.merge_method.identifier)] + [value(a) for a in n.args] + [f"{k}={value(v)}" for k, v in n.kwargs.items()]
            ref = add(" ".join(parts))
        else:
            raise TypeError(type(n))
        done[id(n)] = ref
        return ref

    node(recipe)
    return "\n".join([f"version {MECHA_FORMAT_VERSION}"] + instructions)


# The `.mecha` line grammar (format 0.1.0), as this parser reads it:
#     line   := command { blank arg }            command := dict | literal | model | merge
#     arg    := [ name "=" ] value               name    := any run of characters without blank, quote or "="
#     value  := string | "&" index | null | true | false | number
#     string := '"' { "\" any-char | any char but '"' and "\" } '"'          (a backslash keeps the next character as it is)
#     number := a python float literal if it holds "." or an exponent, else a python int literal
# Blank lines and lines that start with "#" are skipped; "&n" names the result of instruction n (0-based, earlier lines only).
_ARG = re.compile(r"""
    \s*
    (?: (?P<name> [^\s"=]+ ) = (?= \S ) )?          # optional  name=
    (?: " (?P<string> (?: \\. | [^"\\] )* ) "       # a quoted string ...
      | (?P<bare> [^\s"]+ ) )                       # ... or a bare word: reference, constant or number
    (?= \s | $ )
""", re.VERBOSE | re.DOTALL)
_UNESCAPE = re.compile(r"\\(.)", re.DOTALL)
_NUMBER_AS_FLOAT = re.compile(r"[.eE]")


def _scan(line: str):
    """One instruction line -> (command, [(name or None, kind, text)]) with kind in {"string", "bare"}."""
    pos, out = 0, []
    end = len(line.rstrip())
    while pos < end:
        m = _ARG.match(line, pos)
        if m is None:
            raise ValueError("mismatched quotes in input" if '"' in line[pos:] else f"cannot read {line[pos:].strip()!r}")
        if m.group("string") is not None:
            out.append((m.group("name"), "string", _UNESCAPE.sub(r"\1", m.group("string"))))
        else:
            out.append((m.group("name"), "bare", m.group("bare")))
        pos = m.end()
    if not out or out[0][0] is not None or out[0][1] != "bare":
        raise ValueError("expected a command")
    return out[0][2], out[1:]


def _cmd_dict(positional, named):
    return dict(*positional, **named)


def _cmd_literal(positional, named):
    return LiteralRecipeNode(positional[0])


def _cmd_model(positional, named):
    config = named.get("model_config")
    return ModelRecipeNode(pathlib.Path(positional[0]), None if config in (None, "structural") else config, named.get("merge_space", "weight"))


def _cmd_merge(positional, named):
    method, *rest = positional
    if method == "convert_singleton":           # the three-line spelling of a python scalar (serialize() above): back to the scalar
        lit = rest[0]
        inner = lit.value if isinstance(lit, LiteralRecipeNode) else lit
        return LiteralRecipeNode(inner["key"]) if isinstance(inner, Mapping) and "key" in inner else value_to_node(inner)
    return resolve(method)(*rest, **named)


_COMMANDS = {"dict": _cmd_dict, "literal": _cmd_literal, "model": _cmd_model, "merge": _cmd_merge}


def deserialize(recipe) -> RecipeNode:
    """Recreate a recipe graph from `.mecha` text (the format serialization.py:26-140 reads; grammar above).  Methods are
    looked up by identifier in this package's registry; ``convert_singleton`` of a literal ``{"key": v}`` is the scalar ``v``."""
    lines = recipe.split("\n") if not isinstance(recipe, list) else recipe
    if not lines or not lines[0].startswith("version"):
        raise RuntimeError("bad format: expected version at line 1")
    if lines[0] != f"version {MECHA_FORMAT_VERSION}":
        raise RuntimeError(f"bad recipe version: got {lines[0]}, expected version {MECHA_FORMAT_VERSION}")
    results: List[Any] = []

    def value(kind: str, text: str, index: int):
        if kind == "string":
            return text
        if text in _CONSTANTS:
            return _CONSTANTS[text]
        try:
            if text[0] == "&":
                ref = int(text[1:])
                if not 0 <= ref < len(results):
                    raise ValueError(f"reference {text} out of bounds")
                return results[ref]
            return float(text) if _NUMBER_AS_FLOAT.search(text) else int(text)
        except ValueError as e:
            raise ValueError(f"argument {index}: {e}") from None

    for line_num, line in enumerate(lines[1:], 2):
        line = line.strip()
        if not line or line[0] == "#":
            continue
        try:
            command, args = _scan(line)
            handler = _COMMANDS.get(command)
            if handler is None:
                raise ValueError(f"unknown command: {command}")
            positional = [value(kind, text, i) for i, (name, kind, text) in enumerate(args) if name is None]
            named = {name: value(kind, text, i) for i, (name, kind, text) in enumerate(args) if name is not None}
            results.append(handler(positional, named))
        except ValueError as e:
            raise ValueError(f"line {line_num}: {e}.\n    {line}") from None
    return results[-1]
