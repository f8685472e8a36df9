"""
Minimal pipeline layer: enough of ``lenskit.pipeline`` for ``pipelines/als-implicit.toml``
and ``pipelines/iknn-explicit.toml`` to load and run UNCHANGED.

Mirrors ``Component`` config handling (src/lenskit/pipeline/components.py:65-199), the
``std:topn`` / ``std:topn-predict`` bases (src/lenskit/pipeline/_common.py:176-251),
``PipelineBuilder.from_config`` (``_builder.py:673-806``: ``[options] base``,
``[components.<name>] class | code, config``), ``Pipeline.train`` with one spawned
``SeedSequence`` child per trainable component (``_impl.py:316-372``) and ``Pipeline.run``.

Class paths starting with ``lenskit.`` resolve to the real LensKit when it is importable
and to this package's mirror (``lkpy_amd.``) otherwise, which is what lets the reference's
TOML files name ``lenskit.als.ImplicitMFScorer`` / ``lenskit.knn.ItemKNNScorer``.
"""

from __future__ import annotations

import importlib
import inspect
from dataclasses import replace
from pathlib import Path
from typing import Any, get_type_hints

import numpy as np

from .data import ItemList, RecQuery
from .training import Trainable, TrainingOptions


class Component:
    """
    Base class of pipeline components: ``Cls()``, ``Cls(config_obj)``, ``Cls(dict)`` or
    ``Cls(**kwargs)`` (components.py:126-142); the config class comes from the ``config``
    annotation.
    """

    config: Any = None

    def __init__(self, config: object | None = None, **kwargs: Any):
        cfg_cls = self._config_class()
        if config is None:
            config = cfg_cls(**kwargs) if cfg_cls is not None else None
        elif kwargs:
            raise RuntimeError("cannot supply both a configuration object and kwargs")
        elif cfg_cls is not None and not isinstance(config, cfg_cls):
            config = cfg_cls.model_validate(config) if hasattr(cfg_cls, "model_validate") \
                else cfg_cls(**config)
        self.config = config

    @classmethod
    def _config_class(cls):
        for klass in cls.__mro__:
            ann = klass.__dict__.get("__annotations__", {})
            if "config" in ann:
                hints = get_type_hints(klass)
                ct = hints.get("config")
                return ct if inspect.isclass(ct) else None
        return None

    def dump_config(self) -> dict:
        cfg = self.config
        if cfg is None:
            return {}
        return cfg.model_dump() if hasattr(cfg, "model_dump") else dict(vars(cfg))

    def __repr__(self):
        return f"<{self.__class__.__name__} {self.dump_config()}>"


def import_path_string(path: str):
    "``module.Class`` or ``module:Class`` -> object (``_types.py:284-309``)."
    if ":" in path:
        mod_name, name = path.split(":", 1)
    else:
        mod_name, name = path.rsplit(".", 1)
    if mod_name == "lenskit" or mod_name.startswith("lenskit."):
        try:
            mod = importlib.import_module(mod_name)
            return getattr(mod, name)
        except Exception:
            mod_name = "lkpy_amd" + mod_name[len("lenskit"):]
    mod = importlib.import_module(mod_name)
    return getattr(mod, name)


class Node:
    def __init__(self, name, component=None, wiring=None, kind="component"):
        self.name, self.component, self.kind = name, component, kind
        self.wiring = wiring or {}

    def __repr__(self):
        return f"<Node {self.name} ({self.kind})>"


class Pipeline:
    def __init__(self, name: str | None = None):
        self.name = name
        self.nodes: dict[str, Node] = {}
        self.aliases: dict[str, str] = {}
        self.default: str | None = None

    # -- construction -----------------------------------------------------------
    def create_input(self, name):
        self.nodes[name] = Node(name, kind="input")
        return name

    def add_component(self, name, comp, config=None, **wiring):
        if inspect.isclass(comp):
            comp = comp(config) if config is not None else comp()
        self.nodes[name] = Node(name, comp, wiring)
        return name

    def replace_component(self, name, comp, config=None):
        "keeps the node's slot and wiring (``_builder.py:441-443``)"
        node = self.nodes[name]
        if inspect.isclass(comp):
            comp = comp(config) if config is not None else comp()
        node.component = comp
        node.kind = "component"

    def use_first_of(self, name, *sources):
        self.nodes[name] = Node(name, kind="first-of", wiring={"sources": list(sources)})
        return name

    def alias(self, name, target):
        self.aliases[name] = target

    def node(self, name: str) -> Node:
        return self.nodes[self.aliases.get(name, name)]

    # -- standard bases (src/lenskit/pipeline/_common.py:176-251) -----------------
    @classmethod
    def std_topn(cls, name=None, options=None) -> "Pipeline":
        from .basic import TopNRanker, TrainingItemsCandidateSelector, UserTrainingHistoryLookup

        options = options or {}
        p = cls(name)
        p.create_input("query")
        p.create_input("items")
        p.create_input("n")
        p.add_component("history-lookup", UserTrainingHistoryLookup, query="query")
        p.add_component("candidate-selector", TrainingItemsCandidateSelector,
                        query="history-lookup")
        p.use_first_of("candidates", "items", "candidate-selector")
        p.nodes["scorer"] = Node("scorer", None, {"query": "history-lookup",
                                                  "items": "candidates"}, kind="placeholder")
        p.add_component("ranker", TopNRanker, {"n": options.get("default_length")},
                        items="scorer", n="n")
        p.alias("recommender", "ranker")
        p.default = "recommender"
        return p

    @classmethod
    def std_topn_predict(cls, name=None, options=None) -> "Pipeline":
        from .basic import BiasScorer, FallbackScorer

        options = options or {}
        p = cls.std_topn(name, options)
        if options.get("fallback_predictor", True) is False:
            p.alias("rating-predictor", "scorer")
        else:
            p.add_component("fallback-predictor", BiasScorer, query="history-lookup",
                            items="candidates")
            p.add_component("rating-merger", FallbackScorer, primary="scorer",
                            backup="fallback-predictor")
            p.alias("rating-predictor", "rating-merger")
        return p

    @classmethod
    def from_config(cls, cfg: dict) -> "Pipeline":
        meta = cfg.get("meta", {})
        options = dict(cfg.get("options", {}))
        base = options.pop("base", None)
        if base == "std:topn":
            pipe = cls.std_topn(meta.get("name"), options)
        elif base == "std:topn-predict":
            pipe = cls.std_topn_predict(meta.get("name"), options)
        elif base is None:
            pipe = cls(meta.get("name"))
        else:
            raise ValueError(f"unsupported pipeline base {base}")
        for name, spec in cfg.get("components", {}).items():
            path = spec.get("class", spec.get("code"))  # `class` <-> `code` alias
            comp_cls = import_path_string(path)
            config = spec.get("config", None)
            if name in pipe.nodes:
                pipe.replace_component(name, comp_cls, config)
            else:
                pipe.add_component(name, comp_cls, config, **spec.get("inputs", {}))
        return pipe

    @classmethod
    def load_config(cls, path) -> "Pipeline":
        "``Pipeline.load_config(toml)`` (cli/pipeline/_load.py:35)."
        import tomli

        path = Path(path)
        with open(path, "rb") as f:
            if path.suffix == ".toml":
                cfg = tomli.load(f)
            elif path.suffix == ".json":
                import json

                cfg = json.load(f)
            else:
                import yaml

                cfg = yaml.safe_load(f)
        return cls.from_config(cfg)

    # -- training (src/lenskit/pipeline/_impl.py:316-372) -------------------------
    def train(self, data, options: TrainingOptions | None = None) -> None:
        options = options or TrainingOptions()
        # _impl.py:346-352: only a SEED is wrapped and spawned; None, a Generator or a
        # BitGenerator is handed to every component unchanged
        if isinstance(options.rng, np.random.SeedSequence):
            seed = options.rng
        elif options.rng is None or isinstance(options.rng, (np.random.Generator,
                                                             np.random.BitGenerator)):
            seed = None
        else:
            seed = np.random.SeedSequence(options.rng)
        for node in self.nodes.values():
            comp = node.component
            if comp is None or not isinstance(comp, Trainable):
                continue
            if comp.is_trained() and not options.retrain:
                continue  # skipped components consume no child seed (_impl.py:360-361)
            # one spawned child per component actually trained, in node order (_impl.py:363-366)
            c_opts = options if seed is None else replace(options, rng=seed.spawn(1)[0])
            comp.train(data, c_opts)

    # -- execution ------------------------------------------------------------------
    def run(self, node: str | None = None, /, **inputs):
        name = node or self.default
        cache: dict[str, Any] = {}
        return self._eval(self.aliases.get(name, name), inputs, cache)

    def run_all(self, *nodes, **inputs):
        cache: dict[str, Any] = {}
        return {n: self._eval(self.aliases.get(n, n), inputs, cache) for n in nodes}

    def _eval(self, name, inputs, cache):
        if name in cache:
            return cache[name]
        node = self.nodes[name]
        if node.kind == "input":
            val = inputs.get(name)
        elif node.kind == "first-of":
            val = None
            for src in node.wiring["sources"]:
                val = self._eval(src, inputs, cache)
                if val is not None:
                    break
        elif node.kind == "placeholder" or node.component is None:
            raise RuntimeError(f"pipeline node {name} has no component")
        else:
            sig = inspect.signature(node.component.__call__)
            kwargs = {}
            for pname, src in node.wiring.items():
                v = self._eval(src, inputs, cache)
                if pname in sig.parameters:
                    kwargs[pname] = v
            val = node.component(**kwargs)
        cache[name] = val
        return val


def topn_pipeline(scorer, *, predicts_ratings: bool = False, n: int | None = None,
                  name: str | None = None) -> Pipeline:  # fmt: skip
    p = Pipeline.std_topn_predict(name, {"default_length": n}) if predicts_ratings \
        else Pipeline.std_topn(name, {"default_length": n})
    p.replace_component("scorer", scorer)
    return p


def predict_pipeline(scorer, *, fallback: bool = True, n: int | None = None,
                     name: str | None = None) -> Pipeline:  # fmt: skip
    p = Pipeline.std_topn_predict(name, {"default_length": n, "fallback_predictor": fallback})
    p.replace_component("scorer", scorer)
    return p
