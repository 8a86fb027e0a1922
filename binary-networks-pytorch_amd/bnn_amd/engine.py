"""Recipe engine: a YAML list of binarisation steps -> ``BConfig`` -> ``prepare_binary_model``.

Host-side mirror of the reference's ``bnn/engine.py:23-79`` (``BinaryChef``): same constructor,
``len()``, ``get_num_steps()``, ``run_step(model, i)`` and ``next(model)``.  Differences, all in the
direction of robustness: names are resolved through a registry instead of ``eval`` (so a recipe
cannot execute code), no ``easydict`` dependency, and both ``name`` and ``NAME`` keys are accepted
(the shipped ``examples/recepies/xnor-net.yaml:6`` uses ``NAME`` and crashes upstream).

    step0:
      pre_activation:  {name: "BasicInputBinarizer"}
      post_activation: {name: "BasicScaleBinarizer"}
      weight:          {name: "XNORWeightBinarizer", args: {compute_alpha: True, center_weights: True}}
      ignore_layer_names: ["_last_", "_first_"]
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List

import torch.nn as nn
import yaml

from . import ops
from .bconfig import BConfig, Identity
from .binarize import prepare_binary_model

_SECTIONS = (("pre_activation", "activation_pre_process"),
             ("post_activation", "activation_post_process"),
             ("weight", "weight_pre_process"))


def _default_registry() -> Dict[str, Callable[..., nn.Module]]:
    reg: Dict[str, Callable[..., nn.Module]] = {name: getattr(ops, name) for name in ops.__all__
                                                if isinstance(getattr(ops, name), type)}
    reg.update({"Identity": Identity, "nn.Identity": nn.Identity, "torch.nn.Identity": nn.Identity})
    return reg


class BinaryChef:
    """Converts a model step by step according to a YAML recipe.

    >>> chef = BinaryChef("recipe.yaml")
    >>> for _ in range(len(chef)):
    ...     model = chef.next(model)      # (train between the steps)
    """

    def __init__(self, config: str, user_modules: List[Callable[..., nn.Module]] = []) -> None:
        with open(config) as fh:
            raw = yaml.safe_load(fh) or {}
        self.config: List[Dict[str, Any]] = [raw[k] for k in raw.keys()]
        self.current_step = 0
        self._registry = _default_registry()
        for mod in user_modules:  # custom binarizers, addressed by class name like upstream
            self._registry[mod.__name__] = mod

    def __len__(self) -> int:
        return len(self.config)

    def get_num_steps(self) -> int:
        return len(self)

    def _factory(self, section: Dict[str, Any], where: str):
        name = section.get("name", section.get("NAME"))
        if name is None:
            raise KeyError(f"recipe section '{where}' has no 'name'")
        try:
            target = self._registry[name]
        except KeyError:
            raise NameError(f"unknown binarizer '{name}' in recipe section '{where}' "
                            f"(pass it through user_modules=[...])") from None
        args = section.get("args")
        if args:
            if not hasattr(target, "with_args"):
                raise TypeError(f"'{name}' takes no recipe arguments")
            target = target.with_args(**args)
        return target

    def step_bconfig(self, step: int) -> BConfig:
        cfg = self.config[step]
        return BConfig(**{field: self._factory(cfg[key], key) for key, field in _SECTIONS})

    def run_step(self, model: nn.Module, step: int) -> nn.Module:
        assert len(self) > step
        cfg = self.config[step]
        return prepare_binary_model(model, bconfig=self.step_bconfig(step),
                                    ignore_layers_name=list(cfg.get("ignore_layer_names", [])))

    def next(self, model: nn.Module) -> nn.Module:  # noqa: A003
        self.current_step += 1
        return self.run_step(model, self.current_step - 1)
