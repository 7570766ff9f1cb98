"""Operator surface of a flow layer (reference surface: inf/layers/flowlayer.py:7-51).

A layer maps `forward(input, context) -> (output, log|det J| per sample)`, undoes it with `reverse(input, context)` and
reports `logdet(input, context)`.  Layers with a self-normalised gradient (ModifiedGradFlowLayer) take one more
argument, `compute_expensive`, which selects the exact dense computation; preprocessing layers are only a marker (their
log-det is left out of FlowSequential.non_preprocessing_logdet)."""
import abc

from torch import nn

_SURFACE = ("forward", "reverse", "logdet")


def _abstract(name, with_switch):
    if with_switch:
        def method(self, input, context=None, compute_expensive=False):
            raise NotImplementedError(name)
    else:
        def method(self, input, context=None):
            raise NotImplementedError(name)
    method.__name__ = name
    return abc.abstractmethod(method)


class FlowLayer(nn.Module, metaclass=abc.ABCMeta):
    """Abstract: forward, reverse, logdet."""


class ModifiedGradFlowLayer(FlowLayer):
    """Abstract: the same three methods with the `compute_expensive` switch."""


for _name in _SURFACE:
    setattr(FlowLayer, _name, _abstract(_name, False))
    setattr(ModifiedGradFlowLayer, _name, _abstract(_name, True))
FlowLayer.__abstractmethods__ = frozenset(_SURFACE)
ModifiedGradFlowLayer.__abstractmethods__ = frozenset(_SURFACE)


class PreprocessingFlowLayer(FlowLayer):
    """Marker base class (dequantisation, normalisation)."""


def mark_expensive(func):
    """Tag a method as the exact / expensive computation (flowlayer.py:49-51)."""
    func._expensive_computation = True
    return func
