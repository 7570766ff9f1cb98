"""Operator surface of a flow layer (reference: inf/layers/flowlayer.py:7-51).

forward(input, context) -> (output, log|det J|); reverse(input, context) -> input of forward;
logdet(input, context).  ModifiedGradFlowLayer adds the `compute_expensive` switch that selects
the exact (dense) computation instead of the self-normalised one.
"""
import abc

import torch.nn as nn


class FlowLayer(nn.Module, metaclass=abc.ABCMeta):
    @abc.abstractmethod
    def forward(self, input, context=None):
        ...

    @abc.abstractmethod
    def reverse(self, input, context=None):
        ...

    @abc.abstractmethod
    def logdet(self, input, context=None):
        ...


class ModifiedGradFlowLayer(FlowLayer):
    @abc.abstractmethod
    def forward(self, input, context=None, compute_expensive=False):
        ...

    @abc.abstractmethod
    def reverse(self, input, context=None, compute_expensive=False):
        ...

    @abc.abstractmethod
    def logdet(self, input, context=None, compute_expensive=False):
        ...


class PreprocessingFlowLayer(FlowLayer):
    """Marker base class: layers whose log-det is excluded from non_preprocessing_logdet."""


def mark_expensive(func):
    """Tag a method as the exact/expensive computation (flowlayer.py:49-51)."""
    func._expensive_computation = True
    return func
