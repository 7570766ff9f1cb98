"""Only the base class FlowSequential filters on (reference: inf/layers/activations.py);
the activation layers themselves are outside the hot path."""
from .flowlayer import FlowLayer


class FlowActivationLayer(FlowLayer):
    pass
