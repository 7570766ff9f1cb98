"""MI355X-native mirror of the `inf` package surface of girish-lab/Inverse-Flow that sits on the
inverse-of-convolution hot path: FlowLayer ABCs, inv_flow_* layers,
SelfNormConv, FlowSequential.  Everything else of the reference package is out of scope."""
