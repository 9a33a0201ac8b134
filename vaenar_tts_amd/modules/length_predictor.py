"""DenseLengthPredictor mirror (/root/reference/modules/length_predictor.py:30-42)."""
from ._base import EngineModule, check


class DenseLengthPredictor(EngineModule):
    var_prefix = "length_predictor"

    def __init__(self, activation, name='lengthPredictor', engine=None):
        super().__init__(name, engine)
        self.activation = activation

    def __call__(self, inputs, input_lengths, training=None):
        """length_predictor.py:35-42: [B,T,D] -> float lengths [B] (device)."""
        e = self.engine
        x = self._f32(inputs)
        B, T, _ = x.shape
        lens = self._i32(input_lengths, B, T)
        out = e.empty((B,))
        e.call("vnr_length_predictor_fwd", x.ptr, lens.ptr, B, T, out.ptr)
        return out

    call = __call__
