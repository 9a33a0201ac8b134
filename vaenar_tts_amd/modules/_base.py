"""Shared plumbing of the module mirrors: every module is a thin view on one Engine
(the engine owns the weights, the stream and the workspace)."""
import ctypes as C

import numpy as np

from .._lib import DeviceArray, check


class EngineModule:
    def __init__(self, name, engine):
        if engine is None:
            raise ValueError("%s needs engine= (vaenar_tts_amd._lib.Engine); there is no CPU path" % name)
        self.name = name
        self.engine = engine

    # helpers -----------------------------------------------------------------
    def _i32(self, x, n=None, fill=None):
        """int32 device vector; None -> filled with `fill` (the reference's 'lengths=None' default)."""
        if x is None:
            x = np.full(n, fill, np.int32)
        return self.engine.asarray(x, np.int32)

    def _f32(self, x):
        return self.engine.asarray(x, np.float32)

    @staticmethod
    def _ptr(a):
        return None if a is None else C.c_void_p(a.ptr)

    def _no_training(self, training):
        if training:
            raise NotImplementedError(
                "%s: training=True is not built for this module (no dropout / batch statistics inside it; "
                "the backward pass is the next row, DESIGN.md)" % self.name)

    def _set_training(self, training, dropout_seed=None):
        """The reference's `training=` argument: Dropout on (counter-based masks keyed by the engine option
        "dropout_seed"), BatchNormalization on batch statistics + moving update.  Forward only."""
        self.engine.set_option("training", 1 if training else 0)
        if training and dropout_seed is not None:
            self.engine.set_option("dropout_seed", int(dropout_seed) & 0x7FFFFFFF)
