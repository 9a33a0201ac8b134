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
                "%s: training=True (dropout / batch statistics / backward) is not built yet; "
                "see DESIGN.md section 'next'" % self.name)
