"""Shared plumbing of the module mirrors: every module is a thin view on one Engine
(the engine owns the weights, the stream and the workspace)."""
import ctypes as C

import numpy as np

from .._lib import DeviceArray, check


class EngineModule:
    var_prefix = None        # object-graph attribute of this layer in models.VAENAR (models.py:16-65)

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

    def _training(self, training, dropout_seed=None):
        """The reference's `training=` argument as a context: Dropout on (counter-based masks keyed by the engine option
        "dropout_seed"), BatchNormalization on batch statistics + moving update for the calls inside the `with` block; the
        option is reset on the way out even when the call fails (a handle left in training mode would run the next inference
        with Dropout and batch statistics)."""
        return _TrainingScope(self.engine, training, dropout_seed)

    @property
    def trainable_variables(self):
        """This layer's slice of `model.trainable_variables` (Keras order)."""
        from ..variables import model_variables
        return model_variables(self.engine, self.engine.hps, True, self.var_prefix, self.engine.has_posterior())

    @property
    def variables(self):
        from ..variables import model_variables
        return model_variables(self.engine, self.engine.hps, False, self.var_prefix, self.engine.has_posterior())


class _TrainingScope:
    def __init__(self, engine, training, dropout_seed):
        self.engine, self.training, self.seed = engine, bool(training), dropout_seed

    def __enter__(self):
        self.engine.set_option("training", 1 if self.training else 0)
        if self.training and self.seed is not None:
            self.engine.set_option("dropout_seed", int(self.seed) & 0x7FFFFFFF)
        return self

    def __exit__(self, *exc):
        self.engine.set_option("training", 0)
        return False
