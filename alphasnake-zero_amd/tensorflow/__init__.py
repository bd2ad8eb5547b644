"""Minimal stand-in so the reference's unmodified entry scripts import: train.py line 1 does
`import tensorflow as tf` and probes for a Google-Cloud TPU inside a bare try/except (train.py:15-27); with this
package on sys.path the probe raises and the script continues with TPU = None.  Nothing else of TensorFlow is
provided: the net lives in utils/alpha_nnet.py + snake_engine (HIP kernels)."""


class _NoTPU:
    def __getattr__(self, name):
        raise RuntimeError("TensorFlow is not part of this build: no TPU / tf.* functionality (using the MI355X engine)")


class _ClusterResolver:
    @staticmethod
    def TPUClusterResolver(*a, **k):
        raise RuntimeError("no Google-Cloud TPU in this build")


class _Distribute:
    cluster_resolver = _ClusterResolver()
    experimental = _NoTPU()


distribute = _Distribute()
config = _NoTPU()
tpu = _NoTPU()
__version__ = "0.0-shim"
