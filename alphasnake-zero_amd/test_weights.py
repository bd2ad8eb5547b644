"""Weight statistics of a saved model -- this build's counterpart of the reference's ``test_weights.py`` script
(test_weights.py:4-12): shape, smallest and largest entry and sum of squares of every array of ``get_weights()``.
Not a pytest file: ``python test_weights.py [<model name>]``."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def report(weights, out=sys.stdout):
    for w in weights:
        print(w.shape, file=out)
        print("Min weight:", np.min(w), "Max weight:", np.max(w), file=out)
        print("Sum of squres (L2)", np.sum(np.power(w, 2)), file=out)
        print(file=out)


def main(argv):
    from utils.alpha_nnet import AlphaNNet
    name = argv[0] if argv else input("\nEnter the model name:\n")
    report(AlphaNNet(model_name="models/" + name + ".h5").v_net.get_weights())


if __name__ == "__main__":
    main(sys.argv[1:])
