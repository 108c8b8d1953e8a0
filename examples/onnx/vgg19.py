"""vgg19 from the ONNX model zoo; see zoo.py for options."""
import sys

import zoo

if __name__ == "__main__":
    zoo.main(["vgg19"] + sys.argv[1:])
