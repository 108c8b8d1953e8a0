"""roberta through the ONNX importer; see ../text_models.py for options."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                ".."))
import text_models  # noqa: E402

if __name__ == "__main__":
    text_models.main(["roberta"] + sys.argv[1:])
