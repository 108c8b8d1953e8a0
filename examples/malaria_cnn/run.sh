#!/bin/bash
# python train_cnn.py cnn malaria --synthetic -m 3     # no download needed
python train_cnn.py cnn malaria -dir "${1:-/tmp/malaria}"
