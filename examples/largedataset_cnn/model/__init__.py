"""``from model import cnn, alexnet, resnet, xceptionnet`` as in scripts
written against the reference's examples/*/model directory: the networks are
the library's model zoo (singa_amd/models)."""
