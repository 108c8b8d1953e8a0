"""Model zoo: the networks of the reference's examples/cnn/model and
examples/mlp as importable modules.  ``create(name, **kwargs)`` builds one by
name ('cnn', 'alexnet', 'resnet18' ... 'resnet152', 'xceptionnet', 'mlp')."""
from . import alexnet, cnn, mlp, resnet, xceptionnet  # noqa: F401


def create(name, **kwargs):
    name = name.lower()
    if name.startswith("resnet"):
        return getattr(resnet, name)(**kwargs)
    if name in ("xception", "xceptionnet"):
        return xceptionnet.create_model(**kwargs)
    return {"cnn": cnn, "alexnet": alexnet, "mlp": mlp}[name].create_model(
        **kwargs)
