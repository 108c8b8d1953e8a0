"""Three-stage CNN for 128x128 cell thumbnails: (conv3 + ReLU, pool2) with
32 / 64 / 64 filters, fc128 + ReLU, fc(num_classes).  Layer plan of the
reference's examples/malaria_cnn/model/cnn.py:25-58; the distributed update
modes come from the shared classifier base."""
from singa_amd import layer
from singa_amd.models.common import Classifier


class CNN(Classifier):

    def __init__(self, num_classes=2, num_channels=3, input_size=128):
        super(CNN, self).__init__(num_classes, input_size)
        self.stages = []
        for i, filters in enumerate((32, 64, 64)):
            conv = layer.Conv2d(filters, 3, padding=0, activation="RELU")
            pool = layer.MaxPool2d(2, 2, padding=0)
            # attributes register the sublayers (names conv1.., pooling1..)
            setattr(self, "conv%d" % (i + 1), conv)
            setattr(self, "pooling%d" % (i + 1), pool)
            self.stages.append((conv, pool))
        self.flatten = layer.Flatten()
        self.linear1 = layer.Linear(128)
        self.relu = layer.ReLU()
        self.linear2 = layer.Linear(num_classes)

    def forward(self, x):
        for conv, pool in self.stages:
            x = pool(conv(x))
        return self.linear2(self.relu(self.linear1(self.flatten(x))))


def create_model(pretrained=False, **kwargs):
    return CNN(**kwargs)


__all__ = ["CNN", "create_model"]
