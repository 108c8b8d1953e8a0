"""Sentiment classifier: fused multi-layer RNN/LSTM/GRU over embedded tokens,
then two dense layers (counterpart of the reference's
examples/rnn/imdb_model.py:23-58)."""
import os
import sys

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)),
                             "..", "..")]

from singa_amd import autograd, layer, model  # noqa: E402


class IMDBModel(model.Model):

    def __init__(self, hidden_size, mode="lstm", return_sequences=False,
                 bidirectional=False, num_layers=1, dropout=0.0):
        super().__init__()
        self.lstm = layer.CudnnRNN(hidden_size=hidden_size, batch_first=True,
                                   rnn_mode=mode,
                                   return_sequences=return_sequences,
                                   num_layers=num_layers, dropout=dropout,
                                   bidirectional=bidirectional)
        self.l1 = layer.Linear(64)
        self.l2 = layer.Linear(2)

    def forward(self, x):
        y = self.lstm(x)
        y = autograd.reshape(y, (y.shape[0], -1))
        y = autograd.relu(self.l1(y))
        return self.l2(y)

    def train_one_batch(self, x, y):
        out = self.forward(x)
        loss = autograd.softmax_cross_entropy(out, y)
        self.optimizer(loss)
        return out, loss

    def set_opt(self, optimizer):
        self.optimizer = optimizer
