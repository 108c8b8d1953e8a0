"""Answer-selection models: encode a question and a stack of [positive ;
negative] answers, score both by cosine similarity (reference
examples/qabot/qabot_model.py:23-152 has one class per pooling variant; here
one encoder class takes the pooling as an argument, with the reference's class
names kept as thin factories)."""
import os
import sys

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)),
                             "..", "..")]

from singa_amd import autograd, layer, model  # noqa: E402


class _Pair(model.Model):
    """Shared scoring head: split answers into the two halves and compare."""

    def score(self, q, a_batch):
        n = q.shape[0]
        a_pos, a_neg = autograd.split(a_batch, 0, [n, n])
        return autograd.cossim(q, a_pos), autograd.cossim(q, a_neg)


class QAModel_mlp(_Pair):

    def __init__(self, hidden_size):
        super().__init__()
        self.linear_q = layer.Linear(hidden_size)
        self.linear_a = layer.Linear(hidden_size)

    def forward(self, q, a_batch):
        q = self.linear_q(autograd.reshape(q, (q.shape[0], -1)))
        a = self.linear_a(autograd.reshape(a_batch, (a_batch.shape[0], -1)))
        return self.score(q, a)


class QAEncoder(_Pair):
    """BiLSTM encoders with pooling in {'last', 'mean', 'max'} over time."""

    def __init__(self, hidden_size, pooling="max", q_seq=None, a_seq=None,
                 num_layers=1, bidirectional=True):
        super().__init__()
        self.hidden_size = hidden_size
        self.pooling = pooling
        seqs = pooling != "last"
        self.lstm_q = layer.CudnnRNN(hidden_size=hidden_size, batch_first=True,
                                     num_layers=num_layers,
                                     bidirectional=bidirectional,
                                     return_sequences=seqs)
        self.lstm_a = layer.CudnnRNN(hidden_size=hidden_size, batch_first=True,
                                     num_layers=num_layers,
                                     bidirectional=bidirectional,
                                     return_sequences=seqs)
        if pooling == "max":
            self.q_pool = layer.MaxPool2d((q_seq, 1))
            self.a_pool = layer.MaxPool2d((a_seq, 1))

    def _pool(self, x, pool):
        if self.pooling == "last":
            return x                                        # (bs, 2*hid)
        if self.pooling == "mean":
            return autograd.reduce_mean(x, [1], keepdims=0)
        x = autograd.reshape(x, (x.shape[0], 1, x.shape[1], x.shape[2]))
        x = pool(x)                                         # (bs, 1, 1, 2*hid)
        return autograd.reshape(x, (x.shape[0], x.shape[3]))

    def forward(self, q, a_batch):
        q = self._pool(self.lstm_q(q), getattr(self, "q_pool", None))
        a = self._pool(self.lstm_a(a_batch), getattr(self, "a_pool", None))
        return self.score(q, a)


def QAModel(hidden_size, num_layers=1, bidirectional=True,
            return_sequences=False):
    return QAEncoder(hidden_size, "last", num_layers=num_layers,
                     bidirectional=bidirectional)


def QAModel_mean(hidden_size, bidirectional=True, return_sequences=True):
    return QAEncoder(hidden_size, "mean", bidirectional=bidirectional)


def QAModel_maxpooling(hidden_size, q_seq, a_seq, num_layers=1,
                       bidirectional=True, return_sequences=True):
    return QAEncoder(hidden_size, "max", q_seq, a_seq, num_layers,
                     bidirectional)
