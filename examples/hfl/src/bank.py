"""The UCI Bank-Marketing table for the federated example: one-hot encoding
of its categorical columns, an 80/20 split, one shard per client, min-max
scaling fitted on a client's own training shard (counterpart of the
reference's examples/hfl/src/bank.py:29-97).

    python -m src.bank N [--non-iid] [--source bank-additional-full.csv]

writes ``data/bank_{train,test}_<i>.csv`` for i < N.  Without the UCI file
(it cannot be downloaded here) ``--synthetic`` first generates a table of the
same 21 columns -- ten numeric, ten categorical, the yes/no outcome -- from
fixed marginals and a logistic outcome model, so every later step runs on
data of the real shape.
"""
import argparse
import os

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
DATA_DIR = os.path.join(HERE, "..", "data")
SOURCE = "bank-additional-full.csv"

CATEGORICAL = {
    "job": ["admin.", "blue-collar", "entrepreneur", "housemaid",
            "management", "retired", "self-employed", "services", "student",
            "technician", "unemployed", "unknown"],
    "marital": ["divorced", "married", "single", "unknown"],
    "education": ["basic.4y", "basic.6y", "basic.9y", "high.school",
                  "illiterate", "professional.course", "university.degree",
                  "unknown"],
    "default": ["no", "yes", "unknown"],
    "housing": ["no", "yes", "unknown"],
    "loan": ["no", "yes", "unknown"],
    "contact": ["cellular", "telephone"],
    "month": ["mar", "apr", "may", "jun", "jul", "aug", "sep", "oct", "nov",
              "dec"],
    "day_of_week": ["mon", "tue", "wed", "thu", "fri"],
    "poutcome": ["failure", "nonexistent", "success"],
}
COLUMNS = ["age", "job", "marital", "education", "default", "housing", "loan",
           "contact", "month", "day_of_week", "duration", "campaign", "pdays",
           "previous", "poutcome", "emp.var.rate", "cons.price.idx",
           "cons.conf.idx", "euribor3m", "nr.employed", "y"]


def synthesize(path, rows=8000, seed=11):
    """A table with the columns, value sets and separator of the UCI file."""
    rng = np.random.RandomState(seed)
    df = pd.DataFrame({
        "age": rng.randint(18, 90, rows),
        "duration": rng.gamma(2.0, 130.0, rows).astype(int),
        "campaign": 1 + rng.poisson(1.5, rows),
        "pdays": np.where(rng.rand(rows) < 0.9, 999, rng.randint(0, 30, rows)),
        "previous": rng.poisson(0.2, rows),
        "emp.var.rate": rng.choice([-3.4, -1.8, -0.1, 1.1, 1.4], rows),
        "cons.price.idx": np.round(92.2 + 2.5 * rng.rand(rows), 3),
        "cons.conf.idx": np.round(-50.8 + 24.0 * rng.rand(rows), 1),
        "euribor3m": np.round(0.6 + 4.4 * rng.rand(rows), 3),
        "nr.employed": rng.choice([4963.6, 5017.5, 5099.1, 5191.0, 5228.1],
                                  rows),
    })
    for col, values in CATEGORICAL.items():
        df[col] = rng.choice(values, rows)
    # outcome: long calls, earlier success, students / retired and low
    # euribor say yes more often
    z = (0.004 * (df["duration"] - 260) - 0.6 * (df["euribor3m"] - 3.0)
         + 1.5 * (df["poutcome"] == "success")
         + 0.8 * df["job"].isin(["student", "retired"])
         - 0.3 * (df["contact"] == "telephone") - 1.2)
    p = 1.0 / (1.0 + np.exp(-z))
    df["y"] = np.where(rng.rand(rows) < p, "yes", "no")
    df[COLUMNS].to_csv(path, sep=";", index=False)
    return path


def encode(df):
    """Numeric columns as they are, every other column one-hot, column order
    kept."""
    parts = []
    for col in df.columns.values:
        if pd.api.types.is_numeric_dtype(df[col]):
            parts.append(df[col])
        else:
            parts.append(pd.get_dummies(df[col], prefix=col, dtype=np.int64))
    return pd.concat(parts, axis=1)


def normalize(x_train, x_test):
    """Min-max scaling fitted on the training rows."""
    lo, hi = x_train.min(axis=0), x_train.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    return ((x_train - lo) / span).astype(np.float32), \
        ((x_test - lo) / span).astype(np.float32)


def split(num, source=None, out_dir=None, non_iid=False, seed=0):
    """Encode the table, split 80/20 and deal both parts to ``num`` clients.
    ``non_iid``: rows are ordered by outcome and by job before dealing, so
    the clients' label shares differ (0 % to 100 % yes at the extremes)."""
    out_dir = out_dir or DATA_DIR
    source = source or os.path.join(out_dir, SOURCE)
    df = pd.read_csv(source, sep=";")
    df["y"] = (df["y"] == "yes").astype(int)
    data = encode(df)
    data = data.sample(frac=1.0, random_state=seed).reset_index(drop=True)
    cut = int(0.8 * len(data))
    train, test = data[:cut], data[cut:]
    if non_iid:
        train = train.sort_values(["y"], kind="stable")
        test = test.sort_values(["y"], kind="stable")
    train.to_csv(os.path.join(out_dir, "bank_train_.csv"), index=False)
    test.to_csv(os.path.join(out_dir, "bank_test_.csv"), index=False)
    per_train, per_test = len(train) // num, len(test) // num
    print("train_per_client:", per_train)
    print("test_per_client:", per_test)
    for i in range(num):
        train[i * per_train:(i + 1) * per_train].to_csv(
            os.path.join(out_dir, "bank_train_%d.csv" % i), index=False)
        test[i * per_test:(i + 1) * per_test].to_csv(
            os.path.join(out_dir, "bank_test_%d.csv" % i), index=False)
    return per_train, per_test


def load(device_id, data_dir=None):
    """-> train_x, train_y, val_x, val_y, num_classes of client
    ``device_id``."""
    data_dir = data_dir or DATA_DIR
    train = pd.read_csv(os.path.join(data_dir, "bank_train_%d.csv" % device_id))
    test = pd.read_csv(os.path.join(data_dir, "bank_test_%d.csv" % device_id))
    train_x = np.array(train.drop(["y"], axis=1), dtype=np.float32)
    val_x = np.array(test.drop(["y"], axis=1), dtype=np.float32)
    train_y = np.array(train["y"], dtype=np.int32)
    val_y = np.array(test["y"], dtype=np.int32)
    train_x, val_x = normalize(train_x, val_x)
    return train_x, train_y, val_x, val_y, 2


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("num", type=int, help="number of clients")
    ap.add_argument("--source", default=None,
                    help="the UCI file (default: data/%s)" % SOURCE)
    ap.add_argument("--out", default=None, help="output directory")
    ap.add_argument("--non-iid", action="store_true", dest="non_iid")
    ap.add_argument("--synthetic", action="store_true",
                    help="generate the source table first")
    ap.add_argument("--rows", type=int, default=8000)
    args = ap.parse_args(argv)
    out = args.out or DATA_DIR
    os.makedirs(out, exist_ok=True)
    source = args.source or os.path.join(out, SOURCE)
    if args.synthetic:
        synthesize(source, args.rows)
    split(args.num, source, out, args.non_iid)


if __name__ == "__main__":
    main()
