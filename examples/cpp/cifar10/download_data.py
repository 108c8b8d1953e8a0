#!/usr/bin/env python3
"""Fetch and unpack the CIFAR-10 binary version next to the C++ programs
(``cifar-10-batches-bin/``).  The programs fall back to synthetic images when
the directory is absent, so this step is optional; it needs network access.

    python download_data.py [bin|py]
"""
import os
import sys
import tarfile
import urllib.request

URLS = {
    "bin": "http://www.cs.toronto.edu/~kriz/cifar-10-binary.tar.gz",
    "py": "http://www.cs.toronto.edu/~kriz/cifar-10-python.tar.gz",
}


def main(kind="bin", dest="."):
    if kind not in URLS:
        raise SystemExit("usage: download_data.py [bin|py]")
    name = "cifar-10-batches-" + kind
    if os.path.isdir(os.path.join(dest, name)):
        print(name, "is already there")
        return
    archive = os.path.join(dest, os.path.basename(URLS[kind]))
    print("downloading", URLS[kind])
    urllib.request.urlretrieve(URLS[kind], archive)
    with tarfile.open(archive) as tar:
        members = [m for m in tar.getmembers()
                   if not os.path.isabs(m.name) and ".." not in m.name]
        tar.extractall(dest, members)
    os.remove(archive)
    print("unpacked into", os.path.join(dest, name))


if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["bin"]))
