#!/usr/bin/env python3
"""Cut a release: compute the next version from the latest git tag (or from
``singa_amd.__version__`` when the history has no tag yet), write it into
``singa_amd/__init__.py`` and ``setup.py``, commit and tag.

    python tool/release/release.py patch          # 0.1.0 -> 0.1.1
    python tool/release/release.py minor          # 0.1.3 -> 0.2.0
    python tool/release/release.py rc             # 0.2.0 -> 0.2.0-rc1 -> -rc2
    python tool/release/release.py stable         # 0.2.0-rc2 -> 0.2.0
    python tool/release/release.py minor --dry-run

Role parity: reference tool/release/release.py (same release kinds; the
reference edits its conda recipe, this tree carries the version in the
package).
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
KINDS = ("major", "minor", "patch", "rc", "stable")


def parse(tag):
    """'1.2.3' or '1.2.3-rc4' -> (1, 2, 3, None | 4)"""
    m = re.fullmatch(r"v?(\d+)\.(\d+)\.(\d+)(?:-rc(\d+))?", tag.strip())
    if not m:
        raise ValueError("malformed version %r" % tag)
    major, minor, patch, rc = m.groups()
    return int(major), int(minor), int(patch), None if rc is None else int(rc)


def render(version):
    major, minor, patch, rc = version
    text = "%d.%d.%d" % (major, minor, patch)
    return text if rc is None else "%s-rc%d" % (text, rc)


def bump(version, kind):
    major, minor, patch, rc = version
    if kind in ("major", "minor", "patch"):
        if rc is not None:
            raise ValueError("%s is a release candidate: finish it with "
                             "'stable' (or cut another 'rc') first" %
                             render(version))
        if kind == "major":
            return major + 1, 0, 0, None
        if kind == "minor":
            return major, minor + 1, 0, None
        return major, minor, patch + 1, None
    if kind == "rc":
        return major, minor, patch, 1 if rc is None else rc + 1
    if kind == "stable":
        if rc is None:
            raise ValueError("%s is not a release candidate" % render(version))
        return major, minor, patch, None
    raise ValueError("unknown release kind %r" % kind)


def current_version():
    r = subprocess.run(["git", "describe", "--abbrev=0", "--tags"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                       text=True)
    if r.returncode == 0 and r.stdout.strip():
        return parse(r.stdout)
    with open(os.path.join(ROOT, "singa_amd", "__init__.py")) as f:
        return parse(re.search(r'__version__ = "([^"]+)"', f.read()).group(1))


def rewrite(path, pattern, replacement):
    with open(path) as f:
        text = f.read()
    new, n = re.subn(pattern, replacement, text, count=1)
    if n != 1:
        raise RuntimeError("no version line in %s" % path)
    with open(path, "w") as f:
        f.write(new)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("kind", choices=KINDS)
    ap.add_argument("-y", dest="confirmed", action="store_true",
                    help="do not ask before committing and tagging")
    ap.add_argument("--dry-run", action="store_true",
                    help="print the next version and change nothing")
    args = ap.parse_args(argv)
    old = current_version()
    new = render(bump(old, args.kind))
    print("%s -> %s" % (render(old), new))
    if args.dry_run:
        return new
    if not args.confirmed and \
            input("commit and tag %s? [y/N] " % new).lower() != "y":
        print("aborted")
        return None
    # PEP 440 spells a candidate 1.2.3rc1
    pep440 = new.replace("-rc", "rc")
    rewrite(os.path.join(ROOT, "singa_amd", "__init__.py"),
            r'__version__ = "[^"]+"', '__version__ = "%s"' % pep440)
    rewrite(os.path.join(ROOT, "setup.py"), r'version="[^"]+"',
            'version="%s"' % pep440)
    subprocess.check_call(["git", "commit", "-am", "Release " + new], cwd=ROOT)
    subprocess.check_call(["git", "tag", "-a", new, "-m", "Release " + new],
                          cwd=ROOT)
    print("tagged %s; push with: git push --follow-tags" % new)
    return new


if __name__ == "__main__":
    try:
        main()
    except ValueError as e:
        sys.exit(str(e))
