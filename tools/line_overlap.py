"""Line-overlap metric between this package's Python modules and the files of
the same role in a reference checkout: the share of normalised, non-trivial
code lines (comments and docstrings removed) that also occur verbatim in the
reference file.  Usage: line_overlap.py [--show module.py] [reference_root]."""
import io
import re
import sys
import tokenize

PAIRS = [("singa_amd/model.py", "python/singa/model.py"),
         ("singa_amd/layer.py", "python/singa/layer.py"),
         ("singa_amd/tensor.py", "python/singa/tensor.py"),
         ("singa_amd/autograd.py", "python/singa/autograd.py"),
         ("singa_amd/opt.py", "python/singa/opt.py"),
         ("singa_amd/initializer.py", "python/singa/initializer.py"),
         ("singa_amd/utils/__init__.py", "python/singa/utils.py"),
         ("singa_amd/device.py", "python/singa/device.py"),
         ("singa_amd/data.py", "python/singa/data.py"),
         ("singa_amd/image_tool.py", "python/singa/image_tool.py"),
         ("singa_amd/sonnx.py", "python/singa/sonnx.py"),
         ("singa_amd/snapshot.py", "python/singa/snapshot.py")]
TRIVIAL = {"else:", "pass", "return", "try:", ")", "}", "]", "break",
           "continue", "finally:"}


def code_lines(path):
    src = open(path).read()
    lines = src.split("\n")
    drop, prev = set(), None
    for t in tokenize.generate_tokens(io.StringIO(src).readline):
        if t.type == tokenize.COMMENT:
            row = t.start[0] - 1
            lines[row] = lines[row][:t.start[1]]
        starts_stmt = prev is None or prev.type in (
            tokenize.INDENT, tokenize.NEWLINE, tokenize.NL, tokenize.DEDENT)
        if t.type == tokenize.STRING and starts_stmt:
            drop.update(range(t.start[0] - 1, t.end[0]))
        if t.type not in (tokenize.NL, tokenize.COMMENT):
            prev = t
    out = []
    for i, text in enumerate(lines):
        text = re.sub(r"\s+", " ", text.strip())
        if i not in drop and text and text not in TRIVIAL:
            out.append(text)
    return out


def main(argv):
    show = None
    if "--show" in argv:
        k = argv.index("--show")
        show = argv[k + 1]
        del argv[k:k + 2]
    root = argv[1] if len(argv) > 1 else "/root/reference"
    here = __file__.rsplit("/tools/", 1)[0]
    for mine, theirs in PAIRS:
        a = code_lines(here + "/" + mine)
        b = set(code_lines(root + "/" + theirs))
        hits = [l for l in a if l in b]
        print("%-28s %4d / %4d = %4.1f%%" % (mine, len(hits), len(a),
                                             100.0 * len(hits) / max(len(a), 1)))
        if show and mine.endswith(show):
            for l in hits:
                print("    " + l)


if __name__ == "__main__":
    main(list(sys.argv))
