"""sha256 over every file under include/loops/ (path + contents, sorted) -- the C++ header API, i.e. everything the example
programs can include (the C ABI header include/loops_amd.h is not part of it) -- plus this repository's own example drivers
(examples/) and the script that builds them: identifies the sources a set of example binaries was built from."""
import hashlib, os, sys

repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
root = os.path.join(repo, "include", "loops")


def digest() -> str:
    h = hashlib.sha256()
    for base, dirs, files in sorted(os.walk(root)):
        dirs.sort()
        for f in sorted(files):
            p = os.path.join(base, f)
            h.update(os.path.relpath(p, root).encode())
            h.update(open(p, "rb").read())
    extra = [os.path.join(repo, "scripts", "build_reference_examples.sh")]
    for base, dirs, files in sorted(os.walk(os.path.join(repo, "examples"))):
        dirs.sort()
        extra += [os.path.join(base, f) for f in sorted(files)]
    for p in extra:
        h.update(os.path.relpath(p, repo).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(digest())
