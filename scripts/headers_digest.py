"""sha256 over every file under include/loops/ (path + contents, sorted) -- the C++ header API, i.e. everything the example
programs can include (the C ABI header include/loops_amd.h is not part of it): identifies the header set a binary was built from."""
import hashlib, os, sys

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "loops")


def digest() -> str:
    h = hashlib.sha256()
    for base, dirs, files in sorted(os.walk(root)):
        dirs.sort()
        for f in sorted(files):
            p = os.path.join(base, f)
            h.update(os.path.relpath(p, root).encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(digest())
