#!/usr/bin/env python3
"""Identity of the kernel sources: sha256 (first 16 hex digits) over every file that goes into libmcgp_hip.so --
csrc/*.hip, csrc/*.h, this script, the Makefile and include/mcgp.h --, each as name + NUL + contents, sorted by name.

The Makefile compiles it into the library (-DMCGP_SOURCE_HASH, exported as mcgp_build_hash() and as the marker
string "MCGP_BUILD_HASH=<hash>" in the file); monte_carlo_gp_amd/_native.py computes the same value from the tree
and refuses a library that carries another one.  Prints the hash."""
import hashlib
import os

CSRC = os.path.dirname(os.path.abspath(__file__))


def sources():
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
            if f.endswith(('.hip', '.h')) or f in ('Makefile', 'source_hash.py')]
    srcs.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), 'include', 'mcgp.h'))
    return srcs


def source_hash():
    h = hashlib.sha256()
    for path in sorted(sources(), key=os.path.basename):
        h.update(os.path.basename(path).encode() + b'\0')
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(source_hash())
