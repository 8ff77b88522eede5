#!/usr/bin/env python3
"""A/B builds of libmi355env.so: one translation unit recompiled with extra flags, linked with the product's other objects.

    python scripts/build_variant.py <name> <unit.hip> [extra hipcc flags ...]   ->  gymnasium_amd/csrc/libmi355env_<name>.so

Run the variant with MI355ENV_LIBRARY=<path> (gymnasium_amd/_native.py).  Experiment infrastructure: nothing in the package uses it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gymnasium_amd.csrc import build as B  # noqa: E402


def main():
    name, unit, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build(verbose=False)  # the product's objects must be current
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = os.path.join(B.HERE, f"{os.path.splitext(unit)[0]}_{name}.o")
    cmd = [hipcc, f"--offload-arch={B.ARCH}", *B.FLAGS, *B.TU_FLAGS.get(unit, []), *extra, "-c", "-o", obj, os.path.join(B.HERE, unit)]
    subprocess.run(cmd, check=True, cwd=B.HERE)
    objs = [obj if src == unit else os.path.join(B.HERE, os.path.splitext(src)[0] + ".o") for src in B.SOURCES]
    out = os.path.join(B.HERE, f"libmi355env_{name}.so")
    subprocess.run([hipcc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out] + objs, check=True, cwd=B.HERE)
    print(out)


if __name__ == "__main__":
    main()
