#!/usr/bin/env python3
"""A/B builds of libmi355env.so: one translation unit recompiled with extra flags, linked with the product's other objects.

    python scripts/build_variant.py <name> <unit.hip>[,<unit2.hip>...] [extra hipcc flags ...]   ->  gymnasium_amd/csrc/libmi355env_<name>.so

Run the variant with MI355ENV_LIBRARY=<path> (gymnasium_amd/_native.py).  Experiment infrastructure: nothing in the package uses it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gymnasium_amd.csrc import build as B  # noqa: E402


def main():
    name, units, extra = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    B.build(verbose=False)  # the product's objects must be current
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    jobs, variant = [], {}
    for unit in units:  # the units compile side by side
        obj = variant[unit] = os.path.join(B.HERE, f"{os.path.splitext(unit)[0]}_{name}.o")
        cmd = [hipcc, f"--offload-arch={B.ARCH}", *B.FLAGS, *B.TU_FLAGS.get(unit, []), *extra, "-c", "-o", obj, os.path.join(B.HERE, unit)]
        jobs.append((cmd, subprocess.Popen(cmd, cwd=B.HERE)))
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    objs = [variant.get(src, os.path.join(B.HERE, os.path.splitext(src)[0] + ".o")) for src in B.SOURCES]
    out = os.path.join(B.HERE, f"libmi355env_{name}.so")
    subprocess.run([hipcc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out] + objs, check=True, cwd=B.HERE)
    print(out)


if __name__ == "__main__":
    main()
