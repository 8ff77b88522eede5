"""Print per-kernel register / LDS / scratch usage of libmi355env.so (reads the gfx950 code object out of the fat binary)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def extract_all(path):
    """Every gfx950 code object of the file: a library linked from several translation units carries one offload bundle per unit."""
    blob = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, at = [], blob.find(magic)
    while at >= 0:
        (n,) = struct.unpack_from("<Q", blob, at + len(magic))
        pos = at + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, pos)
            triple = blob[pos + 24:pos + 24 + tl].decode(errors="replace")
            pos += 24 + tl
            if "gfx950" in triple:
                out.append(blob[at + off:at + off + size])
        at = blob.find(magic, at + len(magic))
    if not out:
        raise SystemExit("no gfx950 code object in " + path)
    return out


def extract(path):
    return extract_all(path)[0]


def resources(so):
    """[{name (demangled), vgpr_count, vgpr_spill_count, sgpr_count, group_segment_fixed_size, private_segment_fixed_size}, ...] of every gfx950 kernel"""
    notes = ""
    for co in extract_all(so):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co), f.flush()
            notes += subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    rows, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "group_segment_fixed_size":  # first of these keys in a kernel's (alphabetically sorted) metadata block
            if "vgpr_count" in cur:
                rows.append(cur)
            cur = {}
        if k == "name" and "name" in cur:  # argument names inside .args
            continue
        cur[k] = v
    if "vgpr_count" in cur:
        rows.append(cur)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    out = []
    for r, nm in zip(rows, names):
        d = {k: int(v) for k, v in r.items() if k != "name"}
        d["name"] = re.sub(r"\(.*", "", nm.replace("(anonymous namespace)::", ""))
        out.append(d)
    return out


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gymnasium_amd", "csrc", "libmi355env.so")
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    notes = ""
    for co in extract_all(so):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co), f.flush()
            notes += subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    rows, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(name|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "group_segment_fixed_size":  # first of these keys in a kernel's (alphabetically sorted) metadata block
            if "vgpr_count" in cur:
                rows.append(cur)
            cur = {}
        if k == "name" and "name" in cur:  # argument names inside .args
            continue
        cur[k] = v
    if "vgpr_count" in cur:
        rows.append(cur)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'vgpr':>5} {'spill':>5} {'sgpr':>5} {'lds':>7} {'scratch':>8}  kernel")
    for r, nm in zip(rows, names):
        nm = nm.replace("(anonymous namespace)::", "")
        nm = re.sub(r"\(.*", "", nm)
        if pat and not pat.search(nm):
            continue
        print(f"{r.get('vgpr_count','?'):>5} {r.get('vgpr_spill_count','?'):>5} {r.get('sgpr_count','?'):>5} {r.get('group_segment_fixed_size','?'):>7} "
              f"{r.get('private_segment_fixed_size','?'):>8}  {nm}")


if __name__ == "__main__":
    main()
